#!/usr/bin/env python
"""Summarises a rocprofv3 (ROCm 7.2) rocpd SQLite result (`--kernel-trace --stats`, default output format)
as a per-kernel table: calls, total / average / min / max duration, grid, workgroup, VGPR/AGPR/LDS.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/r01_x_kernel_stats.md
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, grid_x, grid_y, grid_z, workgroup_x, count(*), sum(duration), avg(duration), min(duration), "
        "max(duration), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size) "
        "from kernels group by name, grid_x, grid_y, grid_z, workgroup_x order by sum(duration) desc").fetchall()
    total = sum(r[6] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of `{path}`\n")
    print("| kernel | grid (threads) | wg | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds B |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, gx, gy, gz, wx, calls, tot, avg, mn, mx, vg, ag, sg, lds in rows:
        short = name if len(name) < 90 else name[:87] + "..."
        print(f"| `{short}` | {gx}x{gy}x{gz} | {wx} | {calls} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | "
              f"{mx / 1e3:.2f} | {100.0 * tot / total:.1f} | {vg} | {ag} | {sg} | {lds} |")


if __name__ == "__main__":
    main(sys.argv[1])

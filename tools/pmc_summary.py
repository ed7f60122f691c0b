#!/usr/bin/env python
"""Summarises the rocprofv3 passes written by tools/profile.sh.

    python tools/pmc_summary.py gpurun_out/prof_<tag> profiles/<dir> [kernel-substring]

Writes <dir>/kernel_stats.csv (copy of the --kernel-trace --stats pass), <dir>/pmc_summary.md (mean counter value per
dispatch for every kernel and counter) and <dir>/traffic.json: HBM bytes per launch of the dominant kernel from the
separate FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts 64 B
per 128-B request of a wide coalesced stream: doubled; both counters are in KiB).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def counters(directory):
    out = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> values
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                out[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    needle = sys.argv[3] if len(sys.argv) > 3 else "conv3x3_board_kernel"
    os.makedirs(dst, exist_ok=True)
    for path in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(path, os.path.join(dst, "kernel_stats.csv"))

    lines = [f"# rocprofv3 PMC passes (`tools/profile.sh`), mean per dispatch\n"]
    totals = collections.defaultdict(list)
    for pass_name in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_tcp", "pmc_tcc", "pmc_wait"):
        data = counters(os.path.join(src, pass_name))
        for kernel in sorted(data, key=lambda k: -sum(len(v) for v in data[k].values())):
            lines.append(f"\n## {pass_name}: `{kernel}`\n\n| counter | dispatches | mean |\n|---|---|---|")
            for name in sorted(data[kernel]):
                v = data[kernel][name]
                lines.append(f"| {name} | {len(v)} | {sum(v) / len(v):.6g} |")
                if needle in kernel:
                    totals[name] += v
    with open(os.path.join(dst, "pmc_summary.md"), "w") as f:
        f.write("\n".join(lines) + "\n")

    mean = {k: sum(v) / len(v) for k, v in totals.items() if v}
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        traffic = {
            "kernel": needle, "fetch_size_kib": mean["FETCH_SIZE"], "write_size_kib": mean["WRITE_SIZE"],
            "hbm_bytes_per_launch": (2.0 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024.0,
            "note": "FETCH_SIZE doubled (gfx950 wide-stream correction, MI355X_MICROARCH.md HBM section); separate --pmc passes",
            "batch": int(os.environ.get("BENCH_BATCH", "256")), "channels": int(os.environ.get("BENCH_CHANNELS", "256")),
        }
        for extra in ("GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY",
                      "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum",
                      "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCC_REQ_sum", "TCC_EA0_RDREQ_sum",
                      "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum", "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD",
                      "SQ_INSTS_VMEM_WR", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM"):
            if extra in mean:
                traffic[extra] = mean[extra]
        with open(os.path.join(dst, "traffic.json"), "w") as f:
            json.dump(traffic, f, indent=1)
        print(json.dumps(traffic))


if __name__ == "__main__":
    main()

"""Small and mid batches at 20x256 with one / two / three submissions in flight (PCIe-inclusive, board records, the loop inside the
library: dgnn_forward_pipelined_timed_n) -- VERDICT r05 item 2's sizes -- with split launches side by side within the waiter budget
(what ships) and chained one behind the other (test build, DGNN_SPLIT_BESIDE=0: rounds 2-5).

    python tools/r06_sweep_small.py          # runs itself twice per variant, one JSON line per (variant, size)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import numpy as np
    import __graft_entry__ as graft
    graft.package()
    from dream_go_amd import api as shipped, synth
    api = shipped.load_full()
    blocks, channels = int(os.environ.get("SWEEP_BLOCKS", "20")), int(os.environ.get("SWEEP_CHANNELS", "256"))
    net = api.Network.from_json(synth.to_json(synth.make_tensors(blocks, channels, seed=1)))
    board = api.Board(7.5)
    board.place(0, 3, 3); board.place(1, 15, 15); board.place(0, 15, 3)
    for n in [int(v) for v in os.environ.get("SWEEP_SIZES", "32,48,64,80,96,112,128").split(",")]:
        records = np.concatenate([board.record(1, i % 8) for i in range(n)])
        spaces = [net.get_workspace(n) for _ in range(3)]
        row = {"beside": os.environ.get("DGNN_SPLIT_BESIDE", "1"), "crowd": os.environ.get("DGNN_SPLIT_CROWD", "1"), "net": f"{blocks}x{channels}", "boards": n, "kernel": spaces[0].conv_kernel()}
        for alive in (1, 2, 3):
            api.forward_pipelined_timed_n(spaces[:alive], "records", records, n, 50)
            row[f"alive_{alive}"] = round(api.forward_pipelined_timed_n(spaces[:alive], "records", records, n, 600))
        row["towers"] = net.tower_stats()
        print(json.dumps(row), flush=True)
        for w in spaces:
            w.release()
    net.destroy()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        variants = [v.split(":") for v in os.environ.get("SWEEP_VARIANTS", "1:1,0:1,1:1,0:1").split(",")]      # beside:crowd
        for beside, crowd in variants:
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, DGNN_SPLIT_BESIDE=beside, DGNN_SPLIT_CROWD=crowd), check=True)

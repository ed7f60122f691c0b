#!/usr/bin/env python
"""Where the wall time of ONE device batch goes: dgnn_forward_stage_times for a range of batch sizes, one
submission at a time and two alive (how a leaf pool drives the device).  One JSON line per (batch, in flight).

    python tools/stage_times.py [--blocks 20 --channels 256] [--batches 32,64,96,128,160,192,224,256] [--steps 40] > stage_times.jsonl
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--batches", default="32,64,96,128,160,192,224,256")
    ap.add_argument("--capacity", type=int, default=256, help="batch size of the workspaces (the pool's max_batch_size)")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--in-flight", default="1,2")
    args = ap.parse_args()
    import __graft_entry__ as graft
    graft.package()
    from dream_go_amd import api, synth
    api = api.load_full()          # dgnn_forward_stage_times is a measurement entry of include/dgnn_test.h: the test build of the library
    import bench
    net = api.Network.from_json(synth.to_json(synth.make_tensors(args.blocks, args.channels, seed=1)))
    positions = bench.real_game_positions(api)
    records = np.concatenate([positions[(7 * i) % len(positions)][0].record(positions[(7 * i) % len(positions)][1], i % 8) for i in range(args.capacity)])
    flights = [int(v) for v in args.in_flight.split(",")]
    spaces = [net.get_workspace(args.capacity) for _ in range(max(flights))]
    for n in [int(v) for v in args.batches.split(",")]:
        for k in flights:
            out = api.forward_stage_times(spaces[:k], records[:n * api.BOARD_RECORD_BYTES], n, args.steps)
            out.update({"batch": n, "in_flight": k, "net": f"{args.blocks}x{args.channels}", "steps": args.steps, "kernel": spaces[0].conv_kernel(),
                        "boards_per_sec": n * k / (out["host"]["submit_to_answers_us"] * 1e-6) if k == 1 else None})
            print(json.dumps(out), flush=True)
    for w in spaces:
        w.release()
    net.destroy()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Batch-size sweep of the hot path, the counterpart of src/dg_tests/benches/batch_sizes.rs:41-68 (batch 1 ... 256,
Bernoulli(0.2) features).  One JSON line per batch size: evals/s with inputs resident in HBM, and the synchronous
host-buffer rate (PCIe-inclusive).

    python tools/sweep.py [--blocks 20 --channels 256] > profiles/<round>/batch_sweep.jsonl
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.package()
from dream_go_amd import api, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batches", type=str, default="1,2,4,8,16,32,64,128,256")
    ap.add_argument("--in-flight", action="store_true",
                    help="also: two and three submissions of that size in flight (what a leaf pool keeps on a device, predictors/nn.rs:64-67), "
                         "PCIe-inclusive, the loop inside the library (dgnn_forward_pipelined_timed)")
    args = ap.parse_args()
    net = api.Network.from_json(synth.to_json(synth.make_tensors(args.blocks, args.channels, seed=1)))
    for batch in [int(b) for b in args.batches.split(",")]:
        features = synth.make_features(batch, seed=batch)
        with net.get_workspace(batch) as ws:
            ws.upload(features)
            for _ in range(3):
                ws.forward_resident()
            ws.sync()
            total_ms, tower_ms, launches = ws.forward_timed(args.steps)
            for _ in range(3):
                api.forward(ws, features)
            laps = []
            for _ in range(9):                      # the median call: one descheduled call (the boxes grant a process 16 of their 256
                t0 = time.perf_counter()            # CPUs under a CFS quota) used to halve a five-call mean
                api.forward(ws, features)
                laps.append(time.perf_counter() - t0)
            dt = 5 * sorted(laps)[4]
            line = {
                "net": f"{args.blocks}x{args.channels}", "batch": batch, "kernel": ws.conv_kernel(),
                "evals_per_sec_resident": batch * args.steps / (total_ms * 1e-3), "ms_per_pass": total_ms / args.steps,
                "residual_conv_us": tower_ms / max(launches, 1) * 1e3,
                "evals_per_sec_host_buffers": batch * 5 / dt,
            }
            if args.in_flight:
                packed = api.features_pack(features, batch)
                with net.get_workspace(batch) as ws2, net.get_workspace(batch) as ws3:
                    line["evals_per_sec_two_in_flight"] = api.forward_pipelined_timed_n([ws, ws2], "packed", packed, batch, 2 * args.steps)
                    line["evals_per_sec_three_in_flight"] = api.forward_pipelined_timed_n([ws, ws2, ws3], "packed", packed, batch, 3 * args.steps)
            print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""PCIe-inclusive throughput of the host side (scope row f-1): many caller threads with small requests (the shape of
the reference's policy-play mode, BASELINE.json configs[4]: every game thread evaluates <= 8 symmetries of one
position per move, src/libdg_mcts/self_play.rs:370-405) coalesced into device batches by LeafPool.

    python tools/bench_pool.py [--threads 64] [--boards 8] [--calls 40] [--max-batch 256] [--blocks 20 --channels 256]

Prints one JSON line.  NOT the headline metric (bench.py measures the hot path with inputs resident in HBM).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.package()
from dream_go_amd import api, synth  # noqa: E402
api = api.load_full()          # this tool uses entry points of include/dgnn_test.h (raw batcher / single layers): the test build of the library


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=64)
    ap.add_argument("--boards", type=int, default=8)
    ap.add_argument("--calls", type=int, default=40)
    ap.add_argument("--max-batch", type=int, default=256)
    ap.add_argument("--max-batches", type=int, default=2)
    ap.add_argument("--workers", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--channels", type=int, default=256)
    args = ap.parse_args()

    net = api.Network.from_json(synth.to_json(synth.make_tensors(args.blocks, args.channels, seed=1)))
    feats = synth.make_features(args.boards, seed=5)
    batcher = api.Batcher(net, args.max_batch, args.max_batches)
    pool = api.LeafPool(batcher, args.workers)
    pool.predict(feats, args.boards)                       # warm: workspace, weight upload

    def game():
        for _ in range(args.calls):
            pool.predict(feats, args.boards)

    before = batcher.stats()
    threads = [threading.Thread(target=game) for _ in range(args.threads)]
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    [t.start() for t in threads]
    [t.join() for t in threads]
    dt = time.perf_counter() - t0
    cpu = time.process_time() - cpu0
    after = batcher.stats()
    boards = after["boards"] - before["boards"]
    batches = after["batches"] - before["batches"]
    pool.close()
    print(json.dumps({
        "metric": "pool_leaf_evals_per_sec_pcie_inclusive", "value": boards / dt, "unit": "evals/s",
        "callers": args.threads, "boards_per_call": args.boards, "calls": args.threads * args.calls,
        "device_batches": batches, "mean_batch": boards / max(batches, 1), "max_batch": args.max_batch,
        "max_batches_in_flight": args.max_batches, "net": f"{args.blocks}x{args.channels}",
        "cpu_cores_busy": cpu / dt,        # the whole process: callers (Python threads), pool workers, HIP runtime threads
    }))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Where the time of a layer goes in tower_split_kernel: wave 0's s_memtime stamps (start, K loop, epilogue, published) of every
(layer, board, channel block), dumped by DGNN_SPLIT_TRACE=<file> (last launch wins).

    DGNN_SPLIT_TRACE=/tmp/t.bin python tools/split_trace.py --batch 128 [--blocks 20 --channels 256]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--batch", type=int, default=128)
    args = ap.parse_args()
    path = os.environ.setdefault("DGNN_SPLIT_TRACE", "/tmp/split_trace.bin")
    import __graft_entry__ as graft
    graft.package()
    from dream_go_amd import api, synth
    net = api.Network.from_json(synth.to_json(synth.make_tensors(args.blocks, args.channels, seed=1)))
    features = synth.make_features(args.batch, seed=2)
    with net.get_workspace(args.batch) as ws:
        assert ws.conv_kernel() == "tower_split_kernel", ws.conv_kernel()
        for _ in range(3):
            api.forward(ws, features)
    raw = open(path, "rb").read()
    count, batch, groups, ns = np.frombuffer(raw[:16], dtype=np.int32)
    t = np.frombuffer(raw[16:], dtype=np.uint64).reshape(count, batch, groups, 4).copy()
    prefetched = (t[..., 2] >> np.uint64(63)).astype(bool)
    t[..., 2] &= np.uint64((1 << 63) - 1)
    t = t.astype(np.float64)
    t0 = t[0, :, :, 0].min()
    span = t[-1, :, :, 3].max() - t0
    print(f"layers {count}, boards {batch}, workgroups per board {groups} (NS = {ns}); launch span {span:.0f} ticks")
    print("per layer, mean over boards (ticks): prologue | K loop | epilogue+publish | gap to next layer's start | first chunk prefetched")
    for nb in range(groups):
        pro = (t[1:, :, nb, 1] - t[1:, :, nb, 0]).mean()
        loop = (t[1:, :, nb, 2] - t[1:, :, nb, 1]).mean()
        epi = (t[1:, :, nb, 3] - t[1:, :, nb, 2]).mean()
        gap = (t[1:, :, nb, 0] - t[:-1, :, nb, 3]).mean()
        period = (t[-1, :, nb, 3] - t[1, :, nb, 0]).mean() / (count - 1)
        print(f"  block {nb}: {pro:8.0f} | {loop:8.0f} | {epi:8.0f} | {gap:6.0f} | {prefetched[1:-1, :, nb].mean():.2f}   period {period:.0f}")
    lag = (t[1:, :, 1:, 0] - t[1:, :, :1, 0]).mean(axis=(0, 1))
    print("start of a layer relative to block 0 (ticks):", np.round(lag))


if __name__ == "__main__":
    main()

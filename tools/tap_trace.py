#!/usr/bin/env python
"""Where the taps of ONE chunk (chunk 1 of every layer) start, wave by wave, in tower_split_kernel -- for a library built with
tools/build_variant.sh <name> -DDGNN_TAP_STAMPS=<layers of the launch> (2 * blocks + 1).  Measurement only.

    DGNN_LIBRARY=gpurun_libs/libdgnn_<name>.so python tools/tap_trace.py --batch 32 --blocks 9 --channels 128
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=9)
    ap.add_argument("--channels", type=int, default=128)
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    path = os.environ.setdefault("DGNN_SPLIT_TRACE", "/tmp/tap_trace.bin")
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
    words = np.frombuffer(raw[16:], dtype=np.uint64)
    regular = int(count) * int(batch) * int(groups) * 4
    taps = words[regular:].reshape(count, batch, groups, 12, 12).astype(np.float64)
    used = taps[..., 0] > 0                                          # waves that exist and layers with at least two chunks
    print(f"layers {count}, boards {batch}, workgroups per board {groups} (NS = {ns}); waves per workgroup with stamps: {int(used[1].any(axis=(0, 1)).sum())}")
    for kind, layers in (("no skip (conv_1 of a block)", range(1, count, 2)), ("skip (conv_2)", range(2, count, 2))):
        t = taps[list(layers)]
        u = used[list(layers)]
        d = np.diff(t[..., :11], axis=-1)                            # tap 0..8 durations, then tap 8 end -> barrier passed
        print(kind + ": mean cycles per tap of chunk 1, by wave (taps 0..8 | wait at the chunk boundary)")
        for w in range(12):
            if not u[..., w].any():
                continue
            m = d[..., w, :][u[..., w]].mean(axis=0)
            print(f"  wave {w:2d}: " + " ".join(f"{x:5.0f}" for x in m[:9]) + f" | {m[9]:6.0f}   sum {m.sum():7.0f}")
        # skew between the waves of a workgroup at the start of tap 0
        start = t[..., 0]
        valid = u.all(axis=-1) if u.all() else u
        spread = (np.where(u, start, np.nan).max(axis=-1) - np.where(u, start, np.nan).min(axis=-1))
        print(f"  spread of the waves' tap-0 start within a workgroup: mean {np.nanmean(spread):.0f} cycles")


if __name__ == "__main__":
    main()

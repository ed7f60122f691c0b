#!/usr/bin/env python
"""Soak test of tower_board_kernel against one launch per layer: the same weights, many different inputs and repeated passes,
outputs compared bit for bit.  The tower kernel hands activations from layer to layer inside a workgroup (own stores, own
loads, workgroup-scope ordering); an ordering bug would show up as a rare mismatch, so this runs many more passes than the
unit tests do.    python tools/tower_stress.py [--iters 40]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.package()
from dream_go_amd import api, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    bad = 0
    ap.add_argument("--split", action="store_true", help="the split tower (a board over several workgroups) instead")
    args = ap.parse_args()
    shapes = ((20, 256, 256), (9, 128, 256), (6, 64, 77), (20, 256, 200))
    expect = "tower_board_kernel"
    if args.split:
        shapes = ((20, 256, 128), (20, 256, 64), (20, 256, 32), (20, 256, 90), (9, 128, 128), (9, 128, 32), (20, 256, 7), (20, 256, 16), (9, 128, 17), (20, 256, 1))
        expect = "tower_split_kernel"
    for blocks, channels, batch in shapes:
        text = synth.to_json(synth.make_tensors(blocks, channels, seed=7))
        os.environ["DGNN_TOWER"] = "1"
        tower_net = api.Network.from_json(text)
        tower = tower_net.get_workspace(batch).__enter__()
        os.environ["DGNN_TOWER"] = "0"
        layer_net = api.Network.from_json(text)
        layers = layer_net.get_workspace(batch).__enter__()
        assert tower.conv_kernel() == expect and layers.conv_kernel() == "conv3x3_board_kernel"
        mismatches = 0
        for i in range(args.iters):
            features = synth.make_features(batch, seed=1000 + i)
            a = api.forward(tower, features).unwrap()
            b = api.forward(layers, features).unwrap()
            same = np.array_equal(a[0].view(np.uint16), b[0].view(np.uint16)) and np.array_equal(a[1].view(np.uint16), b[1].view(np.uint16))
            # and the tower kernel against itself, back to back on resident inputs
            tower.upload(features)
            for _ in range(3):
                tower.forward_resident()
            tower.sync()
            again = tower.download()
            same = same and np.array_equal(again.policy.view(np.uint16), a[1].view(np.uint16))
            mismatches += 0 if same else 1
        print(f"{blocks}x{channels} batch {batch}: {args.iters} inputs, {mismatches} mismatches")
        bad += mismatches
        tower.release(); layers.release()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""settle F(2x2, 3x3) Winograd with data.  The prior question is numerics, not traffic: the transformed
operands would be fp16 MFMA operands.  For the tower's residual convolutions (every one of them) the oracle's numpy statement is
run twice -- direct form and Winograd form with fp16 U / V and fp32 accumulation (oracle/oracle.py: _conv3x3_winograd) -- on the
round-3 parity workloads (sharpened networks, positions of the committed games) and the outputs are compared.  CPU only.

    python tools/winograd_numerics.py [--blocks 20 --channels 256 --boards 16]
"""
import argparse
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.package()
from dream_go_amd import api, synth  # noqa: E402
from oracle import oracle  # noqa: E402


def positions(count):
    out = []
    for game in json.load(open(os.path.join(ROOT, "tests", "golden", "real_games.json")))["games"]:
        board = api.Board(7.5)
        for color, vertex in re.findall(r";([BW])\[([a-s]{2})\]", game["moves"]):
            c = 0 if color == "B" else 1
            board.place(c, ord(vertex[0]) - 97, ord(vertex[1]) - 97)
            out.append((board.clone(), 1 - c))
    picks = np.linspace(3, len(out) - 1, count).astype(int)
    return np.concatenate([out[i][0].get_features(out[i][1], k % 8).reshape(-1) for k, i in enumerate(picks)]).view(np.uint16)


def compare(net, features, boards, label):
    f32 = lambda bits: np.asarray(bits).view(np.float16).astype(np.float32)
    dv, dp = oracle.forward_numpy(net, features, boards)
    wv, wp = oracle.forward_numpy(net, features, boards, tower_conv=oracle._conv3x3_winograd)
    cv, cp = net.forward(features, boards)                      # the C oracle: how far two DIRECT forms are apart (summation order only)
    line = {
        "workload": label, "boards": boards,
        "winograd_vs_direct": {"max_abs_value": float(np.abs(f32(wv) - f32(dv)).max()), "max_abs_policy": float(np.abs(f32(wp) - f32(dp)).max())},
        "direct_vs_direct": {"max_abs_value": float(np.abs(f32(cv) - f32(dv)).max()), "max_abs_policy": float(np.abs(f32(cp) - f32(dp)).max())},
        "policy_max_range": [float(f32(dp).reshape(boards, 362).max(axis=1).min()), float(f32(dp).reshape(boards, 362).max(axis=1).max())],
    }
    print(json.dumps(line), flush=True)
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--boards", type=int, default=16)
    args = ap.parse_args()
    oracle.build()
    real = positions(args.boards)
    for label, tensors, features in (
        ("He-init, alpha 0.5, Bernoulli(0.2) planes (the bench workload)", synth.make_tensors(args.blocks, args.channels, seed=1), synth.make_features(args.boards)),
        ("He-init, alpha 0.5, real positions", synth.make_tensors(args.blocks, args.channels, seed=1), real),
        ("sharpened (alpha 0.1, dense gains 24 / 16), real positions", synth.make_sharp_tensors(args.blocks, args.channels, seed=2020, policy_gain=24.0, value_gain=16.0), real),
        ("sharpened / 4 (gains 6 / 4), real positions", synth.make_sharp_tensors(args.blocks, args.channels, seed=2020, policy_gain=6.0, value_gain=4.0), real),
    ):
        t, _ = oracle.load_weights(synth.to_json(tensors))
        compare(oracle.OracleNetwork(t), features, args.boards, f"{args.blocks}x{args.channels}: {label}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Generates tests/golden/planes_real_games.npz: the 32 feature planes of `features::V1` (src/libdg_go/utils/features.rs:99-251) as
the INDEPENDENT restatement oracle/features.py computes them -- flood fills, simulated moves, whole-position super-ko -- on every
5th position of the three committed reference games (tests/golden/real_games.json; src/dg_tests/tests/real_games.rs), replayed the
way the reference's harness replays them (common/mod.rs:37-77), for both colours to move and all eight symmetries.

    python tests/golden/make_planes.py            (about five minutes of pure Python; needs nothing but numpy)

Per entry: `masks` [361] uint32 (bit c of word 19 y + x set where plane c is non-zero at that point of the TRANSFORMED tensor),
`komi` fp16 bits of the colour-plane constant, and (`game`, `stones`, `color`, `symmetry`) saying which position it is.  The two
product statements are compared with these words: csrc/go.cpp features_v1 on the CPU (tests/test_features_oracle.py) and
csrc/features.hip board_records_kernel on the device (tests/test_features_gpu.py).  `--wrong-order A,B` swaps two planes: what a
misreading of the plane order looks like (the tests show that such a fixture FAILS them)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import features as F  # noqa: E402

EVERY = 5


def entries(every=EVERY, swap=None, limit=None):
    games = json.load(open(os.path.join(HERE, "real_games.json")))["games"]
    for g, game in enumerate(games):
        for n, (board, _, _) in enumerate(F.replay(game["moves"])):
            stones = n + 1
            if stones % every != 0 or (limit is not None and stones > limit):
                continue
            for color in (F.BLACK, F.WHITE):
                mask, komi = F.plane_masks(board, color)
                if swap is not None:
                    a, b = swap
                    bit_a, bit_b = (mask >> a) & 1, (mask >> b) & 1
                    mask = (mask & ~np.uint32((1 << a) | (1 << b))) | (bit_a << b).astype(np.uint32) | (bit_b << a).astype(np.uint32)
                for symmetry in range(8):
                    yield g, stones, color, symmetry, F.apply_symmetry(mask, symmetry), komi
        assert F.position_text(board) == game["final_position"], game["name"]      # (the replay itself is pinned by the reference's hash)


def main():
    swap = None
    if "--wrong-order" in sys.argv:
        swap = tuple(int(v) for v in sys.argv[sys.argv.index("--wrong-order") + 1].split(","))
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(HERE, "planes_real_games.npz")
    rows = list(entries(swap=swap))
    np.savez_compressed(
        out,
        game=np.array([r[0] for r in rows], dtype=np.int16), stones=np.array([r[1] for r in rows], dtype=np.int16),
        color=np.array([r[2] for r in rows], dtype=np.int8), symmetry=np.array([r[3] for r in rows], dtype=np.int8),
        masks=np.stack([r[4] for r in rows]).astype(np.uint32), komi=np.array([r[5] for r in rows], dtype=np.uint16))
    masks = np.stack([r[4] for r in rows])
    print(f"{len(rows)} entries -> {out} ({os.path.getsize(out)} bytes); bits per plane:",
          [int(((masks >> c) & 1).sum()) for c in range(32)])


if __name__ == "__main__":
    main()

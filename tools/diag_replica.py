"""GPU diagnostic: one board replicated B times must give B identical outputs; report which tile rows deviate."""
import collections
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402

graft.package()
from dream_go_amd import api  # noqa: E402
api = api.load_full()          # this tool uses entry points of include/dgnn_test.h (raw batcher / single layers): the test build of the library


def run(batch, cin, cout, variant, seed):
    rng = np.random.default_rng(seed)
    board = np.maximum(rng.normal(0, 1, (1, 361, cin)), 0).astype(np.float16)
    x = np.repeat(board, batch, axis=0)
    w = rng.normal(0, np.sqrt(2.0 / (9 * cin)), (cout, 3, 3, cin)).astype(np.float16)
    b = rng.normal(0, 0.1, (cout,)).astype(np.float16)
    for k in ("DGNN_CONV", "DGNN_CONV_FM"):
        os.environ.pop(k, None)
    if variant == "v1":
        os.environ["DGNN_CONV"] = "v1"
        bm = 128
    else:
        os.environ["DGNN_CONV_FM"] = variant
        bm = 32 * int(variant)
    y = api.layer_conv3x3(x, batch, 19, cin, w, b, cout, relu=True).reshape(batch, 361, cout)
    # majority vote per (point, cout)
    vals = y.astype(np.int32)
    med = np.median(vals, axis=0).astype(np.int32)
    bad = np.argwhere(vals != med[None])
    print(f"{variant:3s} B={batch} {cin}->{cout} seed {seed}: {len(bad)} deviating outputs of {y.size}", flush=True)
    if len(bad):
        flat = bad[:, 0] * 361 + bad[:, 1]
        rows = flat % bm
        print("    row-in-tile histogram (row: count):", sorted(collections.Counter(rows.tolist()).items())[:40])
        print("    row mod 16:", sorted(collections.Counter((rows % 16).tolist()).items()))
        print("    fm index  :", sorted(collections.Counter(((rows % (bm // 2)) // 16).tolist()).items()))
        print("    cout mod 64 // 16 (fn):", sorted(collections.Counter(((bad[:, 2] % 64) // 16).tolist()).items()),
              " cout mod 4:", sorted(collections.Counter((bad[:, 2] % 4).tolist()).items()))
        print("    distinct (point, cout) pairs:", len({(p, c) for _, p, c in bad.tolist()}), " boards:", len(np.unique(bad[:, 0])))
        d = vals[tuple(bad.T)] - med[bad[:, 1], bad[:, 2]]
        print("    deviation in fp16 ulps:", sorted(collections.Counter(d.tolist()).items()))


if __name__ == "__main__":
    for variant in ("v1", "6", "8"):
        for seed in (1, 2):
            run(256, 256, 256, variant, seed)

"""GPU diagnostic: does a board's convolution result depend on where the board sits in the batch?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402

graft.package()
from dream_go_amd import api  # noqa: E402
api = api.load_full()          # this tool uses entry points of include/dgnn_test.h (raw batcher / single layers): the test build of the library


def run(batch, cin, cout, variant, alpha):
    rng = np.random.default_rng(batch)
    x = np.maximum(rng.normal(0, 1, (batch, 361, cin)), 0).astype(np.float16)
    w = rng.normal(0, np.sqrt(2.0 / (9 * cin)), (cout, 3, 3, cin)).astype(np.float16)
    b = rng.normal(0, 0.1, (cout,)).astype(np.float16)
    z = np.maximum(rng.normal(0, 1, (batch, 361, cout)), 0).astype(np.float16)
    perm = rng.permutation(batch)
    for k in ("DGNN_CONV", "DGNN_CONV_FM"):
        os.environ.pop(k, None)
    if variant == "v1":
        os.environ["DGNN_CONV"] = "v1"
    else:
        os.environ["DGNN_CONV_FM"] = variant
    a1, a2 = (alpha, 1 - alpha) if alpha else (1.0, 0.0)
    kw = dict(alpha1=a1, alpha2=a2, relu=True)
    y0 = api.layer_conv3x3(x, batch, 19, cin, w, b, cout, z=z if alpha else None, **kw).reshape(batch, 361, cout)
    y1 = api.layer_conv3x3(x[perm], batch, 19, cin, w, b, cout, z=z[perm] if alpha else None, **kw).reshape(batch, 361, cout)
    bad = np.argwhere(y1 != y0[perm])
    msg = f"{variant:3s} B={batch} {cin}->{cout} alpha={alpha}: {len(bad)} of {y0.size} differ"
    if len(bad):
        bm = 32 * int(variant) if variant != "v1" else 128
        flat1 = bad[:, 0] * 361 + bad[:, 1]                      # position index in the permuted run
        flat0 = perm[bad[:, 0]] * 361 + bad[:, 1]                # same board position in the original run
        msg += (f"; boards {len(np.unique(bad[:, 0]))}, points {np.unique(bad[:, 1])[:10]}, couts {np.unique(bad[:, 2])[:10]}"
                f"; row-in-tile run1 {np.unique(flat1 % bm)[:12]} run0 {np.unique(flat0 % bm)[:12]}"
                f"; sample {y1.view(np.float16)[tuple(bad[0])]} vs {y0.view(np.float16)[perm][tuple(bad[0])]}")
    print(msg, flush=True)


if __name__ == "__main__":
    for variant in ("v1", "4", "6", "8"):
        for alpha in (None, 0.5):
            run(64, 256, 256, variant, alpha)

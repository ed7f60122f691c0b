"""GPU diagnostic: change ONLY board k+1's data (positions fixed); which outputs of the other boards change?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402

graft.package()
from dream_go_amd import api  # noqa: E402
api = api.load_full()          # this tool uses entry points of include/dgnn_test.h (raw batcher / single layers): the test build of the library


def setv(variant):
    for k in ("DGNN_CONV", "DGNN_CONV_FM"):
        os.environ.pop(k, None)
    if variant == "v1":
        os.environ["DGNN_CONV"] = "v1"
    else:
        os.environ["DGNN_CONV_FM"] = variant


def main():
    batch, cin, cout = 128, 256, 256
    rng = np.random.default_rng(3)
    x = np.maximum(rng.normal(0, 1, (batch, 361, cin)), 0).astype(np.float16)
    w = rng.normal(0, np.sqrt(2.0 / (9 * cin)), (cout, 3, 3, cin)).astype(np.float16)
    b = rng.normal(0, 0.1, (cout,)).astype(np.float16)
    x2 = x.copy()
    odd = np.arange(1, batch, 2)
    x2[odd] = np.maximum(rng.normal(0, 1, (len(odd), 361, cin)), 0).astype(np.float16)   # replace every odd board
    x3 = x.copy()
    x3[odd] = 0                                                                              # ... or zero it
    even = np.arange(0, batch, 2)
    for variant in ("v1", "4", "6", "8"):
        setv(variant)
        bm = 128 if variant == "v1" else 32 * int(variant)
        y = api.layer_conv3x3(x, batch, 19, cin, w, b, cout, relu=True).reshape(batch, 361, cout)
        for name, xx in (("random neighbours", x2), ("zero neighbours", x3)):
            yy = api.layer_conv3x3(xx, batch, 19, cin, w, b, cout, relu=True).reshape(batch, 361, cout)
            bad = np.argwhere(yy[even] != y[even])
            print(f"{variant:3s} {name}: {len(bad)} outputs of unchanged boards changed", flush=True)
            if len(bad):
                board = even[bad[:, 0]]
                flat = board * 361 + bad[:, 1]
                # distance (in positions) to the nearest changed board
                dist = np.minimum(bad[:, 1] + 1, 361 - bad[:, 1])
                same_tile = [(int(f // bm) == int((f - d) // bm)) or (int(f // bm) == int((f + d) // bm)) for f, d in zip(flat, dist)]
                print("     points", bad[:12, 1].tolist(), "dist-to-changed-board", dist[:12].tolist(),
                      "shares a tile with changed rows:", same_tile[:12], flush=True)


if __name__ == "__main__":
    main()

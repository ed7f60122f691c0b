"""GPU diagnostic: same real-valued input, repeated runs -> bitwise identical?  Also board-shift dependence."""
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
    for variant in ("v1", "6", "8"):
        setv(variant)
        runs = [api.layer_conv3x3(x, batch, 19, cin, w, b, cout, relu=True).reshape(batch, 361, cout) for _ in range(4)]
        same = [int(np.count_nonzero(runs[0] != r)) for r in runs[1:]]
        # shift by one board: prepend a different board
        extra = np.maximum(rng.normal(0, 1, (1, 361, cin)), 0).astype(np.float16)
        xs = np.concatenate([extra, x[:-1]], axis=0)
        ys = api.layer_conv3x3(xs, batch, 19, cin, w, b, cout, relu=True).reshape(batch, 361, cout)
        shift = np.argwhere(ys[1:] != runs[0][:-1])
        # reverse board order
        yr = api.layer_conv3x3(x[::-1].copy(), batch, 19, cin, w, b, cout, relu=True).reshape(batch, 361, cout)
        rev = np.argwhere(yr[::-1] != runs[0])
        print(f"{variant:3s}: repeat-run differences {same}; shift-by-one-board differences {len(shift)}; reversed-order differences {len(rev)}", flush=True)
        for name, bad in (("shift", shift), ("rev", rev)):
            if len(bad):
                print(f"     {name}: (board, point, cout) {bad[:8].tolist()}", flush=True)


if __name__ == "__main__":
    main()

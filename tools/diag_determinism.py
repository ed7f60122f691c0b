"""GPU diagnostic: is the tile kernel deterministic and exact under full load?  Integer-valued data make every
product and sum exact, so the generic kernel is a bit-exact reference for the tile kernel at any size."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402

graft.package()
from dream_go_amd import api  # noqa: E402
api = api.load_full()          # this tool uses entry points of include/dgnn_test.h (raw batcher / single layers): the test build of the library


def run(batch, cin, cout, fm, seed):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 4, (batch, 19, 19, cin)).astype(np.float16)
    w = (rng.integers(-3, 4, (cout, 3, 3, cin)) / 16.0).astype(np.float16)
    b = (rng.integers(-8, 9, (cout,)) / 8.0).astype(np.float16)
    z = rng.integers(0, 4, (batch, 19, 19, cout)).astype(np.float16)
    os.environ["DGNN_CONV"] = "v1"
    want = api.layer_conv3x3(x, batch, 19, cin, w, b, cout, alpha1=0.5, alpha2=0.5, z=z, relu=False)
    del os.environ["DGNN_CONV"]
    os.environ["DGNN_CONV_FM"] = str(fm)
    outs = [api.layer_conv3x3(x, batch, 19, cin, w, b, cout, alpha1=0.5, alpha2=0.5, z=z, relu=False) for _ in range(3)]
    for i, got in enumerate(outs):
        bad = np.flatnonzero(got != want)
        msg = f"B={batch} {cin}->{cout} fm={fm} run {i}: {bad.size} mismatches of {got.size}"
        if bad.size:
            pos, co = bad // cout, bad % cout
            bm = 32 * fm
            msg += (f"; positions {np.unique(pos)[:12]} tiles {np.unique(pos // bm)[:12]} rows-in-tile {np.unique(pos % bm)[:16]}"
                    f" couts {np.unique(co)[:16]}; sample got/want "
                    f"{got.view(np.float16)[bad[:4]]} / {want.view(np.float16)[bad[:4]]}")
        print(msg, flush=True)
    print("   runs identical:", all(np.array_equal(outs[0], o) for o in outs[1:]), flush=True)


if __name__ == "__main__":
    for batch in (16, 64, 256):
        for fm in (4, 6, 8):
            run(batch, 256, 256, fm, seed=batch * 10 + fm)

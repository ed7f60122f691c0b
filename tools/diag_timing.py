"""Prints where the workgroups of the residual convolution spend their cycles (DGNN_CONV_TIMING=1 instrumentation; board kernel).

    [DGNN_CONV_ABLATE=1|2|3|4|8] [DGNN_BOARD_SPLIT=4] python tools/diag_timing.py [batch]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DGNN_CONV_TIMING"] = "1"
os.environ.setdefault("DGNN_CONV", "board")
os.environ.setdefault("DGNN_CONV_FM", "6")
import __graft_entry__ as graft  # noqa: E402

graft.package()
from dream_go_amd import api  # noqa: E402
api = api.load_full()          # this tool uses entry points of include/dgnn_test.h (raw batcher / single layers): the test build of the library

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
c = 256
rng = np.random.default_rng(1)
x = np.maximum(rng.normal(0, 1, (batch, 19, 19, c)), 0).astype(np.float16)
w = rng.normal(0, np.sqrt(2.0 / (9 * c)), (c, 3, 3, c)).astype(np.float16)
b = rng.normal(0, 0.1, (c,)).astype(np.float16)
for alpha in (None, 0.5):
    z = x if alpha is not None else None
    a1, a2 = (alpha, 1 - alpha) if alpha is not None else (1.0, 0.0)
    print(f"skip={'yes' if alpha is not None else 'no'}", flush=True)
    api.layer_conv3x3(x, batch, 19, c, w, b, c, alpha1=a1, alpha2=a2, z=z, relu=True)

#!/usr/bin/env python
"""One residual convolution (256 channels, batch 8), with and without a skip input, through the per-layer board kernel with eight
waves (DGNN_BOARD_NS=8) and with the wide split (DGNN_BOARD_NS=16: four waves x two channel tiles, MFMAs from inline asm): the
outputs must agree bit for bit; differing elements are listed with their tile coordinates.   python tools/layer_cmp_wide.py"""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def child(ns, path):
    os.environ["DGNN_BOARD_NS"] = ns
    os.environ["DGNN_CONV"] = "board"
    import __graft_entry__ as graft
    graft.package()
    from dream_go_amd import api
    rng = np.random.default_rng(3)
    batch, c = 8, 256
    out = {}
    for case, alpha in (("noskip", None), ("skip", 0.5)):
        x = np.maximum(rng.normal(0, 1, (batch, 19, 19, c)), 0).astype(np.float16)
        w = rng.normal(0, np.sqrt(2.0 / (9 * c)), (c, 3, 3, c)).astype(np.float16)
        b = rng.normal(0, 0.1, (c,)).astype(np.float16)
        z = np.maximum(rng.normal(0, 1, (batch, 19, 19, c)), 0).astype(np.float16) if alpha is not None else None
        a1, a2 = (alpha, 1 - alpha) if alpha is not None else (1.0, 0.0)
        y = api.layer_conv3x3(x, batch, 19, c, w, b, c, alpha1=a1, alpha2=a2, z=z, relu=True)
        out[case] = np.asarray(y).view(np.uint16).reshape(batch, 361, c).copy()
        if z is not None: out["z"] = z.reshape(batch, 361, c).view(np.uint16).copy()
    np.savez(path, **out)
if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1], sys.argv[2]); sys.exit(0)
    for ns in ("8", "16"):
        subprocess.run([sys.executable, os.path.abspath(__file__), ns, f"/tmp/lc_{ns}.npz"], check=True, stderr=subprocess.DEVNULL)
    a, b = np.load("/tmp/lc_8.npz"), np.load("/tmp/lc_16.npz")
    for case in ("noskip", "skip"):
        d = np.argwhere(a[case] != b[case])
        print(case, "differing elements:", len(d), "of", a[case].size)
        if len(d):
            print("  boards", np.unique(d[:, 0]).tolist()[:10], "channels", np.unique(d[:, 2]).tolist()[:40])
            pos = np.unique(d[:, 1]); rows = (pos // 19) * 20 + pos % 19
            print("  row tiles", np.unique(rows // 32).tolist(), "n positions", len(pos))
            for (bb, pp, cc) in d[:12]:
                ya = float(a[case][bb, pp, cc:cc+1].view(np.float16)[0]); yb = float(b[case][bb, pp, cc:cc+1].view(np.float16)[0])
                zz = float(a["z"][bb, pp, cc:cc+1].view(np.float16)[0]) if case == "skip" else 0.0
                zn = [float(a["z"][bb, pp, k:k+1].view(np.float16)[0]) for k in (cc+1, cc+2, cc+3, cc+8, cc+32)] if case == "skip" else []
                print("   ", bb, pp, cc, "shipped %.4f wide %.4f diff %.4f  z*0.5 = %.4f  neighbours z*0.5: %s" % (ya, yb, ya - yb, 0.5 * zz, [round(0.5 * v, 4) for v in zn]))

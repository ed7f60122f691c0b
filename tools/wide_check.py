#!/usr/bin/env python
"""The wide work split of the tower (default at 256 channels: four waves x two channel tiles, MFMAs from inline asm) against the
eight-wave one (DGNN_TOWER_WIDE=0): the same weights and inputs in two processes, outputs compared bit for bit (the summation order per accumulator is the same).
    python tools/wide_check.py            # parent: runs itself twice and compares
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(path):
    import __graft_entry__ as graft
    graft.package()
    from dream_go_amd import api, synth
    out = {}
    # 256 channels: whole boards (batch > 128: Split<16> / Split<8>) and boards split in two (batch 65..128: Split<20> / Split<4>);
    # 128 channels: whole boards (batch > 128: Split<20> / Split<4>)
    for blocks, channels, batch in ((20, 256, 256), (3, 256, 200), (20, 256, 131), (20, 256, 128), (4, 256, 77), (9, 128, 256), (9, 128, 150)):
        net = api.Network.from_json(synth.to_json(synth.make_tensors(blocks, channels, seed=5 + blocks)))
        with net.get_workspace(batch) as ws:
            for rep in range(3):
                features = synth.make_features(batch, seed=100 * blocks + channels + rep)
                value, policy = api.forward(ws, features).unwrap()
                out[f"v{blocks}x{channels}_{batch}_{rep}"] = value.view(np.uint16).copy()
                out[f"p{blocks}x{channels}_{batch}_{rep}"] = policy.view(np.uint16).copy()
        net.destroy()
    np.savez(path, **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    paths = []
    for wide in ("0", "1"):
        path = f"/tmp/wide_check_{wide}.npz"
        env = dict(os.environ, DGNN_TOWER_WIDE=wide)
        subprocess.run([sys.executable, os.path.abspath(__file__), path], check=True, env=env, stderr=subprocess.DEVNULL)
        paths.append(path)
    a, b = np.load(paths[0]), np.load(paths[1])
    bad = 0
    for key in a.files:
        diff = int(np.count_nonzero(a[key] != b[key]))
        bad += diff
        if diff:
            if key.startswith("p"):
                boards = np.unique(np.nonzero(a[key].reshape(-1, 362) != b[key].reshape(-1, 362))[0])
            else:
                boards = np.nonzero(a[key] != b[key])[0]
            print(f"{key}: {diff} of {a[key].size} words differ, boards {boards.tolist()}")
    print("wide == shipped, bit for bit" if bad == 0 else f"MISMATCH: {bad} words")
    sys.exit(1 if bad else 0)

#!/usr/bin/env python
"""Does splitting a 256-board batch into independent halves on two HIP streams beat one launch per layer?  Two workspaces
(own streams) with 128 boards each, passes enqueued alternately, against one workspace with 256.  Prints evals/s for both.
(A diagnostic for DESIGN.md section 4.3: 256 workgroups that start and finish every layer together put their prologue,
their HBM write burst and their epilogue at the same instants.)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.package()
from dream_go_amd import api, synth  # noqa: E402


def rate(workspaces, boards, steps):
    for ws in workspaces:
        ws.forward_resident()
    for ws in workspaces:
        ws.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for ws in workspaces:
            ws.forward_resident()
    for ws in workspaces:
        ws.sync()
    return boards * steps / (time.perf_counter() - t0)


def main():
    net = api.Network.from_json(synth.to_json(synth.make_tensors(20, 256, seed=1)))
    for parts in (1, 2, 4):
        per = 256 // parts
        workspaces = []
        for i in range(parts):
            ws = net.get_workspace(per).__enter__()
            ws.upload(synth.make_features(per, seed=10 + i))
            workspaces.append(ws)
        best = max(rate(workspaces, 256, 30) for _ in range(3))
        print(f"{parts} stream(s) x {per} boards: {best:.0f} evals/s")
        for ws in workspaces:
            ws.release()


if __name__ == "__main__":
    main()

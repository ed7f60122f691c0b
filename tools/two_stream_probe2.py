#!/usr/bin/env python
"""Two 256-board passes in flight at once (two workspaces = two streams) against the same passes one after the other.
    python tools/two_stream_probe2.py [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
graft.package()
from dream_go_amd import api, synth

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
net = api.Network.from_json(synth.to_json(synth.make_tensors(20, 256, seed=1)))
f = synth.make_features(batch, seed=3)
a = net.get_workspace(batch).__enter__()
b = net.get_workspace(batch).__enter__()
for w in (a, b):
    w.upload(f)
    w.forward_resident(); w.sync()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    a.forward_resident(); a.sync()
    b.forward_resident(); b.sync()
t_serial = time.perf_counter() - t0
t0 = time.perf_counter()
for _ in range(n):
    a.forward_resident(); b.forward_resident()
    a.sync(); b.sync()
t_both = time.perf_counter() - t0
t0 = time.perf_counter()
a.forward_resident()
for _ in range(n):
    b.forward_resident(); a.sync(); a.forward_resident(); b.sync()
a.sync()
t_pipe = time.perf_counter() - t0
print(f"batch {batch}: one after the other {2 * n * batch / t_serial:.0f} evals/s; both enqueued, then both awaited {2 * n * batch / t_both:.0f}; "
      f"always one queued behind the running one {(2 * n + 1) * batch / t_pipe:.0f}")

# the same through host buffers: packed features in, fp16 outputs out (dgnn_forward_submit_packed / dgnn_forward_wait)
import ctypes
import numpy as np
u32p, u16p = ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint16)
packed = np.ascontiguousarray(api.features_pack(f, batch), dtype=np.uint32).reshape(-1)
pair = (a, b)
vals = [np.empty(batch, dtype=np.uint16) for _ in pair]
pols = [np.empty(batch * 362, dtype=np.uint16) for _ in pair]
submit = lambda i: api.lib.dgnn_forward_submit_packed(pair[i].handle, packed.ctypes.data_as(u32p), batch)
wait = lambda i: api.lib.dgnn_forward_wait(pair[i].handle, vals[i].ctypes.data_as(u16p), pols[i].ctypes.data_as(u16p))
assert submit(0) == 0 and wait(0) == 0 and submit(1) == 0 and wait(1) == 0
t_sub, t_wait = [], []
t0 = time.perf_counter()
submit(0)
for c in range(1, 2 * n):
    t1 = time.perf_counter(); submit(c & 1); t2 = time.perf_counter(); wait((c - 1) & 1); t3 = time.perf_counter()
    t_sub.append(t2 - t1); t_wait.append(t3 - t2)
wait((2 * n - 1) & 1)
dt = time.perf_counter() - t0
print(f"packed host buffers, two workspaces in turn: {2 * n * batch / dt:.0f} evals/s; submit call {1e3 * np.mean(t_sub):.3f} ms (max {1e3 * np.max(t_sub):.3f}), wait call {1e3 * np.mean(t_wait):.3f} ms")
t0 = time.perf_counter()
for c in range(2 * n):
    submit(0); wait(0)
dt = time.perf_counter() - t0
print(f"packed host buffers, one workspace, call after call: {2 * n * batch / dt:.0f} evals/s")

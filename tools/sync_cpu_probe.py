"""How much CPU does one thread burn waiting for the device?  N synchronous passes of 20x256 batch 256 (3.3 ms each) from one
thread: process CPU time / wall time.  ~1.0 = the wait spins, ~0.05 = it sleeps.   python tools/sync_cpu_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.package()
from dream_go_amd import api, synth
net = api.Network.from_json(synth.to_json(synth.make_tensors(20, 256, seed=1)))
f = synth.make_features(256)
packed = api.features_pack(f, 256)
with net.get_workspace(256) as ws:
    api.forward_packed(ws, packed, 256)
    c0, w0 = time.process_time(), time.perf_counter()
    for _ in range(100):
        api.forward_packed(ws, packed, 256)
    c, w = time.process_time() - c0, time.perf_counter() - w0
print(f"100 passes: wall {w:.3f} s, cpu {c:.3f} s, cpu/wall {c / w:.2f}")

"""Where the workgroups of the whole-board tower (tower_board_kernel) spend their cycles, layer by layer: DGNN_TOWER_TIMING=<file>
makes every layer write wave-level s_memtime stamps (start, K loop, epilogue, end); this prints the means over workgroups and waves.

    python tools/tower_trace.py [batch] [blocks] [channels]
"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PATH = "/tmp/dgnn_tower_timing.bin"
os.environ["DGNN_TOWER_TIMING"] = PATH
import __graft_entry__ as graft  # noqa: E402

graft.package()
from dream_go_amd import api, synth  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 20
channels = int(sys.argv[3]) if len(sys.argv) > 3 else 256
net = api.Network.from_json(synth.to_json(synth.make_tensors(blocks, channels, seed=5)))
features = synth.make_features(batch, seed=6)
with net.get_workspace(batch) as ws:
    print("kernel:", ws.conv_kernel())
    for _ in range(4):
        api.forward(ws, features).unwrap()
raw = open(PATH, "rb").read()
layers, wgs, waves, _ = struct.unpack("4i", raw[:16])
t = np.frombuffer(raw[16:], dtype=np.uint64).reshape(layers, wgs, waves, 4).astype(np.int64)
live = t[..., 3] > 0                                       # variants with fewer than 8 waves leave slots empty
t = np.where(live[..., None], t, 0)
n = live.sum(axis=(1, 2)).clip(min=1)
pro = ((t[..., 1] - t[..., 0]) * live).sum(axis=(1, 2)) / n
loop = ((t[..., 2] - t[..., 1]) * live).sum(axis=(1, 2)) / n
epi = ((t[..., 3] - t[..., 2]) * live).sum(axis=(1, 2)) / n
# gap: from a layer's end to the next layer's start, per wave
gap = np.zeros(layers)
for l in range(1, layers):
    gap[l] = (((t[l, ..., 0] - t[l - 1, ..., 3]) * live[l]).sum()) / n[l]
total = np.zeros(layers)
for l in range(layers):
    total[l] = pro[l] + loop[l] + epi[l] + gap[l]
print(f"{layers} layers x {wgs} workgroups; cycles per wave (s_memtime ticks)")
print("layer   gap  prologue   K loop  per k-step  epilogue   total")
for l in range(layers):
    ksteps = 9 * (1 if l == 0 and layers % 2 == 1 else channels // 32)
    print(f"{l:5d} {gap[l]:6.0f} {pro[l]:8.0f} {loop[l]:9.0f} {loop[l] / ksteps:9.0f} {epi[l]:9.0f} {total[l]:9.0f}")
first, last = t[0, ..., 0][live[0]].min(), t[-1, ..., 3][live[-1]].max()
print(f"means over the residual layers: gap {gap[1:].mean():.0f}  prologue {pro[1:].mean():.0f}  K loop {loop[1:].mean():.0f}  epilogue {epi[1:].mean():.0f}"
      f"  total {total[1:].mean():.0f}; launch {last - first} ticks first start -> last end")

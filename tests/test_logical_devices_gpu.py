"""The SINGLE-PROCESS multi-device code with eight lanes on one GPU (no 8-GPU node has ever been available).

The reference drives every device from one process (src/libdg_mcts/predictors/nn.rs:64-67,87-89: Device::all(), `2 x num_devices`
batches alive, weights copied to each device on first use, tensor.rs:123-141).  The library's forms of that -- dgnn_pool_create_devices,
dgnn_broadcast_weights_devices, per-device weight images / workspaces / chains -- had never run with more than one device.  The TEST
build maps LOGICAL devices onto the GPUs present (DGNN_FAKE_DEVICES=8: logical d -> physical d % count, csrc/kernels.h) and stands in
for the one grouped ncclBroadcast, which needs distinct GPUs, by device-to-device copies; everything else is the code an 8-GPU node runs:

  * eight per-device weight images reserved, filled by the "broadcast", committed -- each checksums to the host image;
  * one leaf pool over eight lanes (the reference's form): every lane evaluates batches, every answer is the bits of the same
    position evaluated directly on lane 0;
  * eight pools, one per lane, each with its own transposition table, games `id mod 8` (the partitioning of configs[3])
    played concurrently from eight threads: every game id played exactly once, the same records as the same games played on lane 0 alone;
  * a weight refresh across all eight lanes under pool traffic: pause -> broadcast -> resume, afterwards every lane answers from the
    new weights.
configs[3] itself (512 games over 8 MI355X) stays UNMEASURED: this proves the control flow, not a scaling curve.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import json, os, sys, threading, time
sys.path.insert(0, %(root)r)
import __graft_entry__ as graft
graft.package()
import numpy as np
from dream_go_amd import api as shipped, synth
api = shipped.load_full()
report = {"devices": api.Device.len(), "torch_free": True}
N = report["devices"]
text_a, text_b = synth.to_json(synth.make_tensors(4, 128, seed=8100)), synth.to_json(synth.make_tensors(4, 128, seed=8101))
net = api.Network.from_json(text_a)

# ---- the "broadcast": device 0 uploads over PCIe, the other seven receive device to device; every copy checksums to the host image
net.broadcast_weights_devices(num_devices=N)
report["checksums_agree"] = len({net.weight_checksum_device(d) for d in range(N)} | {net.weight_checksum()}) == 1

# ---- one pool over eight lanes (nn.rs:84-108): answers bit-equal to lane 0's own, every lane used
feats = np.ascontiguousarray(synth.make_features(256, seed=8102)).reshape(256, -1)
with net.get_workspace(256, device=0) as ws:
    v, p = api.forward(ws, feats.reshape(-1)).unwrap()
want_v, want_p = v.view(np.uint16).copy(), p.view(np.uint16).reshape(256, 362).copy()
with net.get_workspace(16, device=N - 1) as ws:                 # a workspace on the LAST lane, directly
    v, p = api.forward(ws, feats[:16].reshape(-1)).unwrap()
    report["last_lane_bit_equal"] = bool(np.array_equal(v.view(np.uint16), want_v[:16]) and np.array_equal(p.view(np.uint16).reshape(16, 362), want_p[:16]))
    report["last_lane_device"] = ws.device
pool = api.LeafPool.over_devices(net, None, max_batch_size=16, batches_per_device=2, workers_per_device=2)
wrong = []
def caller(t):
    rng = np.random.default_rng(8200 + t)
    for _ in range(40):
        rows = rng.integers(0, 256, size=8)
        got = pool.predict(feats[rows].reshape(-1), 8)
        for r, g in zip(rows, got):
            if int(np.asarray(g._value).view(np.uint16)) != int(want_v[r]) or not np.array_equal(g._policy.view(np.uint16), want_p[r]):
                wrong.append(int(r))
threads = [threading.Thread(target=caller, args=(t,)) for t in range(32)]
[t.start() for t in threads]; [t.join() for t in threads]
st = pool.stats()
report["one_pool"] = {"lanes": st["lanes"], "lane_boards": st["lane_boards"][:N], "lane_batches": st["lane_batches"][:N], "wrong": len(wrong), "boards": st["boards"]}
pool.close()

# ---- eight pools, games id mod 8, each lane its own table; against the same games on lane 0 alone
def play(lane, first, stride, count):
    pool = api.LeafPool.over_devices(net, [lane], max_batch_size=64, batches_per_device=2, workers_per_device=2, boards=True)
    out = api.selfplay_go(count, 6, 24, threads_per_search=1, seed=77, pool=pool, cache=None, return_moves=True, first_game=first, game_stride=stride,
                          cache_per_game=5000, leaves_per_thread=4, host_threads=2)
    pool.close()
    return out
games = 2 * N
alone = play(0, 0, 1, games)["record"]
sharded, lock = {}, threading.Lock()
def shard(lane):
    out = play(lane, lane, N, games // N)
    with lock:
        for i in range(games // N):
            sharded[lane + i * N] = out["record"][i]
threads = [threading.Thread(target=shard, args=(d,)) for d in range(N)]
[t.start() for t in threads]; [t.join() for t in threads]
report["sharded_games"] = {"ids": sorted(sharded), "same_records": bool(all(np.array_equal(sharded[g], alone[g]) for g in range(games)))}

# ---- refresh across the lanes under traffic: pause -> new host image -> broadcast -> resume
net_b = api.Network.from_json(text_b)
with net_b.get_workspace(8, device=0) as ws:
    v, p = api.forward(ws, feats[:8].reshape(-1)).unwrap()
new_v, new_p = v.view(np.uint16).copy(), p.view(np.uint16).reshape(8, 362).copy()
image_b = net_b.weight_image_host().copy()
pool = api.LeafPool.over_devices(net, None, max_batch_size=16, batches_per_device=2, workers_per_device=2)
stop, after, t_resumed = threading.Event(), [], [None]
def traffic():
    while not stop.is_set():
        t = time.perf_counter()
        got = pool.predict(feats[:8].reshape(-1), 8)
        after.append((t, all(int(np.asarray(g._value).view(np.uint16)) == int(new_v[i]) and np.array_equal(g._policy.view(np.uint16), new_p[i]) for i, g in enumerate(got)),
                      all(int(np.asarray(g._value).view(np.uint16)) == int(want_v[i]) and np.array_equal(g._policy.view(np.uint16), want_p[i]) for i, g in enumerate(got))))
threads = [threading.Thread(target=traffic) for _ in range(16)]
[t.start() for t in threads]
time.sleep(0.3)
pool.pause()
net.set_weight_image(0, image_b.ctypes.data, image_b.size, src_on_device=False)     # the root's copy and the host image
net.broadcast_weights_devices(num_devices=N)                                          # ... and the other seven
pool.resume()
t_resumed[0] = time.perf_counter()
time.sleep(0.3)
stop.set()
[t.join() for t in threads]
st = pool.stats()
pool.close()
late = [s for s in after if s[0] >= t_resumed[0]]
report["refresh"] = {"answers": len(after), "after_resume": len(late), "all_new_after_resume": bool(late) and all(s[1] for s in late),
                     "never_a_mixture": all(s[1] or s[2] for s in after), "old_before": any(s[2] for s in after),
                     "checksums_agree": len({net.weight_checksum_device(d) for d in range(N)} | {net_b.weight_checksum()}) == 1,
                     "lanes_used": sum(1 for b in st["lane_batches"][:N] if b > 0)}
net.destroy(); net_b.destroy()
print("REPORT " + json.dumps(report), flush=True)
"""


def test_eight_logical_devices_in_one_process():
    env = dict(os.environ, DGNN_FAKE_DEVICES="8", HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert run.returncode == 0, (run.stdout[-1500:], run.stderr[-3000:])
    report = json.loads([line for line in run.stdout.splitlines() if line.startswith("REPORT ")][-1][7:])
    print(report)
    assert report["devices"] == 8 and report["checksums_agree"]
    assert report["last_lane_bit_equal"] and report["last_lane_device"] == 7
    one = report["one_pool"]
    assert one["lanes"] == 8 and one["wrong"] == 0 and all(b > 0 for b in one["lane_batches"]) and sum(one["lane_boards"]) == one["boards"]
    assert report["sharded_games"]["ids"] == list(range(16)) and report["sharded_games"]["same_records"]
    r = report["refresh"]
    assert r["all_new_after_resume"] and r["never_a_mixture"] and r["old_before"] and r["checksums_agree"] and r["lanes_used"] == 8
    assert "gave up" not in run.stderr


def test_the_shipped_library_does_not_know_logical_devices():
    """DGNN_FAKE_DEVICES is read by the test build only: the shipped library reports the devices that are there."""
    code = ("import sys; sys.path.insert(0, %r)\nimport __graft_entry__ as g; g.package()\nfrom dream_go_amd import api\nimport torch\n"
            "print('COUNT', api.Device.len(), torch.cuda.device_count())\n" % ROOT)
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, DGNN_FAKE_DEVICES="8"), cwd=ROOT)
    assert run.returncode == 0, run.stderr[-2000:]
    shipped, real = [int(v) for v in [l for l in run.stdout.splitlines() if l.startswith("COUNT")][-1].split()[1:]]
    assert shipped == real

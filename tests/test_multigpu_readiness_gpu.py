"""What ONE GPU box can prove about the N > 1 run (no 8-GPU node has ever been available):

  * the device the library opens for a rank IS the device the launcher means: LOCAL_RANK, HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES
    re-numbering, torch's numbering and the library's agree (PCI bus ids), and the CPU pinning reads the same GPU's NUMA node;
  * `bench.py --gpus 1 --force-dist` executes the path of the N > 1 run with one rank: torch.distributed over RCCL ("nccl"), the
    unique id through it, dgnn_rccl_comm_create, dgnn_broadcast_weights, per-rank gathers, the one JSON line.
Model: one process per GPU, src/libdg_mcts/predictors/nn.rs:64-67,87-89 (the reference: one process, Device::all()).
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

PROBE = r"""
import json, os, sys
sys.path.insert(0, %r)
import __graft_entry__ as graft
graft.package()
import torch
from dream_go_amd import api, replicas, synth
rank = int(os.environ.get("LOCAL_RANK", "0"))
out = {"count": api.Device.len(), "torch_count": torch.cuda.device_count()}
if out["count"] > rank:
    out["bus"] = api.Device(rank).pci_bus_id().lower()
    props = torch.cuda.get_device_properties(rank)
    out["torch_bus"] = ("%%04x:%%02x:%%02x.0" %% (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)) if hasattr(props, "pci_bus_id") else None
    net = api.Network.from_json(synth.to_json(synth.make_tensors(1, 32, seed=3)))
    with net.get_workspace(2, device=rank) as ws:
        out["workspace_device"] = ws.device
        value, policy = api.forward(ws, synth.make_features(2, seed=4)).unwrap()
        out["policy_sum"] = float(policy.astype("float32").reshape(2, 362).sum(axis=1)[0])
    net.destroy()
    out["physical"] = replicas.visible_gpu_index(rank)
    out["numa"] = replicas.gpu_numa_node(rank)
print(json.dumps(out))
""" % ROOT


def probe(env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(env_extra)
    run = subprocess.run([sys.executable, "-c", PROBE], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert run.returncode == 0, run.stderr[-2000:]
    return json.loads(run.stdout.strip().splitlines()[-1])


def test_the_device_a_rank_opens_is_the_one_the_launcher_means():
    plain = probe({"LOCAL_RANK": "0"})
    assert plain["count"] >= 1 and plain["count"] == plain["torch_count"]
    assert plain["torch_bus"] is None or plain["bus"][-7:] == plain["torch_bus"][-7:]      # torch's device 0 and the library's are one GPU
    assert plain["workspace_device"] == 0 and abs(plain["policy_sum"] - 1.0) < 5e-3
    assert plain["physical"] == 0
    # the launcher re-numbers: physical GPU 0 as the only visible device, by either variable -- same GPU, still device 0 of the process
    for name in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        seen = probe({"LOCAL_RANK": "0", name: "0"})
        assert seen["count"] == 1 and seen["bus"] == plain["bus"] and seen["physical"] == 0 and seen["numa"] == plain["numa"]
    # a rank whose device is not visible must not silently open another one
    hidden = probe({"LOCAL_RANK": "1", "HIP_VISIBLE_DEVICES": "0"})
    assert hidden["count"] == 1 and "bus" not in hidden
    if plain["count"] >= 2:                                            # (only on a multi-GPU box) device 1 alone: it becomes device 0, another bus id
        other = probe({"LOCAL_RANK": "0", "HIP_VISIBLE_DEVICES": "1"})
        assert other["count"] == 1 and other["bus"] != plain["bus"] and other["physical"] == 1


def test_bench_force_dist_runs_the_multi_rank_path_with_one_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["MASTER_PORT"] = "29577"
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--lean", "--steps", "5", "--warmup", "2"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert run.returncode == 0, run.stderr[-2000:]
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and len(line["per_rank_evals_per_sec"]) == 1
    b = line["weight_broadcast"]
    assert b["ranks"] == 1 and b["checksums_agree"] is True and "rccl" in b["backend"] and b["bytes"] > 40_000_000      # the 20x256 image: 52 MB
    assert line["roofline"]["frac"] > 0.3

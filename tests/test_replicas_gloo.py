"""The N > 1 path on CPU: two processes, gloo backend.  Covers the sharding, the max-over-ranks clock, and the
weight-image broadcast + checksum agreement that bench.py performs with RCCL on the GPUs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    graft.package()
    from dream_go_amd import api, replicas, synth

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank starts from DIFFERENT weights of the same architecture; rank 0's must win
        net = api.Network.from_json(synth.to_json(synth.make_tensors(1, 32, num_samples=2, seed=10 + rank)))
        before = net.weight_checksum()
        agree_before = replicas.all_agree(dist, torch, before)
        image = torch.from_numpy(net.weight_image_host())          # view of the library's host image
        replicas.broadcast_image(dist, image, src=0)
        after = net.weight_checksum()
        agree_after = replicas.all_agree(dist, torch, after)
        assert replicas.fnv1a(net.weight_image_host()) == after    # python restatement of the checksum

        units = list(replicas.shard(11, world, rank))
        total = replicas.sum_over_ranks(dist, torch, len(units))
        slowest = replicas.max_over_ranks(dist, torch, 1.0 + rank)
        results[rank] = dict(before=before, after=after, agree_before=agree_before, agree_after=agree_after,
                             units=units, total=total, slowest=slowest,
                             seed=replicas.rank_seed(5, rank))
    finally:
        dist.destroy_process_group()


def test_two_rank_replicas_over_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    with ctx.Manager() as manager:
        results = manager.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, results)) for port in [_free_port()] for r in range(world)]
        [p.start() for p in procs]
        [p.join(180) for p in procs]
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        r0, r1 = results[0], results[1]
    assert r0["before"] != r1["before"] and not r0["agree_before"]
    assert r0["after"] == r0["before"] == r1["after"] and r0["agree_after"] and r1["agree_after"]
    assert r0["units"] + r1["units"] == list(range(11)) and r0["total"] == 11
    assert r0["slowest"] == r1["slowest"] == 2.0
    assert r0["seed"] != r1["seed"]


def test_shard_covers_everything_once():
    from dream_go_amd import replicas
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 8, 512, 1000003 % 4096):
            seen = [u for r in range(world) for u in replicas.shard(n, world, r)]
            assert seen == list(range(n))
            sizes = [len(replicas.shard(n, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert [replicas.owner_of_game(g, 8) for g in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]


def _selfplay_worker(rank, world, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    graft.package()
    from dream_go_amd import api, replicas

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = replicas.selfplay_sharded(api, dist, torch, num_games=5, num_moves=6, num_rollouts=12, threads_per_search=1, seed=21,
                                        return_moves=True)
        results[rank] = dict(ids=out["game_ids"], record=out["record"].tolist(), job=out["job"], evals=out["evals"])
    finally:
        dist.destroy_process_group()


def test_games_sharded_over_two_ranks_are_the_games_one_rank_plays():
    """SURVEY.md section 8e / BASELINE.json configs[3]: games are sharded `id mod world`; with one search thread and the seeded
    hash predictor a game is a function of (seed, game id), so two ranks together play exactly what one rank plays alone, and
    the job counters are the sums."""
    from dream_go_amd import api, replicas
    world = 2
    ctx = mp.get_context("spawn")
    with ctx.Manager() as manager:
        results = manager.dict()
        procs = [ctx.Process(target=_selfplay_worker, args=(r, world, port, results)) for port in [_free_port()] for r in range(world)]
        [p.start() for p in procs]
        [p.join(180) for p in procs]
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        r0, r1 = dict(results[0]), dict(results[1])
    alone = replicas.selfplay_sharded(api, None, torch, num_games=5, num_moves=6, num_rollouts=12, threads_per_search=1, seed=21,
                                      return_moves=True)
    assert r0["ids"] == [0, 2, 4] and r1["ids"] == [1, 3] and alone["game_ids"] == [0, 1, 2, 3, 4]
    merged = {g: rec for r in (r0, r1) for g, rec in zip(r["ids"], r["record"])}
    assert [merged[g] for g in range(5)] == alone["record"].tolist()
    assert r0["job"] == r1["job"] and r0["job"]["games"] == 5 and r0["job"]["moves"] == 30 and r0["job"]["world"] == 2
    assert r0["job"]["evals"] == r0["evals"] + r1["evals"] == alone["job"]["evals"]
    assert [replicas.games_of_rank(5, 8, r)[0] for r in range(8)] == [1, 1, 1, 1, 1, 0, 0, 0]


def test_pin_to_compact_cpus_takes_a_slice_per_rank(monkeypatch):
    """replicas.pin_to_compact_cpus: a process whose CPU-time quota is smaller than the CPUs it may run on is pinned to `quota`
    neighbouring CPUs, rank r of a node to the r-th slice; nothing happens when it is not oversubscribed or when switched off."""
    import os
    from dream_go_amd import replicas
    if not hasattr(os, "sched_setaffinity"):
        pytest.skip("no scheduler affinity on this platform")
    allowed = list(range(64))
    chosen = {}
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(allowed))
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: chosen.__setitem__("cpus", list(cpus)))
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert replicas.pin_to_compact_cpus(0, quota=16) == list(range(16)) and chosen["cpus"] == list(range(16))
    assert replicas.pin_to_compact_cpus(3, quota=16) == list(range(48, 64))
    assert replicas.pin_to_compact_cpus(5, quota=16) == list(range(16, 32))          # more ranks than slices: wrap around
    assert replicas.pin_to_compact_cpus(0, quota=64) is None                           # not oversubscribed
    # a single-rank job takes the idlest slice: here somebody else keeps CPUs 0..19 busy
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setattr(replicas, "_cpu_busy_fractions", lambda interval=0.05: {c: (0.9 if c < 20 else 0.0) for c in allowed})
    assert replicas.pin_to_compact_cpus(0, quota=16) == list(range(32, 48))
    monkeypatch.setenv("DGNN_PIN_CPUS", "0")
    assert replicas.pin_to_compact_cpus(0, quota=16) is None


def _fake_sysfs(root, gpu_nodes, cpus_per_node=64):
    """A /sys with a KFD topology of one CPU node per NUMA node followed by the GPUs (gpu_nodes[i] = NUMA node of GPU i), their
    render nodes and the nodes' cpulists -- the files replicas.gpu_numa_node / numa_node_cpus read."""
    numa = sorted(set(gpu_nodes))
    topo = os.path.join(root, "class", "kfd", "kfd", "topology", "nodes")
    for i, n in enumerate(numa):                                                     # CPU agents come first, no SIMDs
        os.makedirs(os.path.join(topo, str(i)))
        open(os.path.join(topo, str(i), "properties"), "w").write("cpu_cores_count 64\nsimd_count 0\ndrm_render_minor 0\n")
        os.makedirs(os.path.join(root, "devices", "system", "node", f"node{n}"))
        open(os.path.join(root, "devices", "system", "node", f"node{n}", "cpulist"), "w").write(f"{n * cpus_per_node}-{(n + 1) * cpus_per_node - 1}\n")
    for g, n in enumerate(gpu_nodes):
        d = os.path.join(topo, str(len(numa) + g))
        os.makedirs(d)
        open(os.path.join(d, "properties"), "w").write(f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {128 + g}\n")
        dev = os.path.join(root, "class", "drm", f"renderD{128 + g}", "device")
        os.makedirs(dev)
        open(os.path.join(dev, "numa_node"), "w").write(f"{n}\n")


def test_ranks_are_pinned_to_the_numa_node_of_their_gpu(monkeypatch, tmp_path):
    """a rank's CPUs come from the NUMA node its GPU hangs on (KFD topology -> render node -> numa_node), the ranks
    of one node take that node's slices in rank order, HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES re-numbering is honoured."""
    from dream_go_amd import replicas
    if not hasattr(os, "sched_setaffinity"):
        pytest.skip("no scheduler affinity on this platform")
    root = str(tmp_path)
    _fake_sysfs(root, [0, 0, 0, 0, 1, 1, 1, 1])                                      # an 8-GPU box: GPUs 0-3 on socket 0, 4-7 on socket 1
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(128)))
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: None)
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    for name in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(name, raising=False)
    assert [replicas.gpu_numa_node(r, root) for r in range(8)] == [0, 0, 0, 0, 1, 1, 1, 1]
    assert replicas.numa_node_cpus(1, root) == list(range(64, 128))
    # 16 CPUs of quota per rank: ranks 0-3 take socket 0's four slices, ranks 4-7 socket 1's
    got = [replicas.pin_to_compact_cpus(r, quota=16, sysfs_root=root) for r in range(8)]
    assert got[0] == list(range(0, 16)) and got[3] == list(range(48, 64))
    assert got[4] == list(range(64, 80)) and got[7] == list(range(112, 128))
    # the launcher hands every rank ONE visible device: rank 5 sees physical GPU 5 as its device 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "5")
    assert replicas.visible_gpu_index(0) == 5 and replicas.gpu_numa_node(0, root) == 1
    # ROCR re-numbers first, HIP on top of it: HIP device 1 -> ROCR entry 1 -> physical GPU 6
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "4,6")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1")
    assert replicas.visible_gpu_index(1) == 6 and replicas.gpu_numa_node(1, root) == 1
    assert replicas.visible_gpu_index(2) is None and replicas.gpu_numa_node(2, root) is None      # not a visible device
    # a node that lists SMT siblings as a second range ("0-63,128-191"): a rank's slice is one thread on each of 16 neighbouring cores
    open(os.path.join(root, "devices", "system", "node", "node0", "cpulist"), "w").write("0-63,128-191\n")
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)))
    assert replicas.numa_node_cpus(0, root) == list(range(64)) + list(range(128, 192))
    assert replicas.numa_node_cpus(0, root, primary_threads=16) == list(range(64))
    assert replicas.pin_to_compact_cpus(2, quota=16, sysfs_root=root) == list(range(32, 48))
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(128)))
    # no topology (this container): the numeric slices, as before
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    assert replicas.gpu_numa_node(0, os.path.join(root, "nowhere")) is None
    assert replicas.pin_to_compact_cpus(3, quota=16, sysfs_root=os.path.join(root, "nowhere")) == list(range(48, 64))


def _mismatch_worker(rank, world, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    graft.package()
    from dream_go_amd import api, replicas, synth

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        same = api.Network.from_json(synth.to_json(synth.make_tensors(1, 32, num_samples=2, seed=40 + rank)))
        other = api.Network.from_json(synth.to_json(synth.make_tensors(1 + rank, 32, num_samples=2, seed=50)))   # rank 1: another depth
        results[rank] = dict(same=replicas.descriptors_agree(dist, torch, same.image_descriptor()),
                             other=replicas.descriptors_agree(dist, torch, other.image_descriptor()),
                             bytes=(same.image_descriptor()[0], other.image_descriptor()[0]))
    finally:
        dist.destroy_process_group()


def test_ranks_with_different_architectures_are_refused_before_anything_moves():
    """dgnn_broadcast_weights trusted every rank to hold the same num_bytes; a rank with another
    architecture hung the collective.  The library now all-gathers (image bytes, architecture hash) first and EVERY rank returns
    DGNN_BAD_ARGUMENT on a mismatch (csrc/rccl.cpp); this is the same exchange over gloo (replicas.descriptors_agree, what
    bench.py --dry runs): both ranks see the refusal, nobody posts a broadcast, nobody hangs."""
    world = 2
    ctx = mp.get_context("spawn")
    with ctx.Manager() as manager:
        results = manager.dict()
        procs = [ctx.Process(target=_mismatch_worker, args=(r, world, port, results)) for port in [_free_port()] for r in range(world)]
        [p.start() for p in procs]
        [p.join(180) for p in procs]
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        out = {r: dict(results[r]) for r in range(world)}
    assert out[0]["same"] and out[1]["same"]                       # other weights, same architecture: fine
    assert not out[0]["other"] and not out[1]["other"]             # refused on BOTH ranks
    assert out[0]["bytes"][1] != out[1]["bytes"][1]

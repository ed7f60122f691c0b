"""Multi-GPU plumbing of the leaf-evaluation path: independent replicas, one process per GPU.

The reference treats GPUs as replicas fed whole batches round-robin and uploads the weights to each device from
the host (src/libdg_mcts/predictors/nn.rs:87-89, src/libdg_nn/tensor.rs:123-141); no activation ever crosses
devices.  Here every rank owns one engine; leaf batches (and, above them, games: game_id mod world) are sharded
statically, nothing is exchanged per step, and the only collective is ONE broadcast of the packed weight image
from the rank that loaded the file (RCCL over xGMI with backend "nccl"; gloo in the CPU tests).
"""
import numpy as np


def shard(num_units, world, rank):
    """Static contiguous sharding of independent units (games / leaf batches): -> range for this rank."""
    base, extra = divmod(int(num_units), int(world))
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def owner_of_game(game_id, world):
    """Games are pinned to a GPU so that a game's transposition cache and batches stay local."""
    return int(game_id) % int(world)


def rank_seed(base_seed, rank):
    """Distinct synthetic inputs per rank (ranks must not evaluate identical batches)."""
    return int(base_seed) + 7919 * int(rank)


def broadcast_image(dist, image, src=0):
    """In-place broadcast of the weight image (a torch uint8 tensor on the rank's device, or on CPU for gloo)."""
    dist.broadcast(image, src=src)
    return image


def all_agree(dist, torch, value, device="cpu"):
    """True iff every rank holds the same 63-bit integer (e.g. an image checksum)."""
    mine = torch.tensor([int(value) & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=device)
    gathered = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, mine)
    return len({int(g.item()) for g in gathered}) == 1


def descriptors_agree(dist, torch, descriptor, device="cpu"):
    """What dgnn_broadcast_weights does with RCCL before it moves the image, for a host-side broadcast (gloo, dry runs): every
    rank's (image bytes, architecture hash) is all-gathered; True on EVERY rank only if all ranks hold the same architecture and
    packing.  A rank with another network would otherwise post a broadcast of another size and hang the collective."""
    mine = torch.tensor([int(v) & 0x7FFFFFFFFFFFFFFF for v in descriptor], dtype=torch.int64, device=device)
    gathered = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, mine)
    return all(bool((g == gathered[0]).all().item()) for g in gathered)


def max_over_ranks(dist, torch, seconds, device="cpu"):
    """The job's wall time is the slowest rank's."""
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, torch, count, device="cpu"):
    t = torch.tensor([int(count)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def fnv1a(data):
    """The checksum libdgnn_hip.so uses for weight images (dgnn_network_weight_checksum)."""
    h = 1469598103934665603
    for byte in np.asarray(data, dtype=np.uint8).tobytes():
        h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def games_of_rank(num_games, world, rank):
    """Games are sharded `game_id mod world` (owner_of_game) -> (count, first_game, game_stride) for dgnn_selfplay_go."""
    num_games, world, rank = int(num_games), int(world), int(rank)
    count = (num_games - rank + world - 1) // world if rank < num_games else 0
    return count, rank, world


def selfplay_sharded(api, dist, torch, num_games, num_moves, num_rollouts, threads_per_search, seed, pool=None, cache=None,
                     return_moves=False, return_sgf=False, device="cpu", num_parallel=0, ex_it_rollouts=0, leaves_per_thread=1, cache_per_game=0,
                     host_threads=0, opening=None, window=None, on_game=None):
    """One self-play job over all ranks (BASELINE.json configs[3]): every rank plays its games against its own engine, pool and
    transposition cache, nothing is exchanged while the games run, the counters are summed and the clock is the slowest
    rank's.  `dist` may be None (one rank).  Returns this rank's result with the job totals under "job"."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    count, first, stride = games_of_rank(num_games, world, rank)
    if dist is not None:
        dist.barrier()
    if count > 0:
        out = api.selfplay_go(count, num_moves, num_rollouts, threads_per_search, seed=seed, pool=pool, cache=cache,
                              return_moves=return_moves, return_sgf=return_sgf, first_game=first, game_stride=stride,
                              num_parallel=num_parallel, ex_it_rollouts=ex_it_rollouts, leaves_per_thread=leaves_per_thread, cache_per_game=cache_per_game,
                              host_threads=host_threads, opening=opening, window=window, on_game=on_game)
    else:
        out = {"seconds": 0.0, "games": 0, "moves": 0, "evals": 0, "conflicts": 0, "cache_hits": 0, "games_ended": 0, "ex_it_searches": 0}
    out["game_ids"] = [first + i * stride for i in range(count)]
    keys = ("games", "moves", "evals", "conflicts", "cache_hits", "games_ended", "ex_it_searches")
    if dist is not None:
        job = {k: sum_over_ranks(dist, torch, out[k], device=device) for k in keys}
        job["seconds"] = max_over_ranks(dist, torch, out["seconds"], device=device)
    else:
        job = {k: out[k] for k in keys}
        job["seconds"] = out["seconds"]
    if window is not None:             # a timed window over games with replacement: the ranks' counters summed, the longest window's clock
        mine = out.get("window") or {"seconds": 0.0}
        wkeys = ("moves", "evals", "cache_hits", "conflicts", "games_started", "games_completed", "games_ended", "stones_completed", "ex_it_searches")
        if dist is not None:
            job["window"] = {k: sum_over_ranks(dist, torch, mine.get(k, 0), device=device) for k in wkeys}
            job["window"]["seconds"] = max_over_ranks(dist, torch, mine["seconds"], device=device)
        else:
            job["window"] = {k: mine.get(k, 0) for k in wkeys}
            job["window"]["seconds"] = mine["seconds"]
    job["world"] = world
    out["job"] = job
    return out


def _parse_cpulist(text):
    """"0-15,128-143" -> [0, ..., 15, 128, ..., 143]."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def visible_gpu_index(local_rank):
    """The PHYSICAL GPU (its position in the KFD topology, the order ROCr / HIP enumerate devices in) behind device `local_rank` of
    this process: HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES re-number the devices a process sees (a list of
    physical indices; ROCR is applied first, HIP on top of it).  UUID entries ("GPU-...") cannot be resolved here: None."""
    import os
    index = int(local_rank)
    for name in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):      # innermost re-numbering first
        value = os.environ.get(name)
        if value is None or value.strip() == "":
            continue
        if name == "CUDA_VISIBLE_DEVICES" and os.environ.get("HIP_VISIBLE_DEVICES"):
            continue                                                                           # (HIP_ takes precedence over CUDA_)
        entries = [e.strip() for e in value.split(",")]
        if index >= len(entries) or not entries[index].lstrip("-").isdigit():
            return None
        index = int(entries[index])
        if index < 0:
            return None
    return index


def gpu_numa_node(local_rank=0, sysfs_root="/sys"):
    """NUMA node of the GPU this rank opens, from the KFD topology (the nodes with SIMDs are the GPUs, in the runtime's order; each
    names its DRM render node, whose PCI device carries `numa_node`) -- the figure `rocm-smi --showtoponuma` prints.  None when the
    tree is not there (no GPU, a container without /sys/class/kfd) or says -1."""
    import os
    gpu = visible_gpu_index(local_rank)
    if gpu is None:
        return None
    nodes_dir = os.path.join(sysfs_root, "class", "kfd", "kfd", "topology", "nodes")
    try:
        names = sorted(os.listdir(nodes_dir), key=lambda n: int(n) if n.isdigit() else 1 << 30)
    except OSError:
        return None
    gpus = []
    for name in names:
        try:
            props = dict(line.split()[:2] for line in open(os.path.join(nodes_dir, name, "properties")) if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            gpus.append(props)
    if gpu >= len(gpus):
        return None
    minor = gpus[gpu].get("drm_render_minor")
    if minor is None:
        return None
    try:
        node = int(open(os.path.join(sysfs_root, "class", "drm", f"renderD{minor}", "device", "numa_node")).read().strip())
    except (OSError, ValueError):
        return None
    return node if node >= 0 else None


def numa_node_cpus(node, sysfs_root="/sys", primary_threads=0):
    """The CPUs of a NUMA node.  primary_threads > 0: when the node lists its CPUs as several ranges ("0-63,128-191": the cores' first
    hardware threads, then their SMT siblings) and the first range alone holds that many CPUs, only the first range -- a rank's slice
    is then one thread on each of `quota` neighbouring cores, not a handful of cores and their siblings (whose other halves may
    belong to somebody else's busy threads: /proc/stat calls such a sibling idle)."""
    import os
    try:
        text = open(os.path.join(sysfs_root, "devices", "system", "node", f"node{int(node)}", "cpulist")).read()
    except (OSError, ValueError):
        return []
    ranges = [r for r in text.strip().split(",") if r]
    try:
        if primary_threads > 0 and len(ranges) > 1 and len(_parse_cpulist(ranges[0])) >= primary_threads:
            return _parse_cpulist(ranges[0])
        return _parse_cpulist(text)
    except ValueError:
        return []


def pin_to_compact_cpus(local_rank=0, quota=None, sysfs_root="/sys"):
    """Keeps this process's threads on a compact set of CPUs, as many as its CPU-time quota pays for.

    The GPU boxes of the pool show 256 logical CPUs (2 sockets x 64 cores x 2) and grant a container 16 CPUs of TIME
    (cgroup cpu.max): unpinned, the 256 game threads of a self-play job roam over every core of both sockets, run with cold
    caches after every wait and are throttled in bursts -- measured on the bench's self-play leg: 13.3 cores of CPU time for
    143 k rollouts/s unpinned, 6.0 cores for 195 k rollouts/s on CPUs 0-15 (tools/affinity_probe.sh).  Rank r of a node takes
    the r-th slice of the allowed CPUs (in their numeric order: on these boxes the first half is socket 0, where GPUs 0-3
    hang, and a slice of 16 is 16 physical cores of two neighbouring CCDs).  Returns the chosen CPUs, or None when the process
    is not oversubscribed (quota >= allowed CPUs) or the platform has no affinity call.  DGNN_PIN_CPUS=0 switches it off.
    NUMA (round 5): where the KFD topology says which NUMA node the rank's GPU hangs on (gpu_numa_node: HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES honoured), the slices are cut from THAT node's CPUs, and the ranks whose GPUs share a node take its
    slices in rank order -- the pinned host buffers, the records a rank uploads and the answers it downloads then stay on the
    socket the GPU's PCIe root is on.  Without the topology the numeric order stands in for it, as before."""
    import os
    if os.environ.get("DGNN_PIN_CPUS", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = sorted(os.sched_getaffinity(0))
    node = gpu_numa_node(local_rank, sysfs_root)
    if node is not None and quota is None:
        quota_hint = 16
        try:
            q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                quota_hint = max(1, int(int(q) / int(period)))
        except (OSError, ValueError):
            pass
    else:
        quota_hint = quota or 16
    if node is not None:
        local = [c for c in numa_node_cpus(node, sysfs_root, primary_threads=quota_hint) if c in set(allowed)]
        if local:
            # my turn among the node's ranks: the lower local ranks whose GPU sits on the same node
            world_local = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
            turn = sum(1 for r in range(min(int(local_rank), world_local)) if gpu_numa_node(r, sysfs_root) == node)
            allowed, local_rank = local, turn
    if quota is None:
        quota = len(allowed)
        try:
            q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                quota = max(1, int(int(q) / int(period)))
        except (OSError, ValueError):
            pass
        quota = max(1, quota // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))     # the node's ranks share the container's quota
    if quota >= len(allowed):
        return None
    slices = max(1, len(allowed) // quota)
    index = local_rank % slices
    if int(os.environ.get("WORLD_SIZE", "1")) == 1:
        # alone on (a share of) the machine: other tenants may already sit on the low-numbered cores -- take the idlest slice
        busy = _cpu_busy_fractions()
        if busy:
            load = [sum(busy.get(c, 0.0) for c in allowed[i * quota:(i + 1) * quota]) for i in range(slices)]
            quiet = [i for i in range(slices) if load[i] < 0.1 * quota]          # less than a tenth of the slice in use
            index = quiet[0] if quiet else min(range(slices), key=lambda i: load[i])   # the first quiet one: near GPU 0's socket
    chosen = allowed[index * quota:(index + 1) * quota]
    os.sched_setaffinity(0, chosen)
    return chosen


def _cpu_busy_fractions(interval=0.05):
    """Busy fraction per logical CPU over a short interval, from /proc/stat ({} where there is none)."""
    import time

    def snapshot():
        out = {}
        try:
            for line in open("/proc/stat"):
                if line.startswith("cpu") and line[3].isdigit():
                    f = line.split()
                    v = [int(x) for x in f[1:9]]
                    out[int(f[0][3:])] = (sum(v), v[3] + v[4])          # total, idle + iowait
        except (OSError, ValueError, IndexError):
            return {}
        return out
    a = snapshot()
    if not a:
        return {}
    time.sleep(interval)
    b = snapshot()
    busy = {}
    for cpu, (total, idle) in b.items():
        if cpu in a and total > a[cpu][0]:
            busy[cpu] = 1.0 - (idle - a[cpu][1]) / (total - a[cpu][0])
    return busy

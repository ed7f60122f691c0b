#!/usr/bin/env python
"""Throughput harness for the leaf-evaluation path (the counterpart of the reference's `dream_go --bench`
forward line, src/dg/bench/forward.rs:28-47, and of src/dg_tests/benches/batch_sizes.rs:41-68).

    python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` with N > 1 and no launcher in the environment SPAWNS the N ranks itself (one process per GPU under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`); started under a launcher
(RANK / WORLD_SIZE set) it is one of the ranks.  It never prints an `n_gpus` other than the one asked for.

One "step" = one pass of the hot path (dg_nn::forward's device work, graph.rs:134-152) over one batch of
synthetic positions that is already resident in HBM.  Workload at every N: BASELINE.json configs[2] --
19x19, 20-block / 256-filter network, batch 256 per GPU -- the configuration the metric's roofline target is
quoted on.  Multi-GPU: independent replicas, one process per GPU, units sharded across ranks with no data-path
collective ("weak" scaling); the only collective is one RCCL broadcast of the packed weight image at start-up,
issued by libdgnn_hip.so itself (dgnn_broadcast_weights): every rank but 0 starts from DIFFERENT weights of the
same architecture, so checksum agreement afterwards proves that RCCL moved the image.

Beside the engine number the line carries MEASURED self-play legs -- real Go boards, the search, the leaf pool and the engine
together (BASELINE.json's metric is games/hour + evals/s; SURVEY.md 8d "reports the measured ratio"):
  whole_games          the steady state of WHOLE games on this run's own clock: --whole-games games in flight with replacement,
                       --num-rollout 800, a ramp and a timed window (dgnn_selfplay_window); headline moves per second, games per hour
                       as counted (with the count's sigma) and from the moves; `tower_passes`: the device passes by tower form
  selfplay_midgame     stones 61..66 of 256 games (+ the same under the reference's batching rule, `selfplay_midgame_reference_rule`)
  policy_play          --num-rollout 1 WITHOUT --ex-it (1 024 games in flight): 8 evaluations per move
  policy_play_ex_it    BASELINE.json configs[4] AS WRITTEN: --ex-it --num-rollout 1 with --num-ex-it-rollout 1600 (the reference's default)
  selfplay             (only with --selfplay-opening-leg 1) the first stones of games started together: a host-bound corner
and, PCIe-inclusive, `value_reference_contract` (fp16 planes in host memory in, host buffers out, two workspaces alive: the
contract of dg_nn::forward, graph.rs:130-156), `value_pcie_inclusive` (board records / packed planes / three alive) and
`reference_style` (the `dream_go --bench` walk).  `cpu_baseline` is the FASTER of two CPU statements of the reference's semantics
(the OpenMP C port that is also the parity checker; torch-CPU on the trainer's graph), the other one beside it.  The process pins
itself to as many neighbouring CPUs as its cgroup quota pays for before any thread exists (replicas.pin_to_compact_cpus;
DGNN_PIN_CPUS=0 switches it off); the CPU baseline runs unpinned.

`--dry` runs the same control flow without a GPU (gloo, the deterministic CPU predictor behind the leaf pool): it
exists so that the N > 1 path is covered by the CPU tests; its numbers mean nothing and the line says "dry": true.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP16_MFMA_TFLOPS = 2500.0   # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--cpu-sample", type=int, default=256, help="boards the CPU oracle evaluates for cpu_baseline (0 = skip)")
    ap.add_argument("--selfplay-games", type=int, default=256, help="games in flight PER GPU in the measured self-play leg (0 = skip the leg)")
    ap.add_argument("--selfplay-moves", type=int, default=6, help="stones per game in the self-play leg (bounds its duration)")
    ap.add_argument("--selfplay-rollouts", type=int, default=800, help="--num-rollout of the self-play leg (BASELINE.json: 800)")
    ap.add_argument("--selfplay-midgame", type=int, default=60,
                    help="second self-play leg: the searched stones start after this many stones of policy play (0 = skip that leg)")
    ap.add_argument("--selfplay-leaves", type=int, default=16,
                    help="leaves a search thread keeps on the device (dgnn_search_run_batched): one thread x 16 leaves = the reference's default of 16 "
                         "parallel probes per search tree (--num-threads, libdg_utils/config.rs:143-149; rounds 1-3 ran 8)")
    ap.add_argument("--selfplay-batches", type=int, default=3,
                    help="device batches the leaf pool keeps alive per GPU.  The reference keeps 2 (nn.rs:64-67: the reference-rule leg runs "
                         "that).  A third keeps the device supplied across the host's turn-around between batches and on partial batches; it "
                         "puts more leaves in flight, but with coalescing a position evaluated twice in flight is a follower, not a second "
                         "evaluation (DESIGN.md 11.7, 13.5; round 5, same box: steady-state whole games 0.97 -> 1.00 of `value`, 14.2)")
    ap.add_argument("--selfplay-workers", type=int, default=0,
                    help="pool worker threads per GPU; 0 (default) = one more than batches alive: a free worker takes the next batch while another "
                         "is still handing 256 answers to their games (round 5, same box, two runs each: mid-game leg 0.95 -> 0.985 of `value`, "
                         "steady-state whole games 0.97 -> 0.99 with two alive; DESIGN.md 14.2)")
    ap.add_argument("--selfplay-coalesce", type=int, default=1,
                    help="1 (default): a request for a position that is already queued or on the device follows it instead of being evaluated "
                         "again (dgnn_pool_set_coalescing: what the shared transposition table does, one evaluation earlier); 0: the reference's behaviour")
    ap.add_argument("--selfplay-back-to-back", type=int, default=300, metavar="LEAD_US",
                    help="> 0 (default 300): while a batch occupies the device a PARTIAL batch waits until this many microseconds before that batch "
                         "is expected to finish and leaves with what has queued up meanwhile (dgnn_pool_set_back_to_back; a full batch goes at "
                         "once); 0: the reference's rule, whatever is queued the moment a slot is free")
    ap.add_argument("--selfplay-snap", default="auto", metavar="LO,HI",
                    help="with --selfplay-back-to-back: a partial batch of LO < n < HI boards is cut to its LO oldest (dgnn_pool_set_back_to_back's snapping). "
                         "Default \"auto\" = half the batch size, 7/8 of it (128,224): such a batch runs as a split tower of 1.6 ms instead of a whole-board "
                         "tower of 3 ms that leaves a part of the chip idle.  With the split towers of round 6, same box: 24 games in flight 70.2 -> "
                         "74.9 k boards/s, 32 games 74.6 -> 75.7 k, 48 and 96 games unchanged (profiles/r06_final/batch_cap_with_few_games.txt, "
                         "snap_with_many_games.txt; round 4 measured a loss: its split towers were chained and sized alone).  \"\" or \"0,0\": off")
    ap.add_argument("--selfplay-host-threads", type=int, default=4,
                    help="host threads PER GPU the games of the self-play leg run on as fibers (0 = one OS thread per game, the reference's model)")
    ap.add_argument("--selfplay-planes", choices=["device", "host"], default="device",
                    help="where the 32 feature planes are built: on the device from 896-byte board records (default), or on the host like the reference")
    ap.add_argument("--whole-games", type=int, default=96,
                    help="games in flight PER GPU, with replacement, in the steady-state WHOLE-GAME leg (the setting of tools/selfplay_go.py; 0 = skip the leg)")
    ap.add_argument("--selfplay-opening-leg", type=int, default=0,
                    help="1: also run the OPENING-stones leg (`selfplay`: the first --selfplay-moves stones of games started together; 1.3 s of a "
                         "host-bound corner -- games in flight share their first positions -- that rounds 1-5 ran by default; VERDICT r05 item 7)")
    ap.add_argument("--whole-games-seconds", type=float, default=120.0, help="the timed window of the whole-game leg (round 5: 90 s, which sees 12-18 endings)")
    ap.add_argument("--whole-games-ramp", type=float, default=20.0, help="untimed seconds in front of that window")
    ap.add_argument("--whole-games-records", default="", help="write the move lists of the whole-game leg's games to this .npz (diagnosis: which seeded games ended, and when)")
    ap.add_argument("--policy-play-games", type=int, default=1024,
                    help="games in flight PER GPU in the policy-play leg (BASELINE.json configs[4]: --num-rollout 1; 0 = skip the leg)")
    ap.add_argument("--policy-play-seconds", type=float, default=10.0, help="the timed window of the policy-play leg (after 2 untimed seconds)")
    ap.add_argument("--ex-it-rollouts", type=int, default=1600,
                    help="--num-ex-it-rollout of the policy_play_ex_it leg: BASELINE.json configs[4] AS WRITTEN (--ex-it --num-rollout 1); the reference's "
                         "default is 1600 (src/libdg_utils/config.rs:130); 0 = skip the leg")
    ap.add_argument("--ex-it-seconds", type=float, default=15.0, help="the timed window of the policy_play_ex_it leg (after 5 untimed seconds)")
    ap.add_argument("--sustained-seconds", type=float, default=5.0, help="`value_sustained`: the passes of `value` repeated for at least this long (0 = skip)")
    ap.add_argument("--reference-rule-leg", type=int, default=1,
                    help="1 (default): the mid-game leg is run a second time under the REFERENCE's batching rule -- no coalescing, whatever is queued "
                         "the moment a slot is free (worker_thread.rs:131-140) -- and reported beside the tuned one; 0: skip")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the weight broadcast even with one rank (self-test of the N > 1 path)")
    ap.add_argument("--lean", action="store_true", help="only the engine measurement (profiling passes): no PCIe-inclusive, reference-style or self-play legs")
    ap.add_argument("--dry", action="store_true", help="no GPU: gloo + the CPU hash predictor behind the leaf pool (CPU tests of the N > 1 path)")
    args = ap.parse_args(argv)
    args.selfplay_batches = max(1, args.selfplay_batches)
    if args.lean:
        args.selfplay_games, args.cpu_sample, args.whole_games, args.policy_play_games, args.sustained_seconds, args.ex_it_rollouts = 0, 0, 0, 0, 0.0, 0
    return args


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: become the launcher.  The reference needs none -- one process drives every
    device (src/libdg_mcts/predictors/nn.rs:64-67,87-89) -- so a bare invocation must not silently measure one GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def effective_cpus():
    """The cores this process may really use: the scheduler affinity, capped by the cgroup CPU quota (a GPU box of the pool
    shows 256 logical CPUs but grants 16 CPUs of time, /sys/fs/cgroup/cpu.max = "1600000 100000")."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(text, features, batch, sample):
    """The CPU oracle (a port of the reference semantics; the reference has no CPU NN path) on the host cores."""
    from oracle import oracle
    oracle.lib().dgo_set_num_threads(effective_cpus())            # one thread per usable core
    tensors, _ = oracle.load_weights(text)
    net = oracle.OracleNetwork(tensors)
    sample = min(sample, batch)
    f = features.reshape(batch, 11552)[:sample].reshape(-1)
    net.forward(f[:11552], 1)                       # warm (thread pool, page faults)
    t0 = time.perf_counter()
    value, policy = net.forward(f, sample)
    dt = time.perf_counter() - t0
    return {
        "value": sample / dt, "unit": "evals/s", "cores": int(oracle.lib().dgo_num_threads()), "kind": "port",
        "sample": f"{sample} boards of the same batch, one dgo_forward call (oracle/dgnn_oracle.c, OpenMP), {dt:.1f} s",
    }, (value, policy)


def cpu_second_opinion(text, features, batch, sample, threads):
    """a second CPU figure beside the OpenMP port -- the same network stated from the trainer's graph in torch-CPU fp32
    (F.conv2d; oracle/torch_statement.py, the statement tests/test_third_statement.py holds against the oracle), on as many threads."""
    import torch
    from oracle import oracle
    from oracle.torch_statement import trainer_graph_forward
    tensors, _ = oracle.load_weights(text)
    sample = min(sample, batch)
    f = np.ascontiguousarray(features.reshape(batch, 11552)[:sample].reshape(-1))
    before = torch.get_num_threads()
    torch.set_num_threads(max(1, int(threads)))
    try:
        with torch.no_grad():
            trainer_graph_forward(tensors, f[:11552], 1, round_layers=True)          # warm
            t0 = time.perf_counter()
            value, policy = trainer_graph_forward(tensors, f, sample, round_layers=True)
            dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(before)
    return {"value": sample / dt, "unit": "evals/s", "cores": int(threads), "kind": "port",
            "sample": f"{sample} boards, torch-CPU fp32 F.conv2d with fp16 rounding per layer (oracle/torch_statement.py), {dt:.1f} s"}, (value, policy)


def real_game_positions(api):
    """Every position of the three committed reference games (tests/golden/real_games.json; the SGF fixtures of
    src/dg_tests/tests/real_games.rs) as (board, colour to move)."""
    import re
    out = []
    for game in json.load(open(os.path.join(ROOT, "tests", "golden", "real_games.json")))["games"]:
        board = api.Board(7.5)
        for color, vertex in re.findall(r";([BW])\[([a-s]{2})\]", game["moves"]):
            c = 0 if color == "B" else 1
            board.place(c, ord(vertex[0]) - 97, ord(vertex[1]) - 97)
            out.append((board.clone(), 1 - c))
    return out


class Job:
    """The ranks of this run: torch.distributed for the barrier / the max-over-ranks clock, nothing on the data path."""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}: refusing to report a different n_gpus")
        self.dry = args.dry
        self.dist = None
        self.torch = None
        self.device = self.local_rank
        self.tensor_device = "cpu"

    def start(self, force):
        import torch   # device plumbing + torch.distributed only; imported before libdgnn_hip.so is loaded (one HIP runtime)
        self.torch = torch
        if not self.dry:
            visible = torch.cuda.device_count()
            if visible < self.world:
                raise SystemExit(f"--gpus {self.world} but only {visible} GPU(s) visible")
            self.tensor_device = f"cuda:{self.device}"
        if self.world > 1 or force:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            import torch.distributed as dist
            if self.dry:
                dist.init_process_group(backend="gloo")
            else:
                torch.cuda.set_device(self.device)
                dist.init_process_group(backend="nccl", device_id=torch.device(self.tensor_device))
            self.dist = dist

    def barrier(self, ws=None):
        if self.dist is not None:
            self.dist.barrier()
        if not self.dry:
            self.torch.cuda.synchronize(self.device)
        if ws is not None:
            ws.sync()

    def gather_int(self, value):
        if self.dist is None:
            return [int(value)]
        t = self.torch.tensor([int(value) & 0x7FFFFFFFFFFFFFFF], dtype=self.torch.int64, device=self.tensor_device)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [int(x.item()) for x in out]

    def gather_float(self, value):
        if self.dist is None:
            return [float(value)]
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.tensor_device)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(x.item()) for x in out]

    def broadcast_bytes(self, payload, size):
        """rank 0's bytes to everybody (the RCCL unique id travels this way; any host channel would do)."""
        if self.dist is None:
            return bytes(payload)
        t = self.torch.zeros(size, dtype=self.torch.uint8, device=self.tensor_device)
        if self.rank == 0:
            t.copy_(self.torch.frombuffer(bytearray(payload), dtype=self.torch.uint8))
        self.dist.broadcast(t, src=0)
        return bytes(t.cpu().numpy().tobytes())

    def finish(self):
        if self.dist is not None:
            import ctypes
            sys.stdout.flush()
            ctypes.CDLL(None).fflush(None)   # every rank empties its C stdio (RCCL's banner) BEFORE the barrier rank 0 prints after
            self.dist.barrier()
            self.dist.destroy_process_group()


def broadcast_weights(job, api, net):
    """ONE broadcast of the packed weight image from rank 0, by the library's own RCCL (dgnn_broadcast_weights); in a dry run a
    gloo broadcast of the host image.  Every rank but 0 holds different weights before it."""
    before = job.gather_int(net.weight_checksum())
    t0 = time.perf_counter()
    if job.dry:
        from dream_go_amd import replicas
        if not replicas.descriptors_agree(job.dist, job.torch, net.image_descriptor()):
            raise RuntimeError("the ranks do not hold the same architecture / packing: refusing to broadcast")
        image = job.torch.from_numpy(net.weight_image_host())      # view of the library's host image
        job.dist.broadcast(image, src=0)
        after = job.gather_int(net.weight_checksum())
        backend, nbytes, comm_ranks = "gloo (dry run, host image)", int(image.numel()), job.world
    else:
        ident = job.broadcast_bytes(api.Comm.unique_id() if job.rank == 0 else b"", api.RCCL_UNIQUE_ID_BYTES)
        comm = api.Comm.create(job.device, job.world, job.rank, ident)
        comm_ranks = comm.info()["num_ranks"]
        net.broadcast_weights(comm, root=0)
        comm.destroy()
        after = job.gather_int(net.weight_checksum_device(job.device))
        host = job.gather_int(net.weight_checksum())
        if len(set(host)) != 1 or host != after:
            raise RuntimeError("host and device weight images disagree after the broadcast")
        backend, nbytes = "rccl (libdgnn_hip.so: ncclBroadcast over a communicator built from a unique id)", net.weight_image(job.device)[1]
    seconds = time.perf_counter() - t0
    if len(set(after)) != 1 or after[0] != before[0]:
        raise RuntimeError(f"weight image differs across ranks after the broadcast: {after} (rank 0 had {before[0]})")
    return {"backend": backend, "bytes": nbytes, "ranks": comm_ranks, "checksums_agree": True,
            "distinct_images_before": len(set(before)), "seconds_incl_comm_setup": seconds}


def selfplay_pool(job, api, net, args, coalesce=None, back_to_back=None, batches=None):
    """The leaf pool of every self-play leg: one device (or its stand-in), `--selfplay-batches` batches alive, coalescing and
    back-to-back batching as the arguments say (coalesce / back_to_back / batches override them: the reference-rule leg)."""
    boards = args.selfplay_planes == "device"
    alive = args.selfplay_batches if batches is None else batches
    workers = max(args.selfplay_workers, alive) if (args.selfplay_workers > 0 or batches is not None) else alive + 1
    if job.dry and boards:
        # no GPU: the CPU stand-in behind a stand-in for the device (a batch of n boards takes 500 + 10.8 n us, one at a time: the
        # 20x256 engine's 3.3 ms at 256 boards) -- what is measured is the HOST side alone, rollouts per second and core
        pool = api.LeafPool.over_dry_device(base_us=500.0, per_board_us=10.8, max_batch_size=args.batch, batches_per_lane=alive, workers_per_lane=workers)
    elif job.dry:
        pool = api.LeafPool.over_callbacks(api.lib.dgnn_hash_predictor(), None, num_lanes=1, max_batch_size=args.batch)
    else:
        pool = api.LeafPool.over_devices(net, [job.device], max_batch_size=args.batch, batches_per_device=alive, workers_per_device=workers, boards=boards)
    coalesce = (bool(args.selfplay_coalesce) if coalesce is None else coalesce) and pool.takes_boards   # (the games of a leg share one transposition table)
    if coalesce:
        pool.set_coalescing(True)
    lead = args.selfplay_back_to_back if back_to_back is None else back_to_back
    if lead > 0:
        text = getattr(args, "selfplay_snap", "")
        snap = [args.batch // 2, args.batch * 7 // 8] if text == "auto" else [int(v) for v in text.split(",")] if text else [0, 0]
        if args.batch < 64 or snap[0] <= 0:
            snap = [0, 0]
        pool.set_back_to_back(lead, snap[0], snap[1])
        rule_snap = snap
    else:
        rule_snap = [0, 0]
    return pool, {"coalesce": coalesce, "back_to_back": lead, "boards": boards, "alive": alive, "workers": workers, "snap": rule_snap}


def pool_rule_text(args, rule):
    return (f"{rule['alive']} device batches alive, {rule['workers']} pool workers" +
            (f", partial batches back to back (lead {rule['back_to_back']} us" + (f", {rule['snap'][0]} < n < {rule['snap'][1]} boards cut to {rule['snap'][0]}" if rule.get("snap", [0, 0])[0] > 0 else "") + ")"
             if rule["back_to_back"] > 0 else ", a free slot takes what is queued (the reference's rule)") +
            (", positions in flight coalesced" if rule["coalesce"] else ", no coalescing (the reference: every miss is evaluated)"))


class EdgeProbe:
    """What the library does not count, read at the two edges of a dgnn_selfplay_window (from the watcher thread): the pool's device
    boards / batches / followers, the process's CPU time, the wall clock."""

    def __init__(self, pool, net=None):
        self.pool, self.net, self.edges = pool, net, {}

    def __call__(self, edge):
        st = self.pool.stats()
        self.edges[edge] = {"boards": st["boards"], "batches": st["batches"], "coalesced": self.pool.coalesced(),
                            "cpu": time.process_time(), "wall": time.perf_counter()}
        if self.net is not None:
            self.edges[edge].update(self.net.tower_stats())    # device passes by the form their residual tower took (dgnn_network_tower_stats)

    def delta(self, key):
        return self.edges[1][key] - self.edges[0][key] if 0 in self.edges and 1 in self.edges else 0


def window_leg(job, api, replicas, net, args, in_flight, rollouts, ramp, seconds, staggered, seed, ex_it_rollouts=0):
    """`in_flight` games per GPU WITH REPLACEMENT (self_play.rs:478-495) for a fixed wall time: `ramp` untimed seconds, then a
    window of `seconds` whose counters the library notes itself (dgnn_selfplay_window), then the job stops.  staggered: game i of a
    rank's first `in_flight` starts i / in_flight of the way into a game that policy play (--num-rollout 1, not timed) has played to
    its end, so that from the first second on the games in flight are at every stage of a game -- the steady state of a long job --
    and whole games END inside the window at the rate they would in hour six."""
    pool, rule = selfplay_pool(job, api, net, args)
    cache = api.LruCache(200_001, num_shards=8)
    opening, prep_seconds, seeded_by = None, 0.0, None
    if staggered:
        t0 = time.perf_counter()
        # the games to start inside: whole games SEARCHED by this very network (tools/make_searched_games.py; positions to start from,
        # nothing else), so that the games in flight are a sample of the steady state from the first second.  Without the file (another
        # network, a dry run): games played out by policy play -- ~100 stones shorter, and none of them near its end when the window opens
        seeds = os.path.join(ROOT, "profiles", "searched_games_20x256.npz")
        record = None
        if os.path.exists(seeds) and not job.dry:
            try:
                z = np.load(seeds)
                if (int(z["blocks"]), int(z["channels"]), int(z["rollouts"])) == (args.blocks, args.channels, rollouts) and z["record"].shape[0] >= in_flight:
                    record = np.asarray(z["record"][:in_flight], dtype=np.int32)
                    seeded_by = "whole games searched by this network (profiles/searched_games_20x256.npz, tools/make_searched_games.py)"
            except Exception:   # noqa: BLE001
                record = None
        if record is None:
            ahead = replicas.selfplay_sharded(api, job.dist, job.torch, num_games=in_flight * job.world, num_moves=0, num_rollouts=1, threads_per_search=1,
                                              seed=seed, pool=pool, cache=api.LruCache(50_001, num_shards=8), device=job.tensor_device,
                                              host_threads=args.selfplay_host_threads, return_moves=True)
            record = ahead["record"]
            seeded_by = "games played out by policy play (--num-rollout 1, not timed)"
        opening = np.array(record, copy=True)
        for i in range(opening.shape[0]):
            stones = int((opening[i] >= 0).sum())
            opening[i, stones * i // opening.shape[0]:] = -1
        prep_seconds = time.perf_counter() - t0
    probe = EdgeProbe(pool, None if job.dry else net)
    # the games' records leave the way the reference's do (self_play.rs:471-497: a channel whose reader prints them): counted and let go of
    taken = {"records": 0, "whole": 0, "bytes": 0}

    def take_record(_game, text, whole):
        taken["records"] += 1
        taken["whole"] += int(whole)
        taken["bytes"] += len(text)
    out = replicas.selfplay_sharded(api, job.dist, job.torch, num_games=4096 * job.world, num_moves=0, num_rollouts=rollouts, threads_per_search=1,
                                    seed=seed + 1, pool=pool, cache=cache, device=job.tensor_device, leaves_per_thread=args.selfplay_leaves,
                                    host_threads=args.selfplay_host_threads, opening=opening, num_parallel=in_flight, window=(ramp, seconds, probe),
                                    return_moves=bool(staggered and args.whole_games_records), on_game=take_record, ex_it_rollouts=ex_it_rollouts)
    if staggered and args.whole_games_records and job.rank == 0 and "record" in out:
        np.savez_compressed(args.whole_games_records, record=np.asarray(out["record"][:4 * in_flight], dtype=np.int16), opening=np.asarray(opening, dtype=np.int16),
                            ramp=ramp, seconds=seconds)
    batch_times = pool.batch_times()
    pool.close()
    w = out["job"]["window"]
    sec = max(w["seconds"], 1e-9)
    boards = sum(job.gather_int(probe.delta("boards")))
    batches = sum(job.gather_int(probe.delta("batches")))
    followers = sum(job.gather_int(probe.delta("coalesced")))
    cores = job.gather_float(probe.delta("cpu") / max(probe.delta("wall"), 1e-9))
    stones = w["stones_completed"] / max(w["games_completed"], 1)
    leg = {
        "seconds": sec, "ramp_seconds": ramp, "preparation_seconds": prep_seconds, "games_in_flight_per_gpu": in_flight, "seeded_by": seeded_by,
        "moves": w["moves"], "moves_per_sec": w["moves"] / sec,
        "games_completed": w["games_completed"], "games_ended_by_passes": w["games_ended"], "games_started": w["games_started"],
        "mean_stones_per_completed_game": stones if w["games_completed"] else None,
        # whole games that ENDED inside the window, per hour of window -- and the same figure from the window's moves (less noisy: a
        # window sees a few dozen endings): moves per second x 3600 / the mean stones of the games that ended
        "games_per_hour_measured": w["games_completed"] * 3600.0 / sec,
        # (a count of a dozen or two: its own one-sigma counting noise, so that a reader does not take 12 against 18 endings for a change)
        "games_per_hour_measured_sigma": max(w["games_completed"], 1) ** 0.5 * 3600.0 / sec,
        "games_per_hour_from_moves": (w["moves"] / sec) * 3600.0 / stones if w["games_completed"] else None,
        "answers_per_sec": w["evals"] / sec,                          # what the searches were answered (table misses; followers of positions in flight included)
        "rollouts_per_sec": (w["evals"] + w["cache_hits"]) / sec,
        "evals_per_move": w["evals"] / max(w["moves"], 1),
        "cache_hit_ratio": w["cache_hits"] / max(w["cache_hits"] + w["evals"], 1),
        "device_boards_per_sec": boards / sec, "mean_device_batch": boards / max(batches, 1),
        "followers_per_sec": followers / sec,                         # boards that followed a position in flight instead of being evaluated
        "device_batch_us_by_size": {str(k): v for k, v in batch_times.items()},
        # rank 0's device passes inside the window by the form their residual tower took: one launch with a workgroup per board, one launch
        # with a board split over several workgroups (batches too small to fill the chip), one launch per layer (the fallback of a split
        # launch that gave up: expected 0, as is split_gave_up -- round 5's line had nine such give-ups on stderr)
        "tower_passes": None if job.dry else {k: probe.delta(k) for k in ("whole_board", "split", "per_layer", "split_gave_up")},
        "host_cpus_busy": cores[0], "per_rank_host_cpus_busy": cores, "host_cpus": effective_cpus(), "host_threads": args.selfplay_host_threads,
        "probe_conflicts": w["conflicts"],
        "ex_it_searches": w.get("ex_it_searches", 0), "ex_it_searches_per_sec": w.get("ex_it_searches", 0) / sec,
        # (rank 0's, ramp and the games the window's end cut short included: that the records ARE produced and handed over, not a rate)
        "records_streamed": {"records": taken["records"], "of_whole_games": taken["whole"], "megabytes": taken["bytes"] / 1e6},
    }
    return leg, rule


def whole_games_leg(job, api, replicas, net, args):
    """BASELINE.json's first metric on this run's own clock: self-play games per hour, whole games (self_play.rs:423-499)."""
    leg, rule = window_leg(job, api, replicas, net, args, args.whole_games, args.selfplay_rollouts, args.whole_games_ramp, args.whole_games_seconds,
                           staggered=True, seed=20261001)
    leg["what"] = (f"steady state of WHOLE games: {args.whole_games} games in flight per GPU with replacement, --num-rollout {args.selfplay_rollouts}, 1 search thread x "
                   f"{args.selfplay_leaves} leaves per game, games as fibers on {args.selfplay_host_threads} host threads, shared transposition table 200001 x 8, "
                   f"{pool_rule_text(args, rule)}; game i of the first {args.whole_games} starts i / {args.whole_games} of the way into a finished game -- {leg['seeded_by']} -- "
                   f"so games at every stage are in flight from the start and END inside the window; {args.whole_games_ramp:.0f} s ramp, then the timed window; "
                   "random-init weights (such a net plays 600-700 stones per game: games_per_hour_at_250_stones is the same moves per second at a trained net's game length)")
    leg["games_per_hour_at_250_stones"] = leg["moves_per_sec"] * 3600.0 / 250.0
    # Labelled cross-checks, builder-run files: (1) the same steady state measured over TEN minutes with tools/selfplay_go.py --steady 20,600
    # -- the figure this leg's moves per second are to be compared with, and the game length a window of 90 s cannot see: the games that
    # end inside it were seeded by policy play and are ~100 stones shorter than games searched from the empty board; (2) round 4's file:
    # 96 games STARTED TOGETHER, whose searches share 40 % of their leaves -- not comparable, kept to say so.
    if not job.dry:
        for key, name, note in (
                ("steady_state_profile", os.path.join("r06_final", "selfplay_steady_96g_600s.json"),
                 "the same leg over a 600 s window (tools/selfplay_go.py --games 96 --steady 20,600); a builder-run file beside the measurement, not the measurement"),
                ("started_together_r04", os.path.join("r04_final", "selfplay_full_games_96g_l16_h4_sparse_b2b300.json"),
                 "96 games started together from the empty board and played to their end, no replacement (round 4): correlated games, 40 % of their "
                 "leaves followed a position in flight -- a different quantity")):
            path = os.path.join(ROOT, "profiles", name)
            if not os.path.exists(path):
                continue
            try:
                c = json.load(open(path))
                stones = c.get("mean_stones_per_completed_game") or c.get("mean_stones_per_game")
                leg.setdefault("cross_check_committed", {})[key] = {
                    "games_per_hour_measured": c["games_per_hour_measured"], "moves_per_sec": c["moves_per_sec"], "mean_stones_per_game": stones,
                    "source": "profiles/" + name.replace(os.sep, "/") + ": " + note}
                if key == "steady_state_profile" and stones:
                    # this run's moves per second at the game length the ten-minute window measured
                    leg["games_per_hour_at_profile_length"] = leg["moves_per_sec"] * 3600.0 / stones
            except Exception:   # noqa: BLE001
                pass
    return leg


def policy_play_leg(job, api, replicas, net, args):
    """BASELINE.json configs[4]: --num-rollout 1 (self_play.rs:370-405), a thousand games in flight, every move one 8-symmetry
    evaluation through the pool."""
    leg, rule = window_leg(job, api, replicas, net, args, args.policy_play_games, 1, 2.0, args.policy_play_seconds, staggered=False, seed=20261003)
    leg["what"] = (f"policy play WITHOUT expert iteration (--num-rollout 1, no --ex-it; the mode behind the 59-67 k games/hour of BASELINE.md): {args.policy_play_games} games in flight per GPU with replacement from the empty board, every move the "
                   f"8-symmetry evaluation of one position, games as fibers on {args.selfplay_host_threads} host threads, {pool_rule_text(args, rule)}; 2 s ramp, then the timed window "
                   "(games of 500-700 stones take about a minute here: few END inside a 10 s window, the rate is moves per second)")
    leg["games_per_hour_at_250_stones"] = leg["moves_per_sec"] * 3600.0 / 250.0
    return leg


def policy_play_ex_it_leg(job, api, replicas, net, args):
    """BASELINE.json configs[4] AS WRITTEN: `--ex-it --num-rollout 1`.  Every move is drawn from the 8-symmetry evaluation of its position
    (self_play.rs:370-405); on an undecided position (|value| <= 0.8) one move in twenty is re-searched from scratch with
    --num-ex-it-rollout rollouts (is_good_candidate / Player::ex_it, self_play.rs:277-319,394-399; default 1600, config.rs:130) and the
    search's visit distribution and value go into the record.  So a move costs 8 evaluations and, one time in twenty, a whole search:
    ~ 90 evaluations per move on average, most of them leaves of the ~ 5 % of games that are inside a search at any moment -- which each
    keep only --selfplay-leaves positions in flight: device batches are what those few searches and the 8-board policy calls add up to."""
    leg, rule = window_leg(job, api, replicas, net, args, args.policy_play_games, 1, 5.0, args.ex_it_seconds, staggered=False, seed=20261005,
                           ex_it_rollouts=args.ex_it_rollouts)
    leg["what"] = (f"configs[4] as written (--ex-it --num-rollout 1, --num-ex-it-rollout {args.ex_it_rollouts}): {args.policy_play_games} games in flight per GPU with "
                   f"replacement from the empty board; every move the 8-symmetry evaluation of one position, and on undecided positions one move in twenty a fresh "
                   f"search of {args.ex_it_rollouts} rollouts (1 search thread x {args.selfplay_leaves} leaves) whose visit distribution goes into the record; games as fibers on "
                   f"{args.selfplay_host_threads} host threads, {pool_rule_text(args, rule)}; 5 s ramp, then the timed window")
    leg["num_ex_it_rollouts"] = args.ex_it_rollouts
    leg["moves_per_ex_it_search"] = leg["moves"] / max(leg["ex_it_searches"], 1)
    leg["games_per_hour_at_250_stones"] = leg["moves_per_sec"] * 3600.0 / 250.0
    return leg


def selfplay_leg(job, api, replicas, net, args, fast_forward=0, reference_rule=False):
    """Whole pipeline, measured: Go boards -> board records (planes on the device) -> search (1 thread x K leaves per game) ->
    transposition cache -> leaf pool -> engine, `selfplay_games` games in flight per GPU for `selfplay_moves` searched stones each
    (self_play.rs:423-459; rollouts per move follow the win rate, :234-241).  Games are sharded `id mod world`.
    fast_forward > 0: every game is first played that many stones deep by policy play (--num-rollout 1, self_play.rs:369-405; not
    timed) and the searched stones start from there: a MID-GAME leg (fewer table hits, narrower trees, more ladders and captures).
    reference_rule: the pool batches like the reference's -- no coalescing of positions in flight, a free slot takes whatever is
    queued (pool/worker_thread.rs:131-140, predictors/nn.rs:64-67) -- so the figure is comparable with rounds 1-3 and with the
    reference's own behaviour."""
    games = args.selfplay_games * job.world
    cache = api.LruCache(200_001, num_shards=8)                                    # predictors/nn.rs:29,57
    pool, rule = selfplay_pool(job, api, net, args, coalesce=False if reference_rule else None, back_to_back=0 if reference_rule else None,
                               batches=2 if reference_rule else None)        # (nn.rs:64-67: 2 batches alive per device, a worker each)
    coalesce, boards = rule["coalesce"], rule["boards"]
    opening = None
    if fast_forward > 0:
        ahead = replicas.selfplay_sharded(api, job.dist, job.torch, num_games=games, num_moves=fast_forward, num_rollouts=1, threads_per_search=1,
                                          seed=20260930, pool=pool, cache=api.LruCache(50_001, num_shards=8), device=job.tensor_device,
                                          host_threads=args.selfplay_host_threads, return_moves=True)
        opening = ahead["record"]
        before = pool.stats()
    else:
        before = {"boards": 0, "batches": 0}
    cpu0, wall0 = time.process_time(), time.perf_counter()
    out = replicas.selfplay_sharded(api, job.dist, job.torch, num_games=games, num_moves=fast_forward + args.selfplay_moves,
                                    num_rollouts=args.selfplay_rollouts, threads_per_search=1, seed=20260930, pool=pool, cache=cache,
                                    device=job.tensor_device, leaves_per_thread=args.selfplay_leaves, host_threads=args.selfplay_host_threads,
                                    opening=opening)
    cpu_seconds = time.process_time() - cpu0                                                  # this rank's threads, user + system
    cpu_cores_busy = cpu_seconds / max(time.perf_counter() - wall0, 1e-9)
    rank_rollouts = out["evals"] + out["cache_hits"]                                          # this rank's own games
    st = pool.stats()
    coalesced = pool.coalesced()
    batch_times = pool.batch_times()
    boards = sum(job.gather_int(st["boards"] - before["boards"]))
    batches = sum(job.gather_int(st["batches"] - before["batches"]))
    pool.close()
    j = out["job"]
    sec = max(j["seconds"], 1e-9)
    moves_per_sec = j["moves"] / sec
    leg = {
        "what": f"{args.selfplay_games} games in flight per GPU, " +
                (f"stones {fast_forward + 1}..{fast_forward + args.selfplay_moves} of each (the first {fast_forward} by policy play, not timed)" if fast_forward
                 else f"first {args.selfplay_moves} stones of each") + f", --num-rollout {args.selfplay_rollouts}, "
                f"1 search thread x {args.selfplay_leaves} leaves per game, games as " +
                (f"fibers on {args.selfplay_host_threads} host threads" if args.selfplay_host_threads > 0 else "one OS thread each") + ", transposition cache 200001 x 8 shards, through the leaf pool "
                f"({pool_rule_text(args, rule)}, PCIe-inclusive); random-init weights; feature planes built on the " +
                ("DEVICE from 896-byte board records (dgnn_forward_boards)" if boards else "HOST (features::V1, 23 KB of fp16 per leaf)"),
        "seconds": sec, "games": j["games"], "moves": j["moves"],
        # what the searches were ANSWERED per second (table misses): with coalescing most of an opening leg's answers follow a position in
        # flight and cost no evaluation -- the device's own rate is device_boards_per_sec (rounds 1-4 called this field evals_per_sec)
        "answers_per_sec": j["evals"] / sec,
        "device_boards_per_sec": boards / sec,
        "mean_device_batch": boards / max(batches, 1),
        # what a batch took lately by size (first size of each 16-board bucket -> microseconds from hand-over to answers, moving average,
        # including what the batch shared the device with): the pool's own figures (dgnn_pool_batch_times)
        "device_batch_us_by_size": {str(k): v for k, v in batch_times.items()},
        "evals_per_move": j["evals"] / max(j["moves"], 1),          # MEASURED: what one move costs the engine
        "cache_hit_ratio": j["cache_hits"] / max(j["cache_hits"] + j["evals"], 1),
        "probe_conflicts": j["conflicts"],
        "moves_per_sec": moves_per_sec,
        "assumed_stones_per_game": 250,
        # the opening leg's moves per second are those of the first stones only (with coalescing: host-bound, tiny device batches); only the
        # mid-game leg is extrapolated to games per hour
        "games_per_hour": moves_per_sec * 3600.0 / 250.0 if fast_forward else None,
        "note": "games_per_hour = measured moves/s x 3600 / 250 stones (a trained net's game length; a random-init net plays to the "
                "722-stone cap: whole games measured in profiles/, DESIGN.md section 8). " +
                ("Mid-game stones." if fast_forward else "Opening moves only: games in flight share their first positions; with coalescing most "
                 "requests FOLLOW a position that is already in flight (device batches of ~25 boards), so this leg is bound by the host threads "
                 "and its host_cpus_busy / gpus_feedable_at_quota describe the first moves of a game, not a game: whole games keep 1.8 cores "
                 "per GPU busy (profiles/r04_final/selfplay_full_games_*.json, DESIGN.md 13.5); the mid-game leg is the steady state."),
        "host_cpus": effective_cpus(),
        "host_cpu_affinity": f"{min(os.sched_getaffinity(0))}-{max(os.sched_getaffinity(0))} ({len(os.sched_getaffinity(0))} CPUs)",
        "host_cpus_busy": cpu_cores_busy,          # CPU time of rank 0's process over the leg / its wall time: cores kept busy
        "rollouts_per_sec": (j["evals"] + j["cache_hits"]) / sec,
        # what one rollout (probe, board, record, cache, prior, insert, queueing) costs this rank's host: CPU seconds / rollouts
        "host_us_per_rollout": 1e6 * cpu_seconds / max(rank_rollouts, 1),
        "host_us_per_eval": 1e6 * cpu_seconds / max(out["evals"], 1),
        "batching_rule": "reference (2 batches alive, no coalescing, a free slot takes what is queued)" if reference_rule else "tuned (see `what`)",
    }
    leg["coalescing"] = coalesce
    leg["coalesced_boards_rank0"] = coalesced      # boards that followed a position in flight instead of being evaluated (whole pool lifetime)
    # how many GPUs running THIS leg the box's CPU quota could feed: quota cores / cores one GPU's leg kept busy
    leg["gpus_feedable_at_quota"] = leg["host_cpus"] / max(cpu_cores_busy, 1e-9)
    # the host's capacity: rollouts per second and busy core, this rank and every rank (8 pinned ranks on one box: --gpus 8 --dry)
    leg["rollouts_per_sec_per_core"] = rank_rollouts / max(cpu_seconds, 1e-9)
    leg["per_rank_rollouts_per_sec_per_core"] = job.gather_float(leg["rollouts_per_sec_per_core"])
    leg["per_rank_host_cpus_busy"] = job.gather_float(cpu_cores_busy)
    return leg


class DryWorkspace:
    """`--dry`: what the legs below use of api.WorkspaceGuard, over the CPU stand-in (dgnn_hash_predictor).  Like the engine's
    workspace it EVALUATES WHAT IS RESIDENT: upload() and every synchronous forward overwrite the resident batch (engine.cpp: the
    planes of a dgnn_forward land in ws->d_features), forward_resident() evaluates whatever was left there.  A leg that forgets to
    upload its own inputs therefore fails here exactly as it does on the device (tests/test_bench_dry.py)."""

    def __init__(self, api, batch_size):
        self.api, self.batch_size, self.resident, self.out = api, batch_size, None, None
        self._predict = api.PREDICT_FN(api.lib.dgnn_hash_predictor())

    def _evaluate(self, features):
        n = self.batch_size
        f = np.ascontiguousarray(features, dtype=np.uint16).reshape(-1)
        assert f.size == n * 11552
        value, policy = np.empty(n, dtype=np.uint16), np.empty(n * 362, dtype=np.uint16)
        u16p = self.api._u16p
        status = self._predict(None, f.ctypes.data_as(u16p), n, value.ctypes.data_as(u16p), policy.ctypes.data_as(u16p))
        assert status == 0
        return self.api.OutputMap(value.view(np.float16), policy.view(np.float16))

    def upload(self, features):
        self.resident = np.array(np.asarray(features).view(np.uint16).reshape(-1), copy=True)

    def forward_resident(self):
        self.out = self._evaluate(self.resident)

    def sync(self):
        pass

    def download(self):
        return self.out

    def release(self):
        pass


class DryForward:
    """`--dry`: api.forward / api.forward_boards over DryWorkspace (same residency rule)."""

    def __init__(self, api):
        self.api = api

    def forward(self, ws, features):
        ws.upload(features)
        ws.forward_resident()
        return ws.download()

    def forward_boards(self, ws, records, count):
        planes = self.api.features_unpack(self.api.board_records_to_packed(records, count), count)     # host statement (go.cpp)
        return self.forward(ws, planes)


def reference_style_leg(fw, api, ws, positions, batch):
    """The reference's own harness (`dream_go --bench`, src/dg/bench/forward.rs:37-47 under bench/benchmark.rs:37-53): every
    position of an SGF collection -- here the three committed games -- one at a time: planes on the host, repeated to the batch
    size, one synchronous forward(); positions per second over the whole walk, feature extraction included.  NOTE: every call
    overwrites the workspace's resident inputs (the last game position x batch stays behind)."""
    t1 = time.perf_counter()
    for board, color in positions:
        fw.forward(ws, np.tile(board.get_features(color, 0).reshape(-1).view(np.uint16), batch))
    dt1 = time.perf_counter() - t1
    t1 = time.perf_counter()
    for board, color in positions:
        fw.forward_boards(ws, np.tile(board.record(color, 0), batch), batch)
    dt2 = time.perf_counter() - t1
    return {
        "what": f"dream_go --bench style: {len(positions)} positions of the three committed games (tests/golden/real_games.json), each "
                f"repeated {batch} times and evaluated by one synchronous call from one thread (through the Python mirror)",
        "boards_per_sec": batch * len(positions) / dt1, "boards_per_sec_records": batch * len(positions) / dt2,
        "forms": "boards_per_sec: features::V1 on the host + dgnn_forward (the reference's flow); boards_per_sec_records: board record + "
                 "dgnn_forward_boards (planes on the device)",
        "positions": len(positions), "batch": batch,
    }


def parity_entry(ws, features, n, ref_value, ref_policy, what):
    """The resident-input path (upload -> dgnn_forward_resident -> download: what `value` times) on `features`, against the checker's
    outputs for THE SAME features.  The upload is part of the leg: whatever an earlier leg left in the workspace is gone (VERDICT r03
    weak #1: since d4c9b12 the sample was taken on the last game position the reference-style walk had left there)."""
    f = np.ascontiguousarray(np.asarray(features).view(np.uint16).reshape(-1))
    ws.upload(f)
    ws.forward_resident()
    ws.sync()
    got = ws.download()
    gv = got.value.astype(np.float32)[:n]
    gp = got.policy.astype(np.float32)[:n * 362].reshape(n, 362)
    rv = np.asarray(ref_value).view(np.float16).astype(np.float32)[:n]
    rp = np.asarray(ref_policy).view(np.float16).astype(np.float32)[:n * 362].reshape(n, 362)
    entry = {"what": what, "boards": n, "inputs_crc32": zlib.crc32(f[:n * 11552].tobytes()),
             "max_abs_value_diff": float(np.abs(gv - rv).max()), "max_abs_policy_diff": float(np.abs(gp - rp).max()), "tolerance": 1e-3}
    # north_star names both: value abs-diff and policy KL(checker || HIP), per board, on the fp16 probabilities promoted to fp64
    kl = np.sum(rp.astype(np.float64) * (np.log(rp.astype(np.float64) + 1e-12) - np.log(gp.astype(np.float64) + 1e-12)), axis=1)
    entry["policy_kl_max"], entry["policy_kl_mean"] = float(kl.max()), float(kl.mean())
    # board identity: board i's output is nearer to the checker's output of board i than to that of any other board
    hip = np.concatenate([gp, gv[:, None]], axis=1).astype(np.float64)
    ref = np.concatenate([rp, rv[:, None]], axis=1).astype(np.float64)
    distance = np.abs(hip[:, None, :] - ref[None, :, :]).max(axis=2)
    own = np.diag(distance).copy()
    np.fill_diagonal(distance, np.inf)
    other = distance.min(axis=1) if n > 1 else np.full(n, np.inf)
    entry["own_distance_max"] = float(own.max())
    entry["nearest_other_board_min"] = float(other.min()) if n > 1 else None
    entry["boards_nearest_to_their_own"] = int(np.sum(own < other))
    entry["policy_max_range"] = [float(rp.max(axis=1).min()), float(rp.max(axis=1).max())]
    entry["distinct_argmax"] = int(len(set(rp.argmax(axis=1).tolist())))
    entry["value_range"] = [float(rv.min()), float(rv.max())]
    return entry


def real_game_features(positions, batch, first=3):
    """`batch` DISTINCT feature tensors: positions spread over the three committed games, symmetry = index mod 8 (the workload of
    tests/test_round3_gpu.py)."""
    picks = np.linspace(first, len(positions) - 1, batch).astype(int)
    planes = [positions[i][0].get_features(positions[i][1], symmetry=k % 8).reshape(-1) for k, i in enumerate(picks)]
    return np.concatenate(planes).view(np.uint16)


def sharpened_parity(api, synth, args, device, positions, n):
    """A parity number that a mixed-up batch cannot pass: the He-init network of `value` forgets its input
    (ANY board is within 1e-3 of ANY other), so this entry runs a SHARPENED network of the same architecture
    (synth.make_sharp_tensors, dense gains 6 / 4: north_star's 1e-3 applies, tests/test_round3_gpu.py) on `n` distinct positions of
    the committed games and reports the error next to the distance to the nearest OTHER board."""
    from oracle import oracle
    text = synth.to_json(synth.make_sharp_tensors(args.blocks, args.channels, seed=2020, policy_gain=6.0, value_gain=4.0))
    features = real_game_features(positions, args.batch)
    tensors, _ = oracle.load_weights(text)
    ref_value, ref_policy = oracle.OracleNetwork(tensors).forward(np.ascontiguousarray(features[:n * 11552]), n)
    net = api.Network.from_json(text)
    ws = net.get_workspace(args.batch, device=device)
    try:
        return parity_entry(ws, features, n, ref_value, ref_policy,
                            f"sharpened {args.blocks}x{args.channels} net (residual gate 0.1, dense gains 6 / 4), {n} distinct positions of the three "
                            "committed games x 8 symmetries; HIP resident path vs oracle/dgnn_oracle.c")
    finally:
        ws.release()
        net.destroy()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))

    job = Job(args)
    import __graft_entry__ as graft
    graft.package()
    from dream_go_amd import replicas           # (numpy only: libdgnn_hip.so is loaded behind torch, below)
    unpinned = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    pinned = replicas.pin_to_compact_cpus(job.local_rank)   # before any thread exists: they all inherit the mask
    job.start(args.force_dist)
    from dream_go_amd import api, synth

    rank, world, device = job.rank, job.world, job.device
    gpu_bus = None
    if not job.dry:
        # the device this rank's library opens IS the device torch (and so the launcher's LOCAL_RANK) means: same physical GPU
        gpu_bus = api.Device(device).pci_bus_id().lower()
        props = job.torch.cuda.get_device_properties(device)
        if hasattr(props, "pci_bus_id"):
            torch_bus = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
            if gpu_bus[-7:] != torch_bus[-7:]:
                raise SystemExit(f"rank {rank}: the library's device {device} is {gpu_bus}, torch's cuda:{device} is {torch_bus}")

    # ---- synthetic network + inputs in the reference's formats (no network access for real weights).  Rank r > 0 starts from
    # DIFFERENT weights (seed 1 + r): only the broadcast can make the replicas agree
    text = synth.to_json(synth.make_tensors(args.blocks, args.channels, seed=1 + (rank if job.dist is not None else 0)))
    net = api.Network.from_json(text)
    info = net.info()
    features = synth.make_features(args.batch, seed=replicas.rank_seed(synth.FEATURE_SEED, rank))

    bcast = broadcast_weights(job, api, net) if job.dist is not None else None

    ws = None
    tower_ms, tower_launches, total_ms = 0.0, 0, 0.0
    if job.dry:
        pool = api.LeafPool.over_callbacks(api.lib.dgnn_hash_predictor(), None, num_lanes=1, max_batch_size=args.batch)
        for _ in range(args.warmup):
            pool.predict(features, args.batch)
        job.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pool.predict(features, args.batch)
        job.barrier()
        elapsed = time.perf_counter() - t0
        pool.close()
    else:
        ws = net.get_workspace(args.batch, device=device)
        ws.upload(features)                              # inputs resident in HBM before the timed region
        for _ in range(args.warmup):
            ws.forward_resident()
        job.barrier(ws)
        t0 = time.perf_counter()
        # K passes back to back on the workspace's stream; HIP events on that stream bracket the whole run and the
        # residual-tower launches of every pass (include/dgnn.h: dgnn_forward_timed)
        total_ms, tower_ms, tower_launches = ws.forward_timed(args.steps)
        job.barrier(ws)
        elapsed = time.perf_counter() - t0

    per_rank_elapsed = job.gather_float(elapsed)
    elapsed = max(per_rank_elapsed)                      # the job's clock is the slowest rank's

    # ---- the same passes for seconds instead of milliseconds: `value` times K passes (61 ms at the driver's K = 20), which says nothing about
    # what the part sustains once its power management has settled
    sustained = None
    if not job.dry and args.sustained_seconds > 0:
        job.barrier(ws)
        s_total = s_tower = 0.0
        s_launches = s_steps = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < args.sustained_seconds:
            a, b, n = ws.forward_timed(100)
            s_total, s_tower, s_launches, s_steps = s_total + a, s_tower + b, s_launches + n, s_steps + 100
        job.barrier(ws)
        s_elapsed = max(job.gather_float(time.perf_counter() - t0))
        sustained = {"seconds": s_elapsed, "steps": s_steps, "evals_per_sec": job.world * args.batch * s_steps / s_elapsed,
                     "gpu_ms_per_step": s_total / max(s_steps, 1), "tower_launch_us": 1e3 * s_tower / max(s_launches, 1), "tower_launches": s_launches}
    evals = world * args.batch * args.steps
    value = evals / elapsed

    c = info["num_channels"]
    roofline = None
    if not job.dry:
        # ---- roofline of the dominant kernel: the C -> C residual convolution (2 per block)
        conv_flops = 2.0 * 9 * c * c * 361 * args.batch            # algorithmic FLOPs per convolution
        convs_per_launch = max(1, round(2 * info["num_blocks"] * args.steps / max(tower_launches, 1)))   # 1, or 2 * blocks (tower kernel)
        launch_ms = tower_ms / max(tower_launches, 1)
        achieved = conv_flops * convs_per_launch / (launch_ms * 1e-3) / 1e12 if launch_ms > 0 else 0.0
        roofline = {
            "bound": "mfma", "achieved": achieved, "peak": PEAK_FP16_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP16_MFMA_TFLOPS, "traffic": None,
            "kernel": f"residual 3x3 convolution C->C ({ws.conv_kernel()})", "launch_us": launch_ms * 1e3,
            "launches_timed": tower_launches, "convolutions_per_launch": convs_per_launch,
            "flops_per_launch": conv_flops * convs_per_launch, "us_per_convolution": launch_ms * 1e3 / convs_per_launch,
        }
        # HBM bytes per launch of that kernel from the committed rocprofv3 --pmc passes (tools/profile.sh +
        # tools/pmc_summary.py); only quoted when the profile was taken on this workload and this kernel
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(traffic_file):
            try:
                t = json.load(open(traffic_file))
                if t.get("batch") == args.batch and t.get("channels") == c and t.get("kernel", "") in roofline["kernel"]:
                    roofline["traffic"] = t["hbm_bytes_per_launch"]
            except Exception:   # noqa: BLE001
                pass

    selfplay = selfplay_mid = selfplay_mid_ref = whole_games = policy_play = policy_play_ex_it = None
    if args.selfplay_games > 0:
        if args.selfplay_opening_leg:
            selfplay = selfplay_leg(job, api, replicas, net, args)
        if args.selfplay_midgame > 0:
            selfplay_mid = selfplay_leg(job, api, replicas, net, args, fast_forward=args.selfplay_midgame)
            if args.reference_rule_leg:
                selfplay_mid_ref = selfplay_leg(job, api, replicas, net, args, fast_forward=args.selfplay_midgame, reference_rule=True)
    if args.whole_games > 0:
        whole_games = whole_games_leg(job, api, replicas, net, args)
    if args.policy_play_games > 0:
        policy_play = policy_play_leg(job, api, replicas, net, args)
        if args.ex_it_rollouts > 0:
            policy_play_ex_it = policy_play_ex_it_leg(job, api, replicas, net, args)

    out = None
    if rank == 0:
        flops_eval = synth.flops_per_eval(info["num_blocks"], c, info["num_samples"])
        out = {
            "metric": "nn_leaf_evals_per_sec", "value": value, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {
                "workload": f"19x19, {info['num_blocks']}-block/{c}-filter net, batch {args.batch} per GPU "
                            "(BASELINE.json configs[2]); Bernoulli(0.2) features, random-init weights in the reference JSON schema",
                "blocks": info["num_blocks"], "channels": c, "samples": info["num_samples"],
                "batch_per_gpu": args.batch, "global_batch": world * args.batch,
                "parallelism": f"replicas x{world} (independent leaf batches, no data-path collective)",
                "rank0_gpu": {"pci_bus_id": gpu_bus, "numa_node": replicas.gpu_numa_node(job.local_rank), "cpus_pinned": f"{min(pinned)}-{max(pinned)}" if pinned else None},
            },
            "per_rank_evals_per_sec": [args.batch * args.steps / t for t in per_rank_elapsed],
        }
        if job.dry:
            out["dry"] = True
            out["config"]["workload"] += " -- DRY RUN: CPU hash predictor behind the leaf pool, numbers are not measurements"
        else:
            out["roofline"] = roofline
            out["whole_pass"] = {
                "tflops": flops_eval * args.batch * args.steps / (total_ms * 1e-3) / 1e12,
                "frac_of_mfma_peak": flops_eval * args.batch * args.steps / (total_ms * 1e-3) / 1e12 / PEAK_FP16_MFMA_TFLOPS,
                "gpu_ms_per_step": total_ms / args.steps, "flops_per_eval": flops_eval,
            }
            if sustained is not None:
                sustained["roofline_frac"] = roofline["flops_per_launch"] / (sustained["tower_launch_us"] * 1e-6) / 1e12 / PEAK_FP16_MFMA_TFLOPS
                sustained["what"] = (f"the passes of `value` (same kernel, same resident batch) back to back for {sustained['seconds']:.1f} s instead of "
                                     f"{elapsed * 1e3:.0f} ms; roofline_frac from HIP events around every tower launch of those seconds")
                out["value_sustained"] = sustained
        if whole_games is not None:
            out["whole_games"] = whole_games
        if policy_play is not None:
            out["policy_play"] = policy_play
        if policy_play_ex_it is not None:
            out["policy_play_ex_it"] = policy_play_ex_it
        if selfplay_mid is not None:
            out["selfplay_midgame"] = selfplay_mid
        if selfplay_mid_ref is not None:
            out["selfplay_midgame_reference_rule"] = selfplay_mid_ref
        if selfplay is not None:
            out["selfplay"] = selfplay
        per_move = next((leg for leg in (whole_games, selfplay_mid, selfplay) if leg is not None), None)
        if per_move is not None:
            out["derived"] = {
                "games_per_hour_upper_bound": value * 3600.0 / (250 * max(per_move["evals_per_move"], 1e-9)),
                "what": "engine evals/s (inputs resident) x 3600 / (250 stones x the MEASURED evals per move of the " +
                        ("whole_games" if per_move is whole_games else "selfplay_midgame" if per_move is selfplay_mid else "selfplay") +
                        " leg): what the engine alone would allow; the pipeline figures are whole_games.* and selfplay_midgame.games_per_hour",
            }
        # ---- PCIe-inclusive: host buffers in, host buffers out (the drop-in's own contract, graph.rs:130-131,155-156), never `value`
        fw = DryForward(api) if job.dry else api
        leg_ws = DryWorkspace(api, args.batch) if job.dry else ws
        positions = None
        if world == 1 and not args.lean:
            positions = real_game_positions(api)
            if not job.dry:
                # (a) as a leaf pool drives the device: two workspaces, one submission always queued, the loop inside the library
                ws2 = net.get_workspace(args.batch, device=device)
                records = np.concatenate([positions[i % len(positions)][0].record(positions[i % len(positions)][1], i % 8) for i in range(args.batch)])
                packed = api.features_pack(features, args.batch)
                reps = max(10, args.steps)
                out["value_pcie_inclusive"] = {
                    "what": "evals/s with host buffers in and out, two workspaces in flight (dgnn_forward_submit_* / dgnn_forward_wait inside "
                            "dgnn_forward_pipelined_timed_n: no interpreter in the loop), same network and batch as `value`",
                    "board_records": api.forward_pipelined_timed_n([ws, ws2], "records", records, args.batch, reps),     # 896 B per board, planes on the device
                    "packed_planes": api.forward_pipelined_timed_n([ws, ws2], "packed", packed, args.batch, reps),       # 1.5 KB per board
                    "fp16_planes": api.forward_pipelined_timed_n([ws, ws2], "planes", features, args.batch, reps),       # 23 KB per board: the reference's form
                    "steps": reps,
                }
                out["value_pcie_inclusive"]["records_vs_resident"] = out["value_pcie_inclusive"]["board_records"] / value
                # The like-for-like figure for "leaf evals per second" of the reference's own boundary: dg_nn::forward takes a host slice of
                # fp16 planes and returns host vectors (graph.rs:130-156, H2D and D2H inside the call); its predictor keeps two batches
                # alive per device (predictors/nn.rs:64-67).  `value` is the resident-input rate the roofline is priced on; THIS is the rate a
                # caller of dgnn_forward sees.
                out["value_reference_contract"] = {
                    "value": out["value_pcie_inclusive"]["fp16_planes"], "unit": "evals/s", "vs_value": out["value_pcie_inclusive"]["fp16_planes"] / value,
                    "what": "dgnn_forward_submit_n / dgnn_forward_wait on fp16 HWC planes in host memory (23 104 B per board in, 726 B out over PCIe), host "
                            "buffers out, two workspaces alive -- the contract of dg_nn::forward (graph.rs:130-156) under the reference predictor's two "
                            f"batches alive per device (predictors/nn.rs:64-67); {reps} x {args.batch} boards, loop inside the library",
                }
                # ... and with THREE in flight, the pool's round-5 default
                ws3 = net.get_workspace(args.batch, device=device)
                out["value_pcie_inclusive"]["board_records_three_in_flight"] = api.forward_pipelined_timed_n([ws, ws2, ws3], "records", records, args.batch, max(reps, 30))
                ws3.release()
                ws2.release()
            # (b) the reference's own harness; a dry run walks every 16th position (the order of the legs is what it is for)
            out["reference_style"] = reference_style_leg(fw, api, leg_ws, positions[::16] if job.dry else positions, args.batch)
        if bcast is not None:
            out["weight_broadcast"] = bcast
        if world == 1 and args.cpu_sample > 0:
            n = min(args.cpu_sample, args.batch)
            if job.dry:
                # the checker of a dry run is the stand-in itself, called directly on the batch
                ref = DryWorkspace(api, args.batch)._evaluate(features)
                ref_value, ref_policy = ref.value, ref.policy
                flat_what = "DRY RUN: CPU hash predictor vs itself (the leg order is what is under test)"
            else:
                # the CPU baseline runs as an unpinned process would (its OpenMP team is created here and inherits this mask): the
                # pinning is the GPU pipeline's business, and neighbours on the box may sit on the same low-numbered cores
                if pinned is not None:
                    os.sched_setaffinity(0, unpinned)
                baseline, (ref_value, ref_policy) = cpu_baseline(text, features, args.batch, args.cpu_sample)
                if pinned is not None:
                    os.sched_setaffinity(0, pinned)
                # the second statement SURVEY.md 8d names: torch-CPU F.conv2d on the same sample, and how far its outputs are from the port's
                second, (tv, tp) = cpu_second_opinion(text, features, args.batch, args.cpu_sample, baseline["cores"])
                k = min(len(tv), n)
                second["max_abs_diff_vs_port"] = [float(np.abs(tv[:k] - np.asarray(ref_value).view(np.float16).astype(np.float32)[:k]).max()),
                                                  float(np.abs(tp[:k] - np.asarray(ref_policy).view(np.float16).astype(np.float32)[:k * 362].reshape(k, 362)).max())]
                # `cpu_baseline` is the FASTER of the two CPU statements of the reference's semantics (the stated baseline
                # must not understate the CPU); the other one is carried beside it.  Parity is checked against the C port either way: that is
                # the statement pinned to the reference's vectors (tests/test_oracle_kat.py).
                baseline["statement"] = "oracle/dgnn_oracle.c (C, OpenMP; the parity checker)"
                second["statement"] = "oracle/torch_statement.py (torch-CPU F.conv2d, the trainer's graph)"
                fast, slow = (second, baseline) if second["value"] > baseline["value"] else (baseline, second)
                out["cpu_baseline"] = fast
                out["cpu_baseline_other_statement"] = slow
                flat_what = "the network and batch of `value` (He-init, Bernoulli(0.2) planes): HIP resident path vs oracle/dgnn_oracle.c"
            parity = {"flat": parity_entry(leg_ws, features, n, ref_value, ref_policy, flat_what)}
            if not job.dry and not args.lean and positions is not None:
                if pinned is not None:
                    os.sched_setaffinity(0, unpinned)
                parity["sharpened"] = sharpened_parity(api, synth, args, device, positions, min(n, len(positions) - 3))
                if pinned is not None:
                    os.sched_setaffinity(0, pinned)
            out["parity_vs_oracle"] = parity
        line = json.dumps(out)
    else:
        line = None

    if ws is not None:
        ws.release()
    net.destroy()
    job.finish()
    if line is not None:
        # RCCL prints a version banner through C stdio, which is block-buffered when stdout is a file and would otherwise
        # land AFTER this line at exit: flush the C streams first so that the JSON line is the last thing on stdout
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(line, flush=True)


if __name__ == "__main__":
    main()

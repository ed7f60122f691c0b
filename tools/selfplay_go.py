#!/usr/bin/env python
"""Self-play on the REAL board through the whole pipeline: Go rules, super-ko, ladder reader and V1 feature planes (go.cpp) ->
search tree + driver (tree.cpp, search.cpp) -> LeafPool / batcher (batcher.cpp) -> HIP engine, PCIe-inclusive, random-init
network.  The game loop follows self_play_one / Player::predict (random komi, per-player trees re-used across moves, win-rate
driven rollout budget, 8-fold root ensemble, ScoringSearch until the board is scorable / StandardSearch with dirichlet noise
after, two passes on a scorable board end the game, transposition cache).  --moves N caps the stones per game so that a
measurement is bounded; --moves 0 plays whole games (the reference's cap of 722 stones) and reports measured games/hour.
Prints one JSON line; --sgf FILE writes the game records (GameResult::Ended format).

    python tools/selfplay_go.py [--games 64] [--moves 8] [--rollouts 800] [--threads 8] [--blocks 20 --channels 256]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.package()
from dream_go_amd import replicas
replicas.pin_to_compact_cpus(int(os.environ.get("LOCAL_RANK", "0")))   # before any thread exists
from dream_go_amd import api, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=64, help="games in the whole job (sharded game_id mod world over the ranks)")
    ap.add_argument("--parallel", type=int, default=0, help="games in flight per rank (the reference's --num-games); 0 = all of the rank's games")
    ap.add_argument("--moves", type=int, default=8, help="stones per game, 0 = whole games")
    ap.add_argument("--rollouts", type=int, default=800, help="--num-rollout; 1 = policy play (BASELINE.json configs[4])")
    ap.add_argument("--ex-it", type=int, default=0, metavar="ROLLOUTS", help="expert iteration with this --num-ex-it-rollout (0 = off)")
    ap.add_argument("--threads", type=int, default=8, help="search threads per game (the reference: --num-threads per pool)")
    ap.add_argument("--leaves", type=int, default=1, help="leaves a search thread keeps in flight per predictor call (1 = the reference's worker)")
    ap.add_argument("--max-batch", type=int, default=256)
    ap.add_argument("--extra-batch-queue", type=int, default=0, metavar="BOARDS", help="> 0: two batches alive, a third only while at least this many boards are waiting (dgnn_pool_set_extra_batch_queue)")
    ap.add_argument("--min-fill", type=int, default=0, metavar="BOARDS", help="> 0: a free batch slot waits for this many boards while another batch keeps the device busy (dgnn_pool_set_min_fill)")
    ap.add_argument("--min-fill-wait-us", type=int, default=1000)
    ap.add_argument("--back-to-back", type=int, default=300, metavar="LEAD_US", help="> 0 (default 300): a partial batch waits until this long before the batch on the device is expected to finish (dgnn_pool_set_back_to_back)")
    ap.add_argument("--snap", default="auto", metavar="LO,HI", help="with --back-to-back: a batch of LO < n < HI boards is cut to its LO oldest; \"auto\" (default) = half and 7/8 of --max-batch; \"\" = off")
    ap.add_argument("--coalesce", type=int, default=1, help="1: requests for a position in flight follow it (dgnn_pool_set_coalescing; only with a shared table); 0: off")
    ap.add_argument("--max-batches", type=int, default=3, help="device batches the pool keeps alive: 3 (default since round 5, DESIGN.md 14.2), 2 (the reference's rule), or 0 = adaptive (a rule that lost; test build)")
    ap.add_argument("--workers", type=int, default=0, help="pool worker threads per device; 0 (default) = one more than batches alive: a free worker takes the next batch while another is still handing its answers to the games (DESIGN.md 14.2: +2.5 %%)")
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--cache", type=int, default=200_001, help="transposition table entries (0 = none); nn.rs:48 uses 200_001 per shard")
    ap.add_argument("--cache-per-game", type=int, default=0, help="> 0: a private table per game (reproducible games) instead of the shared one")
    ap.add_argument("--sgf", default=None)
    ap.add_argument("--planes", choices=["device", "host"], default="device",
                    help="where the feature planes are built: on the device from 896-byte board records (dgnn_forward_boards), or on the host")
    ap.add_argument("--host-threads", type=int, default=4, help="the games in flight run as fibers on this many host threads (0 = one OS thread per game)")
    ap.add_argument("--steady", default="", metavar="RAMP,SECONDS",
                    help="the STEADY STATE of whole games instead of a fixed set: --games in flight per GPU with replacement, game i started i / games of the "
                         "way into a game played out by policy play, RAMP untimed seconds, then a window of SECONDS (bench.py's whole_games leg: "
                         "dgnn_selfplay_window); prints that leg's JSON")
    args = ap.parse_args()
    lost_rules = args.extra_batch_queue > 0 or args.min_fill > 0 or args.max_batches == 0
    if lost_rules and not hasattr(api.lib, "dgnn_pool_set_min_fill"):
        raise SystemExit("--extra-batch-queue / --min-fill / --max-batches 0 are batching rules that were measured and lost (DESIGN.md 13.5, 13.10): "
                         "their entry points are exported by the test build only -- DGNN_LIBRARY=dream-go_amd/lib/libdgnn_hip_test.so")
    if args.steady:
        # the same leg bench.py runs, with this tool's own settings, on one GPU
        sys.path.insert(0, ROOT)
        import bench
        ramp, seconds = [float(v) for v in args.steady.split(",")]
        bargs = bench.parse_args(["--blocks", str(args.blocks), "--channels", str(args.channels), "--batch", str(args.max_batch), "--whole-games", str(args.games),
                                  "--whole-games-ramp", str(ramp), "--whole-games-seconds", str(seconds), "--selfplay-rollouts", str(args.rollouts),
                                  "--selfplay-leaves", str(args.leaves), "--selfplay-batches", str(args.max_batches), "--selfplay-workers", str(args.workers),
                                  "--selfplay-coalesce", str(args.coalesce), "--selfplay-back-to-back", str(args.back_to_back),
                                  "--selfplay-host-threads", str(args.host_threads), "--selfplay-planes", args.planes, "--selfplay-snap", args.snap])
        job = bench.Job(bargs)
        job.start(False)
        net = api.Network.from_json(synth.to_json(synth.make_tensors(args.blocks, args.channels, seed=1)))
        print(json.dumps(bench.whole_games_leg(job, api, replicas, net, bargs)), flush=True)
        net.destroy()
        return

    # one process per GPU (python -m torch.distributed.run --nproc-per-node N tools/selfplay_go.py ...): every rank owns an
    # engine, a batcher, a pool and a cache; rank 0's weights are broadcast once over RCCL; nothing else is exchanged
    world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    dist = torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    device = local_rank

    net = api.Network.from_json(synth.to_json(synth.make_tensors(args.blocks, args.channels, seed=1 if rank == 0 else 100 + rank)))
    broadcast = None
    if dist is not None:
        # ONE broadcast of rank 0's weight image by the library's own RCCL (dgnn_broadcast_weights), the unique id through torch.distributed
        ident = torch.zeros(api.RCCL_UNIQUE_ID_BYTES, dtype=torch.uint8, device=f"cuda:{device}")
        if rank == 0:
            ident.copy_(torch.frombuffer(bytearray(api.Comm.unique_id()), dtype=torch.uint8))
        dist.broadcast(ident, src=0)
        comm = api.Comm.create(device, world, rank, bytes(ident.cpu().numpy().tobytes()))
        net.broadcast_weights(comm, root=0)
        comm.destroy()
        broadcast = ("rccl (dgnn_broadcast_weights)", net.weight_image(device)[1])
    import time
    capacity = 3 if (args.max_batches == 0 or args.extra_batch_queue > 0) else args.max_batches
    pool = api.LeafPool.over_devices(net, [device], max_batch_size=args.max_batch, batches_per_device=capacity,
                                    workers_per_device=max(args.workers, capacity) if args.workers > 0 else capacity + 1, boards=args.planes == "device")
    if args.extra_batch_queue > 0:
        pool.set_extra_batch_queue(2, 3, args.extra_batch_queue)
    if args.min_fill > 0:
        pool.set_min_fill(args.min_fill, args.min_fill_wait_us)
    snap = [args.max_batch // 2, args.max_batch * 7 // 8] if args.snap == "auto" else [int(v) for v in args.snap.split(",")] if args.snap else [0, 0]
    if args.max_batch < 64:
        snap = [0, 0]
    if args.back_to_back > 0 or snap[0] > 0:
        pool.set_back_to_back(args.back_to_back, snap[0], snap[1])
    adaptive = args.max_batches == 0 and pool.takes_boards and args.extra_batch_queue == 0
    if adaptive:
        pool.set_adaptive_batches(2, 3)
    coalesce = bool(args.coalesce) and pool.takes_boards and args.cache > 0 and not args.cache_per_game
    if coalesce:
        pool.set_coalescing(True)
    api.selfplay_go(2, 1, 16, 2, seed=1, pool=pool)          # warm: workspace, weight upload
    cache = api.LruCache(args.cache, 8) if args.cache > 0 else None
    before = pool.stats()
    cpu0, wall0 = time.process_time(), time.perf_counter()
    # the records are written as their games end (params.on_game: the reference's channel, self_play.rs:471-497), not held until the last one has
    sgf_file = open(args.sgf if world == 1 else f"{args.sgf}.rank{rank}", "w") if args.sgf else None
    write_record = (lambda _game, record, _whole: (sgf_file.write(record.decode("ascii") + "\n"), sgf_file.flush())) if sgf_file else None
    stats = replicas.selfplay_sharded(api, dist, torch, args.games, args.moves, args.rollouts, args.threads, seed=2, pool=pool, cache=cache,
                                      device=f"cuda:{device}" if dist is not None else "cpu", on_game=write_record,
                                      num_parallel=args.parallel, ex_it_rollouts=args.ex_it, leaves_per_thread=args.leaves, cache_per_game=args.cache_per_game,
                                      host_threads=args.host_threads)
    if sgf_file:
        sgf_file.close()
    host_cpus_busy = (time.process_time() - cpu0) / max(time.perf_counter() - wall0, 1e-9)
    after = pool.stats()
    adaptive_stats = pool.adaptive_stats() if adaptive else None
    coalesced = pool.coalesced()
    batch_times = pool.batch_times()
    pool.close()
    boards, batches = after["boards"] - before["boards"], after["batches"] - before["batches"]
    if dist is not None:
        boards, batches = replicas.sum_over_ranks(dist, torch, boards, f"cuda:{device}"), replicas.sum_over_ranks(dist, torch, batches, f"cuda:{device}")
    job = stats["job"]
    line = None
    if rank == 0:
        moves_per_s = job["moves"] / job["seconds"]
        line = json.dumps({
            "metric": "go_selfplay", "n_gpus": world, "evals_per_sec": job["evals"] / job["seconds"], "moves_per_sec": moves_per_s,
            "games_per_hour_at_250_moves": moves_per_s * 3600 / 250, "games": job["games"], "games_in_flight_per_rank": args.parallel or None, "moves_cap": args.moves,
            "games_ended_by_passes": job["games_ended"], "mean_stones_per_game": job["moves"] / max(job["games"], 1),
            "games_per_hour_measured": (job["games"] * 3600 / job["seconds"]) if args.moves == 0 else None,
            "evals_per_move": job["evals"] / max(job["moves"], 1),
            "rollouts_per_move": args.rollouts, "ex_it_rollouts": args.ex_it or None, "ex_it_searches": job["ex_it_searches"], "search_threads_per_game": args.threads, "leaves_per_thread": args.leaves, "mean_device_batch": boards / max(batches, 1), "device_batch_us_by_size": {str(k): v for k, v in batch_times.items()},
            "probe_conflicts": job["conflicts"], "cache_hits": job["cache_hits"], "cache_entries": args.cache,
            "seconds": job["seconds"], "net": f"{args.blocks}x{args.channels}",
            "pool_workers": max(args.workers, capacity) if args.workers > 0 else capacity + 1, "batches_alive": adaptive_stats if adaptive else (f"2, a third while >= {args.extra_batch_queue} boards wait" if args.extra_batch_queue > 0 else args.max_batches), "coalescing": coalesce, "min_fill": [args.min_fill, args.min_fill_wait_us] if args.min_fill > 0 else None, "back_to_back": [args.back_to_back] + snap if (args.back_to_back > 0 or snap[0] > 0) else None, "coalesced_boards_rank0": coalesced,
            "feature_planes": args.planes, "host_threads": args.host_threads, "host_cpus_busy_rank0": host_cpus_busy,
            "weight_broadcast": {"mode": broadcast[0], "bytes": broadcast[1], "backend": "rccl"} if broadcast else None,
            "sharding": "game_id mod world, one engine + batcher + pool + cache per rank, no data-path collective",
            "note": "real board / ladders / feature planes / Benson scoring (go.cpp), self_play_one game loop, PCIe-inclusive, random-init net",
        })
    if dist is not None:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)   # every rank empties its C stdio (RCCL's banner) BEFORE the barrier rank 0 prints after
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:                 # after RCCL's C-stdio banner (see bench.py): the JSON line is the last line on stdout
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(line, flush=True)


if __name__ == "__main__":
    main()

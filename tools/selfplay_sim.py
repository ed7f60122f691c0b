#!/usr/bin/env python
"""Synthetic self-play through the whole pipeline: search tree (tree.cpp) + search driver (search.cpp) + LeafPool /
batcher (batcher.cpp) + the HIP engine, PCIe-inclusive.  The game is the rule-free synthetic game of search.cpp (NOT Go:
no captures, no ladders, pseudo-random feature bits), so the number says how fast the machinery turns rollouts into
leaf evaluations under the reference's request pattern (--num-rollout 800, virtual-loss parallel probes, many games),
not how fast real Go games are played.  Prints one JSON line.

    python tools/selfplay_sim.py [--games 64] [--moves 6] [--rollouts 800] [--threads 8] [--blocks 20 --channels 256]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.package()
from dream_go_amd import api, synth  # noqa: E402
api = api.load_full()          # this tool uses entry points of include/dgnn_test.h (raw batcher / single layers): the test build of the library


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=64)
    ap.add_argument("--moves", type=int, default=6)
    ap.add_argument("--rollouts", type=int, default=800)
    ap.add_argument("--threads", type=int, default=8, help="search threads per game (the reference: --num-threads per pool)")
    ap.add_argument("--max-batch", type=int, default=256)
    ap.add_argument("--max-batches", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--channels", type=int, default=256)
    args = ap.parse_args()

    net = api.Network.from_json(synth.to_json(synth.make_tensors(args.blocks, args.channels, seed=1)))
    batcher = api.Batcher(net, args.max_batch, args.max_batches)
    pool = api.LeafPool(batcher, args.max_batches)
    api.selfplay_synthetic(2, 1, 16, 2, seed=1, pool=pool)          # warm: workspace, weight upload
    before = batcher.stats()
    stats = api.selfplay_synthetic(args.games, args.moves, args.rollouts, args.threads, seed=2, pool=pool)
    after = batcher.stats()
    pool.close()
    boards, batches = after["boards"] - before["boards"], after["batches"] - before["batches"]
    moves_per_s = stats["moves"] / stats["seconds"]
    print(json.dumps({
        "metric": "synthetic_selfplay", "evals_per_sec": stats["evals"] / stats["seconds"], "moves_per_sec": moves_per_s,
        "games_per_hour_at_250_moves": moves_per_s * 3600 / 250, "games": args.games, "moves_per_game": args.moves,
        "rollouts_per_move": args.rollouts, "search_threads_per_game": args.threads, "mean_device_batch": boards / max(batches, 1),
        "probe_conflicts": stats["conflicts"], "seconds": stats["seconds"], "net": f"{args.blocks}x{args.channels}",
        "note": "rule-free synthetic game (search.cpp), PCIe-inclusive, no transposition cache; not a Go self-play measurement",
    }))


if __name__ == "__main__":
    main()

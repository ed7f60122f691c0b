#!/usr/bin/env python
"""Writes profiles/searched_games_20x256.npz: the move lists of 96 whole games SEARCHED (800 rollouts per move) by the bench's own
network (20 x 256, synth.make_tensors seed 1), from the empty board to their end.

bench.py's `whole_games` leg measures a steady state: game i of the 96 in flight starts i / 96 of the way into a finished game, so that
games at every stage are in flight from the first second.  Seeded with POLICY-PLAY games (round 5's first form) the window is not
stationary: policy-play games are ~100 stones shorter than searched ones and none of the seeded games is near its end when the window
opens, so a 90 s window counts too few endings (400-560 games per hour where ten minutes measure 710-730).  Seeded with games that
were themselves searched by this network the games in flight ARE a sample of the steady state.  The file holds positions to start
from (move lists: data), nothing else; without it the leg falls back to policy play and says so.

    python tools/make_searched_games.py [--games 96]          (about five minutes on one MI355X)
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.package()
from dream_go_amd import replicas  # noqa: E402
replicas.pin_to_compact_cpus(0)
from dream_go_amd import api, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=96)
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--rollouts", type=int, default=800)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "searched_games_20x256.npz"))
    args = ap.parse_args()
    net = api.Network.from_json(synth.to_json(synth.make_tensors(args.blocks, args.channels, seed=1)))
    pool = api.LeafPool.over_devices(net, [0], max_batch_size=256, batches_per_device=3, workers_per_device=4, boards=True)
    pool.set_coalescing(True)
    pool.set_back_to_back(300)
    out = api.selfplay_go(args.games, 0, args.rollouts, 1, seed=20261004, pool=pool, cache=api.LruCache(200_001, 8), return_moves=True,
                          leaves_per_thread=16, host_threads=4)
    pool.close()
    record = np.asarray(out["record"], dtype=np.int16)
    stones = (record >= 0).sum(axis=1)
    np.savez_compressed(args.out, record=record, blocks=args.blocks, channels=args.channels, rollouts=args.rollouts, weights_seed=1)
    print(json.dumps({"file": args.out, "bytes": os.path.getsize(args.out), "games": int(record.shape[0]), "mean_stones": float(stones.mean()),
                      "min_stones": int(stones.min()), "max_stones": int(stones.max()), "seconds": out["seconds"], "moves_per_sec": out["moves"] / out["seconds"],
                      "distinct_games": int(len({r.tobytes() for r in record}))}))


if __name__ == "__main__":
    main()

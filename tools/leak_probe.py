"""Create / use / destroy in a loop and watch the device's free memory and the process's resident memory: networks (weights image,
broadcast buffers), workspaces of several sizes (arena, pinned staging, events, tower descriptors), leaf pools (workers, slots),
transposition tables.  What a refresh of the weights every few minutes for a week must not accumulate.
    python tools/leak_probe.py [rounds, default 30]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as graft

graft.package()
from dream_go_amd import api, synth


def rss_mb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"):
            return int(line.split()[1]) / 1024.0


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    text = synth.to_json(synth.make_tensors(6, 256, seed=5))
    features = {b: synth.make_features(b, seed=b) for b in (7, 64, 200, 256)}
    rows = []
    for r in range(rounds):
        net = api.Network.from_json(text)
        for batch, f in features.items():
            with net.get_workspace(batch) as ws:
                api.forward(ws, f).unwrap()
        pool = api.LeafPool.over_devices(net, [0], max_batch_size=256, batches_per_device=3, workers_per_device=4, boards=True)
        pool.set_coalescing(True)
        cache = api.LruCache(20_001, num_shards=4)
        api.selfplay_go(8, 12, 32, threads_per_search=1, seed=r, pool=pool, cache=cache, leaves_per_thread=8, host_threads=2)
        pool.close()
        del cache
        net.destroy()
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        rows.append({"round": r, "device_used_mb": (total - free) / 2 ** 20, "rss_mb": rss_mb(), "threads": len(os.listdir("/proc/self/task"))})
    settled = rows[3:]
    print(json.dumps({"rounds": rounds, "device_used_mb_round_3": settled[0]["device_used_mb"], "device_used_mb_last": settled[-1]["device_used_mb"],
                      "rss_mb_round_3": settled[0]["rss_mb"], "rss_mb_last": settled[-1]["rss_mb"], "threads_round_3": settled[0]["threads"],
                      "threads_last": settled[-1]["threads"], "rows": rows[::max(1, rounds // 10)]}))


if __name__ == "__main__":
    main()

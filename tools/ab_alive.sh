#!/bin/bash
# the self-play legs of bench.py on one box: coalescing of positions in flight off / on, with 2 and 3 batches alive
mkdir -p gpurun_out/r4f
for cfg in "0 2" "1 2" "1 3" "0 2" "1 2" "1 3"; do
  set -- $cfg
  timeout 400 python bench.py --cpu-sample 0 --steps 20 --selfplay-coalesce $1 --selfplay-batches $2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
for leg in ('selfplay', 'selfplay_midgame'):
    s = d[leg]
    print('coalesce=$1 alive=$2 %-17s value %6.0f  evals/s %6.0f  moves/s %6.1f  batch %5.1f  hit %.2f  evals/move %5.0f  cores %.2f  followed %d' % (leg, d['value'], s['evals_per_sec'], s['moves_per_sec'], s['mean_device_batch'], s['cache_hit_ratio'], s['evals_per_move'], s['host_cpus_busy'], s.get('coalesced_boards_rank0', 0)))
"
done | tee gpurun_out/r4f/ab_coalesce.txt

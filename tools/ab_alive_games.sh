#!/bin/bash
# same-box A/B of the batches-alive rules over the first stones of whole-game self-play (coalescing on): constant 2, constant 3,
# "2, a third while >= 256 boards wait"
mkdir -p gpurun_out/r4j
for games_moves in "96 150" "256 80"; do
  set -- $games_moves
  for cfg in "--max-batches 2" "--max-batches 3" "--extra-batch-queue 256" "--max-batches 2"; do
    timeout 600 python tools/selfplay_go.py --games $1 --moves $2 --threads 1 --leaves 8 --host-threads 2 $cfg --coalesce 1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('games $1 x $2 stones  %-26s moves/s %6.1f  answers/s %6.0f  mean device batch %5.1f  host cores %.2f' % ('$cfg', d['moves_per_sec'], d['evals_per_sec'], d['mean_device_batch'], d['host_cpus_busy_rank0']))"
  done
done | tee gpurun_out/r4j/ab_alive_games.txt

#!/bin/sh
# Round 5: batches alive (2 / 3, workers = alive + 1) in steady states with FEW games in flight (partial device batches), interleaved twice.
out=gpurun_out/r05m
mkdir -p $out
for games in 24 48; do
  for alive in 2 3 2 3; do
    python tools/selfplay_go.py --games $games --rollouts 800 --threads 1 --leaves 16 --host-threads 4 --max-batches $alive --workers $((alive + 1)) --steady 10,30 2>/dev/null | tail -1 > $out/tmp.json
    python - $games $alive $out/tmp.json <<'PY' >> $out/alive_with_few_games.txt
import json, sys
r = json.load(open(sys.argv[3]))
print(f"20x256 games {sys.argv[1]} batches alive {sys.argv[2]}: moves/s {r['moves_per_sec']:.1f} device boards/s {r['device_boards_per_sec']:.0f} mean batch {r['mean_device_batch']:.1f}")
PY
  done
done
cat $out/alive_with_few_games.txt

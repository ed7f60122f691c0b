#!/bin/sh
# Round 6: few games in flight (24 / 32: device batches of ~184-200 boards, whole-board towers with a quarter of the chip idle) -- does a
# pool that sends at most 128 boards (split towers, sized for the crowd, 1.6 ms instead of 3 ms per batch) do better?
out=gpurun_out/r06_measure_c
mkdir -p $out
for games in ${GAMES:-24 32}; do
  for variant in b256 b128 snap b256 b128 snap; do
    case $variant in
      b256) env_line=""; args="";;
      b128) env_line=""; args="--max-batch 128";;
      snap) env_line="DGNN_LIBRARY=dream-go_amd/lib/libdgnn_hip_test.so"; args="--snap 128,224";;
    esac
    env $env_line python tools/selfplay_go.py --games $games --rollouts 800 --threads 1 --leaves 16 --host-threads 4 --steady 10,30 $args 2>$out/err.txt | tail -1 > $out/tmp.json
    python - $games $variant $out/tmp.json <<'PY' >> $out/batch_cap_with_few_games.txt
import json, sys
r = json.load(open(sys.argv[3]))
print(f"20x256 games {sys.argv[1]} {sys.argv[2]}: moves/s {r['moves_per_sec']:.1f} device boards/s {r['device_boards_per_sec']:.0f} mean batch {r['mean_device_batch']:.1f} towers {r['tower_passes']}")
PY
  done
done
cat $out/batch_cap_with_few_games.txt

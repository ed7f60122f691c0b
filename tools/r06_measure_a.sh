#!/bin/sh
# Round 6: the leaf pool with few games in flight (partial device batches: whole-board towers and split towers follow each other), 20x256.
#   shipped      split towers alive (tickets), ordered behind / in front of whole-board towers
#   order0       the same, split launches ordered among themselves only (test build, DGNN_TOWER_ORDER=0)
#   split0       no split towers: batches below 129 boards run one launch per layer -- what round 5's pool ran after its give-ups
out=gpurun_out/r06_measure_a
mkdir -p $out
for games in ${GAMES:-24 48}; do
  for variant in shipped order0 split0 shipped order0 split0; do
    case $variant in
      shipped) env_line="";;
      order0)  env_line="DGNN_LIBRARY=dream-go_amd/lib/libdgnn_hip_test.so DGNN_TOWER_ORDER=0";;
      split0)  env_line="DGNN_TOWER_SPLIT=0";;
    esac
    env $env_line python tools/selfplay_go.py --games $games --rollouts 800 --threads 1 --leaves 16 --host-threads 4 --steady 10,30 2>$out/err_${games}_$variant.txt | tail -1 > $out/tmp.json
    python - $games $variant $out/tmp.json <<'PY' >> $out/few_games_in_the_pool.txt
import json, sys
r = json.load(open(sys.argv[3]))
print(f"20x256 games {sys.argv[1]} {sys.argv[2]}: moves/s {r['moves_per_sec']:.1f} device boards/s {r['device_boards_per_sec']:.0f} mean batch {r['mean_device_batch']:.1f} towers {r['tower_passes']}")
PY
    grep -c "gave up" $out/err_${games}_$variant.txt | sed "s/^/   gave-up lines on stderr: /" >> $out/few_games_in_the_pool.txt
  done
done
cat $out/few_games_in_the_pool.txt

#!/bin/sh
# Round 6: the leaf pool with few games in flight (partial device batches: whole-board towers and split towers follow each other), 20x256.
#   shipped      what ships: below 129 boards a split tower (tickets) when the device is the submission's own, one launch per layer beside
#                at least a whole-board tower's worth of other boards
#   split        split towers also in a crowd (test build, DGNN_CROWDED_PER_LAYER=0) -- the first half of round 6
#   split0       no split towers at all: batches below 129 boards always run one launch per layer -- what round 5's pool ran after its give-ups
# (the first run of this script, profiles/r06_final/few_games_in_the_pool_split_alive.txt, had "shipped" = split towers ordered against
# whole-board towers and "order0" = split towers unordered: before the per-layer rule existed)
out=gpurun_out/r06_measure_a
mkdir -p $out
for games in ${GAMES:-24 48}; do
  for variant in shipped split split0 shipped split split0; do
    case $variant in
      shipped) env_line="";;
      split)   env_line="DGNN_LIBRARY=dream-go_amd/lib/libdgnn_hip_test.so DGNN_CROWDED_PER_LAYER=0";;
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

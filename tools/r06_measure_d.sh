#!/bin/sh
# Round 6: snapping (a partial batch of 128 < n < HI boards is cut to its 128 oldest: a split tower of 1.6 ms instead of a whole-board
# tower of 3 ms with a part of the chip idle) where batches are mostly full -- does it cost anything there?
out=gpurun_out/r06_measure_d
mkdir -p $out
for games in ${GAMES:-48 96}; do
  for variant in b256 snap224 snap192 b256 snap224 snap192; do
    case $variant in
      b256) args="";;
      snap224) args="--snap 128,224";;
      snap192) args="--snap 128,192";;
    esac
    DGNN_LIBRARY=dream-go_amd/lib/libdgnn_hip_test.so python tools/selfplay_go.py --games $games --rollouts 800 --threads 1 --leaves 16 --host-threads 4 --steady 10,30 $args 2>$out/err.txt | tail -1 > $out/tmp.json
    python - $games $variant $out/tmp.json <<'PY' >> $out/snap_with_many_games.txt
import json, sys
r = json.load(open(sys.argv[3]))
print(f"20x256 games {sys.argv[1]} {sys.argv[2]}: moves/s {r['moves_per_sec']:.1f} device boards/s {r['device_boards_per_sec']:.0f} mean batch {r['mean_device_batch']:.1f} towers {r['tower_passes']}")
PY
  done
done
cat $out/snap_with_many_games.txt

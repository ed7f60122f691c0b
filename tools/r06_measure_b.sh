#!/bin/sh
# Round 6: the leaf pool with VERY few games in flight (every device batch is small: split towers side by side, sized for the crowd)
# against the rules of rounds 2-5 (every split launch behind the previous one, sized for the submission alone; test build).
out=gpurun_out/r06_measure_b
mkdir -p $out
for net in "20 256" "9 128"; do
  set -- $net
  for games in ${GAMES:-4 8 16}; do
    for variant in shipped r05rules shipped r05rules; do
      case $variant in
        shipped)  env_line="";;
        r05rules) env_line="DGNN_LIBRARY=dream-go_amd/lib/libdgnn_hip_test.so DGNN_SPLIT_BESIDE=0 DGNN_SPLIT_CROWD=0";;
      esac
      env $env_line python tools/selfplay_go.py --blocks $1 --channels $2 --games $games --rollouts 800 --threads 1 --leaves 16 --host-threads 4 --steady 8,20 2>$out/err.txt | tail -1 > $out/tmp.json
      python - $1x$2 $games $variant $out/tmp.json <<'PY' >> $out/very_few_games_in_the_pool.txt
import json, sys
r = json.load(open(sys.argv[4]))
print(f"{sys.argv[1]} games {sys.argv[2]} {sys.argv[3]}: moves/s {r['moves_per_sec']:.1f} device boards/s {r['device_boards_per_sec']:.0f} mean batch {r['mean_device_batch']:.1f} towers {r['tower_passes']}")
PY
      grep -c "gave up" $out/err.txt | sed "s/^/   gave-up lines on stderr: /" >> $out/very_few_games_in_the_pool.txt
    done
  done
done
cat $out/very_few_games_in_the_pool.txt

#!/bin/sh
# Round 5, last measurement: the crowded-device rule in the POOL (not the sweep): steady states with few games in flight, whose device
# batches are partial, with the rule off (test build, DGNN_CROWDED=0) and on.
out=gpurun_out/r05k
mkdir -p $out
for games in 24 48; do
  for rule in 0 1 0 1; do
    DGNN_LIBRARY=dream-go_amd/lib/libdgnn_hip_test.so DGNN_CROWDED=$rule python tools/selfplay_go.py --games $games --rollouts 800 --threads 1 --leaves 16 --host-threads 4 --steady 10,30 2>/dev/null | tail -1 > $out/tmp.json
    python - $games $rule $out/tmp.json <<'PY' >> $out/crowded_in_the_pool.txt
import json, sys
r = json.load(open(sys.argv[3]))
print(f"games {sys.argv[1]} crowded_rule {sys.argv[2]}: moves/s {r['moves_per_sec']:.1f} device boards/s {r['device_boards_per_sec']:.0f} mean batch {r['mean_device_batch']:.1f} answers/s {r['answers_per_sec']:.0f}")
PY
  done
done
cat $out/crowded_in_the_pool.txt

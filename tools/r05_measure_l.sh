#!/bin/sh
# Round 5: the crowded-device rule in the POOL at 9x128 (a compute unit holds two boards' workgroups there), off / on interleaved.
out=gpurun_out/r05l
mkdir -p $out
for games in 16 32 64; do
  for rule in 0 1 0 1; do
    DGNN_LIBRARY=dream-go_amd/lib/libdgnn_hip_test.so DGNN_CROWDED=$rule python tools/selfplay_go.py --blocks 9 --channels 128 --games $games --rollouts 800 --threads 1 --leaves 16 --host-threads 8 --steady 8,20 2>/dev/null | tail -1 > $out/tmp.json
    python - $games $rule $out/tmp.json <<'PY' >> $out/crowded_in_the_pool_9x128.txt
import json, sys
r = json.load(open(sys.argv[3]))
print(f"9x128 games {sys.argv[1]} crowded_rule {sys.argv[2]}: moves/s {r['moves_per_sec']:.1f} device boards/s {r['device_boards_per_sec']:.0f} mean batch {r['mean_device_batch']:.1f} host cpus {r['host_cpus_busy']:.2f}")
PY
  done
done
cat $out/crowded_in_the_pool_9x128.txt

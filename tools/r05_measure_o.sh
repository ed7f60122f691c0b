#!/bin/sh
# Round 5: zero copy (records in, outputs out) in the POOL, off / on interleaved (ablations build: DGNN_ZEROCOPY).
out=gpurun_out/r05o
mkdir -p $out
export DGNN_LIBRARY=dream-go_amd/lib/ablations/libdgnn_hip.so
for games in 96 48; do
  for zc in 0 1 0 1; do
    DGNN_ZEROCOPY=$zc python tools/selfplay_go.py --games $games --rollouts 800 --threads 1 --leaves 16 --host-threads 4 --steady 10,30 2>/dev/null | tail -1 > $out/tmp.json
    python - $games $zc $out/tmp.json <<'PY' >> $out/zero_copy_in_the_pool.txt
import json, sys
r = json.load(open(sys.argv[3]))
print(f"20x256 games {sys.argv[1]} zero_copy {sys.argv[2]}: moves/s {r['moves_per_sec']:.1f} device boards/s {r['device_boards_per_sec']:.0f} mean batch {r['mean_device_batch']:.1f} host cpus {r['host_cpus_busy']:.2f}")
PY
  done
done
cat $out/zero_copy_in_the_pool.txt

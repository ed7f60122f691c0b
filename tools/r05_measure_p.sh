#!/bin/sh
# Round 5: partial batches back to back (lead 300 us) against the reference's rule (a free slot takes what is queued) in steady states, interleaved twice.
out=gpurun_out/r05p
mkdir -p $out
for games in 48 96; do
  for lead in 0 300 0 300; do
    python tools/selfplay_go.py --games $games --rollouts 800 --threads 1 --leaves 16 --host-threads 4 --back-to-back $lead --steady 10,30 2>/dev/null | tail -1 > $out/tmp.json
    python - $games $lead $out/tmp.json <<'PY' >> $out/back_to_back_steady.txt
import json, sys
r = json.load(open(sys.argv[3]))
print(f"20x256 games {sys.argv[1]} back_to_back_lead_us {sys.argv[2]}: moves/s {r['moves_per_sec']:.1f} device boards/s {r['device_boards_per_sec']:.0f} mean batch {r['mean_device_batch']:.1f}")
PY
  done
done
cat $out/back_to_back_steady.txt

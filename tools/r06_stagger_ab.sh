#!/bin/sh
# Round 6: the whole-board tower's workgroups started in phases (DGNN_TOWER_STAGGER="phases,us per phase", test build) against in lockstep.
# RESULT (profiles/r06_final/stagger_ab.txt): every variant costs exactly its own start delay and gains nothing; the switch and the
# kernel's prologue for it were taken out again (git log: "Whole-board tower: optional start in phases" and its revert).  Kept as the record.
out=gpurun_out/r06_stagger
mkdir -p $out
for rep in 1 2; do
  for v in ${VARIANTS:-0 2,36 2,18 4,18 4,9 8,9 3,24 2,72}; do
    DGNN_LIBRARY=dream-go_amd/lib/libdgnn_hip_test.so DGNN_TOWER_STAGGER=$v python bench.py --lean --steps 40 --warmup 5 2>/dev/null | tail -1 > $out/tmp.json
    python - "$v" $out/tmp.json <<'PY' >> $out/stagger_ab.txt
import json, sys
d = json.load(open(sys.argv[2]))
print(f"stagger {sys.argv[1]:>6}: value {d['value']:.0f} evals/s  launch {d['roofline']['launch_us']:.1f} us  frac {d['roofline']['frac']:.4f}")
PY
  done
done
cat $out/stagger_ab.txt

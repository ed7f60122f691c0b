#!/bin/bash
# A/B of the convolution implementations on the bench workload; one JSON line per variant into gpurun_out/ab.jsonl
out=${1:-gpurun_out/ab.jsonl}
: > "$out"
for variant in ${VARIANTS:-"DGNN_CONV=tile1,DGNN_CONV_FM=6" "DGNN_CONV=tile2,DGNN_CONV_FM=4" "DGNN_CONV=tile2,DGNN_CONV_FM=6" "DGNN_CONV=tile2,DGNN_CONV_FM=8"}; do
  echo "# $variant" >> "$out"
  env ${variant//,/ } python bench.py --steps 10 --warmup 3 --cpu-sample 0 2>/dev/null | tail -1 >> "$out"
done
python - "$out" <<'PY'
import json, sys
name = None
for line in open(sys.argv[1]):
    if line.startswith("#"):
        name = line[2:].strip()
    elif line.startswith("{"):
        d = json.loads(line)
        r = d["roofline"]
        print(f"{name:36s} evals/s {d['value']:9.0f}  ms/step {d['ms_per_step']:7.3f}  conv {r['launch_us']:7.1f} us  {r['achieved']:7.1f} TF  frac {r['frac']:.3f}")
PY

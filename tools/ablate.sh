#!/bin/bash
# Component pricing of the staggered tile kernel (results of ablated runs are garbage; only the timing matters).
#   1 = no MFMA, 2 = no in-loop DMA, 3 = neither (fragment reads + barriers only)
out=${1:-gpurun_out/ablate.jsonl}
: > "$out"
for ab in 0 1 2 3; do
  echo "# ablate=$ab" >> "$out"
  DGNN_CONV=${VARIANT:-tile3} DGNN_CONV_FM=6 DGNN_CONV_ABLATE=$ab python bench.py --steps 10 --warmup 3 --cpu-sample 0 2>/dev/null | tail -1 >> "$out"
done
python - "$out" <<'PY'
import json, sys
name = None
for line in open(sys.argv[1]):
    if line.startswith("#"):
        name = line[2:].strip()
    elif line.startswith("{"):
        d = json.loads(line)
        print(f"{name:12s} conv {d['roofline']['launch_us']:7.1f} us")
PY

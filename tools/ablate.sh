#!/bin/bash
# needs a library built with the ablated kernels:  DGNN_ABLATIONS=1 python dream-go_amd/build.py --force
# Component pricing of the residual convolution (results of ablated runs are garbage; only the timing matters).
#   1 = no MFMA, 3 = neither MFMA nor in-loop memory instructions,
#   16 = no barrier at the chunk boundaries (what a barrier-free K loop could gain at most)
# Round 3: the ablated variants exist for the four-slice split only (the wide split has none): DGNN_BOARD_NS=4 selects it.
out=${1:-gpurun_out/ablate.jsonl}
: > "$out"
for ab in 0 1 3 16; do
  echo "# ablate=$ab" >> "$out"
  DGNN_TOWER=0 DGNN_BOARD_NS=${NS:-4} DGNN_CONV=${VARIANT:-board} DGNN_CONV_ABLATE=$ab python bench.py --steps 10 --warmup 3 --cpu-sample 0 2>/dev/null | tail -1 >> "$out"
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

#!/bin/bash
# needs a library built with the ablated kernels:  DGNN_ABLATIONS=1 python dream-go_amd/build.py --force
# Component pricing of the per-layer board kernel at a given batch (results of ablated runs are garbage; only the timing matters):
#   0 = shipped, 1 = no MFMA, 3 = neither MFMA nor in-loop memory instructions, 16 = no chunk barriers
#   tools/ablate2.sh <batch> [blocks channels]      (round 3: the four-slice split, DGNN_BOARD_NS=4; the wide split has no ablated variants)
batch=${1:-128}; blocks=${2:-20}; channels=${3:-256}
for ab in ${ABLATIONS:-0 1 3 16 0}; do
  DGNN_TOWER=0 DGNN_BOARD_NS=${NS:-4} DGNN_CONV_ABLATE=$ab python tools/sweep.py --blocks $blocks --channels $channels --batches $batch --steps 20 2>/dev/null | \
    python -c "import sys,json
for l in sys.stdin:
    d=json.loads(l); print('ablate %2d  B=%3d %-20s conv %7.1f us   %8.3f ms/pass' % ($ab, d['batch'], d['kernel'], d['residual_conv_us'], d['ms_per_pass']))"
done

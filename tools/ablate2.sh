#!/bin/bash
# needs a library built with the ablated kernels:  DGNN_ABLATIONS=1 python dream-go_amd/build.py --force
# Component pricing of the per-layer board kernel at a given batch (results of ablated runs are garbage; only the timing matters):
#   0 = shipped, 32 = K loop without the software pipeline, 1 = no MFMA, 2 = no in-loop DMA, 3 = neither
#   tools/ablate2.sh <batch> [blocks channels]
batch=${1:-128}; blocks=${2:-20}; channels=${3:-256}
for ab in ${ABLATIONS:-0 32 1 2 3 64 0}; do
  DGNN_TOWER=0 DGNN_CONV_ABLATE=$ab python tools/sweep.py --blocks $blocks --channels $channels --batches $batch --steps 20 2>/dev/null | \
    python -c "import sys,json
for l in sys.stdin:
    d=json.loads(l); print('ablate %2d  B=%3d %-20s conv %7.1f us   %8.3f ms/pass' % ($ab, d['batch'], d['kernel'], d['residual_conv_us'], d['ms_per_pass']))"
done

#!/bin/bash
# Same-box A/B of builds of libdgnn_hip.so:  tools/ab_lib.sh <other.so>[,<other2.so>...] <batches> [blocks channels]
# alternates the libraries (B A B A) so that clock drift shows up as spread, not as a difference.
others=$1; batches=$2; blocks=${3:-20}; channels=${4:-256}
for rep in 1 2; do
  for lib in ${others//,/ } ""; do
    DGNN_LIBRARY=$lib python tools/sweep.py --blocks $blocks --channels $channels --batches $batches --steps 30 2>/dev/null | \
      python -c "import sys,json
for l in sys.stdin:
    d=json.loads(l); print('%-28s B=%3d %-20s %9.0f ev/s %8.3f ms' % ('$lib' or 'in-tree', d['batch'], d['kernel'], d['evals_per_sec_resident'], d['ms_per_pass']))"
  done
done

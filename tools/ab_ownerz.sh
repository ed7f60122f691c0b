#!/bin/bash
# same-box A/B: skip rows loaded and waited for by the OWNER of a row tile only (in-tree) vs by every wave (build_variant.sh noownerz -DDGNN_OWNER_Z=0)
mkdir -p gpurun_out/r4i
timeout 500 python -m pytest tests/test_parity_gpu.py tests/test_round2_gpu.py tests/test_round3_gpu.py tests/test_round4_gpu.py -q -m gpu --timeout 200 -p no:cacheprovider 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-160
for rep in 1 2; do
  for lib in gpurun_libs/libdgnn_noownerz.so ""; do
    for shape in "20 256 65,96,128" "9 128 192,256" "6 64 256"; do
      set -- $shape
      DGNN_LIBRARY=$lib timeout 200 python tools/sweep.py --blocks $1 --channels $2 --batches $3 --steps 50 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    d=json.loads(l); print('%-32s %s B=%3d %-20s %9.0f ev/s %8.4f ms' % ('$lib' or 'in-tree', d['net'], d['batch'], d['kernel'], d['evals_per_sec_resident'], d['ms_per_pass']))"
    done
  done
done | tee gpurun_out/r4i/ab_ownerz.txt

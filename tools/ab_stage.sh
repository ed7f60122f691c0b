#!/bin/bash
# same-box A/B: the filter of the one-slice chunked splits staged through registers (in-tree) vs by LDS-DMA (build_variant.sh nostage -DDGNN_STAGE_FILTER=0)
mkdir -p gpurun_out/r4h
timeout 300 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "split_tower_equals or fill_level" --timeout 120 -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2; do
  for lib in gpurun_libs/libdgnn_nostage.so ""; do
    for shape in "9 128 32,64" "20 256 16,32,64" "6 64 64"; do
      set -- $shape
      DGNN_LIBRARY=$lib timeout 200 python tools/sweep.py --blocks $1 --channels $2 --batches $3 --steps 50 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    d=json.loads(l); print('%-32s %s B=%3d %-20s %9.0f ev/s %8.4f ms' % ('$lib' or 'in-tree', d['net'], d['batch'], d['kernel'], d['evals_per_sec_resident'], d['ms_per_pass']))"
    done
  done
done | tee gpurun_out/r4h/ab_stage.txt

#!/bin/bash
# same-box A/B of DMA front-loading variants (tools/build_variant.sh ppt2|ppt4|ppt7) on the small-batch splits
mkdir -p gpurun_out/r4e
for rep in 1 2; do
  for lib in "" gpurun_libs/libdgnn_ppt2.so gpurun_libs/libdgnn_ppt4.so gpurun_libs/libdgnn_ppt7.so; do
    for shape in "9 128 8,16,32,64" "20 256 1,8,16,32"; do
      set -- $shape
      DGNN_LIBRARY=$lib timeout 200 python tools/sweep.py --blocks $1 --channels $2 --batches $3 --steps 50 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    d=json.loads(l); print('%-32s %s B=%3d %-20s %9.0f ev/s %8.4f ms' % ('$lib' or 'in-tree', d['net'], d['batch'], d['kernel'], d['evals_per_sec_resident'], d['ms_per_pass']))"
    done
  done
done | tee gpurun_out/r4e/ab_ppt.txt
DGNN_LIBRARY=gpurun_libs/libdgnn_ppt4.so timeout 300 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "split_tower_equals or fill_level" --timeout 120 2>&1 | tail -3

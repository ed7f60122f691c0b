mkdir -p gpurun_out/r4d
# NOTE (round 6): the switch this script toggles (pair split / head-in-tower / arrival handshake) has been removed from the sources; the script is kept as the record of how the files it names under profiles/ were produced.
timeout 300 python -m pytest tests/test_round4_gpu.py -q -m gpu -k pair_split --timeout 120 > gpurun_out/r4d/pair_test.log 2>&1; echo "pair test rc $?"; tail -3 gpurun_out/r4d/pair_test.log
for rep in 1 2; do
  for pair in 0 1; do
    DGNN_TOWER_PAIR=$pair timeout 200 python tools/sweep.py --batches 256,192,144 --steps 30 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    d=json.loads(l); print('pair=$pair B=%3d %-20s %9.0f ev/s %8.3f ms conv %.1f us' % (d['batch'], d['kernel'], d['evals_per_sec_resident'], d['ms_per_pass'], d['residual_conv_us']))"
  done
done | tee gpurun_out/r4d/ab_pair.txt

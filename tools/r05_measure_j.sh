#!/bin/sh
# Round 5, tenth GPU call: the fourth energy probe (operand order / side), and the batch sweep of the final code with the GPU suite's skip reasons.
out=gpurun_out/r05j
mkdir -p $out
sh tools/mfma_energy.sh 4 mfma_energy4 > $out/mfma_energy4.txt 2>&1
timeout 600 python tools/sweep.py --blocks 20 --channels 256 --batches 1,8,16,32,48,64,80,96,112,128,129,144,160,192,224,256 --in-flight > $out/batch_sweep_20x256.jsonl 2>/dev/null
timeout 300 python tools/sweep.py --blocks 9 --channels 128 --batches 16,32,64,96,128 --in-flight > $out/sweep_9x128_a.jsonl 2>/dev/null
timeout 300 python tools/sweep.py --blocks 9 --channels 128 --batches 16,32,64,96,128 --in-flight > $out/sweep_9x128_b.jsonl 2>/dev/null
python -m pytest tests -m gpu -q -rs > $out/gpu_suite_rs.log 2>&1
tail -2 $out/gpu_suite_rs.log

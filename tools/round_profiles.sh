#!/bin/bash
# Everything a round's profiles/ directory is made of, in one call on the GPU box:  tools/round_profiles.sh <tag>
#   batch sweeps of the three net sizes, the default bench line, rocprofv3 kernel traces (+ --stats) of the bench workload
#   and of the two smaller BASELINE configs, PMC passes of the bench workload (tools/profile.sh), the MFMA ceiling.
tag=${1:-final}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
python tools/sweep.py --blocks 20 --channels 256 --batches 1,2,4,8,16,24,32,48,64,96,128,144,160,176,192,224,256 > $out/batch_sweep_20x256.jsonl 2>/dev/null
python tools/sweep.py --blocks 9 --channels 128 --batches 1,8,16,32,64,96,128,192,256 > $out/batch_sweep_9x128.jsonl 2>/dev/null
python tools/sweep.py --blocks 6 --channels 64 --batches 1,8,16,64,256 > $out/batch_sweep_6x64.jsonl 2>/dev/null
python bench.py > $out/bench.json 2> $out/bench.err
bash tools/profile.sh $tag > /dev/null 2>&1
mkdir -p $out/pmc; python tools/pmc_summary.py gpurun_out/prof_$tag $out/pmc tower_board_kernel > $out/pmc/traffic_line.txt 2>&1
cd /tmp; export TMPDIR=/tmp
for cfg in "20 256 128" "9 128 32" "20 256 16"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$1x$2_b$3 -o t -- python $root/bench.py --blocks $1 --channels $2 --batch $3 --steps 10 --warmup 2 --cpu-sample 0 --selfplay-games 0 > $out/bench_$1x$2_b$3.json 2> /dev/null
  cp $out/trace_$1x$2_b$3/*/t_kernel_stats.csv $out/kernel_stats_$1x$2_b$3.csv 2>/dev/null || cp $out/trace_$1x$2_b$3/t_kernel_stats.csv $out/kernel_stats_$1x$2_b$3.csv
  rm -rf $out/trace_$1x$2_b$3
done
cd $root
sh tools/mfma_peak.sh 4 > $out/mfma_peak.txt 2>&1
ls $out
# round 3: three submissions in flight (the leaf pool's regime), the per-layer stamps of the whole-board tower, the energy probes
python tools/sweep.py --blocks 20 --channels 256 --in-flight --batches 96,112,128,129,136,144,152,160,176,192,224,256 > $out/batch_sweep_in_flight_20x256.jsonl 2>/dev/null
python tools/tower_trace.py 256 > $out/tower_trace_256.txt 2>&1
python tools/tower_trace.py 192 > $out/tower_trace_192.txt 2>&1
sh tools/mfma_energy.sh 4 mfma_energy > $out/mfma_energy.txt 2>&1
sh tools/mfma_energy.sh 4 mfma_energy2 > $out/mfma_energy2.txt 2>&1
ls $out

#!/bin/bash
# The last call of a round, one box: the driver line, then the rocprofv3 passes of the same workload (kernel stats, PMC) and the 40-pass
# lean line they are compared with -- at the round's last commit.
tag=${1:-r05_last}
out=gpurun_out/final_$tag
mkdir -p $out
timeout 600 python bench.py > $out/bench.log 2> $out/bench.err; echo "bench rc $?"
grep '^{"metric"' $out/bench.log | tail -1 > $out/bench.json
timeout 600 bash tools/profile.sh $tag > $out/profile.log 2>&1; echo "profile rc $?"
python tools/pmc_summary.py gpurun_out/prof_$tag $out/pmc tower_board_kernel > $out/pmc_summary.log 2>&1
timeout 300 python bench.py --lean --steps 40 --warmup 5 > $out/bench_lean_40.json 2>/dev/null
python - <<PY
import json
d = json.load(open("$out/bench.json"))
print("value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["launch_us"])
l = json.loads(open("$out/bench_lean_40.json").read().strip().split("\n")[-1])
print("lean 40:", l["value"], l["roofline"]["frac"], l["roofline"]["launch_us"])
PY
head -3 $out/pmc/kernel_stats.csv

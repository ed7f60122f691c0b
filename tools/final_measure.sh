#!/bin/bash
# One box, one call: the driver line, the rocprofv3 passes of the same workload, the batch sweeps, the parity figures.
tag=${1:-r04}
out=gpurun_out/final_$tag
mkdir -p $out
timeout 500 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc $?"
timeout 600 bash tools/profile.sh $tag > $out/profile.log 2>&1; echo "profile rc $?"
python tools/pmc_summary.py gpurun_out/prof_$tag $out/pmc tower_board_kernel > $out/pmc_summary.log 2>&1
timeout 300 python bench.py --lean --steps 40 --warmup 5 > $out/bench_lean_40.json 2>/dev/null
timeout 400 python tools/sweep.py --blocks 20 --channels 256 --batches 1,8,16,32,64,96,128,129,144,160,192,224,256 --in-flight > $out/batch_sweep_20x256.jsonl 2>/dev/null
timeout 300 python tools/sweep.py --blocks 9 --channels 128 --batches 1,8,16,32,64,128,256 > $out/batch_sweep_9x128.jsonl 2>/dev/null
timeout 300 python bench.py --lean --blocks 9 --channels 128 --batch 32 --steps 50 > $out/bench_9x128_b32.json 2>/dev/null
timeout 400 python -m pytest tests/test_round4_gpu.py -q -m gpu -s --timeout 200 -p no:cacheprovider 2>&1 | grep -E "^(sharp|centered)|^   |passed|failed" > $out/parity_ulps.txt
DGNN_SPLIT_INJECT_GIVEUP=2 timeout 100 python - > $out/giveup_latency.txt 2>&1 <<PY
import sys, time
sys.path.insert(0, ".")
import __graft_entry__ as graft
graft.package()
from dream_go_amd import api, synth
net = api.Network.from_json(synth.to_json(synth.make_tensors(2, 256, seed=650)))
features = synth.make_features(16, seed=651)
ws = net.get_workspace(16)
t0 = time.perf_counter(); api.forward(ws, features); t1 = time.perf_counter()
api.forward(ws, features); t2 = time.perf_counter()
print(f"split tower launched without its last eight workgroups (DGNN_SPLIT_INJECT_GIVEUP=2): first pass {1e3 * (t1 - t0):.1f} ms end to end "
      f"(the give-up after 50 ms of s_memrealtime + the pass repeated per layer), next pass {1e3 * (t2 - t1):.2f} ms")
PY
timeout 600 python -m pytest tests -x -q -m gpu --timeout 200 -p no:cacheprovider > $out/gpu_suite.log 2>&1; echo "suite rc $?"; tail -2 $out/gpu_suite.log
python - <<PY
import json
d = json.load(open("$out/bench.json"))
print("value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["launch_us"], "parity", d["parity_vs_oracle"]["flat"]["max_abs_value_diff"], d["parity_vs_oracle"]["sharpened"]["max_abs_policy_diff"])
for leg in ("selfplay", "selfplay_midgame"):
    s = d[leg]; print(leg, s["evals_per_sec"], s["moves_per_sec"], s["mean_device_batch"], s["host_cpus_busy"])
PY
head -3 $out/pmc/kernel_stats.csv

#!/bin/bash
# One box, one call: the driver line, the rocprofv3 passes of the same workload, the batch sweeps, the parity figures, the suite,
# and (round 5) the ten-minute steady state of whole games that bench.py's 90 s window is compared with.
tag=${1:-r05}
out=gpurun_out/final_$tag
mkdir -p $out
timeout 600 python bench.py > $out/bench.log 2> $out/bench.err; echo "bench rc $?"
grep '^{"metric"' $out/bench.log | tail -1 > $out/bench.json
timeout 600 bash tools/profile.sh $tag > $out/profile.log 2>&1; echo "profile rc $?"
python tools/pmc_summary.py gpurun_out/prof_$tag $out/pmc tower_board_kernel > $out/pmc_summary.log 2>&1
timeout 300 python bench.py --lean --steps 40 --warmup 5 > $out/bench_lean_40.json 2>/dev/null
timeout 500 python tools/sweep.py --blocks 20 --channels 256 --batches 1,8,16,32,64,96,128,129,144,160,192,224,256 --in-flight > $out/batch_sweep_20x256.jsonl 2>/dev/null
timeout 300 python tools/sweep.py --blocks 9 --channels 128 --batches 1,8,16,32,64,128,256 > $out/batch_sweep_9x128.jsonl 2>/dev/null
timeout 300 python bench.py --lean --blocks 9 --channels 128 --batch 32 --steps 50 > $out/bench_9x128_b32.json 2>/dev/null
timeout 300 python tools/stage_times.py > $out/stage_times_20x256.jsonl 2>/dev/null
timeout 400 python -m pytest tests/test_round4_gpu.py tests/test_sharp_fixture.py -q -m gpu -s --timeout 300 -p no:cacheprovider 2>&1 | grep -E "^(sharp|centered)|^   |^logits|passed|failed" > $out/parity_ulps.txt
timeout 800 python tools/selfplay_go.py --games 96 --rollouts 800 --threads 1 --leaves 16 --host-threads 4 --steady 20,600 > $out/selfplay_steady_96g_600s.json 2> $out/steady.err; echo "steady rc $?"
timeout 900 python -m pytest tests -q -m gpu --timeout 300 -p no:cacheprovider > $out/gpu_suite.log 2>&1; echo "suite rc $?"; tail -2 $out/gpu_suite.log
python - <<PY
import json
d = json.load(open("$out/bench.json"))
print("value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["launch_us"], "sustained", d["value_sustained"]["evals_per_sec"], d["value_sustained"]["roofline_frac"])
for leg in ("whole_games", "policy_play", "selfplay_midgame", "selfplay_midgame_reference_rule"):
    s = d[leg]; print(leg, "moves/s %.1f dev %.0f (%.3f) batch %.0f cores %.2f" % (s["moves_per_sec"], s["device_boards_per_sec"], s["device_boards_per_sec"] / d["value"], s["mean_device_batch"], s["host_cpus_busy"]))
print("cpu", d["cpu_baseline"]["value"], d.get("cpu_baseline_second_opinion"))
s = json.load(open("$out/selfplay_steady_96g_600s.json"))
print("steady 600 s: moves/s %.1f games/h measured %.0f (from moves %.0f) completed %d stones %.0f dev %.0f batch %.0f cores %.2f" % (s["moves_per_sec"], s["games_per_hour_measured"], s["games_per_hour_from_moves"], s["games_completed"], s["mean_stones_per_completed_game"], s["device_boards_per_sec"], s["mean_device_batch"], s["host_cpus_busy"]))
PY
head -3 $out/pmc/kernel_stats.csv

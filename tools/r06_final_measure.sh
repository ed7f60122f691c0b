#!/bin/bash
# Round 6, one box, one call: the driver's line, the rocprofv3 passes of the same workload (and of configs[1]), the counters rocprofv3
# knows on this box, the ten-minute steady state of whole games, the GPU suite.    bash tools/r06_final_measure.sh [tag] [steady seconds]
tag=${1:-r06}
steady=${2:-600}
out=gpurun_out/final_$tag
mkdir -p $out
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.log 2> $out/bench.err; echo "bench rc $?"
grep '^{"metric"' $out/bench.log | tail -1 > $out/bench.json
echo "gave-up lines on the bench's stderr: $(grep -c 'gave up' $out/bench.err)"
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -o -E "\b(TCP|TCC|SQ|GRBM|TA|TD)_[A-Z0-9_a-z]+" | sort -u) > $out/counters_available.txt
timeout 900 bash tools/profile.sh $tag > $out/profile.log 2>&1; echo "profile rc $?"
BENCH_BATCH=256 BENCH_CHANNELS=256 python tools/pmc_summary.py gpurun_out/prof_$tag $out/pmc tower_board_kernel > $out/pmc_summary.log 2>&1
BENCH_ARGS="--blocks 9 --channels 128 --batch 32" timeout 900 bash tools/profile.sh ${tag}_9x128_b32 > $out/profile_9x128.log 2>&1; echo "profile 9x128 rc $?"
BENCH_BATCH=32 BENCH_CHANNELS=128 python tools/pmc_summary.py gpurun_out/prof_${tag}_9x128_b32 $out/pmc_9x128_b32 tower_split_kernel > $out/pmc_summary_9x128.log 2>&1
timeout 300 python bench.py --lean --steps 40 --warmup 5 > $out/bench_lean_40.json 2>/dev/null
timeout 300 python bench.py --lean --blocks 9 --channels 128 --batch 32 --steps 50 > $out/bench_9x128_b32.json 2>/dev/null
if [ "$steady" -gt 0 ]; then
  timeout $((steady + 200)) python tools/selfplay_go.py --games 96 --rollouts 800 --threads 1 --leaves 16 --host-threads 4 --steady 20,$steady > $out/selfplay_steady_96g_${steady}s.json 2> $out/steady.err; echo "steady rc $?"
  echo "gave-up lines on the steady state's stderr: $(grep -c 'gave up' $out/steady.err)"
fi
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider > $out/gpu_suite.log 2>&1; echo "suite rc $?"; tail -2 $out/gpu_suite.log
python - <<PY
import json
d = json.load(open("$out/bench.json"))
print("value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["launch_us"], "sustained", d["value_sustained"]["evals_per_sec"], d["value_sustained"]["roofline_frac"])
print("reference contract", d["value_reference_contract"]["value"], d["value_reference_contract"]["vs_value"])
for leg in ("whole_games", "policy_play", "policy_play_ex_it", "selfplay_midgame", "selfplay_midgame_reference_rule"):
    s = d[leg]; print(leg, "moves/s %.1f dev %.0f (%.3f) batch %.0f cores %.2f evals/move %.1f towers %s ex-it %s" % (s["moves_per_sec"], s["device_boards_per_sec"], s["device_boards_per_sec"] / d["value"], s["mean_device_batch"], s["host_cpus_busy"], s["evals_per_move"], s.get("tower_passes"), s.get("ex_it_searches")))
w = d["whole_games"]; print("whole games: completed %d, /h measured %.0f +- %.0f, from moves %.0f" % (w["games_completed"], w["games_per_hour_measured"], w["games_per_hour_measured_sigma"], w["games_per_hour_from_moves"] or 0))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["statement"], "| other", d["cpu_baseline_other_statement"]["value"])
PY
head -3 $out/pmc/kernel_stats.csv; cat $out/pmc/traffic.json; cat $out/pmc_9x128_b32/traffic.json

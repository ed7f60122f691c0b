#!/bin/bash
# Round 5, fifth GPU call: the pruned build -- whole -m gpu suite over both libraries, the two-tenant diagnostic, the default bench line.
out=gpurun_out/r05e
mkdir -p $out
(time python -m pytest tests -m gpu -q) > $out/gputest.log 2>&1
grep -E "passed|failed|error" $out/gputest.log | tail -3
grep -E "^FAILED|^ERROR" $out/gputest.log | cut -c1-200 | head -20
python -m pytest tests/test_failure_paths_gpu.py -m gpu -q -s -k "two_processes_running or not_fully_resident" 2>&1 | grep -E "slow_passes|pass with eight|passed|failed" | cut -c1-900 > $out/two_tenants.txt
cat $out/two_tenants.txt
(time python bench.py) > $out/bench.log 2> $out/bench.err
grep real $out/bench.log $out/bench.err
grep '^{"metric"' $out/bench.log | tail -1 > $out/bench.json
python - <<PY
import json
d = json.load(open("$out/bench.json"))
print("value %.0f frac %.3f sustained %.0f (%.3f) pcie %.0f" % (d["value"], d["roofline"]["frac"], d["value_sustained"]["evals_per_sec"], d["value_sustained"]["roofline_frac"], d["value_pcie_inclusive"]["board_records"]))
for leg in ("whole_games", "policy_play", "selfplay_midgame", "selfplay_midgame_reference_rule", "selfplay"):
    s = d[leg]
    print(leg, "moves/s %.1f dev %.0f (%.3f of value) batch %.0f cores %.2f" % (s["moves_per_sec"], s["device_boards_per_sec"], s["device_boards_per_sec"] / d["value"], s["mean_device_batch"], s["host_cpus_busy"]),
          ("games/h %.0f (from moves %.0f), %d completed, %.0f stones" % (s["games_per_hour_measured"], s["games_per_hour_from_moves"] or 0, s["games_completed"], s["mean_stones_per_completed_game"] or 0)) if "games_completed" in s else "")
print("parity", {k: (v["max_abs_value_diff"], v["max_abs_policy_diff"], v["policy_kl_max"]) for k, v in d["parity_vs_oracle"].items()}, d["config"]["rank0_gpu"])
PY

#!/bin/bash
# Round 5, sixth GPU call: the suite after the prune fixes; three batches alive / three workers in the steady-state legs.
out=gpurun_out/r05f
mkdir -p $out
(time python -m pytest tests -m gpu -q) > $out/gputest.log 2>&1
grep -E "passed|failed|error" $out/gputest.log | tail -3
grep -E "^FAILED|^ERROR" $out/gputest.log | cut -c1-200 | head
short="--cpu-sample 0 --sustained-seconds 0 --reference-rule-leg 0 --selfplay-moves 2 --whole-games-ramp 10 --whole-games-seconds 40 --policy-play-seconds 6"
for rep in 1 2; do
  timeout 300 python bench.py $short > $out/bench_b2w2_$rep.json 2>/dev/null
  timeout 300 python bench.py $short --selfplay-workers 3 > $out/bench_b2w3_$rep.json 2>/dev/null
  timeout 300 python bench.py $short --selfplay-batches 3 > $out/bench_b3w3_$rep.json 2>/dev/null
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/bench_b*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    row = [f.split("/")[-1][:14], "value %.0f" % d["value"], "pinned %s" % d["config"]["rank0_gpu"]["cpus_pinned"]]
    for leg in ("selfplay_midgame", "whole_games", "policy_play"):
        s = d.get(leg)
        if s: row.append("%s: moves/s %.1f dev %.0f (%.3f) batch %.0f cores %.2f" % (leg[:9], s["moves_per_sec"], s["device_boards_per_sec"], s["device_boards_per_sec"] / d["value"], s["mean_device_batch"], s["host_cpus_busy"]))
    print(" | ".join(row))
PY

#!/bin/bash
# Same box: the self-play legs and whole games with two pool workers for two batches alive (as shipped) and with three
# (a free worker takes the next batch while another hands 256 answers to their games).  Not run yet: the
# round's GPU budget was spent when the per-size batch times (dgnn_pool_batch_times) pointed at it.
# ROUND 5: run in the form of tools/r05_measure_f.sh (the three settings interleaved twice on one box): profiles/r05_final/ab_batches_workers.txt --
# a third worker + 2.5-3.5 %, a third batch alive + 1-2 % more; both are the tools' defaults now.
#   gpurun --timeout 900 -- 'bash tools/ab_workers.sh gpurun_out/ab_workers'
out=${1:-gpurun_out/ab_workers}
mkdir -p $out
for w in 0 3 0 3; do
  timeout 200 python bench.py --selfplay-workers $w > $out/bench_w$w.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$out/bench_w$w.json"))
for leg in ("selfplay", "selfplay_midgame"):
    s = d[leg]
    print("workers $w", leg, "moves/s %.1f" % s["moves_per_sec"], "device boards/s %.0f" % s["device_boards_per_sec"], "mean batch %.1f" % s["mean_device_batch"],
          "host cores %.2f" % s["host_cpus_busy"], "of value %.3f" % (s["device_boards_per_sec"] / d["value"]))
PY
done
for w in 0 3; do
  timeout 400 python tools/selfplay_go.py --games 96 --moves 0 --threads 1 --leaves 16 --host-threads 12 --workers $w > $out/selfplay_full_games_96g_l16_h12_w$w.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$out/selfplay_full_games_96g_l16_h12_w$w.json"))
print("workers $w whole games: %.0f per hour, %.1f moves/s, mean batch %.1f, host cores %.2f" % (d["games_per_hour_measured"], d["moves_per_sec"], d["mean_device_batch"], d["host_cpus_busy_rank0"]))
PY
done

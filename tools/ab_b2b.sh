#!/bin/bash
# Same box: whole-game self-play over the first N stones with the reference's batch rule and back to back (dgnn_pool_set_back_to_back)
out=${1:-gpurun_out/r4p}; moves=${2:-150}
mkdir -p $out
run() { name=$1; shift; timeout 300 python tools/selfplay_go.py --games 96 --moves $moves --threads 1 --leaves 16 --host-threads 12 "$@" > $out/$name.json 2> $out/$name.err
  python - <<PY
import json
d = json.load(open("$out/$name.json"))
print("$name", "moves/s %.1f" % d["moves_per_sec"], "answers/s %.0f" % d["evals_per_sec"], "mean batch %.1f" % d["mean_device_batch"], "host cores %.2f" % d["host_cpus_busy_rank0"], "followed", d["coalesced_boards_rank0"])
PY
}
run base
run b2b_300 --back-to-back 300
run b2b_300_snap --back-to-back 300 --snap 128,208
run b2b_600_snap --back-to-back 600 --snap 128,208
run base_again

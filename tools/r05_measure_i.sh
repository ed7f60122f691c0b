#!/bin/bash
# Round 5, ninth GPU call: whole-board towers for small batches on a crowded device -- the suite, then the in-flight sweep with and without.
out=gpurun_out/r05i
mkdir -p $out
(time python -m pytest tests -m gpu -q) > $out/gputest.log 2>&1
grep -E "passed|failed|error" $out/gputest.log | tail -3
grep -E "^FAILED|^ERROR" $out/gputest.log | cut -c1-200 | head
B=16,32,48,64,80,96,112,128,144,160
python tools/sweep.py --blocks 20 --channels 256 --batches $B --in-flight > $out/sweep_crowded1.jsonl 2>/dev/null
DGNN_LIBRARY=dream-go_amd/lib/libdgnn_hip_test.so DGNN_CROWDED=0 python tools/sweep.py --blocks 20 --channels 256 --batches $B --in-flight > $out/sweep_crowded0.jsonl 2>/dev/null
python tools/sweep.py --blocks 20 --channels 256 --batches $B --in-flight > $out/sweep_crowded1_again.jsonl 2>/dev/null
python tools/sweep.py --blocks 9 --channels 128 --batches 16,32,64,96,128 --in-flight > $out/sweep_9x128_crowded1.jsonl 2>/dev/null
DGNN_LIBRARY=dream-go_amd/lib/libdgnn_hip_test.so DGNN_CROWDED=0 python tools/sweep.py --blocks 9 --channels 128 --batches 16,32,64,96,128 --in-flight > $out/sweep_9x128_crowded0.jsonl 2>/dev/null
python - <<PY
import json
def load(f):
    return {json.loads(l)["batch"]: json.loads(l) for l in open(f)}
for tag, files in (("20x256", ("sweep_crowded0", "sweep_crowded1", "sweep_crowded1_again")), ("9x128", ("sweep_9x128_crowded0", "sweep_9x128_crowded1"))):
    rows = [load("$out/" + f + ".jsonl") for f in files]
    print(tag, "batch | one at a time | two in flight | three in flight   (each: " + " / ".join(files) + ")")
    for b in sorted(rows[0]):
        print(b, "|", " / ".join("%.0f" % r[b]["evals_per_sec_host_buffers"] for r in rows), "|", " / ".join("%.0f" % r[b]["evals_per_sec_two_in_flight"] for r in rows), "|", " / ".join("%.0f" % r[b]["evals_per_sec_three_in_flight"] for r in rows))
PY

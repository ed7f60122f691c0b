#!/bin/bash
# NOTE (round 6): the switch this script toggles (pair split / head-in-tower / arrival handshake) has been removed from the sources; the script is kept as the record of how the files it names under profiles/ were produced.
# Round 5, fourth GPU call: the arrival handshake (failure-path tests, its price at 9x128 batch 32) and zero-copy I/O (A/B), then the suite.
out=gpurun_out/r05d
mkdir -p $out
python -m pytest tests/test_failure_paths_gpu.py -m gpu -q -s -x 2>&1 | grep -E "pass with eight|worst_ms|passed|failed|Error|assert" | cut -c1-300 > $out/failure_paths.txt
cat $out/failure_paths.txt
for z in 1 0 1 0; do
  DGNN_ZEROCOPY=$z python tools/stage_times.py --batches 64,128,256 --steps 60 > $out/stage_times_zerocopy${z}_$RANDOM.jsonl 2>/dev/null
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/stage_times_zerocopy*.jsonl")):
    for l in open(f):
        d = json.loads(l)
        print(f.split("/")[-1][:22], "b", d["batch"], "fl", d["in_flight"], "h2d %.0f planes %.0f tower %.0f headconv %.0f gemm %.0f softmax %.0f d2h %.0f | s2a %.0f" % (d["h2d_records"], d["planes_kernel"], d["tower"], d["head_conv"], d["policy_gemm"], d["softmax_value"], d["d2h_outputs"], d["host"]["submit_to_answers_us"]),
              "boards/s %.0f" % (d["batch"] * d["in_flight"] / d["host"]["submit_to_answers_us"] * 1e6))
PY
# the handshake's price where a launch is shortest: 9x128 batch 32 (configs[1]) and 20x256 batch 128
for rep in 1 2; do
  python bench.py --lean --blocks 9 --channels 128 --batch 32 --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('9x128 b32 value %.0f launch_us %.1f' % (d['value'], d['roofline']['launch_us']))"
  python bench.py --lean --batch 128 --steps 50 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('20x256 b128 value %.0f launch_us %.1f' % (d['value'], d['roofline']['launch_us']))"
done
short="--cpu-sample 0 --sustained-seconds 0 --reference-rule-leg 0 --whole-games-ramp 10 --whole-games-seconds 40 --policy-play-seconds 6"
for z in 1 0 1 0; do
  DGNN_ZEROCOPY=$z timeout 300 python bench.py $short > $out/bench_zerocopy${z}_$RANDOM.json 2>/dev/null
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/bench_zerocopy*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    row = [f.split("/")[-1][:15], "value %.0f" % d["value"], "pcie %.0f (%.3f)" % (d["value_pcie_inclusive"]["board_records"], d["value_pcie_inclusive"]["board_records"] / d["value"])]
    for leg in ("selfplay_midgame", "whole_games", "policy_play"):
        s = d.get(leg)
        if s: row.append("%s: moves/s %.1f dev %.0f (%.3f) batch %.0f cores %.2f" % (leg[:9], s["moves_per_sec"], s["device_boards_per_sec"], s["device_boards_per_sec"] / d["value"], s["mean_device_batch"], s["host_cpus_busy"]))
    print(" | ".join(row))
PY
(time python -m pytest tests -m gpu -x -q) > $out/gputest.log 2>&1
grep -E "passed|failed|error" $out/gputest.log | tail -3

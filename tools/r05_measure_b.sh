#!/bin/bash
# Round 5, second GPU call: where a device batch's time goes, the chain A/B, workers A/B, split-tower stamps at 64 / 96 boards, energy table.
out=gpurun_out/r05b
mkdir -p $out
python tools/stage_times.py > $out/stage_times_20x256.jsonl 2> $out/stage_times.err
DGNN_CHAIN=2 python tools/stage_times.py --batches 128,192,256 --in-flight 2 > $out/stage_times_20x256_chain2.jsonl 2>> $out/stage_times.err
short="--cpu-sample 0 --sustained-seconds 0 --policy-play-games 0 --reference-rule-leg 0 --whole-games-ramp 10 --whole-games-seconds 40"
for rep in 1 2; do
  for chain in 0 2 1; do
    DGNN_CHAIN=$chain timeout 300 python bench.py $short > $out/bench_chain${chain}_$rep.json 2>/dev/null
  done
done
for w in 0 3; do
  timeout 300 python bench.py $short --whole-games 0 --selfplay-workers $w > $out/bench_workers$w.json 2>/dev/null
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    row = [f.split("/")[-1], "value %.0f" % d["value"], "pcie %.0f" % d["value_pcie_inclusive"]["board_records"]]
    for leg in ("selfplay_midgame", "whole_games"):
        s = d.get(leg)
        if s: row.append("%s: moves/s %.1f dev %.0f (%.3f of value) batch %.0f cores %.2f" % (leg, s["moves_per_sec"], s["device_boards_per_sec"], s["device_boards_per_sec"] / d["value"], s["mean_device_batch"], s["host_cpus_busy"]))
    print(" | ".join(row))
PY
for b in 64 96 128; do
  DGNN_SPLIT_TRACE=/tmp/split_$b.bin python tools/split_trace.py --batch $b > $out/split_trace_20x256_b$b.txt 2>&1
  DGNN_LIBRARY=gpurun_libs/libdgnn_tap41.so DGNN_SPLIT_TRACE=/tmp/tap_$b.bin python tools/tap_trace.py --batch $b --blocks 20 --channels 256 > $out/tap_trace_20x256_b$b.txt 2>&1
done
sh tools/mfma_energy.sh 6 mfma_energy3 > $out/mfma_energy3.txt 2>&1
tail -3 $out/split_trace_20x256_b64.txt; grep final $out/mfma_energy3.txt | tail -6

#!/bin/sh
# A ten-minute soak of the steady state at the shipped settings (the whole-game leg of bench.py over 600 s), with the process's resident
# memory and the device's used VRAM sampled every 10 s: what a long self-play job holds after its first minute it must still hold at
# its tenth.    tools/soak.sh [seconds, default 600] [out dir, default gpurun_out/soak]
#   SOAK_ARGS="--games 1024 --rollouts 1 --threads 1 --leaves 16 --host-threads 4" tools/soak.sh ...   policy play: sixteen games end and start every second
SECS=${1:-600}
OUT=${2:-gpurun_out/soak}
mkdir -p "$OUT"
python tools/selfplay_go.py ${SOAK_ARGS:---games 96 --rollouts 800 --threads 1 --leaves 16 --host-threads 4} --steady 20,$SECS > "$OUT/steady.json" 2> "$OUT/steady.err" &
PID=$!
: > "$OUT/memory.log"
while kill -0 $PID 2>/dev/null; do
    RSS=$(awk '/VmRSS/ {print $2}' /proc/$PID/status 2>/dev/null)
    THREADS=$(awk '/Threads/ {print $2}' /proc/$PID/status 2>/dev/null)
    VRAM=$(rocm-smi --showmeminfo vram 2>/dev/null | awk '/Used/ {print $NF}' | head -1)
    echo "t=$(date +%s) rss_kb=$RSS threads=$THREADS vram_used_bytes=$VRAM" >> "$OUT/memory.log"
    sleep 10
done
wait $PID
echo "rc=$?"
python - "$OUT" <<'PY'
import json, re, sys
out = sys.argv[1]
rows = [dict((k, int(v)) for k, v in re.findall(r"(\w+)=(\d+)", line)) for line in open(out + "/memory.log") if "rss_kb=" in line and re.search(r"rss_kb=\d", line)]
steady = [r for r in rows if r["t"] - rows[0]["t"] >= 60]
leg = json.loads(open(out + "/steady.json").read().strip().split("\n")[-1])
print(json.dumps({"samples": len(rows), "rss_mb_after_the_first_minute": steady[0]["rss_kb"] / 1024, "rss_mb_at_the_end": steady[-1]["rss_kb"] / 1024,
                  "rss_mb_max": max(r["rss_kb"] for r in steady) / 1024, "vram_mb_after_the_first_minute": steady[0].get("vram_used_bytes", 0) / 2**20,
                  "vram_mb_at_the_end": steady[-1].get("vram_used_bytes", 0) / 2**20, "threads": sorted(set(r["threads"] for r in steady)),
                  "moves_per_sec": leg["moves_per_sec"], "games_completed": leg["games_completed"], "games_per_hour_measured": leg["games_per_hour_measured"],
                  "device_boards_per_sec": leg["device_boards_per_sec"], "mean_device_batch": leg["mean_device_batch"], "seconds": leg["seconds"]}))
PY

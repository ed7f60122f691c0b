#!/bin/bash
# What CPU placement does to the bench's self-play leg on a box that shows 256 logical CPUs and grants 16 CPUs of time:
# unpinned, CPUs 0-15, 8 cores + their SMT siblings, 16 cores of the other socket, 32 CPUs.
# Runs bench.py with DGNN_PIN_CPUS=0 under taskset so that the harness's own pinning does not interfere.
export DGNN_PIN_CPUS=0
lscpu | grep -E "NUMA|Socket|Thread|Core" | head -12
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -4
run() { echo "== $1"; shift; "$@" python bench.py --steps 5 --warmup 2 --cpu-sample 0 --selfplay-moves 8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['selfplay']; print({k: round(s[k],2) for k in ('host_cpus_busy','rollouts_per_sec','evals_per_sec','mean_device_batch','games_per_hour','seconds','cache_hit_ratio')}, 'engine', round(d['value']))"; }
run "no affinity" env
run "cpus 0-15" taskset -c 0-15
run "cpus 0-7 + smt siblings 128-135" taskset -c 0-7,128-135
run "cpus 64-79 (other socket)" taskset -c 64-79
run "cpus 0-31" taskset -c 0-31

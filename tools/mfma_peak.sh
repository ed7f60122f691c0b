#!/bin/sh
# Builds and runs tools/mfma_peak.hip on the GPU box (hipcc is there): the sustained MFMA ceiling of the part for constant,
# random and half-zero operands, with the socket power sampled beside it (rocm-smi, twice a second).
#   tools/mfma_peak.sh [seconds per mode, default 6]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SECS=${1:-6}
hipcc --offload-arch=gfx950 -O3 "$ROOT/tools/mfma_peak.hip" -o /tmp/dgnn_mfma_peak 2>/dev/null
( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo; sleep 0.5; done ) > /tmp/dgnn_power.log &
SMI=$!
/tmp/dgnn_mfma_peak "$SECS"
kill $SMI 2>/dev/null || true
echo "--- rocm-smi samples (every 0.5 s over the three modes) ---"
awk 'NR % 2 == 0' /tmp/dgnn_power.log | sed 's/=\+//g' | cut -c1-200

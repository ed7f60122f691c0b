#!/bin/sh
# Builds and runs tools/mfma_energy.hip on the GPU box with the socket power and clock sampled twice a second.
#   tools/mfma_energy.sh [seconds per mode, default 5] [mfma_energy | mfma_energy2]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SECS=${1:-5}
SRC=${2:-mfma_energy}
hipcc --offload-arch=gfx950 -O3 -w "$ROOT/tools/$SRC.hip" -o /tmp/dgnn_mfma_energy
( while true; do echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' ')"; sleep 0.5; done ) > /tmp/dgnn_power.log &
SMI=$!
date +%s.%N | cut -c1-14
/tmp/dgnn_mfma_energy "$SECS"
kill $SMI 2>/dev/null || true
echo "--- rocm-smi samples ---"
sed 's/=\+//g' /tmp/dgnn_power.log | cut -c1-220

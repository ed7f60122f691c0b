#!/bin/sh
# Builds and runs tools/mfma_peak.hip on the GPU box (hipcc is there): the sustained MFMA ceiling of the part.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
hipcc --offload-arch=gfx950 -O3 "$ROOT/tools/mfma_peak.hip" -o /tmp/dgnn_mfma_peak 2>/dev/null
/tmp/dgnn_mfma_peak

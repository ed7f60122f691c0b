#!/bin/sh
# Builds and runs tools/host_bench.cpp against the in-tree library (no GPU needed).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
g++ -O2 -std=c++17 -I"$ROOT/include" "$ROOT/tools/host_bench.cpp" -o /tmp/dgnn_host_bench -L"$ROOT/dream-go_amd/lib" -ldgnn_hip -Wl,-rpath,"$ROOT/dream-go_amd/lib"
/tmp/dgnn_host_bench

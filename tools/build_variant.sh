#!/bin/bash
# A variant build of libdgnn_hip.so for same-box A/B runs (tools/ab_lib.sh, DGNN_LIBRARY=<path>): the two kernel translation units
# that hold the residual-tower body are recompiled with extra flags (pass -DDGNN_ABLATIONS for the measurement hooks: DGNN_SPLIT_TRACE, DGNN_TOWER_TIMING ...), everything else is taken from the in-tree build.
#   tools/build_variant.sh <name> [-DDGNN_PIECES_PER_TAP=4 ...]     ->  gpurun_libs/libdgnn_<name>.so
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; shift
out="$ROOT/gpurun_libs"; mkdir -p "$out/obj_$name"
for tu in conv_board conv_board_b conv_board_tower; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result "$@" -x hip -c "$ROOT/dream-go_amd/csrc/$tu.hip" -o "$out/obj_$name/$tu.o" &
done
wait
objs=""
for o in "$ROOT"/dream-go_amd/lib/*.o; do
  b=$(basename "$o")
  case "$b" in conv_board.o|conv_board_b.o|conv_board_tower.o|conv_tile.o) ;; *) objs="$objs $o";; esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libdgnn_$name.so" $objs "$out/obj_$name/conv_board.o" "$out/obj_$name/conv_board_b.o" "$out/obj_$name/conv_board_tower.o"
echo "$out/libdgnn_$name.so"

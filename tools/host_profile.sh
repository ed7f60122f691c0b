#!/bin/sh
# Builds and runs tools/host_profile.cpp (see its header) and prints where the host's time goes, by function.
#   tools/host_profile.sh [games] [stones]          e.g.  POOL=256 tools/host_profile.sh 256 12
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/dream-go_amd/csrc"
TMP="${TMPDIR:-/tmp}/dgnn_host_profile"
mkdir -p "$TMP"
# the engine entry points the batcher links against: the CPU stand-in of the sanitizer script
sed -n '/^cat > "\$TMP\/stubs.cpp"/,/^STUBS/p' "$ROOT/tools/sanitize_host.sh" | sed '1d;$d' > "$TMP/stubs.cpp"
g++ -O3 -g -std=c++17 -no-pie -fno-omit-frame-pointer -I"$ROOT/include" -I"$SRC" "$ROOT/tools/host_profile.cpp" "$TMP/stubs.cpp" \
    "$SRC/tree.cpp" "$SRC/search.cpp" "$SRC/go.cpp" "$SRC/batcher.cpp" "$SRC/fiber.cpp" "$SRC/formats.cpp" "$SRC/loader.cpp" -lpthread -o "$TMP/host_profile" 2>/dev/null
cd "$TMP"
./host_profile "$@"
python3 "$ROOT/tools/host_profile.py" "$TMP"

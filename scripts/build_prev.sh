#!/bin/bash
# A/B helper: build the library of another commit (default HEAD) as pyopenvino_amd/libpvhip_prev.so, for
#   python scripts/with_lib.py pyopenvino_amd/libpvhip_prev.so bench.py ...   (same box, alternating runs)
# The sources are taken from git (never from the working tree), compiled in a scratch directory.
set -eu
REV=${1:-HEAD}
OUT=${2:-pyopenvino_amd/libpvhip_prev.so}
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d /tmp/pvprev.XXXXXX)
git -C "$R" archive "$REV" pyopenvino_amd/csrc include scripts | tar -x -C "$T"
make -C "$T/pyopenvino_amd/csrc" -j8 ARCH=gfx950 > "$T/build.log" 2>&1 || { tail -20 "$T/build.log"; exit 1; }
cp "$T/pyopenvino_amd/libpvhip.so" "$R/$OUT"
rm -rf "$T"
echo "built $OUT from $REV"

#!/bin/bash
# Build the HIP library of a git revision (or of the working tree: rev = WORK) into build/ab/<name>.so for A/B timing
# on one GPU box (tools/ab.sh).  build/ is git-ignored and travels with gpurun.
set -e
REV=$1; NAME=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p "$T/csrc" "$T/include" "$ROOT/build/ab"
if [ "$REV" = WORK ]; then cp "$ROOT"/fasthevc_amd/csrc/* "$T/csrc/"; cp "$ROOT"/include/* "$T/include/";
else for f in $(git -C "$ROOT" ls-tree --name-only "$REV" fasthevc_amd/csrc/ include/); do git -C "$ROOT" show "$REV:$f" > "$T/$( [ "${f#include/}" != "$f" ] && echo include || echo csrc)/$(basename $f)"; done; fi
sed -i 's|#include "../../include/fasthevc.h"|#include "../include/fasthevc.h"|' "$T"/csrc/* 2>/dev/null || true
cd "$T/csrc"
for s in $(cd "$T/csrc" && ls *.hip | sed "s/.hip//"); do
  X=""; [ $s = k_cnn ] && X="-ffinite-math-only -fno-signed-zeros ${CNN_FLAGS:-}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w $X -I"$T/include" -c $s.hip -o $s.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build/ab/$NAME.so" *.o
rm -rf "$T"
echo "$ROOT/build/ab/$NAME.so"

#!/bin/bash
# PMC passes over tools/layers_bench.py (GPU box).  usage: tools/pmc_layers.sh <outdir> "<counters pass>" ...
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/pass$i" -- python3 "$GRAFT_REPO_ROOT/tools/layers_bench.py" > "$OUT/pass$i.log" 2>&1
  echo "pass $i rc=$?"
done

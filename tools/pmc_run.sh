#!/bin/bash
# PMC passes over a short bench run (GPU box).  usage: tools/pmc_run.sh <outdir> "<counters pass>" ...
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/pass$i" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-host-path --no-stages > "$OUT/pass$i.log" 2>&1
  echo "pass $i rc=$?"
done

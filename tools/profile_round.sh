#!/bin/bash
# One GPU-box call that regenerates everything under profiles/ for a round (run through gpurun from the repo root):
#   tools/profile_round.sh r01
# Writes into gpurun_out/<tag>/; copy the summaries into profiles/ afterwards (tools/profile_collect.py).
set -u
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python3 bench.py > "$OUT/bench_default.log" 2> "$OUT/bench_default.err" || exit 1
timeout -k 10 200 python3 tools/bench_kernels.py > "$OUT/bench_kernels.json" 2> "$OUT/bench_kernels.err" || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-host-path --no-stages > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err" || exit 1
# the secondary kernels (first pass, pre-analysis, both motion searches ...) under the same profiler
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_kernels" -- python3 "$GRAFT_REPO_ROOT/tools/bench_kernels.py" > "$OUT/bench_kernels_under_rocprof.json" 2> "$OUT/bench_kernels_under_rocprof.err" || exit 1
cd "$GRAFT_REPO_ROOT"
bash tools/pmc_run.sh "$OUT/pmc_bench" "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES" || exit 1
bash tools/pmc_kernels.sh "$OUT/pmc_kernels" "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" || exit 1
python3 tools/pmc_summary.py "$OUT/pmc_bench" "$OUT/pmc_bench.json" > /dev/null
python3 tools/pmc_summary.py "$OUT/pmc_kernels" "$OUT/pmc_kernels.json" > /dev/null
echo done

#!/bin/bash
# One GPU-box call while iterating on the i8 variant of the depth kernel: parity under FHEVC_CNN_ARITH=i8, then three timed runs
# (bench.py, device-resident GOP) and optionally the phase table.
set -u
cd "$GRAFT_REPO_ROOT"
export FHEVC_CNN_ARITH=i8
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_random_stress.py -m gpu -x -q > gpurun_out/gpu_tests_i8.log 2>&1 || { tail -30 gpurun_out/gpu_tests_i8.log; exit 1; }
tail -1 gpurun_out/gpu_tests_i8.log
bash tools/ab_env.sh ${AB:-FHEVC_CNN_WG_PER_CU 3} 2>&1 | tee gpurun_out/ab_i8.log
for n in ${PHASES:-}; do
  echo "--- phases i8, $n WG/CU"
  FHEVC_DEBUG_WG_PER_CU=$n timeout -k 10 120 python3 tools/phase_cycles.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/phase_i8_wg$n.log
done

#!/bin/bash
# One GPU-box call while iterating on the i8 variant of the depth kernel: parity under FHEVC_CNN_ARITH=i8, A/B against f16, phases of both.
set -u
cd "$GRAFT_REPO_ROOT"
FHEVC_CNN_ARITH=i8 timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_random_stress.py -m gpu -x -q > gpurun_out/gpu_tests_i8.log 2>&1 || { tail -30 gpurun_out/gpu_tests_i8.log; exit 1; }
tail -1 gpurun_out/gpu_tests_i8.log
bash tools/ab_env.sh FHEVC_CNN_ARITH f16 i8 2>&1 | tee gpurun_out/ab_arith.log
for a in ${PHASES:-i8}; do
  echo "--- phases $a"
  FHEVC_CNN_ARITH=$a timeout -k 10 120 python3 tools/phase_cycles.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/phase_$a.log
done

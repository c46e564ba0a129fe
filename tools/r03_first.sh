#!/bin/bash
# round 3, first GPU-box call: the GPU suite (validates the hygiene fixes), then the depth kernel at 1 / 2 / 3 workgroups per CU
# (how far one workgroup alone fills its SIMDs) and the stamped build's phase table at the same settings
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests.log 2>&1 || { tail -30 gpurun_out/r03_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r03_gpu_tests.log
for n in 3 2 1 3; do
  echo "--- FHEVC_CNN_WG_PER_CU=$n"
  FHEVC_CNN_WG_PER_CU=$n timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-host-path --no-stages --no-variants > /tmp/b.log 2>/tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('/tmp/b.log').read().strip().splitlines()[-1]);print('wg/cu $n: ms/step %.4f  cnn ms %.4f  CTU/s %.4g' % (d['ms_per_step'],d['roofline']['avg_launch_ms'], d['value']))" | tee -a gpurun_out/r03_wg_scaling.log
done
for n in 1 3; do
  echo "--- phases i8, $n WG/CU" | tee -a gpurun_out/r03_phases.log
  FHEVC_DEBUG_WG_PER_CU=$n timeout -k 10 120 python3 tools/phase_cycles.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_phases.log
done

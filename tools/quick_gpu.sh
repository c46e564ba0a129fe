#!/bin/bash
# One GPU-box call while iterating on a kernel: parity tests, a short bench line, the phase diagnostic.
#   gpurun -- 'bash tools/quick_gpu.sh [pytest -k expression]'
set -u
cd "$GRAFT_REPO_ROOT"
K=${1:-}
if [ -n "$K" ]; then
  timeout -k 10 500 python3 -m pytest tests -m gpu -x -q -k "$K" > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
else
  timeout -k 10 500 python3 -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
fi
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_q.log 2> gpurun_out/bench_q.err || { tail gpurun_out/bench_q.err; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/bench_q.log').read().strip().splitlines()[-1]);print('CTU/s %.4g  ms/step %.4f  cnn ms %.4f  frac %.4f' % (d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['frac']))"
timeout -k 10 120 python3 tools/phase_cycles.py > gpurun_out/phase2.log 2>&1
grep -v amdgpu.ids gpurun_out/phase2.log

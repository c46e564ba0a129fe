#!/bin/bash
# A/B timing of library variants on ONE GPU box: tools/ab.sh nameA nameB ... (build/ab/<name>.so, tools/build_variant.sh).
# Each round installs a variant as the in-tree library and runs the bench line; the in-tree library is restored at the end.
cd "$GRAFT_REPO_ROOT"
LIB=fasthevc_amd/lib/libfasthevc_hip.so
cp $LIB /tmp/lib_keep.so
for round in 1 2 3; do
  for v in "$@"; do
    cp build/ab/$v.so $LIB
    timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-host-path --no-stages > /tmp/ab.log 2>/tmp/ab.err || { tail -5 /tmp/ab.err; cp /tmp/lib_keep.so $LIB; exit 1; }
    python3 -c "
import json;d=json.loads(open('/tmp/ab.log').read().strip().splitlines()[-1]);print('$v round $round: ms/step %.4f  cnn ms %.4f  hadamard ms %.4f' % (d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline_hbm_kernel']['avg_launch_ms'] or 0))"
    [ -n "${AB_FLAGS:-}" ] && python3 tools/bench_flags.py 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' ' && echo
  done
done
cp /tmp/lib_keep.so $LIB

#!/bin/bash
# per-phase cycles of library variants with ONE workgroup per CU (a single wave per SIMD: what a phase costs without a twin)
cd "$GRAFT_REPO_ROOT"
LIB=fasthevc_amd/lib/libfasthevc_hip.so
cp $LIB /tmp/lib_keep.so
for v in "$@"; do
  cp build/ab/$v.so $LIB
  echo "== $v, one workgroup per CU"; FHEVC_DEBUG_WG_PER_CU=1 timeout -k 10 120 python3 tools/phase_cycles.py 2>&1 | grep -v amdgpu.ids
  echo "== $v, two workgroups per CU"; timeout -k 10 120 python3 tools/phase_cycles.py 2>&1 | grep -v amdgpu.ids
done
cp /tmp/lib_keep.so $LIB

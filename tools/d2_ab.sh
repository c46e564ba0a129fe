#!/bin/bash
# same-box A/B of variants of the fused two-convolution kernel: tools/d2_ab.sh name ... (build/ab/<name>.so from tools/build_variant.sh), three rounds
cd "$GRAFT_REPO_ROOT"
B=fasthevc_amd/weights/depthnet_family_d2.fhw
for round in 1 2 3; do
  for v in "$@"; do
    echo -n "$v round $round: "
    FHEVC_AB_LIB=build/ab/$v.so FHEVC_LAYERS_BENCH_FUSED=1 FHEVC_LAYERS_BENCH_FRAMES=64 python3 tools/layers_bench.py $B 2>&1 | tail -1
  done
done

#!/bin/bash
# phase decomposition of the fused two-convolution kernel (k_cnn_d2.inc) by SKIP variants (wrong results on purpose): build/ab/d2skip<mask>.so from
#   CNN_FLAGS="-DFHEVC_D2_SKIP=<mask>" tools/build_variant.sh WORK d2skip<mask>     (mask bits: 1 conv1a, 2 conv1b, 4 conv2a, 8 conv2b, 16 conv3a, 32 conv3b, 64 heads)
cd "$GRAFT_REPO_ROOT"
B=fasthevc_amd/weights/depthnet_family_d2.fhw
for v in d2base d2skip1 d2skip2 d2skip4 d2skip8 d2skip16 d2skip32 d2skip64 d2skip127; do
  [ -f build/ab/$v.so ] || continue
  echo -n "$v: "
  FHEVC_AB_LIB=build/ab/$v.so FHEVC_LAYERS_BENCH_FUSED=1 FHEVC_LAYERS_BENCH_FRAMES=64 python3 tools/layers_bench.py $B 2>&1 | tail -1
done

#!/usr/bin/env python3
"""The layer-by-layer path of a family member (k_cnn_layers.inc) on 16 pictures of 1080p, random weights: for rocprofv3 --kernel-trace --stats
(per-launch time of every convolution) and for A/B of the layer kernel.  usage: python tools/layers_bench.py [23,46,92 2 | blob.fhw]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fasthevc_amd import capi, frames, weights  # noqa: E402

if os.environ.get("FHEVC_AB_LIB"):   # a variant library built by tools/build_variant.sh
    capi.LIB_PATH = os.environ["FHEVC_AB_LIB"]
blob = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".fhw") else None   # a trained blob instead of random weights of the given widths
if blob:
    fam = weights.load_any(blob)
    widths, depth = tuple(int(v) for v in fam["widths"]), int(fam["depth"])
else:
    widths = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "23,46,92").split(","))
    depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    fam = weights.random_family(widths, depth, seed=0)
W, H, NF = 1920, 1080, int(os.environ.get("FHEVC_LAYERS_BENCH_FRAMES", "16"))
if os.environ.get("FHEVC_LAYERS_BENCH_FUSED", "0") == "0":   # 1: the library's default dispatch (k_cnn_d2.inc for the x 2 members at 32 / 64 / 96)
    os.environ["FHEVC_FAMILY_LAYERS"] = "1"
base = frames.hetero_luma(W, H)
d8 = torch.from_numpy(np.stack([np.roll(base, 3 * f, axis=1) for f in range(NF)])).cuda()
ctx = capi.Context(W, H, 8, fam, max_frames=NF)
ctx.enable_kernel_timing(True)
out = torch.zeros((NF, ctx.num_ctus, 256), dtype=torch.uint8, device="cuda")
for _ in range(2):
    ctx.predict_frames_device(d8.data_ptr(), 1, W, W * H, NF, out.data_ptr(), None, None)
torch.cuda.synchronize()
ctx.kernel_timing(0, reset=True)
for _ in range(5):
    ctx.predict_frames_device(d8.data_ptr(), 1, W, W * H, NF, out.data_ptr(), None, None)
torch.cuda.synchronize()
ms, n = ctx.kernel_timing(0, reset=True)
print(f"{widths} x {depth}: {ms:.3f} ms per {NF} pictures = {NF * ctx.num_ctus / ms / 1e3:.2f} M CTU/s")
ctx.close()

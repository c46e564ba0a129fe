#!/usr/bin/env python3
"""Phase cycles of the fused two-convolution kernel (k_cnn_d2.inc) from a STAMPED variant library (WRONG depth maps on purpose):
   CNN_FLAGS="-DFHEVC_D2_STAMPS=<wave + 1>" tools/build_variant.sh WORK d2stamp<wave>;  python3 tools/d2_stamps.py build/ab/d2stamp<wave>.so"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fasthevc_amd import capi, frames, weights  # noqa: E402

capi.LIB_PATH = sys.argv[1]
fam = weights.load_any(os.path.join(ROOT, "fasthevc_amd", "weights", "depthnet_family_d2.fhw"))
W, H, NF = 1920, 1080, 64
base = frames.hetero_luma(W, H)
d8 = torch.from_numpy(np.stack([np.roll(base, 3 * f, axis=1) for f in range(NF)])).cuda()
ctx = capi.Context(W, H, 8, fam, max_frames=NF)
out = torch.zeros((NF, ctx.num_ctus, 256), dtype=torch.uint8, device="cuda")
for _ in range(3):
    ctx.predict_frames_device(d8.data_ptr(), 1, W, W * H, NF, out.data_ptr(), None, None)
torch.cuda.synchronize()
st = out[0, :256, :96].cpu().numpy().copy().view(np.uint64).astype(np.float64)   # [workgroup][12]
per = st[:, :9] / st[:, 9:10]
names = ["conv1a", "barrier waits", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "heads | staging", "depth map"]
m = per.mean(axis=0)
print(f"{sys.argv[1]}: {st[:, 9].mean():.1f} CTUs per workgroup, {m.sum():.0f} ticks per CTU")
for n, v in zip(names, m):
    print(f"  {n:16s} {v:8.0f}  {100 * v / m.sum():5.1f} %")

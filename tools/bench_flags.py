#!/usr/bin/env python3
"""Diagnostic: depth CNN launch time with and without the split-flag words (the multi-GPU all-gather payload)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasthevc_amd import capi, frames, weights  # noqa: E402

W, H, NF = 1920, 1080, 64
ctx = capi.Context(W, H, 8, weights.random_weights(0), max_frames=NF)
dev = torch.device("cuda:0")
base = torch.from_numpy(frames.hetero_luma(W, H)).to(dev)
gop = torch.stack([torch.roll(base, 3 * f, 1) for f in range(NF)]).contiguous()
depth = torch.zeros((NF, ctx.num_ctus, 256), dtype=torch.uint8, device=dev)
flags = torch.zeros((NF, ctx.num_ctus), dtype=torch.int32, device=dev)
ctx.enable_kernel_timing(True)
for name, fl in (("no flags", None), ("flags", flags.data_ptr()), ("no flags", None), ("flags", flags.data_ptr())):
    for i in range(25):
        if i == 5:
            ctx.kernel_timing(0, reset=True)
        ctx.predict_frames_device(gop.data_ptr(), 1, W, W * H, NF, depth.data_ptr(), None, None, d_flags=fl)
    torch.cuda.synchronize()
    t = ctx.kernel_timing(0)
    print(name, t)

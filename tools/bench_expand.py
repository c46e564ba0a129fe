#!/usr/bin/env python3
"""Diagnostic: time of the split-flag expansion (the per-step cost every rank pays for the gathered words of all ranks)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasthevc_amd import capi, weights  # noqa: E402

W, H = 1920, 1080
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 512  # 8 ranks x 64 frames
ctx = capi.Context(W, H, 8, weights.random_weights(0), max_frames=64)
dev = torch.device("cuda:0")
flags = torch.randint(0, 1 << 21, (NF, ctx.num_ctus), dtype=torch.int32, device=dev)
depth = torch.zeros((NF, ctx.num_ctus, 256), dtype=torch.uint8, device=dev)
ts = torch.cuda.Stream()  # an explicit stream: a NULL handle means the library's own stream, which torch events do not see
torch.cuda.set_stream(ts)
st = ts.cuda_stream
for _ in range(3):
    ctx.expand_depth_flags_device(flags.data_ptr(), NF, depth.data_ptr(), stream=st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ctx.expand_depth_flags_device(flags.data_ptr(), NF, depth.data_ptr(), stream=st)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"expand {NF} frames: {ms * 1000:.1f} us, {NF * ctx.num_ctus * 256 / ms / 1e6:.0f} GB/s written")

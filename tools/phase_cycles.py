#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the depth CNN kernel (stamped instantiation, NOT the timed kernel)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasthevc_amd import capi, frames, weights  # noqa: E402

W, H, NF = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = capi.Context(W, H, 8, weights.random_weights(0))
dev = torch.device("cuda:0")
base = torch.from_numpy(frames.hetero_luma(W, H)).to(dev)
gop = torch.stack([torch.roll(base, 3 * f, 1) for f in range(NF)]).contiguous()
depth = torch.zeros((NF, ctx.num_ctus, 256), dtype=torch.uint8, device=dev)
lib = ctx.lib
lib.fhevc_debug_cnn_phase_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p]
out = np.zeros(10, np.float64)
for rep in range(3):
    rc = lib.fhevc_debug_cnn_phase_cycles(ctx.h, gop.data_ptr(), 1, W, W * H, NF, depth.data_ptr(), out.ctypes.data)
    assert rc == 0, rc
names = ["prologue", "P1 conv1", "P2 conv2", "P3 conv3", "P4c barrier wait", "P5 depth", "P4a heads", "P4b staging"]
per_ctu = out[:8] / out[8]
tot = per_ctu.sum()
print(f"grid {int(out[9])}, {out[8]:.1f} CTUs per workgroup, {tot:.0f} cycles per CTU per workgroup (wave 0, incl. barrier waits)")
for n, c in zip(names, per_ctu):
    print(f"  {n:16s} {c:9.0f} cycles  {100 * c / tot:5.1f} %")

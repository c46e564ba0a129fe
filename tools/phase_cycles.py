#!/usr/bin/env python3
"""Diagnostic: per-interval cycles of the depth CNN kernel (stamped instantiation, NOT the timed kernel) and the in-kernel clock
(MI355X_MICROARCH.md 'DVFS give-back' item 6: s_memtime span / s_memrealtime span of each workgroup's CTU loop, after >= 2 s of
back-to-back launches of the timed kernel on the bench workload).  usage: python tools/phase_cycles.py [frames=64] [warm seconds=2]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasthevc_amd import capi, frames, weights  # noqa: E402

W, H, NF = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 64
WARM = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
blob = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fasthevc_amd", "weights", "depthnet_v2.fhw")
ctx = capi.Context(W, H, 8, weights.load(blob), max_frames=NF)
dev = torch.device("cuda:0")
base = torch.from_numpy(frames.hetero_luma(W, H)).to(dev)
gop = torch.stack([torch.roll(base, 3 * f, 1) for f in range(NF)]).contiguous()
depth = torch.zeros((NF, ctx.num_ctus, 256), dtype=torch.uint8, device=dev)
had = torch.zeros((NF, ctx.num_ctus), dtype=torch.int32, device=dev)
lib = ctx.lib
lib.fhevc_debug_cnn_phase_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p]
t0 = time.time()
while time.time() - t0 < WARM:   # the timed kernel back to back: the clock the chip settles at under THIS load
    for _ in range(50):
        ctx.predict_frames_device(gop.data_ptr(), 1, W, W * H, NF, depth.data_ptr(), had.data_ptr())
    torch.cuda.synchronize()
os.environ["FHEVC_DEBUG_STAMPS_BY_SLOT"] = "1"
out = np.zeros(39, np.float64)
acc, clocks = [], []
for rep in range(5):
    rc = lib.fhevc_debug_cnn_phase_cycles(ctx.h, gop.data_ptr(), 1, W, W * H, NF, depth.data_ptr(), out.ctypes.data)
    assert rc == 0, rc
    acc.append(out.copy())
    clocks.append(out[10])
out = np.median(np.stack(acc), axis=0)
names = ["prologue", "P1 conv1", "P2 conv2", "P3 conv3 (+Hadamard)", "P4c barrier wait", "P5 depth", "P4a heads", "P4b staging"]
per_ctu = out[:8] / out[8]
tot = per_ctu.sum()
print(f"grid {int(out[9])}, {out[8]:.1f} CTUs per workgroup, {tot:.0f} cycles per CTU per workgroup (wave 0, incl. barrier waits)")
for n, c in zip(names, per_ctu):
    print(f"  {n:22s} {c:9.0f} cycles  {100 * c / tot:5.1f} %")
if out[12 + 8] + out[21 + 8] + out[30 + 8] > 0:
    print("by the workgroup's slot on its CU (the i8 form's conv phases run at s_setprio 1 + slot): cycles per CTU")
    print("  slot  workgroups " + " ".join(f"{n.split()[0]:>9s}" for n in names) + "     total")
    for sl in range(3):
        v = out[12 + 9 * sl:12 + 9 * sl + 8] / out[8]
        print(f"  {sl}     {int(out[12 + 9 * sl + 8]):5d}      " + " ".join(f"{c:9.0f}" for c in v) + f" {v.sum():9.0f}")
print(f"in-kernel clock: median over the workgroups {out[10]:.0f} MHz, slowest workgroup {out[11]:.0f} MHz  (five stamped launches: {', '.join('%.0f' % c for c in clocks)})")
print(f"=> {tot / (out[10] * 1e6) * 1e6:.2f} us per CTU and workgroup; at {int(out[9])} workgroups {NF * ctx.num_ctus / (out[9]) * tot / (out[10] * 1e6) * 1e3:.4f} ms per launch of {NF * ctx.num_ctus} CTUs (stamped build)")

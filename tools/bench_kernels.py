#!/usr/bin/env python3
"""Secondary kernels of the path, timed on the GPU with the library's own HIP events (fhevc_kernel_timing):
AQ pre-analysis (N3), source Hadamard on uint8 and int16 planes, and the 35-mode first pass (A4/A5).
Prints one JSON object; bench.py remains the contract benchmark for the headline metric."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasthevc_amd import capi, frames, weights  # noqa: E402

W, H, NF = 1920, 1080, 64
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
torch.cuda.init()
ctx = capi.Context(W, H, 8, weights.random_weights(0), max_frames=NF)
ctx.enable_kernel_timing(True)
os.environ["FHEVC_FUSE_HADAMARD"] = "0"   # a second context that keeps the stand-alone source-Hadamard launch (the fused form is the default)
ctx_unfused = capi.Context(W, H, 8, weights.random_weights(0), max_frames=NF)
ctx_unfused.enable_kernel_timing(True)
del os.environ["FHEVC_FUSE_HADAMARD"]
base = frames.hetero_luma(W, H)
planes = np.stack([frames.to_pel_plane(np.roll(base, 3 * f, axis=1), 8)[0] for f in range(NF)])
_, org, stride = frames.to_pel_plane(base, 8)
d16 = torch.from_numpy(planes).to(dev)
fs = planes.shape[1] * planes.shape[2]
d8 = torch.from_numpy(np.stack([np.roll(base, 3 * f, axis=1) for f in range(NF)])).to(dev)
out = {"workload": f"{NF} frames {W}x{H}", "reps": REPS}

# --- AQ pre-analysis, all four layers
off = ctx.aq_layout(4)
act = torch.zeros((NF, off[-1]), dtype=torch.float64, device=dev)
for layout, ptr, sb, st, fstride in (("int16 HM planes", d16.data_ptr() + 2 * org, 2, stride, fs), ("uint8 packed", d8.data_ptr(), 1, W, W * H)):
    for _ in range(3):
        ctx.preanalyze_frames_device(ptr, sb, st, fstride, NF, act.data_ptr(), 4)
    torch.cuda.synchronize()
    ctx.kernel_timing(3, reset=True)
    for _ in range(REPS):
        ctx.preanalyze_frames_device(ptr, sb, st, fstride, NF, act.data_ptr(), 4)
    torch.cuda.synchronize()
    ms, n = ctx.kernel_timing(3, reset=True)
    alg = NF * (W * H * sb + off[-1] * 8)
    out[f"preanalyze[{layout}]"] = {"ms": ms, "launches": n, "algorithmic_bytes": alg, "GB/s": alg / ms / 1e6,
                                    "frac_of_8TB/s": alg / ms / 1e6 / 8000}

# --- source Hadamard as its own launch (context without the fusion) and the depth kernel with / without the fused Hadamard
depth = torch.zeros((NF, ctx.num_ctus, 256), dtype=torch.uint8, device=dev)
had = torch.zeros((NF, ctx.num_ctus), dtype=torch.int32, device=dev)
for layout, ptr, sb, st, fstride in (("int16 HM planes", d16.data_ptr() + 2 * org, 2, stride, fs), ("uint8 packed", d8.data_ptr(), 1, W, W * H)):
    for cx in (ctx_unfused, ctx):
        for _ in range(2):
            cx.predict_frames_device(ptr, sb, st, fstride, NF, depth.data_ptr(), had.data_ptr())
        torch.cuda.synchronize()
        cx.kernel_timing(0, reset=True); cx.kernel_timing(1, reset=True)
        for _ in range(REPS):
            cx.predict_frames_device(ptr, sb, st, fstride, NF, depth.data_ptr(), had.data_ptr())
        torch.cuda.synchronize()
        ms_c, _ = cx.kernel_timing(0, reset=True)
        ms_h, n = cx.kernel_timing(1, reset=True)
        alg = NF * (W * H * sb + ctx.num_ctus * 4)
        if cx is ctx_unfused:
            out[f"src_hadamard[{layout}]"] = {"ms": ms_h, "launches": n, "algorithmic_bytes": alg, "GB/s": alg / ms_h / 1e6}
            out[f"depth_cnn i8[{layout}]"] = {"ms": ms_c, "Mctu/s": NF * ctx.num_ctus / ms_c / 1e3}
        else:
            out[f"depth_cnn i8+fused_hadamard[{layout}]"] = {"ms": ms_c, "Mctu/s": NF * ctx.num_ctus / ms_c / 1e3, "stand_alone_hadamard_launches": n}
        # the 16-bit form of the classifier on the same context
        cx.set_cnn_arith("f16")
        for _ in range(2):
            cx.predict_frames_device(ptr, sb, st, fstride, NF, depth.data_ptr(), had.data_ptr())
        torch.cuda.synchronize()
        cx.kernel_timing(0, reset=True); cx.kernel_timing(1, reset=True)
        for _ in range(REPS):
            cx.predict_frames_device(ptr, sb, st, fstride, NF, depth.data_ptr(), had.data_ptr())
        torch.cuda.synchronize()
        ms_c, _ = cx.kernel_timing(0, reset=True)
        cx.kernel_timing(1, reset=True)
        cx.set_cnn_arith("i8")
        out[f"depth_cnn f16{'' if cx is ctx_unfused else '+fused_hadamard'}[{layout}]"] = {"ms": ms_c, "Mctu/s": NF * ctx.num_ctus / ms_c / 1e3}

# --- 35-mode first pass, one picture per call (host-buffer entry point; the kernel time excludes the copies)
for _ in range(2):
    ctx.intra_first_pass(planes[0], org, stride, qp=32)
ctx.kernel_timing(2, reset=True)
for f in range(min(REPS, 8)):
    ctx.intra_first_pass(planes[f], org, stride, qp=32)
ms, n = ctx.kernel_timing(2, reset=True)
# integer work per CTU: 4 levels x 35 modes x 64 tiles x (64 predicted samples + ~575 Hadamard ops)
ops = ctx.num_ctus * 4 * 35 * 64 * (64 + 575)
out["first_pass[1 picture]"] = {"ms": ms, "launches": n, "ctu/s": ctx.num_ctus / ms * 1e3, "approx_int_ops": ops,
                                "Tint-op/s": ops / ms / 1e9}
# the same over a device-resident batch of 8 pictures in one launch
NB = 8
nodes = torch.zeros((NB * ctx.num_ctus * 85, 2), dtype=torch.float64, device=dev)  # 16 bytes per node
for _ in range(2):
    ctx.intra_first_pass_device(d16.data_ptr() + 2 * org, 2, stride, fs, NB, nodes.data_ptr(), qp=32)
torch.cuda.synchronize()
ctx.kernel_timing(2, reset=True)
for _ in range(5):
    ctx.intra_first_pass_device(d16.data_ptr() + 2 * org, 2, stride, fs, NB, nodes.data_ptr(), qp=32)
torch.cuda.synchronize()
ms, n = ctx.kernel_timing(2, reset=True)
out[f"first_pass[{NB} pictures, device batch]"] = {"ms": ms, "launches": n, "ctu/s": NB * ctx.num_ctus / ms * 1e3,
                                                  "Tint-op/s": NB * ops / ms / 1e9}
# --- config 4: source-only motion search per CU node, 16 pictures (15 pairs), search range 4 and 8
mot = torch.zeros((15 * ctx.num_ctus * 85, 4), dtype=torch.int32, device=dev)
for rng_ in (4, 8):
    for _ in range(2):
        ctx.motion_search_device(d16.data_ptr() + 2 * org, 2, stride, fs, 16, mot.data_ptr(), qp=38, search_range=rng_)
    torch.cuda.synchronize()
    ctx.kernel_timing(4, reset=True)
    for _ in range(5):
        ctx.motion_search_device(d16.data_ptr() + 2 * org, 2, stride, fs, 16, mot.data_ptr(), qp=38, search_range=rng_)
    torch.cuda.synchronize()
    ms, n = ctx.kernel_timing(4, reset=True)
    mops = 15 * ctx.num_ctus * (2 * rng_ + 1) ** 2 * 64 * (64 + 575)  # per (vector, 8x8 tile): 64 differences + ~575 Hadamard ops
    out[f"motion_search[15 pairs, range {rng_}]"] = {"ms": ms, "launches": n, "ctu/s": 15 * ctx.num_ctus / ms * 1e3, "approx_int_ops": mops,
                                                     "Tint-op/s": mops / ms / 1e9}
# --- the same search in its SAD mode at HM's SearchRange (k_motion_wide.hip), 5 pictures (4 pairs), int16 planes and uint8 planes
ctx.set_motion_distortion("sad")
for rng_ in (16, 32, 64):
    for _ in range(2):
        ctx.motion_search_device(d16.data_ptr() + 2 * org, 2, stride, fs, 5, mot.data_ptr(), qp=38, search_range=rng_)
    torch.cuda.synchronize()
    ctx.kernel_timing(4, reset=True)
    for _ in range(4):
        ctx.motion_search_device(d16.data_ptr() + 2 * org, 2, stride, fs, 5, mot.data_ptr(), qp=38, search_range=rng_)
    torch.cuda.synchronize()
    ms, n = ctx.kernel_timing(4, reset=True)
    vec = (2 * rng_ + 1) ** 2
    out[f"motion_search_sad_wide[4 pairs, range {rng_}]"] = {"ms": ms, "launches": n, "ctu/s": 4 * ctx.num_ctus / ms * 1e3, "vectors_per_node": vec,
                                                             "node_vectors/s": 4 * ctx.num_ctus * 85 * vec / ms * 1e3,
                                                             "qsad_Tlaneop/s": 4 * ctx.num_ctus * vec * 4096 / 16 / ms / 1e9}
ctx.set_motion_distortion("satd")
print(json.dumps(out, indent=1))
ctx.close()
ctx_unfused.close()

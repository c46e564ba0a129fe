#!/usr/bin/env python3
"""Longer randomised parity sweep than tests/test_gpu_random_stress.py (GPU box): 240 seeded configurations (sizes, bit depths, QPs, blobs
incl. maximum-magnitude ones and random requant shifts), both arithmetic forms of the classifier against the CPU oracle, bit for bit."""
import ctypes as C, sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_py as op
from fasthevc_amd import capi, frames, weights
oracle = op.load_oracle()
bad = 0
for seed in range(1000, 1240):
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(8, 60)) * 8, int(rng.integers(8, 40)) * 8
    bd = int(rng.choice([8, 8, 10, 12]))
    qp = int(rng.integers(0, 52))
    w = weights.random_weights(seed, extreme=bool(seed % 5 == 0))
    if seed % 7 == 0:
        w["shift"] = np.array([int(rng.integers(4, 8)), int(rng.integers(0, 15)), int(rng.integers(0, 15))], np.int32)
    m = frames.HM_MARGIN
    stride = W + 2 * m
    buf = np.zeros((H + 2 * m, stride), np.int16)
    top = (1 << bd) - 1
    kind = seed % 3
    if kind == 0: y = rng.integers(0, top + 1, (H, W))
    elif kind == 1: y = (frames.fractal_luma(W + 8, H + 8, seed=seed)[:H, :W].astype(np.int64) << (bd - 8))
    else: y = np.where((np.add.outer(np.arange(H) // 3, np.arange(W) // 5)) % 2 == 0, 0, top)
    buf[m:m + H, m:m + W] = y
    org = m * stride + m
    cw, ch = frames.ctu_grid(W, H); n = cw * ch
    depth_ref, logits = np.zeros(n * 256, np.uint8), np.zeros(n * 42, np.int32)
    oracle.fho_predict_frame(op.weights_from_arrays(w), op.ptr(buf.reshape(-1), org), stride, W, H, bd, qp, depth_ref, C.c_void_p(logits.ctypes.data))
    had_ref = np.zeros(n, np.int32)
    oracle.fho_frame_src_hadamard(op.ptr(buf.reshape(-1), org), stride, W, H, had_ref)
    for arith in ("i8", "f16"):
        ctx = capi.Context(W, H, bd, w, arith=arith)
        depth, had = ctx.predict_frame(buf, org, stride, qp=qp)
        if not (np.array_equal(depth.reshape(-1), depth_ref) and np.array_equal(had, had_ref)):
            bad += 1; print("MISMATCH", seed, arith, W, H, bd, qp, w["shift"])
        ctx.close()
print("done, mismatches:", bad)

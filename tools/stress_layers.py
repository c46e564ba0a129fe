#!/usr/bin/env python3
"""Randomised parity sweep of the layer-by-layer family path (k_cnn_layers.inc) on the GPU box: random widths (1..128, last a multiple of 4), 1..3
convolutions per block, random shifts, picture sizes, bit depths and QPs, against the CPU oracle (fho_predict_frame_family), bit for bit."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FHEVC_FAMILY_LAYERS"] = "1"
from oracle import oracle_py as op  # noqa: E402
from fasthevc_amd import capi, frames, weights  # noqa: E402

oracle = op.load_oracle()
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for seed in range(2000, 2000 + N):
    rng = np.random.default_rng(seed)
    depth = int(rng.integers(1, 4))
    widths = (int(rng.integers(1, 129)), int(rng.integers(1, 129)), 4 * int(rng.integers(1, 33)))
    W, H = int(rng.integers(8, 40)) * 8, int(rng.integers(8, 30)) * 8
    bd = int(rng.choice([8, 8, 10, 12]))
    qp = int(rng.integers(0, 52))
    fam = weights.random_family(widths, depth, seed=seed)
    if seed % 3 == 0:
        fam["shift"] = rng.integers(3, 12, size=(3, 3)).astype(np.int32)
    if seed % 2 == 0:
        os.environ["FHEVC_LAYERS_NO_FUSE"] = "1"
    else:
        os.environ.pop("FHEVC_LAYERS_NO_FUSE", None)
    if seed % 5 == 0:   # the single-buffered staging instead of the double-buffered LDS-direct one
        os.environ["FHEVC_LAYERS_NO_DBUF"] = "1"
    else:
        os.environ.pop("FHEVC_LAYERS_NO_DBUF", None)
    luma = frames.fractal_luma(W + 8, H + 8, seed=seed)[:H, :W].copy() if seed % 2 else frames.texture16_luma(W, H, seed=seed)
    buf, org, stride = frames.to_pel_plane(luma, bd)
    n = ((W + 63) // 64) * ((H + 63) // 64)
    depth_ref, logits_ref = np.zeros(n * 256, np.uint8), np.zeros(n * 42, np.int32)
    f = op.family_from_arrays(fam)
    oracle.fho_predict_frame_family(C.byref(f), op.ptr(buf.reshape(-1), org), stride, W, H, bd, qp, depth_ref.ctypes.data, logits_ref.ctypes.data)
    ctx = capi.Context(W, H, bd, fam)
    d, _ = ctx.predict_frame(buf, org, stride, qp=qp)
    ctx.close()
    ok = np.array_equal(d.reshape(-1), depth_ref)
    if not ok:
        bad += 1
        print("MISMATCH", seed, widths, depth, W, H, bd, qp, flush=True)
    if (seed - 2000) % 20 == 19:
        print(seed - 2000 + 1, "configurations,", bad, "mismatches", flush=True)
print("done, mismatches:", bad)

#!/usr/bin/env python3
"""TEST/QUALITY INFRASTRUCTURE (uses oracle/_ref): training-set generator: replaces the reference's HARP dump + MATLAB getData pipeline
(Src_HARP/CShow_PredResiReco.h:127-255, matlab/dataExtraction/detectAndClassify32Cu.m).

For seeded synthetic pictures it runs the REFERENCE's own full-RDO decision path (oracle/_ref/libhmref.so:
TEncSlice::compressSlice -> TEncCu::xCompressCU through oracle/ref_rdo_harness.cpp) at several QPs and stores, per
full 64x64 CTU: the 8-bit luma tile and the 16x16 depth map.  Build-container only (needs oracle/_ref).

usage: python tests/quality/make_labels.py --out /tmp/fhevc_labels --frames 64 --workers 8
"""
import argparse
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/quality -> repo root
sys.path.insert(0, ROOT)
from fasthevc_amd import frames  # noqa: E402

QPS = (22, 27, 32, 37)
W, H = 1024, 576  # 16 x 9 full CTUs


def training_picture(seed):
    """Seeded content different from the two pinned evaluation pictures (other seeds, random contrast/brightness)."""
    rng = np.random.default_rng(10_000 + seed)
    kind = seed % 4
    if seed >= 5000:  # further families, flat mid-grey chroma: 1000.. fractal, 2000.. gratings, 3000.. polygons, 4000.. chirps,
        return frames.deadleaves_luma(W, H, seed=70_000 + seed, leaves=int(rng.integers(150, 900))), "flatchroma"  # 5000.. dead leaves
    if seed >= 4000:
        return frames.chirp_luma(W, H, seed=70_000 + seed), "flatchroma"
    if seed >= 3000:
        return frames.polygon_luma(W, H, seed=70_000 + seed), "flatchroma"
    if seed >= 2000:
        return frames.gratings_luma(W, H, seed=70_000 + seed), "flatchroma"
    if seed >= 1000:
        return frames.fractal_luma(W, H, seed=70_000 + seed), "flatchroma"
    if kind == 3:
        y = frames.texture16_luma(W, H, seed=50_000 + seed, frame=int(rng.integers(0, 50))).astype(np.float64)
    else:
        y = frames.hetero_luma(W, H, seed=20_000 + seed).astype(np.float64)
    gain = rng.uniform(0.6, 1.4)
    offs = rng.uniform(-25, 25)
    y = (y - 128.0) * gain + 128.0 + offs
    if rng.random() < 0.3:  # soften some pictures: more large CUs
        y = frames._box_blur(y, 3)
    return np.clip(np.rint(y), 0, 255).astype(np.uint8), ("texture16" if kind == 3 else "hetero")


COSTS = False  # --costs: the cost-recording build of the reference (make -C oracle costs): also store, per CTU and QP, the RD costs of
               # "best non-split mode" and "split" of the 21 nodes xCompressCU compares (cost_q<qp> [numCtus, 21, 2] float32)


def work(seed):
    from oracle import oracle_py as op
    lib = op.bind_rdo(op.load_ref(hook="costs" if COSTS else False))
    luma, kind = training_picture(seed)
    chroma = None if kind == "flatchroma" else tuple(c.astype(np.int16) for c in frames.chroma_planes(kind, W, H))
    buf, org, stride = frames.to_pel_plane(luma, 8)
    cw, ch = W // 64, H // 64
    tiles = luma.reshape(ch, 64, cw, 64).transpose(0, 2, 1, 3).reshape(cw * ch, 64, 64)
    out = {"tiles": tiles}
    for qp in QPS:
        depth, st = op.rdo_encode(lib, buf, org, stride, W, H, 8, qp, chroma=chroma)
        out[f"depth_q{qp}"] = depth.reshape(cw * ch, 16, 16)
        if COSTS:
            out[f"cost_q{qp}"] = op.split_costs(lib, cw * ch).astype(np.float32)
        out[f"stats_q{qp}"] = np.array([st["bits"], st["dist"], st["psnr_y"], st["seconds"]])
    return seed, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="/tmp/fhevc_labels")
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--costs", action="store_true", help="also record the split / no-split RD costs of every node (needs `make -C oracle costs`)")
    args = ap.parse_args()
    global COSTS
    COSTS = args.costs
    os.makedirs(args.out, exist_ok=True)
    t0 = time.time()
    with Pool(args.workers) as pool:
        for i, (seed, out) in enumerate(pool.imap_unordered(work, range(args.first, args.first + args.frames))):
            np.savez_compressed(os.path.join(args.out, f"pic_{seed:05d}.npz"), **out)
            if i % 8 == 0:
                print(f"{i + 1}/{args.frames} pictures, {time.time() - t0:.0f} s", flush=True)
    print("done", time.time() - t0)


if __name__ == "__main__":
    main()

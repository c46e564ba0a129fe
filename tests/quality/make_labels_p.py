#!/usr/bin/env python3
"""TEST/QUALITY INFRASTRUCTURE (uses oracle/_ref/libhmref_p.so): P-picture training labels.  For seeded 4-frame pan clips of
every synthetic family, encode I P P P through the P-slice variant of the reference (vanilla HM-16.14 decision path,
hm_patch/restore_inter.py) at QP {22,27,32,37} (P pictures at QP + 6) and store, per full CTU of POC 2 and POC 3 (P pictures
whose reference is a P picture): the 8-bit luma tile and HM's 16x16 depth map, keyed by the P picture's own QP.
Same file format as make_labels.py, so fasthevc_amd/train/train.py --qps 28,33,38,43 trains on it unchanged.

usage: python tests/quality/make_labels_p.py --out /tmp/fhevc_labels_p --clips 96 --workers 8
"""
import argparse
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fasthevc_amd import frames  # noqa: E402

QPS = (22, 27, 32, 37)
W, H = 1024, 576


def clip(seed):
    rng = np.random.default_rng(90_000 + seed)
    fast = seed >= 2000                  # seeds 2000..: motion of up to 48 (horizontal) / 24 (vertical) samples per picture, for the +-64 search
    vx, vy = (48, 24) if fast else (4, 2)
    if 1000 <= seed < 2000 or seed >= 3000:  # two overlaid motions (structure one way, noise the other), as in the config-4 evaluation clip
        return frames.pan_clip(W, H, 4, seed=70_000 + seed, v_structure=int(rng.integers(-vx, vx + 1)), v_noise=int(rng.integers(-vx, vx + 1)))
    gen = [frames.hetero_luma, frames.texture16_luma, frames.fractal_luma, frames.gratings_luma, frames.polygon_luma,
           frames.chirp_luma, frames.deadleaves_luma][seed % 7]
    px, py = (8 + 3 * vx, 8 + 3 * vy) if fast else (32, 32)
    big = gen(W + 2 * px, H + 2 * py, seed=80_000 + seed)
    dx, dy = int(rng.integers(-vx, vx + 1)), int(rng.integers(-vy, vy + 1))   # global pan per frame
    out = []
    for f in range(4):
        x0, y0 = px + dx * f, py + dy * f
        y = big[y0:y0 + H, x0:x0 + W].astype(np.float64)
        y = y + rng.normal(0, 0.7, size=y.shape) + 0.4 * f        # a little sensor noise and brightness drift
        out.append(np.clip(np.rint(y), 0, 255).astype(np.uint8))
    return out


def work(seed):
    import eval_p
    lib = eval_p.load_p()
    ys = clip(seed)
    cw, ch = W // 64, H // 64
    tiles = lambda y: y.reshape(ch, 64, cw, 64).transpose(0, 2, 1, 3).reshape(cw * ch, 64, 64)
    out = {"tiles": np.concatenate([tiles(ys[2]), tiles(ys[3])])}
    for qp in QPS:
        seq = eval_p.encode_seq(lib, ys, qp)
        out[f"depth_q{qp + 6}"] = np.concatenate([seq[2][0].reshape(cw * ch, 16, 16), seq[3][0].reshape(cw * ch, 16, 16)])
        out[f"stats_q{qp + 6}"] = np.array([[s["bits"], s["psnr_y"], s["seconds"], s["skip_share"]] for _, s in seq])
    return seed, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="/tmp/fhevc_labels_p")
    ap.add_argument("--clips", type=int, default=96)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--workers", type=int, default=8)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    t0 = time.time()
    with Pool(args.workers) as pool:
        for i, (seed, out) in enumerate(pool.imap_unordered(work, range(args.first, args.first + args.clips))):
            np.savez_compressed(os.path.join(args.out, f"pic_{seed:05d}.npz"), **out)
            if i % 8 == 0:
                print(f"{i + 1}/{args.clips} clips, {time.time() - t0:.0f} s", flush=True)
    print("done", time.time() - t0)


if __name__ == "__main__":
    main()

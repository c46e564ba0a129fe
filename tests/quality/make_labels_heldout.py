#!/usr/bin/env python3
"""TEST/QUALITY INFRASTRUCTURE (uses oracle/_ref).  Cost labels (make_labels.py --costs layout) for the HELD-OUT evaluation families -- the zone-plate /
checkerboard mix "ood" (eval_rd.ood_luma), "glyphs" and "waves" (fasthevc_amd/frames.py, round 4) -- none of which any training label ever contained.
Seeds are those of eval_families.py's evaluation pictures (900 000 + 1000 x family index + k), so the study and the BD-rate runs see the same pictures.

usage: python tests/quality/make_labels_heldout.py --out /tmp/fhevc_labels_ho --pictures 8 --workers 8   (needs `make -C oracle costs`)
"""
import argparse
import os
import sys
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fasthevc_amd import frames  # noqa: E402

QPS = (22, 27, 32, 37)
W, H = 1024, 576
HELD_OUT = ("ood", "glyphs", "waves")


def work(job):
    family, k = job
    import eval_families
    from oracle import oracle_py as op
    lib = op.bind_rdo(op.load_ref(hook="costs"))
    luma = eval_families.picture(family, k, W, H)
    buf, org, stride = frames.to_pel_plane(luma, 8)
    cw, ch = W // 64, H // 64
    out = {"tiles": luma.reshape(ch, 64, cw, 64).transpose(0, 2, 1, 3).reshape(cw * ch, 64, 64)}
    for qp in QPS:
        depth, st = op.rdo_encode(lib, buf, org, stride, W, H, 8, qp, chroma=None)
        out[f"depth_q{qp}"] = depth.reshape(cw * ch, 16, 16)
        out[f"cost_q{qp}"] = op.split_costs(lib, cw * ch).astype(np.float32)
        out[f"stats_q{qp}"] = np.array([st["bits"], st["dist"], st["psnr_y"], st["seconds"]])
    return family, k, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="/tmp/fhevc_labels_ho")
    ap.add_argument("--pictures", type=int, default=8)
    ap.add_argument("--workers", type=int, default=8)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    jobs = [(f, k) for f in HELD_OUT for k in range(args.pictures)]
    with Pool(args.workers) as pool:
        for family, k, out in pool.imap_unordered(work, jobs):
            np.savez_compressed(os.path.join(args.out, f"pic_{family}_{k:03d}.npz"), **out)
            print(family, k, flush=True)


if __name__ == "__main__":
    main()

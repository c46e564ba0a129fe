#!/usr/bin/env python3
"""TEST/QUALITY INFRASTRUCTURE: motion features (oracle/fhevc_oracle.c: fho_motion_ctu) for the P-picture label set of
make_labels_p.py.  Per clip and P-picture QP: the 85 motion nodes of every CTU of POC 3 searched in the ORIGINAL POC 2,
next to HM's depth maps of POC 2 (the reference picture's depths) and POC 3 (the label).

usage: python tests/quality/p_features.py --labels /tmp/fhevc_labels_p --out /tmp/pfit/feats.npz [--range 4]
"""
import argparse
import ctypes as C
import glob
import os
import sys
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

MOTION_DTYPE = np.dtype([("satd_zero", np.uint32), ("satd_best", np.uint32), ("cost_best", np.uint32), ("mvx", np.int16), ("mvy", np.int16)])
QPS = (28, 33, 38, 43)


def motion_frame(oracle, cur, ref, qp, rng, bit_depth=8, dist=0):
    """85 motion nodes per CTU of `cur` (uint8 [H, W]) searched in `ref`; dist: 0 Hadamard SATD, 1 SAD (ranges above 8: the wide kernel's mode)"""
    from fasthevc_amd import frames
    H, W = cur.shape
    cb, co, cs = frames.to_pel_plane(cur, bit_depth)
    rb, ro, rs = frames.to_pel_plane(ref, bit_depth)
    cw, ch = (W + 63) // 64, (H + 63) // 64
    out = np.zeros((cw * ch, 85), MOTION_DTYPE)
    sl = oracle.fho_lambda_intra(qp, bit_depth) ** 0.5
    cp, rp = cb.reshape(-1).ctypes.data + 2 * co, rb.reshape(-1).ctypes.data + 2 * ro
    for c in range(cw * ch):
        oracle.fho_motion_ctu_dist(C.c_void_p(cp), cs, C.c_void_p(rp), rs, W, H, c % cw, c // cw, bit_depth, rng, C.c_double(sl), dist,
                                   C.c_void_p(out[c].ctypes.data))
    return out


def work(args):
    path, rng, dist, mc = args
    import make_labels_p
    from oracle import oracle_py as op
    oracle = op.load_oracle()
    seed = int(os.path.basename(path)[4:9])
    ys = make_labels_p.clip(seed)
    d = np.load(path)
    n = d["tiles"].shape[0] // 2
    out = {}
    H, W = ys[3].shape
    oracle.fho_p_motion_compensated_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    for qp in QPS:
        nodes = motion_frame(oracle, ys[3], ys[2], qp, rng, dist=dist)
        out[f"nodes_q{qp}"] = nodes
        prev = np.ascontiguousarray(d[f"depth_q{qp}"][:n].reshape(n, 256))
        if mc:  # the reference picture's depths seen through the motion (fhevc_p_motion_compensated_depth) instead of co-located
            seen = np.zeros_like(prev)
            for c in range(n):
                oracle.fho_p_motion_compensated_depth(nodes[c].ctypes.data, prev.ctypes.data, W, H, c, seen[c].ctypes.data)
            prev = seen
        out[f"prev_q{qp}"] = prev
        out[f"label_q{qp}"] = d[f"depth_q{qp}"][n:].reshape(n, 256)
    return seed, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--labels", default="/tmp/fhevc_labels_p")
    ap.add_argument("--out", default="/tmp/pfit/feats.npz")
    ap.add_argument("--range", type=int, default=4)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--dist", default="satd", choices=("satd", "sad"), help="sad: HM's integer-search distortion (needed for ranges above 8)")
    ap.add_argument("--mc", action="store_true", help="previous depths taken at the motion-compensated position")
    args = ap.parse_args()
    files = sorted(glob.glob(os.path.join(args.labels, "pic_*.npz")))
    acc = {}
    with Pool(args.workers) as pool:
        for i, (seed, out) in enumerate(pool.imap(work, [(f, args.range, 1 if args.dist == "sad" else 0, args.mc) for f in files])):
            for k, v in out.items():
                acc.setdefault(k, []).append(v)
            acc.setdefault("seed", []).append(np.full(out["prev_q28"].shape[0], seed))
            if i % 16 == 0:
                print(i + 1, "/", len(files), flush=True)
    np.savez_compressed(args.out, **{k: np.concatenate(v) for k, v in acc.items()})
    print("wrote", args.out)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""TEST/QUALITY INFRASTRUCTURE, GPU box: the features of p_features.py computed by the PRODUCT path instead of the oracle -- the MI355X's own
motion search (bit-exact with the oracle, and at +-64 about 50 000 times faster than one host core) and fhevc_p_motion_compensated_depth.
Input: the depth maps of the P-picture label files in one compact archive (keys "<seed>_q<qp>": [2 n, 16, 16], POC 2 then POC 3), the clips
are regenerated from their seeds (make_labels_p.clip).

usage: python tests/quality/p_features_gpu.py build/pfit_in/depths_slow.npz gpurun_out/feats_slow.npz --range 64 --dist sad --mc
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fasthevc_amd import capi, frames  # noqa: E402
import make_labels_p  # noqa: E402

QPS = (28, 33, 38, 43)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("depths")
    ap.add_argument("out")
    ap.add_argument("--range", type=int, default=4)
    ap.add_argument("--dist", default="satd", choices=("satd", "sad"))
    ap.add_argument("--mc", action="store_true")
    args = ap.parse_args()
    z = np.load(args.depths)
    seeds = sorted({int(k.split("_")[0]) for k in z.files})
    W, H = make_labels_p.W, make_labels_p.H
    ctx = capi.Context(W, H, 8)
    ctx.set_motion_distortion(args.dist)
    acc = {}
    for i, seed in enumerate(seeds):
        ys = make_labels_p.clip(seed)
        (rb, org, stride), (cb, _, _) = frames.to_pel_plane(ys[2], 8), frames.to_pel_plane(ys[3], 8)
        for qp in QPS:
            d = z[f"{seed}_q{qp}"]
            n = d.shape[0] // 2
            nodes = ctx.motion_search(cb, rb, org, stride, qp=qp, search_range=args.range)
            prev = np.ascontiguousarray(d[:n].reshape(n, 256))
            if args.mc:
                prev = capi.p_motion_compensated_depth(nodes, prev, W, H)
            acc.setdefault(f"nodes_q{qp}", []).append(nodes)
            acc.setdefault(f"prev_q{qp}", []).append(prev)
            acc.setdefault(f"label_q{qp}", []).append(d[n:].reshape(n, 256))
        acc.setdefault("seed", []).append(np.full(n, seed))
        if i % 16 == 0:
            print(i + 1, "/", len(seeds), flush=True)
    np.savez_compressed(args.out, **{k: np.concatenate(v) for k, v in acc.items()})
    print("wrote", args.out, os.path.getsize(args.out))
    ctx.close()


if __name__ == "__main__":
    main()

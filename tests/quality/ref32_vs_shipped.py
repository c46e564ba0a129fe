#!/usr/bin/env python3
"""TEST/QUALITY INFRASTRUCTURE (uses oracle/): the shipped fixed-point network's 32-level decision on the held-out quadrants of
fasthevc_amd/train/train_ref32.py (every 8th label file, CTUs with Depth(CU0_0) != 0, isDiv = Depth != 1:
matlab/dataExtraction/detectAndClassify32Cu.m:11-16, 57-62), so that the reference's single-level classifier and the shipped
network are compared on the same crops.  The shipped network sees the whole CTU and the QP.

usage: python tests/quality/ref32_vs_shipped.py --data /tmp/fhevc_labels --json profiles/r02_reference_32x32_classifier.json
       (adds the key "shipped_fixed_point_32_level" to the JSON written by train_ref32)
"""
import argparse
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fasthevc_amd import weights  # noqa: E402
from oracle import oracle_py as op  # noqa: E402

QPS = (22, 27, 32, 37)


def shipped_32_level_error(data, blob=None, stride=4):
    files = sorted(glob.glob(os.path.join(data, "pic_*.npz")))
    va_files = [f for i, f in enumerate(files) if i % 8 == 0]
    oracle = op.load_oracle()
    ws = op.weights_from_arrays(weights.load(blob or os.path.join(ROOT, "fasthevc_amd", "weights", "depthnet_v1.fhw")))
    wrong = total = 0
    per_qp = {}
    logits = np.zeros(42, np.int32)
    for f in va_files[::stride]:
        z = np.load(f)
        for qp in QPS:
            d = z[f"depth_q{qp}"]
            for c in np.nonzero(d[:, 0, 0] != 0)[0]:
                ctu = (z["tiles"][c].astype(np.int16) - 128).astype(np.int8).reshape(-1)
                oracle.fho_cnn_ctu(ws, np.ascontiguousarray(ctu), qp, logits)
                for k, (oy, ox) in enumerate(((0, 0), (0, 8), (8, 0), (8, 8))):
                    pred = logits[2 * (1 + k) + 1] > logits[2 * (1 + k)]
                    bad = int(pred != (d[c, oy, ox] != 1))
                    wrong += bad
                    total += 1
                    a = per_qp.setdefault(int(qp), [0, 0])
                    a[0] += bad
                    a[1] += 1
    return {"val_error_natural_prior": wrong / total, "crops": total, "val_error_per_qp": {k: v[0] / v[1] for k, v in per_qp.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default="/tmp/fhevc_labels")
    ap.add_argument("--blob", default=None)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    r = shipped_32_level_error(args.data, args.blob)
    print(f"shipped fixed-point network, 32-level decision on held-out quadrants: error {r['val_error_natural_prior']:.4f} ({r['crops']} crops)")
    if args.json:
        rep = json.load(open(args.json)) if os.path.exists(args.json) else {}
        rep["shipped_fixed_point_32_level"] = r
        with open(args.json, "w") as fo:
            json.dump(rep, fo, indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The reference's EXACT single-level classifier, trained and evaluated on our labels (second half).

Reference definition (MATLAB only, no weights shipped): matlab/dataExtraction/detectAndClassify32Cu.m:11-16, 57-62 -- the four 32x32
quadrants of every CTU whose top-left CU has Depth != 0, label isDiv = (Depth of the quadrant's first CU != 1); network
Train...Example.m:75-96 -- input 32x32x3 (luma replicated, zero-centred) -> conv3x3x16 pad 1 + BN + ReLU -> maxpool 2 ->
conv3x3x32 + BN + ReLU -> maxpool 2 -> conv3x3x64 + BN + ReLU -> FC(2) -> softmax; options :195-205 -- SGDM, default rate 0.01 /
momentum 0.9 / mini-batch 128 / L2 1e-4, classes balanced by splitEachLabel (:66-69).  QP is not an input (the reference stores it
with every label, CShow_PredResiReco.h:93, and never reads it).  Its one numeric trace: a saved net named by its validation error,
0.0989 (filteredResults.m:2), dataset and split unknown.

Here: float32 PyTorch, the same layers and options, crops cut from the label files of tests/quality/make_labels.py (the
reference's own full-RDO depth maps), every 8th picture held out (the split of train.py), all four QPs pooled as the reference's
dump would.  Prints the validation error (balanced, and at the natural class prior); tests/quality/ref32_vs_shipped.py adds the 32-level
error of the shipped fixed-point network on the same held-out crops to the JSON.  The reference zeroes the learning rate after the first epoch
(LearnRateDropFactor 0, period 1); the error after epoch 1 is therefore the like-for-like figure, later epochs are extra.

usage: python -m fasthevc_amd.train.train_ref32 --data /tmp/fhevc_labels [--json profiles/r02_reference_32x32_classifier.json]
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
QPS = (22, 27, 32, 37)


def crops_of(files, max_ctus_per_file=None, rng=None):
    """-> x [N, 32, 32] uint8, y [N] (isDiv), qp [N], ctu tiles index for the fixed-point comparison"""
    xs, ys, qs = [], [], []
    for f in files:
        z = np.load(f)
        tiles = z["tiles"]
        for qp in QPS:
            d = z[f"depth_q{qp}"]
            keep = np.nonzero(d[:, 0, 0] != 0)[0]                      # Depth(CU0_0) != 0
            if max_ctus_per_file and len(keep) > max_ctus_per_file:
                keep = rng.choice(keep, max_ctus_per_file, replace=False)
            for (oy, ox) in ((0, 0), (0, 32), (32, 0), (32, 32)):
                xs.append(tiles[keep, oy:oy + 32, ox:ox + 32])
                ys.append((d[keep, oy // 4, ox // 4] != 1).astype(np.int64))  # isDiv = (Depth != 1)
                qs.append(np.full(len(keep), qp))
    return np.concatenate(xs), np.concatenate(ys), np.concatenate(qs)


def net():
    return nn.Sequential(nn.Conv2d(3, 16, 3, padding=1), nn.BatchNorm2d(16), nn.ReLU(), nn.MaxPool2d(2),
                         nn.Conv2d(16, 32, 3, padding=1), nn.BatchNorm2d(32), nn.ReLU(), nn.MaxPool2d(2),
                         nn.Conv2d(32, 64, 3, padding=1), nn.BatchNorm2d(64), nn.ReLU(), nn.Flatten(), nn.Linear(8 * 8 * 64, 2))


def balance(x, y, q, rng):
    n = min((y == 0).sum(), (y == 1).sum())
    idx = np.concatenate([rng.choice(np.nonzero(y == c)[0], n, replace=False) for c in (0, 1)])
    rng.shuffle(idx)
    return x[idx], y[idx], q[idx]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default="/tmp/fhevc_labels")
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--max-train", type=int, default=240000)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    files = sorted(glob.glob(os.path.join(args.data, "pic_*.npz")))
    tr_files = [f for i, f in enumerate(files) if i % 8 != 0]
    va_files = [f for i, f in enumerate(files) if i % 8 == 0]
    xt, yt, qt = crops_of(tr_files, max_ctus_per_file=24, rng=rng)
    xv, yv, qv = crops_of(va_files)
    print(f"{len(files)} pictures: {len(yt)} training crops (isDiv share {yt.mean():.3f}), {len(yv)} validation crops (isDiv share {yv.mean():.3f})", flush=True)
    xt, yt, qt = balance(xt, yt, qt, rng)
    if len(yt) > args.max_train:
        xt, yt, qt = xt[:args.max_train], yt[:args.max_train], qt[:args.max_train]
    xvb, yvb, qvb = balance(xv, yv, qv, rng)
    mean = float(xt.mean())                                           # imageInputLayer: zero-centre normalisation

    def prep(a):
        t = torch.from_numpy(a.astype(np.float32) - mean)[:, None]
        return t.expand(-1, 3, -1, -1)                                 # luma replicated to three channels (CHelper.h:267-280)

    model = net()
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    lossf = nn.CrossEntropyLoss()
    report = {"training_crops": int(len(yt)), "validation_crops_balanced": int(len(yvb)), "validation_crops_natural": int(len(yv)), "epochs": []}

    def evaluate(x, y):
        model.eval()
        wrong = 0
        with torch.no_grad():
            for b in range(0, len(y), 2048):
                wrong += int((model(prep(x[b:b + 2048])).argmax(1).numpy() != y[b:b + 2048]).sum())
        return wrong / len(y)

    t0 = time.time()
    for ep in range(args.epochs):
        model.train()
        perm = rng.permutation(len(yt))
        for b in range(0, len(perm) - 127, 128):
            idx = perm[b:b + 128]
            opt.zero_grad()
            loss = lossf(model(prep(xt[idx])), torch.from_numpy(yt[idx]))
            loss.backward()
            opt.step()
        eb, en = evaluate(xvb, yvb), evaluate(xv, yv)
        per_qp = {int(qp): evaluate(xv[qv == qp], yv[qv == qp]) for qp in QPS}
        report["epochs"].append({"epoch": ep + 1, "val_error_balanced": eb, "val_error_natural_prior": en, "val_error_per_qp": per_qp})
        print(f"epoch {ep + 1}: validation error balanced {eb:.4f}, natural prior {en:.4f}, per QP {per_qp} ({time.time() - t0:.0f} s)", flush=True)

    # (the shipped fixed-point network's 32-level decision on the same held-out quadrants is evaluated by
    # tests/quality/ref32_vs_shipped.py: it needs the CPU oracle, which nothing under fasthevc_amd/ may import)
    report["reference_trace"] = "0.0989 validation error of one Bayesian-optimisation trial (filteredResults.m:2); dataset and split unknown"
    if args.json:
        with open(args.json, "w") as fo:
            json.dump(report, fo, indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""TEST/QUALITY INFRASTRUCTURE (uses oracle/).  Round-4 study: do the features the GPU already computes -- the 35-mode first pass (A4, TEncSearch.cpp:2271-2295:
per-node best cost), the per-8x8 source Hadamard (A6) and the QP -- help the depth decision when used TOGETHER with the classifier's logits?
(Round 1 looked at the first-pass cost ratio ALONE: AUC 0.53-0.85; SURVEY section 8(a) A4 assigns it the role of a classifier feature.)

Per label picture (tests/quality/make_labels.py --costs: CTU tiles + the reference's own no-split / split RD costs of the 21 nodes at four QPs):
  * logits of a weight blob (the trainer's exact integer forward pass = the HIP kernels' arithmetic),
  * first-pass best cost per node at each QP (oracle fho_first_pass_node: all 35 SATDs once, the cost's lambda term added per QP),
  * log ratio parent cost / sum of the four children's costs for the 21 split nodes, the same on the SATDs alone, log of the node's mean SATD per sample.
Then, per split level, a cost-weighted logistic combiner [logit difference, ratios, QP] -> split / no split is fitted on training pictures and the TREE
REGRET (train.py: RD cost of the chosen quad-tree over the cheapest, from the recorded node costs) of hard decisions is compared on the held-out pictures:
logits alone vs logits + features; and the regret left when only decisions surer than a margin are forced (the soft hook), at equal shares of forced nodes.

usage: python tests/quality/feature_study.py --labels /tmp/fhevc_labels_c --blob fasthevc_amd/weights/depthnet_family_d2.fhw --cache /tmp/fhevc_feat --json profiles/r04_feature_study.json
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fasthevc_amd import frames, weights  # noqa: E402
from fasthevc_amd.train import train as T  # noqa: E402

QPS = (22, 27, 32, 37)
MODE_BITS = np.full(35, 6.0)
MODE_BITS[0] = 2.0    # MPM index 0 (PLANAR): flag + 1 bin
MODE_BITS[1] = 3.0    # MPM index 1 (DC)
MODE_BITS[26] = 3.0   # MPM index 2 (VER)


def sqrt_lambda(qp):
    return float(np.sqrt(0.57 * 2.0 ** ((qp - 12.0) / 3.0)))


def first_pass_costs(oracle, op, luma):
    """[ctus, 85, 4] best first-pass cost per node and QP, [ctus, 85] best SATD; nodes 0 = 64, 1..4 = 32 raster, 5..20 = 16 raster, 21..84 = 8 raster"""
    H, W = luma.shape
    buf, org, stride = frames.to_pel_plane(luma, 8)
    cw, ch = (W + 63) // 64, (H + 63) // 64
    cost = np.zeros((cw * ch, 85, 4))
    satd = np.zeros((cw * ch, 85))
    best = op.NodeCost()
    allm = np.zeros(35, np.uint32)
    base = op.ptr(buf.reshape(-1), org)
    for cy in range(ch):
        for cx in range(cw):
            k = 0
            for n, g in ((64, 1), (32, 2), (16, 4), (8, 8)):
                for by in range(g):
                    for bx in range(g):
                        oracle.fho_first_pass_node(base, stride, W, H, cx * 64 + bx * n, cy * 64 + by * n, n, 8, 0.0, C.byref(best), allm.ctypes.data)
                        s = allm.astype(np.float64)
                        satd[cy * cw + cx, k] = s.min()
                        for qi, qp in enumerate(QPS):
                            cost[cy * cw + cx, k, qi] = (s + MODE_BITS * sqrt_lambda(qp)).min()
                        k += 1
    return cost, satd


def picture_features(job):
    path, blob, cache = job
    out = os.path.join(cache, os.path.basename(path))
    if os.path.exists(out):
        return out
    import torch
    from oracle import oracle_py as op
    oracle = op.load_oracle()
    z = np.load(path)
    tiles = z["tiles"]                                   # [144, 64, 64] of a 1024 x 576 picture (16 x 9 CTUs)
    n = len(tiles)
    cw = 16 if n == 144 else int(round(np.sqrt(n * 16 / 9)))
    ch = n // cw
    luma = tiles.reshape(ch, cw, 64, 64).transpose(0, 2, 1, 3).reshape(ch * 64, cw * 64)
    w = weights.load_any(blob)
    fam = "widths" in w
    model = T.DepthNetQ(widths=tuple(int(v) for v in w["widths"]) if fam else (16, 32, 64), depth=int(w["depth"]) if fam else 1,
                        shifts=[int(w["shift"][b][j]) for b in range(3) for j in range(int(w["depth"]))] if fam else None)
    model.load_arrays(w)
    with torch.no_grad():
        x = torch.from_numpy(tiles.astype(np.float32) - 128.0)[:, None]
        l64, l32, l16 = model.heads(model.trunk(x))
        qb = torch.round(model.qp_bias).numpy()           # [3, 52]
    d64 = (l64[:, 1] - l64[:, 0]).numpy()                 # [N]
    d32 = (l32[:, 1] - l32[:, 0]).numpy()                 # [N, 2, 2]
    d16 = (l16[:, 1] - l16[:, 0]).numpy()                 # [N, 4, 4]
    cost, satd = first_pass_costs(oracle, op, luma)
    np.savez_compressed(out, d64=d64, d32=d32, d16=d16, qp_bias=qb, fp_cost=cost.astype(np.float32), fp_satd=satd.astype(np.float32),
                        **{f"cost_q{qp}": z[f"cost_q{qp}"] for qp in QPS})
    return out


def node_features(f, qi, qp):
    """per level: X [nodes, features], y (split is cheaper), w (|delta J|), and the index arrays to put decisions back into tree_regret's grids"""
    fp, sd = f["fp_cost"][:, :, qi].astype(np.float64), f["fp_satd"].astype(np.float64)
    N = len(fp)
    c64, c32, c16, c8 = fp[:, 0], fp[:, 1:5].reshape(N, 2, 2), fp[:, 5:21].reshape(N, 4, 4), fp[:, 21:85].reshape(N, 8, 8)
    s64, s32, s16, s8 = sd[:, 0], sd[:, 1:5].reshape(N, 2, 2), sd[:, 5:21].reshape(N, 4, 4), sd[:, 21:85].reshape(N, 8, 8)
    sum4 = lambda a: a.reshape(N, a.shape[1] // 2, 2, a.shape[2] // 2, 2).sum(axis=(2, 4))
    eps = 1.0
    feats = {}
    qb = f["qp_bias"]
    for lv, d, cp, cc, sp, sc, area in ((64, f["d64"].reshape(N, 1, 1), c64.reshape(N, 1, 1), sum4(c32), s64.reshape(N, 1, 1), sum4(s32), 4096.0),
                                        (32, f["d32"], c32, sum4(c16), s32, sum4(s16), 1024.0),
                                        (16, f["d16"], c16, sum4(c8), s16, sum4(s8), 256.0)):
        li = {64: 0, 32: 1, 16: 2}[lv]
        dd = (d + qb[li, qp]) / (2.0 ** (15 - li))        # LOSS_SCALE units of the trainer
        r_cost = np.log((cp + eps) / (cc + eps))
        r_satd = np.log((sp + eps) / (sc + eps))
        act = np.log((sp + eps) / area)
        # spread of the children's costs: a node whose four children differ much is an edge / detail node
        ch = {64: c32, 32: c16, 16: c8}[lv].reshape(N, d.shape[1], 2, d.shape[2], 2).transpose(0, 1, 3, 2, 4).reshape(N, d.shape[1], d.shape[2], 4)
        spread = np.log((ch.max(axis=3) + eps) / (ch.min(axis=3) + eps))
        feats[lv] = np.stack([dd, r_cost, r_satd, act, spread], axis=-1).reshape(-1, 5)
    return feats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--labels", default="/tmp/fhevc_labels_c")
    ap.add_argument("--blob", default=os.path.join(ROOT, "fasthevc_amd", "weights", "depthnet_family_d2.fhw"))
    ap.add_argument("--cache", default="/tmp/fhevc_feat")
    ap.add_argument("--train-pictures", type=int, default=300)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--heldout", default="/tmp/fhevc_labels_ho", help="cost labels of the held-out families (make_labels_heldout.py): evaluated, never fitted on")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    os.makedirs(args.cache, exist_ok=True)
    files = sorted(glob.glob(os.path.join(args.labels, "pic_*.npz")))
    va = [f for i, f in enumerate(files) if i % 8 == 0]
    tr_all = [f for i, f in enumerate(files) if i % 8 != 0]
    tr = tr_all[::max(1, len(tr_all) // args.train_pictures)][:args.train_pictures]
    with Pool(args.procs) as pool:
        done = pool.map(picture_features, [(f, args.blob, args.cache) for f in va + tr], chunksize=4)
    va_f, tr_f = done[:len(va)], done[len(va):]
    ho = sorted(glob.glob(os.path.join(args.heldout, "pic_*.npz"))) if args.heldout else []
    if ho:
        os.makedirs(os.path.join(args.cache, "heldout"), exist_ok=True)
        with Pool(args.procs) as pool:
            ho_f = pool.map(picture_features, [(f, args.blob, os.path.join(args.cache, "heldout")) for f in ho], chunksize=2)

    from sklearn.linear_model import LogisticRegression
    from sklearn.ensemble import HistGradientBoostingClassifier

    def gather(paths):
        data = {qp: {lv: {"X": [], "dj": []} for lv in (64, 32, 16)} for qp in QPS}
        grids = {qp: [] for qp in QPS}
        for p in paths:
            f = np.load(p)
            for qi, qp in enumerate(QPS):
                feats = node_features(f, qi, qp)
                g = T.cost_grids(f[f"cost_q{qp}"])
                grids[qp].append(g)
                for lv in (64, 32, 16):
                    data[qp][lv]["X"].append(feats[lv])
                    data[qp][lv]["dj"].append((g[f"ns{lv}"] - g[f"sp{lv}"]).reshape(-1))   # > 0: splitting is cheaper
        for qp in QPS:
            grids[qp] = {k: np.concatenate([g[k] for g in grids[qp]]) for k in grids[qp][0]}
            for lv in (64, 32, 16):
                data[qp][lv] = {k: np.concatenate(v) for k, v in data[qp][lv].items()}
        return data, grids

    tr_d, tr_g = gather(tr_f)
    va_d, va_g = gather(va_f)
    norms = T.cost_norms(tr_g)
    report = {"blob": os.path.basename(args.blob), "train_pictures": len(tr_f), "validation_pictures": len(va_f), "features": ["logit difference + QP prior", "log first-pass cost parent / children", "the same on SATD", "log SATD per sample", "log max / min child cost"]}

    def shapes(lv, n):
        return {64: (n,), 32: (n, 2, 2), 16: (n, 4, 4)}[lv]

    def regret(dec, qp, grids=None):
        g = va_g if grids is None else grids
        n = len(g[qp]["ns64"])
        ch, be = T.tree_regret(dec[64].reshape(shapes(64, n)), dec[32].reshape(shapes(32, n)), dec[16].reshape(shapes(16, n)), g[qp])
        return 100.0 * (ch / be - 1.0)

    models = {}
    for kind in ("logistic: logits only", "logistic: logits + features", "boosted trees: logits + features"):
        cols = [0] if "only" in kind else [0, 1, 2, 3, 4]
        res = {}
        for qp in QPS:
            dec, score = {}, {}
            for li, lv in enumerate((64, 32, 16)):
                X, dj = tr_d[qp][lv]["X"][:, cols], tr_d[qp][lv]["dj"]
                ok = np.isfinite(dj) & (dj != 0)
                wgt = np.minimum(np.abs(dj[ok]) / norms[qp][li], 8.0)
                m = HistGradientBoostingClassifier(max_depth=4, max_iter=120, learning_rate=0.1) if kind.startswith("boosted") else LogisticRegression(C=10.0, max_iter=400)
                m.fit(X[ok], (dj[ok] > 0).astype(int), sample_weight=wgt)
                models[(kind, qp, lv)] = m
                p = m.predict_proba(va_d[qp][lv]["X"][:, cols])[:, 1]
                dec[lv], score[lv] = p > 0.5, p
            res[f"q{qp}"] = round(regret(dec, qp), 3)
        report[kind] = res
    # the soft hook: only decisions surer than a margin are forced, HM's own search decides the rest (taken as the cheaper alternative).  How much regret is
    # left at a given share of forced nodes -- with the logit difference as the confidence (what the hook uses today) and with the combiner's probability?
    curves = {}
    for kind, cols in (("logits only (|logit difference|)", [0]), ("boosted trees: logits + features (|p - 0.5|)", [0, 1, 2, 3, 4])):
        rows = {}
        for share in (0.5, 0.6, 0.7, 0.8, 0.9, 1.0):
            per_qp = {}
            for qp in QPS:
                dec = {}
                for lv in (64, 32, 16):
                    X, dj = va_d[qp][lv]["X"], va_d[qp][lv]["dj"]
                    if len(cols) == 1:
                        conf, pred = np.abs(X[:, 0]), X[:, 0] > 0
                    else:
                        pr = models[("boosted trees: logits + features", qp, lv)].predict_proba(X[:, cols])[:, 1]
                        conf, pred = np.abs(pr - 0.5), pr > 0.5
                    thr = np.quantile(conf, 1.0 - share) if share < 1.0 else -1.0
                    dec[lv] = np.where(conf >= thr, pred, dj > 0)          # undecided nodes: the reference's own choice between the two
                per_qp[f"q{qp}"] = round(regret(dec, qp), 3)
            rows[f"{int(100 * share)} % of the nodes forced"] = per_qp
        curves[kind] = rows
    report["regret left when only the surest decisions are forced"] = curves
    # the held-out families (never fitted on, never in a training label): hard decisions and the soft curves, family by family
    if ho:
        held = {}
        for famname in sorted({os.path.basename(p).split("_")[1] for p in ho_f}):
            fd, fg = gather([p for p in ho_f if os.path.basename(p).split("_")[1] == famname])
            rows = {}
            for kind, cols in (("logits only", [0]), ("boosted trees: logits + features", [0, 1, 2, 3, 4])):
                for share in (0.6, 0.8, 1.0):
                    per_qp = {}
                    for qp in QPS:
                        dec = {}
                        for lv in (64, 32, 16):
                            X, dj = fd[qp][lv]["X"], fd[qp][lv]["dj"]
                            if len(cols) == 1:
                                conf, pred = np.abs(X[:, 0]), X[:, 0] > 0
                                thr = np.quantile(np.abs(va_d[qp][lv]["X"][:, 0]), 1.0 - share) if share < 1.0 else -1.0   # the margin that forces this share of the VALIDATION nodes
                            else:
                                m = models[("boosted trees: logits + features", qp, lv)]
                                pr = m.predict_proba(X[:, cols])[:, 1]
                                conf, pred = np.abs(pr - 0.5), pr > 0.5
                                thr = np.quantile(np.abs(m.predict_proba(va_d[qp][lv]["X"][:, cols])[:, 1] - 0.5), 1.0 - share) if share < 1.0 else -1.0
                            dec[lv] = np.where(conf >= thr, pred, dj > 0)
                            per_qp.setdefault("forced", []).append(float((conf >= thr).mean()))
                        per_qp[f"q{qp}"] = round(regret(dec, qp, fg), 3)
                    per_qp["forced"] = round(float(np.mean(per_qp["forced"])), 3)
                    rows[f"{kind}, margin of {int(100 * share)} % on the validation set"] = per_qp
            held[famname] = rows
        report["held-out families (margins fixed on the validation pictures)"] = held
    base = {}
    for qp in QPS:
        base[f"q{qp}"] = round(regret({lv: va_d[qp][lv]["X"][:, 0] > 0 for lv in (64, 32, 16)}, qp), 3)
    report["the blob's own hard decisions"] = base
    print(json.dumps(report, indent=1))
    if args.json:
        json.dump(report, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""TEST/QUALITY INFRASTRUCTURE: fit the P-picture split rule of fhevc_p_depth_range (include/fasthevc.h) on HM's own P-picture
decisions.  Input: feature files of p_features.py (motion nodes of POC 3 searched in the original POC 2, HM's depths of POC 2
and POC 3 at four QPs, from make_labels_p.py clips).  One logistic regression per split level (64->32, 32->16, 16->8) over the
nine integer features the header specifies; prints the weights in the fixed-point form of FHEVC_P_RULE_WEIGHTS
(fasthevc_amd/csrc/fhevc_internal.h): weights Q10, bias Q18.

usage: python tests/quality/fit_p_rule.py /tmp/pfit/feats.npz [/tmp/pfit/feats2.npz ...]
"""
import sys

import numpy as np
from sklearn.linear_model import LogisticRegression
from sklearn.metrics import roc_auc_score

QPS = (28, 33, 38, 43)


def ilog2_q8(x):
    return np.floor(256 * np.log2(np.maximum(np.asarray(x, np.float64), 1))).astype(np.int64)


def features(d, qp, lvl):
    """-> (X [n, 9] in 1/256 units, y split labels, clip seeds); nodes HM did not reach (label depth < level) are left out"""
    nodes, lab, prev = d[f"nodes_q{qp}"], d[f"label_q{qp}"].reshape(-1, 16, 16), d[f"prev_q{qp}"].reshape(-1, 16, 16)
    cnt, sz, off, coff = 1 << lvl, 16 >> lvl, (0, 1, 5, 21)[lvl], (1, 5, 21)[lvl]
    X, Y, V = [], [], []
    for by in range(cnt):
        for bx in range(cnt):
            node = off + by * cnt + bx
            ch = [coff + (2 * by + j) * 2 * cnt + 2 * bx + i for j in range(2) for i in range(2)]
            J, S, Z = (nodes[k][:, node].astype(np.int64) for k in ("cost_best", "satd_best", "satd_zero"))
            Jc = sum(nodes["cost_best"][:, c].astype(np.int64) for c in ch)
            Sc = sum(nodes["satd_best"][:, c].astype(np.int64) for c in ch)
            mvd = sum(((nodes["mvx"][:, c] != nodes["mvx"][:, node]) | (nodes["mvy"][:, c] != nodes["mvy"][:, node])).astype(np.int64) for c in ch)
            pb = prev[:, by * sz:(by + 1) * sz, bx * sz:(bx + 1) * sz].reshape(len(prev), -1)
            pmax, pmin = pb.max(1).astype(np.int64), pb.min(1).astype(np.int64)
            norm = 512 * (6 - lvl) + (qp * 256) // 6
            f = np.stack([ilog2_q8(S + 1) - norm, ilog2_q8(np.maximum(J - Jc, 0) + 1) - norm, ilog2_q8(Sc + 1) - norm, ilog2_q8(Z + 1) - ilog2_q8(S + 1),
                          (pmax > lvl) * 256, (pmin > lvl) * 256, (pmax > lvl + 1) * 256, mvd * 64, np.full(len(J), qp * 8)], 1)
            dl = lab[:, by * sz, bx * sz]
            ok = (nodes["cost_best"][:, node] != 0xFFFFFFFF) & (dl >= lvl)
            X.append(f[ok]); Y.append((dl > lvl)[ok]); V.append(d["seed"][ok])
    return np.concatenate(X), np.concatenate(Y), np.concatenate(V)


def main():
    sets = [np.load(p) for p in sys.argv[1:]]
    rows = []
    for lvl in range(3):
        parts = [features(d, qp, lvl) for d in sets for qp in QPS]
        X = np.concatenate([p[0] for p in parts]).astype(np.float64) / 256
        y = np.concatenate([p[1] for p in parts])
        seed = np.concatenate([p[2] for p in parts])
        test = seed % 5 == 0
        clf = LogisticRegression(C=1.0, max_iter=2000, class_weight="balanced").fit(X[~test], y[~test])
        print(f"level {lvl}: {len(y)} nodes, split share {y.mean():.3f}, held-out AUC {roc_auc_score(y[test], clf.decision_function(X[test])):.3f}")
        rows.append(list(np.round(clf.coef_[0] * 1024).astype(int)) + [int(round(clf.intercept_[0] * (1 << 18)))])
    print("#define FHEVC_P_RULE_WEIGHTS { " + ", ".join("{ " + ", ".join(str(v) for v in r) + " }" for r in rows) + " }")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Trainer for the integer-valued depth classifier -- replaces the reference's MATLAB
scripts (matlab/dataExtraction/Train...Example.m:75-96, 195-205; labels detectAndClassify32Cu.m:11-62).

The network is trained directly in the fixed-point domain the HIP kernel runs in (HISTORY.md section 4): latent float
weights are rounded to int8 in the forward pass (straight-through gradients), activations are
clamp(floor((acc + b) >> s), 0, 255).  What the trainer evaluates is therefore bit-for-bit what
fasthevc_amd/csrc/k_cnn.hip and oracle/fhevc_oracle.c compute from the exported FHW1 blob.

Labels (per full CTU, per QP) come from the reference's own full-RDO depth maps (tests/quality/make_labels.py):
  s64 = depth(0,0) >= 1;  s32[q] = depth(quadrant origin) >= 2, counted only where s64;  s16[b] = depth(block origin)
  == 3, counted only where its quadrant is split -- the 32-level rule is the reference's isDiv = (Depth != 1) under
  Depth(CU0_0) != 0 (detectAndClassify32Cu.m:11-16, 57-62).

usage: python -m fasthevc_amd.train.train --data /tmp/fhevc_labels --out fasthevc_amd/weights/depthnet_v1.fhw
"""
import argparse
import glob
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as Fn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fasthevc_amd import weights as W  # noqa: E402

QPS = (22, 27, 32, 37)
SHIFTS = (6, 7, 8)
LOSS_SCALE = (2.0 ** -15, 2.0 ** -14, 2.0 ** -13)  # int logits -> loss units, per level (64, 32, 16)


def ste_round(x):
    return x + (torch.round(x) - x).detach()


def ste_floor(x):
    return x + (torch.floor(x) - x).detach()


class DepthNetQ(nn.Module):
    """widths (16, 32, 64), depth 1: the Train...Example.m network (FHW1 blob).  Any other widths / depth: a member of the reference's
    Bayesian-optimisation family (Optimize...Example.m:103-106, 233-259: `depth` convolutions per block; FHW3 blob)."""

    def __init__(self, seed=0, widths=(16, 32, 64), depth=1, shifts=None):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.widths, self.depth = tuple(widths), depth
        self.family = self.widths != (16, 32, 64) or depth != 1
        c3 = self.widths[2]
        self.convs, self.biases, self.shifts = nn.ParameterList(), nn.ParameterList(), []
        ci = 1
        for b in range(3):
            for j in range(depth):
                fan = 9 * ci
                # start at about the scale the base network starts at (21 at fan-in 9, 9 at 144, 12.5 at 288)
                amp = 21.0 if ci == 1 else 110.0 / np.sqrt(fan) * (1.9 if fan >= 288 else 1.0)
                self.convs.append(nn.Parameter(torch.randn(self.widths[b], ci, 3, 3, generator=g) * amp))
                self.biases.append(nn.Parameter(torch.zeros(self.widths[b])))
                self.shifts.append((6 if ci == 1 else 7 if fan <= 160 else 8) if shifts is None else shifts[len(self.shifts)])
                ci = self.widths[b]
        hs = np.sqrt(64.0 / c3)   # FC heads over c3 channels: keep the logits' scale
        self.h64 = nn.Parameter(torch.randn(2, c3, 8, 8, generator=g) * 6.0 * hs)   # [cls][c][y][x] (exported as [cls][y][x][c])
        self.h32 = nn.Parameter(torch.randn(2, c3, 8, 8, generator=g) * 8.0 * hs)
        self.h16 = nn.Parameter(torch.randn(2, c3, 4, 4, generator=g) * 8.0 * hs)
        self.bh64 = nn.Parameter(torch.zeros(2))
        self.bh32 = nn.Parameter(torch.zeros(2))
        self.bh16 = nn.Parameter(torch.zeros(2))
        self.qp_bias = nn.Parameter(torch.zeros(3, 52))

    @staticmethod
    def q8(w):
        return ste_round(torch.clamp(w, -127, 127))

    def trunk(self, x):
        """x: [B,1,64,64] float holding integers -128..127 -> a3 [B,c3,16,16] integers 0..255"""
        k = 0
        for b in range(3):
            for j in range(self.depth):
                z = Fn.conv2d(x, self.q8(self.convs[k]), ste_round(self.biases[k]), padding=1)
                if j == self.depth - 1 and b < 2:
                    z = Fn.max_pool2d(z, 2)
                x = torch.clamp(ste_floor(z / float(1 << self.shifts[k])), 0, 255)
                k += 1
        return x

    def heads(self, a3):
        """integer logits without the QP prior: l64 [B,2], l32 [B,2,2,2], l16 [B,2,4,4]"""
        pooled = 4.0 * Fn.avg_pool2d(a3, 2)
        l64 = Fn.conv2d(pooled, self.q8(self.h64), ste_round(self.bh64)).flatten(1)
        l32 = Fn.conv2d(a3, self.q8(self.h32), ste_round(self.bh32), stride=8)
        l16 = Fn.conv2d(a3, self.q8(self.h16), ste_round(self.bh16), stride=4)
        return l64, l32, l16

    def load_arrays(self, d):
        """start from an exported blob (weights.load / weights.load_any arrays) of the same shape"""
        with torch.no_grad():
            t = lambda a: torch.from_numpy(np.asarray(a).astype(np.float32))
            if not self.family:
                for k, (wn, bn, ci) in enumerate((("w1", "b1", 1), ("w2", "b2", 16), ("w3", "b3", 32))):
                    self.convs[k].copy_(t(d[wn]).reshape(self.widths[k], ci, 3, 3)); self.biases[k].copy_(t(d[bn]))
                assert tuple(int(v) for v in d["shift"]) == tuple(self.shifts)
            else:
                k = 0
                for b in range(3):
                    for j in range(self.depth):
                        self.convs[k].copy_(t(d[f"w{b}{j}"]).reshape(self.convs[k].shape)); self.biases[k].copy_(t(d[f"b{b}{j}"]))
                        assert int(np.asarray(d["shift"])[b, j]) == self.shifts[k]
                        k += 1
            c3 = self.widths[2]
            self.h64.copy_(t(d["wh64"]).reshape(2, 8, 8, c3).permute(0, 3, 1, 2)); self.bh64.copy_(t(d["bh64"]))
            self.h32.copy_(t(d["wh32"]).reshape(2, 8, 8, c3).permute(0, 3, 1, 2)); self.bh32.copy_(t(d["bh32"]))
            self.h16.copy_(t(d["wh16"]).reshape(2, 4, 4, c3).permute(0, 3, 1, 2)); self.bh16.copy_(t(d["bh16"]))
            self.qp_bias.copy_(t(d["qp_bias"]).reshape(3, 52))

    def export(self):
        r = lambda t: torch.round(t.detach()).to(torch.int64).numpy()
        q = lambda t: np.clip(r(t), -127, 127).astype(np.int8)
        heads = {
            "wh64": q(self.h64).transpose(0, 2, 3, 1).copy(), "bh64": r(self.bh64).astype(np.int32),
            "wh32": q(self.h32).transpose(0, 2, 3, 1).copy(), "bh32": r(self.bh32).astype(np.int32),
            "wh16": q(self.h16).transpose(0, 2, 3, 1).copy(), "bh16": r(self.bh16).astype(np.int32),
            "qp_bias": r(self.qp_bias).astype(np.int32),
        }
        bias = lambda k: np.clip(r(self.biases[k]), -W.BIAS_LIMIT, W.BIAS_LIMIT).astype(np.int32)
        if not self.family:
            out = {"shift": np.array(self.shifts, np.int32), "w1": q(self.convs[0]).reshape(16, 3, 3), "b1": bias(0),
                   "w2": q(self.convs[1]), "b2": bias(1), "w3": q(self.convs[2]), "b3": bias(2)}
            out.update(heads)
            return out
        out = {"widths": np.array(self.widths, np.int32), "depth": self.depth, "shift": np.zeros((3, 3), np.int32)}
        k = 0
        for b in range(3):
            for j in range(self.depth):
                out[f"w{b}{j}"], out[f"b{b}{j}"] = q(self.convs[k]), bias(k)
                out["shift"][b, j] = self.shifts[k]
                k += 1
        out.update(heads)
        return out


def labels_from_depth(depth):
    """depth [N,16,16] uint8 -> s64 [N], s32 [N,2,2], s16 [N,4,4] (int64 0/1) and validity masks m32, m16"""
    d = torch.from_numpy(depth.astype(np.int64))
    s64 = (d[:, 0, 0] >= 1).long()
    s32 = (d[:, ::8, ::8] >= 2).long()
    s16 = (d[:, ::4, ::4] == 3).long()
    m32 = s64[:, None, None].expand_as(s32).bool()
    m16 = s32.repeat_interleave(2, 1).repeat_interleave(2, 2).bool() & s64[:, None, None].bool()
    return s64, s32, s16, m32, m16


def load_data(path, val_every=8, costs=False):
    """-> (tiles, depth maps per QP[, cost grids per QP]) of the training and of the validation pictures.  costs: the label files carry
    cost_q<qp> [N, 21, 2] (make_labels.py --costs): turned into delta grids by cost_grids()."""
    files = sorted(glob.glob(os.path.join(path, "pic_*.npz")))
    tr, va = [], []
    for i, f in enumerate(files):
        z = np.load(f)
        item = (z["tiles"], {qp: z[f"depth_q{qp}"] for qp in QPS}, {qp: z[f"cost_q{qp}"] for qp in QPS} if costs else None)
        (va if i % val_every == 0 else tr).append(item)

    def cat(items):
        tiles = np.concatenate([t for t, _, _ in items])
        depth = {qp: np.concatenate([d[qp] for _, d, _ in items]) for qp in QPS}
        if not costs:
            return tiles, depth
        return tiles, depth, {qp: cost_grids(np.concatenate([c[qp] for _, _, c in items])) for qp in QPS}
    return cat(tr), cat(va), len(files)


# ---- cost-sensitive training (round 3): what a wrong split decision COSTS, from the reference's own RD costs ------------------------------
# z-order -> raster of the quadrants / 16x16 blocks of a CTU (the recorder numbers its nodes as HM's z-scan does)
_Q_YX = [(q >> 1, q & 1) for q in range(4)]
_B_YX = [(2 * ((b >> 2) >> 1) + ((b & 3) >> 1), 2 * ((b >> 2) & 1) + (b & 1)) for b in range(16)]


def cost_grids(cost):
    """cost [N, 21, 2] (no-split, split; NaN = not evaluated) -> dict of float32 arrays: ns64 / sp64 [N], ns32 / sp32 [N, 2, 2],
    ns16 / sp16 [N, 4, 4] in raster order (NaN -> both 0: no preference, no weight)."""
    c = np.nan_to_num(cost.astype(np.float32), nan=0.0, posinf=0.0)
    bad = ~np.isfinite(cost).all(axis=2)
    c[bad] = 0.0
    out = {"ns64": c[:, 0, 0], "sp64": c[:, 0, 1]}
    g32 = np.zeros((len(c), 2, 2, 2), np.float32)
    g16 = np.zeros((len(c), 4, 4, 2), np.float32)
    for q, (y, x) in enumerate(_Q_YX):
        g32[:, y, x] = c[:, 1 + q]
    for b, (y, x) in enumerate(_B_YX):
        g16[:, y, x] = c[:, 5 + b]
    out["ns32"], out["sp32"], out["ns16"], out["sp16"] = g32[..., 0], g32[..., 1], g16[..., 0], g16[..., 1]
    return out


def flip_costs(g):
    return {k: (v if v.ndim == 1 else v[:, :, ::-1]) for k, v in g.items()}


def cost_norms(tr_c):
    """mean |J_no_split - J_split| per QP and level over the training set: the unit of the loss weights"""
    return {qp: tuple(float(np.abs(g[f"ns{l}"] - g[f"sp{l}"]).mean()) + 1e-9 for l in (64, 32, 16)) for qp, g in tr_c.items()}


def tree_regret(p64, p32, p16, g):
    """RD cost of the quad-tree the decisions p64 [N], p32 [N,2,2], p16 [N,4,4] (bool: split) pick, against the cheapest tree, from the
    recorded node costs g (cost_grids): J(16 node) = split ? sp16 : ns16; J(32 node) = split ? sum of its four 16 nodes : ns32;
    J(CTU) = split ? sum of its quadrants : ns64 (split-flag bits and context effects of the neighbours ignored: the four-way sums
    stand in for the recorded split costs).  -> (sum of chosen costs, sum of cheapest costs) over the CTUs that have all 21 nodes."""
    def tree(s64, s32, s16):
        j16 = np.where(s16, g["sp16"], g["ns16"])
        j32 = np.where(s32, j16.reshape(-1, 2, 2, 2, 2).sum(axis=(2, 4)), g["ns32"])
        return np.where(s64, j32.sum(axis=(1, 2)), g["ns64"])
    ok = (g["ns64"] > 0) & (g["ns32"] > 0).all(axis=(1, 2)) & (g["ns16"] > 0).all(axis=(1, 2))
    # the cheapest tree bottom-up under the same approximation
    o16 = g["sp16"] < g["ns16"]
    j16 = np.where(o16, g["sp16"], g["ns16"]).reshape(-1, 2, 2, 2, 2).sum(axis=(2, 4))
    o32 = j16 < g["ns32"]
    o64 = np.where(o32, j16, g["ns32"]).sum(axis=(1, 2)) < g["ns64"]
    chosen, best = tree(p64, p32, p16), tree(o64, o32, o16)
    return float(chosen[ok].sum()), float(best[ok].sum())


def batch_loss_costs(model, x, costs, idx, norms, stats=None, wcap=8.0):
    """cost-sensitive form: EVERY node the reference evaluated is a sample (not only those on its best path: a classifier that splits
    wrongly above must still decide sensibly below), target = the cheaper alternative, weight = what the other one costs more, in units
    of the level's mean difference (capped): near-ties, where either decision is fine, stop driving the loss."""
    a3 = model.trunk(x)
    l64, l32, l16 = model.heads(a3)
    total = 0.0
    for qp in QPS:
        g = costs[qp]
        qb = ste_round(model.qp_bias[:, qp])
        z = ((l64 + torch.stack([torch.zeros(()), qb[0]])[None, :]) * LOSS_SCALE[0],
             ((l32 + torch.stack([torch.zeros(()), qb[1]])[None, :, None, None]) * LOSS_SCALE[1]).permute(0, 2, 3, 1),
             ((l16 + torch.stack([torch.zeros(()), qb[2]])[None, :, None, None]) * LOSS_SCALE[2]).permute(0, 2, 3, 1))
        pred = []
        for li, lv in enumerate((64, 32, 16)):
            d = torch.from_numpy(g[f"ns{lv}"][idx] - g[f"sp{lv}"][idx])       # > 0: splitting is cheaper
            y = (d > 0).long()
            w = torch.clamp(d.abs() / norms[qp][li], max=wcap)
            ce = Fn.cross_entropy(z[li].reshape(-1, 2), y.reshape(-1), reduction="none")
            total = total + (ce * w.reshape(-1)).mean()
            pred.append((z[li][..., 1] > z[li][..., 0]).detach().numpy())
        if stats is not None:
            gi = {k: v[idx] for k, v in g.items()}
            ch, be = tree_regret(pred[0], pred[1], pred[2], gi)
            st = stats.setdefault(qp, np.zeros(2))
            st += np.array([ch, be])
    return total / len(QPS)


def batch_loss(model, x, depths, idx, stats=None):
    a3 = model.trunk(x)
    l64, l32, l16 = model.heads(a3)
    total = 0.0
    for qp in QPS:
        s64, s32, s16, m32, m16 = labels_from_depth(depths[qp][idx])
        qb = ste_round(model.qp_bias[:, qp])
        z64 = (l64 + torch.stack([torch.zeros(()), qb[0]])[None, :]) * LOSS_SCALE[0]
        z32 = (l32 + torch.stack([torch.zeros(()), qb[1]])[None, :, None, None]) * LOSS_SCALE[1]
        z16 = (l16 + torch.stack([torch.zeros(()), qb[2]])[None, :, None, None]) * LOSS_SCALE[2]
        total = total + Fn.cross_entropy(z64, s64)
        if m32.any():
            total = total + Fn.cross_entropy(z32.permute(0, 2, 3, 1)[m32], s32[m32])
        if m16.any():
            total = total + Fn.cross_entropy(z16.permute(0, 2, 3, 1)[m16], s16[m16])
        if stats is not None:
            with torch.no_grad():
                p64 = (z64[:, 1] > z64[:, 0]).long()
                p32 = (z32[:, 1] > z32[:, 0]).long()
                p16 = (z16[:, 1] > z16[:, 0]).long()
                st = stats.setdefault(qp, np.zeros(6))
                st += np.array([(p64 == s64).sum().item(), s64.numel(), (p32 == s32)[m32].sum().item(), int(m32.sum()),
                                (p16 == s16)[m16].sum().item(), int(m16.sum())], np.float64)
    return total / len(QPS)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default="/tmp/fhevc_labels")
    ap.add_argument("--out", default=os.path.join(ROOT, "fasthevc_amd", "weights", "depthnet_v1.fhw"))
    ap.add_argument("--epochs", type=int, default=12)
    ap.add_argument("--batch", type=int, default=96)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--qps", default="", help="comma-separated slice QPs of the label files (default 22,27,32,37; P-picture labels: 28,33,38,43)")
    ap.add_argument("--depth", type=int, default=0, help="1..3: train the member of the reference's Bayesian-optimisation family with that many convolutions "
                    "per block (widths round(32 / sqrt(depth)), x2, x4: Optimize...Example.m:233-259) and write an FHW3 blob; 0: the 16 / 32 / 64 network (FHW1)")
    ap.add_argument("--widths", default="", help="override the family member's widths, e.g. 32,64,128")
    ap.add_argument("--costs", action="store_true", help="cost-sensitive training on label files with the reference's split / no-split RD costs "
                    "(make_labels.py --costs): every evaluated node is a sample, weighted by what the wrong decision costs")
    ap.add_argument("--init", default="", help="start from this blob (same shape) instead of a random initialisation")
    ap.add_argument("--lr-scale", type=float, default=1.0)
    ap.add_argument("--eval-only", action="store_true", help="with --init and --costs: print the blob's tree regret on the validation pictures and exit")
    args = ap.parse_args()
    if args.qps:
        global QPS
        QPS = tuple(int(v) for v in args.qps.split(","))
    torch.set_num_threads(args.threads)
    torch.manual_seed(args.seed)
    tr, va, nfiles = load_data(args.data, costs=args.costs)
    (tr_t, tr_d), (va_t, va_d) = tr[:2], va[:2]
    tr_c, va_c = (tr[2], va[2]) if args.costs else (None, None)
    norms = cost_norms(tr_c) if args.costs else None
    print(f"{nfiles} pictures: {len(tr_t)} training CTUs, {len(va_t)} validation CTUs, QPs {QPS}", flush=True)
    if args.costs:
        print("mean |J_no_split - J_split| per level (64, 32, 16): " + "  ".join(f"q{qp}: " + " / ".join(f"{v:.0f}" for v in n) for qp, n in norms.items()), flush=True)
    if args.depth or args.widths:
        depth = args.depth or 1
        widths = tuple(int(v) for v in args.widths.split(",")) if args.widths else W.family_widths(depth)
        model = DepthNetQ(args.seed, widths, depth)
        print(f"family member: widths {widths}, {depth} convolution(s) per block, shifts {model.shifts}", flush=True)
    else:
        model = DepthNetQ(args.seed)
    if args.init:
        model.load_arrays(W.load_any(args.init))
        print("initialised from", args.init, flush=True)
    conv_w = list(model.convs)
    conv_b = list(model.biases)
    head_w = [model.h64, model.h32, model.h16]
    head_b = [model.bh64, model.bh32, model.bh16, model.qp_bias]
    base = [v * args.lr_scale for v in (0.4, 15.0, 0.4, 400.0)]
    opt = torch.optim.Adam([{"params": conv_w, "lr": base[0]}, {"params": conv_b, "lr": base[1]},
                            {"params": head_w, "lr": base[2]}, {"params": head_b, "lr": base[3]}], betas=(0.9, 0.99))
    steps_per_epoch = len(tr_t) // args.batch
    total_steps = steps_per_epoch * args.epochs
    xt = torch.from_numpy(tr_t.astype(np.float32) - 128.0)[:, None]
    xv = torch.from_numpy(va_t.astype(np.float32) - 128.0)[:, None]
    step, t0 = 0, time.time()

    def validate():
        model.eval()
        stats, vl = {}, 0.0
        with torch.no_grad():
            for b in range(0, len(va_t), 256):
                idx = np.arange(b, min(b + 256, len(va_t)))
                if args.costs:
                    vl += batch_loss_costs(model, xv[idx], va_c, idx, norms, stats).item() * len(idx)
                else:
                    vl += batch_loss(model, xv[idx], va_d, idx, stats).item() * len(idx)
        if args.costs:  # RD cost of the predicted quad-trees over the cheapest ones: the offline stand-in for the BD-rate loss of hard decisions
            msg = "tree regret " + " ".join(f"q{qp}: {100.0 * (s[0] / s[1] - 1.0):.3f} %" for qp, s in stats.items())
        else:
            msg = "acc " + " ".join(f"q{qp}: 64 {s[0] / max(s[1], 1):.3f} 32 {s[2] / max(s[3], 1):.3f} 16 {s[4] / max(s[5], 1):.3f}" for qp, s in stats.items())
        return vl / len(va_t), msg

    if args.eval_only:
        print("validation: loss %.4f %s" % validate(), flush=True)
        return
    for ep in range(args.epochs):
        perm = torch.randperm(len(tr_t))
        model.train()
        run = 0.0
        for b in range(steps_per_epoch):
            idx = perm[b * args.batch:(b + 1) * args.batch]
            x = xt[idx]
            flip = bool(torch.rand(()) < 0.5)  # horizontal flip keeps the block grid: labels flip with it
            if flip:
                x = torch.flip(x, dims=[3])
                depths = {qp: tr_d[qp][:, :, ::-1] for qp in QPS}
            else:
                depths = tr_d
            f = 0.5 * (1 + np.cos(np.pi * step / total_steps))
            for gidx, grp in enumerate(opt.param_groups):
                grp["lr"] = base[gidx] * (0.03 + 0.97 * f)
            if args.costs:
                loss = batch_loss_costs(model, x, {qp: flip_costs(g) for qp, g in tr_c.items()} if flip else tr_c, idx.numpy(), norms)
            else:
                loss = batch_loss(model, x, depths, idx.numpy())
            opt.zero_grad()
            loss.backward()
            opt.step()
            run += loss.item()
            step += 1
            if b % 100 == 99:
                print(f"  ep {ep} step {b + 1}/{steps_per_epoch} loss {run / 100:.4f} ({time.time() - t0:.0f} s)", flush=True)
                run = 0.0
        vl, msg = validate()
        print(f"epoch {ep}: val loss {vl:.4f} {msg} ({time.time() - t0:.0f} s)", flush=True)
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        if model.family:
            with open(args.out, "wb") as fo:
                fo.write(W.pack_family(model.export()))
        else:
            W.save(args.out, model.export())
    print("saved", args.out)


if __name__ == "__main__":
    main()

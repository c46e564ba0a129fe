#!/usr/bin/env python3
"""TEST/QUALITY INFRASTRUCTURE (uses oracle/_ref/libhmref_p.so).  Config 4 (encoder_lowdelay_P_main.cfg geometry): a 2-frame
1080p pan clip, POC 0 as an I slice and POC 1 as a P slice through the reference's own compressSlice with HM-16.14's
inter checks restored (hm_patch/restore_inter.py; the reference as shipped aborts on P slices).

"Inter-CU depth reuse" (BASELINE.json config 4): the depth search of POC 1 is restricted to a window around the
co-located depth of POC 0 through the soft hook (depth_min / depth_max per 4x4 unit).  Reports the BD-rate of POC 1 and
the time in compressSlice against the unrestricted search, over QP {22,27,32,37} (P picture at QP + 6 as
TEncSlice::initEncSlice derives it from the cfg's GOP entry).  Decision stage only: no in-loop filters, TMVP off.

usage: python tests/quality/eval_p.py [--size 1920x1080] [--json out.json]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fasthevc_amd import frames  # noqa: E402
from eval_rd import bd_rate  # noqa: E402

P_SO = os.path.join(ROOT, "oracle", "_ref", "libhmref_p.so")


def load_p():
    from oracle import oracle_py as op
    libdl = C.CDLL(None)
    libdl.dlopen.restype = C.c_void_p
    libdl.dlopen.argtypes = [C.c_char_p, C.c_int]
    h = libdl.dlopen(P_SO.encode(), os.RTLD_LAZY | os.RTLD_LOCAL)
    if not h:
        raise OSError("cannot dlopen " + P_SO + " (make -C oracle pvar)")
    lib = op.bind_rdo(C.CDLL(P_SO, handle=h))
    lib.href_rdo_encode_next_p.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_void_p] * 4
    return lib


SPEED = 3  # px / frame of the clip's two overlaid motions (--speed; 3 = the pinned clip)
CLIP = "overlaid"   # --clip global: ONE motion -- a hetero picture panned by (SPEED, SPEED / 2) samples per picture with a little sensor noise,
                    # the case "inter-CU depth reuse through the motion" is made for (overlaid: two transparent motions in opposite directions,
                    # which no single vector per CU compensates)


def pan_clip(W, H, nframes=2, seed=1234):
    if CLIP == "global":
        rng = np.random.default_rng(seed)
        px, py = 8 + nframes * abs(SPEED), 8 + nframes * (abs(SPEED) // 2)
        big = frames.hetero_luma(W + 2 * px, H + 2 * py, seed=seed)
        out = []
        for f in range(nframes):
            x0, y0 = px + SPEED * f, py + (SPEED // 2) * f
            y = big[y0:y0 + H, x0:x0 + W].astype(np.float64) + rng.normal(0, 0.7, size=(H, W)) + 0.4 * f
            out.append(np.clip(np.rint(y), 0, 255).astype(np.uint8))
        return out
    return frames.pan_clip(W, H, nframes, seed, v_structure=SPEED, v_noise=SPEED)


def cnn_ranges(oracle, ws, y, qp, margin_split, margin_stop):
    """depth range of one picture from a weight blob through the CPU oracle (the GPU path is bit-exact with it)"""
    from oracle import oracle_py as op
    H, W = y.shape
    n = ((W + 63) // 64) * ((H + 63) // 64)
    cw = (W + 63) // 64
    buf, org, stride = frames.to_pel_plane(y, 8)
    pred = np.zeros(n * 256, np.uint8)
    logits = np.zeros(n * 42, np.int32)
    oracle.fho_predict_frame(ws, op.ptr(buf.reshape(-1), org), stride, W, H, 8, qp, pred, C.c_void_p(logits.ctypes.data))
    dmin, dmax = np.zeros((n, 256), np.uint8), np.zeros((n, 256), np.uint8)
    for c in range(n):
        vw, vh = min(64, W - (c % cw) * 64), min(64, H - (c // cw) * 64)
        oracle.fho_depth_range_from_logits(np.ascontiguousarray(logits[c * 42:(c + 1) * 42]), vw, vh, margin_split, margin_stop, dmin[c], dmax[c])
    return dmin, dmax


def mc_depth(oracle, nodes, prev_depth, W, H):
    """the previous picture's depths seen through the motion nodes (oracle twin of fhevc_p_motion_compensated_depth)"""
    n = nodes.shape[0]
    prev = np.ascontiguousarray(prev_depth, np.uint8).reshape(n, 256)
    out = np.zeros((n, 256), np.uint8)
    oracle.fho_p_motion_compensated_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    for c in range(n):
        oracle.fho_p_motion_compensated_depth(nodes[c].ctypes.data, prev.ctypes.data, W, H, c, out[c].ctypes.data)
    return out


def mc_node_depth(oracle, nodes, prev, W, H):
    """the reference picture's depths asked per CU node of the current grid (oracle twin of fhevc_p_node_depth; the round-4 numbers under profiles/ were
    produced by a Python prototype of the same walk, checked equal to the library's function on random nodes)"""
    n = nodes.shape[0]
    prev = np.ascontiguousarray(prev, np.uint8).reshape(n, 256)
    out = np.zeros((n, 256), np.uint8)
    oracle.fho_p_node_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    for c in range(n):
        oracle.fho_p_node_depth(nodes[c].ctypes.data, prev.ctypes.data, W, H, c, out[c].ctypes.data)
    return out


def motion_ranges(oracle, rule, cur, ref, prev_depth, qp, search_range=4, dist=0, mc=False, window_only=None, hybrid=None):
    """depth range of a P picture from the source-only motion search (oracle twin of k_motion.hip) and the previous picture's
    depths, through the oracle twin of fhevc_p_depth_range (the GPU path is bit-exact with both)"""
    import p_features
    H, W = cur.shape
    cw = (W + 63) // 64
    nodes = p_features.motion_frame(oracle, cur, ref, qp, search_range, dist=dist)
    n = nodes.shape[0]
    dmin, dmax = np.zeros((n, 256), np.uint8), np.zeros((n, 256), np.uint8)
    prev = np.ascontiguousarray(prev_depth, np.uint8).reshape(n, 256)
    if mc == "node":  # ... at the displaced centre of each CU node, snapped to the current picture's CU grid
        prev = mc_node_depth(oracle, nodes, prev, W, H)
    elif mc:  # "inter-CU depth reuse" through the motion: the depths at the displaced position instead of the co-located ones
        prev = mc_depth(oracle, nodes, prev, W, H)
    if window_only is not None:  # no rule: the (motion-compensated) depths +- a window
        if len(window_only) > 2:  # block form: [min, max] of the depths inside each 16x16 block +- the window (a displaced map is not aligned to the CU grid)
            b = prev.reshape(n, 4, 4, 4, 4).astype(int)
            lo = np.broadcast_to(b.min(axis=(2, 4), keepdims=True), b.shape).reshape(n, 256)
            hi = np.broadcast_to(b.max(axis=(2, 4), keepdims=True), b.shape).reshape(n, 256)
            return np.clip(lo - window_only[0], 0, 3).astype(np.uint8), np.clip(hi + window_only[1], 0, 3).astype(np.uint8)
        return np.clip(prev.astype(int) - window_only[0], 0, 3).astype(np.uint8), np.clip(prev.astype(int) + window_only[1], 0, 3).astype(np.uint8)
    for c in range(n):
        vw, vh = min(64, W - (c % cw) * 64), min(64, H - (c // cw) * 64)
        oracle.fho_p_depth_range(nodes[c].ctypes.data, prev[c].ctypes.data, vw, vh, qp, C.byref(rule), dmin[c].ctypes.data, dmax[c].ctypes.data)
    if hybrid is not None:  # round-4 prototype: [p - lo, p] everywhere, one level DEEPER than the reference depth p only where the rule (run without its window,
        p = prev.astype(int)   # t_stop = -t_deeper) still allows the split beyond p
        hi = np.where(dmax.astype(int) > p, p + 1, p)
        return np.clip(p - hybrid, 0, 3).astype(np.uint8), np.clip(hi, 0, 3).astype(np.uint8)
    return dmin, dmax


def encode_seq(lib, ys, qp, window=None, cnn=None, motion=None):
    """I P P ...: POC 0 at qp, the P pictures at qp + 6.  window = (levels below, levels above) the co-located depth of
    the PREVIOUS picture (None = unrestricted search; the first P picture is unrestricted when its reference is the I
    picture: intra depths say little about inter depths).  -> per picture (depth [n,256], stats)."""
    from oracle import oracle_py as op
    H, W = ys[0].shape
    n = ((W + 63) // 64) * ((H + 63) // 64)
    u = np.full((H // 2, W // 2), 128, np.int16)
    out = []
    buf, org, stride = frames.to_pel_plane(ys[0], 8)
    d, st = op.rdo_encode(lib, buf, org, stride, W, H, 8, qp, chroma=(u, u))
    out.append((d, {"bits": st["coded_bits"], "psnr_y": st["psnr_y"], "seconds": st["seconds"], "skip_share": 0.0}))
    for f in range(1, len(ys)):
        buf, org, stride = frames.to_pel_plane(ys[f], 8)
        fmin = fmax = None
        if cnn is not None and f >= 2:  # classifier for inter pictures: (oracle, weights, margin_split, margin_stop)
            fmin, fmax = cnn_ranges(cnn[0], cnn[1], ys[f], qp + 6, cnn[2], cnn[3])
            fmin, fmax = np.ascontiguousarray(fmin), np.ascontiguousarray(fmax)
        elif motion is not None and f >= 2:  # (oracle, rule): motion features + the previous P picture's depths
            fmin, fmax = motion_ranges(motion[0], motion[1], ys[f], ys[f - 1], out[-1][0], qp + 6, **(motion[2] if len(motion) > 2 else {}))
            fmin, fmax = np.ascontiguousarray(fmin), np.ascontiguousarray(fmax)
        elif window is not None and f >= 2:
            prev = out[-1][0].astype(int)
            fmin = np.ascontiguousarray(np.clip(prev - window[0], 0, 3).astype(np.uint8))
            fmax = np.ascontiguousarray(np.clip(prev + window[1], 0, 3).astype(np.uint8))
        d1 = np.zeros(n * 256, np.uint8)
        s1 = np.zeros(8)
        rc = lib.href_rdo_encode_next_p(buf.reshape(-1).ctypes.data + 2 * org, u.ctypes.data, u.ctypes.data, stride, W, H, 8, qp + 6, f,
                                        None if fmin is None else fmin.ctypes.data, None if fmax is None else fmax.ctypes.data,
                                        d1.ctypes.data, s1.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"href_rdo_encode_next_p failed: {rc}")
        out.append((d1.reshape(n, 256), {"bits": float(s1[6]), "psnr_y": float(10 * np.log10(255.0 * 255.0 / (s1[4] / (W * H)))),
                                        "seconds": float(s1[3]), "skip_share": float(s1[7])}))
    return out


def _run_qp(job):
    """one slice QP: the anchor (unrestricted search) and every variant; runs in its own process (HM is single-threaded)"""
    global SPEED, CLIP
    qp, variants, (W, H), nframes, SPEED, CLIP = job
    from oracle import oracle_py as op
    from fasthevc_amd import capi
    lib, oracle = load_p(), op.load_oracle()
    ys = pan_clip(W, H, nframes)
    tail = lambda seq: (sum(s["bits"] for _, s in seq[2:]), float(np.mean([s["psnr_y"] for _, s in seq[2:]])), sum(s["seconds"] for _, s in seq[2:]))
    anchor = encode_seq(lib, ys, qp)
    out = {"anchor": tail(anchor), "pictures": [dict(s, depth_hist=np.bincount(d.reshape(-1), minlength=4).tolist()) for d, s in anchor]}
    for name, v in variants.items():
        if v["kind"] == "window":
            seq = encode_seq(lib, ys, qp, window=tuple(v["window"]))
        else:  # the shipped rule (fhevc_p_rule_default) with optional overrides of its thresholds / window
            rule = op.PRule.from_buffer_copy(bytes(capi.p_rule_default_wide() if v.get("wide") else capi.p_rule_default()))
            for l in range(3):
                if "t_split" in v:
                    rule.t_split[l] = int(v["t_split"][l] * (1 << 18))
                if "t_stop" in v:
                    rule.t_stop[l] = int(v["t_stop"][l] * (1 << 18))
            rule.window = v.get("window", rule.window)
            if "w" in v:  # a refitted rule: weights [3][10] in the fixed-point form of FHEVC_P_RULE_WEIGHTS
                for l in range(3):
                    for i in range(10):
                        rule.w[l][i] = int(v["w"][l][i])
            opts = {"search_range": v.get("search_range", 4), "dist": 1 if v.get("dist") == "sad" else 0, "mc": v.get("mc", False),
                    "window_only": tuple(v["mc_window"]) if "mc_window" in v else None, "hybrid": v.get("hybrid")}
            seq = encode_seq(lib, ys, qp, motion=(oracle, rule, opts))
        out[name] = tail(seq)
        out[name + ":agreement"] = [float((seq[f][0] == anchor[f][0]).mean()) for f in range(2, nframes)]
    return qp, out


DEFAULT_VARIANTS = {
    "same_depth": {"kind": "window", "window": [0, 0]},
    "window_pm1": {"kind": "window", "window": [1, 1]},
    "motion_rule": {"kind": "rule"},   # what TEncFastDepth runs with FHEVC_P_MODE=motion: fhevc_p_rule_default()
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--json", default=None)
    ap.add_argument("--variants", default=None, help="JSON: {name: {kind: window|rule, window: ..., t_split: [3], t_stop: [3]}}")
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--clip", default="overlaid", choices=("overlaid", "global"), help="overlaid: the pinned clip's two opposite motions; global: one pan of a hetero picture")
    ap.add_argument("--speed", type=int, default=3, help="px / frame of the clip's motions (beyond the search range: what the rule does when the source-only search cannot reach the true motion)")
    args = ap.parse_args()
    from multiprocessing import Pool
    W, H = (int(v) for v in args.size.split("x"))
    variants = json.loads(args.variants) if args.variants else DEFAULT_VARIANTS
    qps = (22, 27, 32, 37)
    with Pool(args.workers) as pool:
        res = dict(pool.map(_run_qp, [(qp, variants, (W, H), args.frames, args.speed, args.clip) for qp in qps]))
    ra, pa = [res[q]["anchor"][0] for q in qps], [res[q]["anchor"][1] for q in qps]
    ta = sum(res[q]["anchor"][2] for q in qps)
    report = {"clip": f"{W}x{H} {'pan clip (frames.pan_clip: two overlaid opposite motions' if args.clip == 'overlaid' else 'hetero picture under ONE global pan ('}, {args.speed} px / frame), {args.frames} frames I P P ..., P pictures at QP + 6; restricted pictures: POC >= 2 "
                      "(their reference picture is a P picture); decision stage: bits counted by encodeCtu, luma PSNR before the in-loop filters",
              "qp": list(qps), "anchor": [res[q]["anchor"] for q in qps], "pictures": [res[q]["pictures"] for q in qps], "variants": {}, "summary": {}}
    for name, v in variants.items():
        pts = [res[q][name] for q in qps]
        bd = bd_rate(ra, pa, [p[0] for p in pts], [p[1] for p in pts])
        tr = ta / sum(p[2] for p in pts)
        report["variants"][name] = {"definition": v, "points": pts, "bd_rate_percent": bd, "time_ratio": tr,
                                    "depth_agreement_with_full_rdo": [float(np.mean(res[q][name + ":agreement"])) for q in qps]}
        report["summary"][name] = f"BD-rate {bd:+.2f} % at {tr:.2f}x less time in compressSlice"
        print(f"{name:24s} BD-rate of the restricted P pictures {bd:+6.2f} %   decision time {tr:5.2f}x faster", flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()

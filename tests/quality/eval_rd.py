#!/usr/bin/env python3
"""TEST/QUALITY INFRASTRUCTURE (uses oracle/ and oracle/_ref).  Quality guard: BD-rate of depth-map-driven encoding against the reference's full RDO.

For the two pinned 1080p pictures (or crops of them) and QP {22,27,32,37}:
  anchor   = the reference's full-RDO decision path (oracle/_ref/libhmref.so, oracle/ref_rdo_harness.cpp)
  test     = the same path with the hm_patch hook (libhmref_hook.so) driven by a depth map: the classifier's (the CPU
             oracle evaluates the FHW1 blob; the HIP kernel is bit-exact with it), or a constant depth (trivial floors)
Rate = bits counted by TEncCu::encodeCtu over the picture (true CABAC state), distortion = luma PSNR of the
reconstruction before the in-loop filters.  The full encoder (NAL/SEI, deblocking, SAO) cannot be built here (OpenCV), so
this is the decision-stage BD-rate, not a bitstream BD-rate -- stated next to every number.

usage: python tests/quality/eval_rd.py --weights fasthevc_amd/weights/depthnet_v1.fhw [--crop 1024x576]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/quality -> repo root
sys.path.insert(0, ROOT)
from fasthevc_amd import frames, weights  # noqa: E402

QPS = (22, 27, 32, 37)


def ood_luma(width, height, seed=777):
    """Held-out family (no generator of the training labels produces anything like it): a zone plate (radial chirp),
    checkerboards whose scale changes per band, and a smooth ramp."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    r2 = (xx - width * 0.35) ** 2 + (yy - height * 0.5) ** 2
    y = 128.0 + 70.0 * np.cos(r2 / (2.0 * 900.0)) * np.exp(-r2 / (2 * (0.45 * height) ** 2))
    band = (xx > width * 0.62)
    scale = np.choose(np.minimum((yy / (height / 5)).astype(int), 4), [4, 8, 16, 32, 64])
    chk = ((xx // scale + yy // scale) % 2) * 2.0 - 1.0
    y = np.where(band, 128.0 + 55.0 * chk, y)
    y += 25.0 * (xx / width - 0.5) + rng.normal(0, 1.5, size=y.shape)
    return np.clip(np.rint(y), 0, 255).astype(np.uint8)


def bd_rate(r_anchor, p_anchor, r_test, p_test):
    """Bjontegaard delta rate (%), cubic fit of log-rate over PSNR, integrated over the common PSNR interval."""
    la, lt = np.log(np.asarray(r_anchor, float)), np.log(np.asarray(r_test, float))
    pa, pt = np.asarray(p_anchor, float), np.asarray(p_test, float)
    ca, ct = np.polyfit(pa, la, 3), np.polyfit(pt, lt, 3)
    lo, hi = max(pa.min(), pt.min()), min(pa.max(), pt.max())
    ia, it = np.polyint(ca), np.polyint(ct)
    avg = ((np.polyval(it, hi) - np.polyval(it, lo)) - (np.polyval(ia, hi) - np.polyval(ia, lo))) / (hi - lo)
    return (np.exp(avg) - 1.0) * 100.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", default=os.path.join(ROOT, "fasthevc_amd", "weights", "depthnet_v1.fhw"))
    ap.add_argument("--crop", default="1920x1080")
    ap.add_argument("--content", default="hetero,texture16")
    ap.add_argument("--floors", action="store_true", help="also evaluate the constant-depth maps")
    ap.add_argument("--margins", default="", help="comma-separated soft-hook margins to sweep: m (both sides) or split:stop (e.g. 8000,16000:0)")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    from oracle import oracle_py as op
    ref = op.bind_rdo(op.load_ref())
    hook = op.bind_rdo(op.load_ref(hook=True))
    oracle = op.load_oracle()
    w = weights.load(args.weights)
    ws = op.weights_from_arrays(w)
    cw_, ch_ = (int(v) for v in args.crop.split("x"))
    report = {}
    for name in args.content.split(","):
        if name in ("ood", "fractal", "gratings", "polygon", "chirp", "deadleaves"):
            full = {"ood": ood_luma, "fractal": lambda w, h: frames.fractal_luma(w, h, seed=424242),
                    "chirp": lambda w, h: frames.chirp_luma(w, h, seed=424242),
                    "deadleaves": lambda w, h: frames.deadleaves_luma(w, h, seed=424242),
                    "gratings": lambda w, h: frames.gratings_luma(w, h, seed=424242),
                    "polygon": lambda w, h: frames.polygon_luma(w, h, seed=424242)}[name](1920, 1080)
            cu = cv = np.full((540, 960), 128, np.uint8)
        else:
            full = frames.hetero_luma(1920, 1080) if name == "hetero" else frames.texture16_luma(1920, 1080)
            cu, cv = frames.chroma_planes(name, 1920, 1080)
        luma = full[:ch_, :cw_].copy()
        chroma = (cu[:ch_ // 2, :cw_ // 2].astype(np.int16), cv[:ch_ // 2, :cw_ // 2].astype(np.int16))
        buf, org, stride = frames.to_pel_plane(luma, 8)
        H, Wd = luma.shape
        n = ((Wd + 63) // 64) * ((H + 63) // 64)
        rows = {"anchor": [], "cnn": []}
        if args.floors:
            for c in range(4):
                rows[f"const{c}"] = []
        margins = [tuple(int(x) for x in (v.split(":") if ":" in v else (v, v))) for v in args.margins.split(",") if v]
        for mg in margins:
            rows[f"cnn_margin{mg[0]}:{mg[1]}"] = []
        agree = []
        for qp in QPS:
            d_anchor, s_anchor = op.rdo_encode(ref, buf, org, stride, Wd, H, 8, qp, chroma=chroma)
            rows["anchor"].append((s_anchor["coded_bits"], s_anchor["psnr_y"], s_anchor["seconds"]))
            pred = np.zeros(n * 256, np.uint8)
            logits = np.zeros(n * 42, np.int32)
            import ctypes as C
            oracle.fho_predict_frame(ws, op.ptr(buf.reshape(-1), org), stride, Wd, H, 8, qp, pred, C.c_void_p(logits.ctypes.data))
            pred = pred.reshape(n, 256)
            for mg in margins:  # soft hook: RDO decides wherever the classifier's margin is small
                dmin, dmax = np.zeros((n, 256), np.uint8), np.zeros((n, 256), np.uint8)
                cwn0 = (Wd + 63) // 64
                for c in range(n):
                    vw, vh = min(64, Wd - (c % cwn0) * 64), min(64, H - (c // cwn0) * 64)
                    oracle.fho_depth_range_from_logits(np.ascontiguousarray(logits[c * 42:(c + 1) * 42]), vw, vh, mg[0], mg[1], dmin[c], dmax[c])
                _, s_m = op.rdo_encode(hook, buf, org, stride, Wd, H, 8, qp, forced_depth=dmin, chroma=chroma, forced_depth_max=dmax)
                rows[f"cnn_margin{mg[0]}:{mg[1]}"].append((s_m["coded_bits"], s_m["psnr_y"], s_m["seconds"], float((dmin != dmax).mean())))
            _, s_cnn = op.rdo_encode(hook, buf, org, stride, Wd, H, 8, qp, forced_depth=pred, chroma=chroma)
            rows["cnn"].append((s_cnn["coded_bits"], s_cnn["psnr_y"], s_cnn["seconds"]))
            inpic = np.ones((n, 16, 16), bool)  # compare in-picture units only
            cwn = (Wd + 63) // 64
            for c in range(n):
                vw, vh = min(64, Wd - (c % cwn) * 64), min(64, H - (c // cwn) * 64)
                inpic[c, vh // 4:, :] = False
                inpic[c, :, vw // 4:] = False
            m = inpic.reshape(n, 256)
            agree.append((float((pred[m] == d_anchor[m]).mean()), float(np.abs(pred[m].astype(int) - d_anchor[m].astype(int)).mean()),
                          np.bincount(pred[m], minlength=4).tolist(), np.bincount(d_anchor[m], minlength=4).tolist()))
            if args.floors:
                for c in range(4):
                    _, s_c = op.rdo_encode(hook, buf, org, stride, Wd, H, 8, qp, forced_depth=np.full((n, 256), c, np.uint8), chroma=chroma)
                    rows[f"const{c}"].append((s_c["coded_bits"], s_c["psnr_y"], s_c["seconds"]))
            print(f"{name} qp{qp}: anchor {s_anchor['coded_bits']:.0f} b {s_anchor['psnr_y']:.3f} dB {s_anchor['seconds']:.2f} s | "
                  f"cnn {s_cnn['coded_bits']:.0f} b {s_cnn['psnr_y']:.3f} dB {s_cnn['seconds']:.2f} s | unit agreement {agree[-1][0]:.3f}", flush=True)
        ra, pa = [r[0] for r in rows["anchor"]], [r[1] for r in rows["anchor"]]
        rep = {"anchor": rows["anchor"], "agreement": agree}
        for k, v in rows.items():
            if k == "anchor":
                continue
            rep[k] = {"points": v, "bd_rate_percent": bd_rate(ra, pa, [r[0] for r in v], [r[1] for r in v]),
                      "time_ratio": float(np.sum([r[2] for r in rows["anchor"]]) / np.sum([r[2] for r in v]))}
            print(f"{name}: {k}: BD-rate {rep[k]['bd_rate_percent']:+.2f} %  decision time {rep[k]['time_ratio']:.2f}x faster")
        report[name] = rep
    if args.json:
        with open(args.json, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()

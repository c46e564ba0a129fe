#!/usr/bin/env python3
"""TEST/QUALITY INFRASTRUCTURE (uses oracle/ and oracle/_ref).  The consumer of the 35-mode first pass: HM's
estIntraPredLumaQT takes its candidate list from the source-only first pass (fhevc_intra_first_pass_candidates; the CPU oracle evaluates it here, the
HIP kernel is bit-exact with it) instead of running its own 35-mode Hadamard pass on reconstructed neighbours (hm_patch: FHEVC_FIRST_PASS=1).
Per family: BD-rate and time in compressSlice against the reference's full RDO, for
  first_pass        candidate lists only (depth search unrestricted)
  depth             the shipped depth hook alone (depthnet_v2.fhw at its default margins)
  depth+first_pass  both
usage: python tests/quality/eval_first_pass.py --pictures 3 --json profiles/r03_first_pass_consumer.json
"""
import argparse
import ctypes as C
import json
import os
import sys
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fasthevc_amd import frames, weights  # noqa: E402
from eval_families import FAMILIES, QPS, picture  # noqa: E402

MARGINS = (100000, 64000)


def work(job):
    family, k, (W, H), blob, extra = job
    os.environ["FHEVC_FIRST_PASS_EXTRA"] = str(extra)   # read when the harness builds the encoder of a geometry: before the first encode of this process
    from oracle import oracle_py as op
    ref, hook, oracle = op.bind_rdo(op.load_ref()), op.bind_rdo(op.load_ref(hook=True)), op.load_oracle()
    oracle.fho_first_pass_candidates_ctu.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]
    ws = op.weights_from_arrays(weights.load(blob))
    luma = picture(family, k, W, H)
    buf, org, stride = frames.to_pel_plane(luma, 8)
    u = np.full((H // 2, W // 2), 128, np.int16)
    cw, n = (W + 63) // 64, ((W + 63) // 64) * ((H + 63) // 64)
    out = {}
    for qp in QPS:
        _, sa = op.rdo_encode(ref, buf, org, stride, W, H, 8, qp, chroma=(u, u))
        out[("anchor", qp)] = (sa["coded_bits"], sa["psnr_y"], sa["seconds"])
        cand = np.zeros((n, 85, 8), np.uint8)
        sl = oracle.fho_lambda_intra(qp, 8) ** 0.5
        for c in range(n):
            oracle.fho_first_pass_candidates_ctu(C.c_void_p(buf.reshape(-1).ctypes.data + 2 * org), stride, W, H, c % cw, c // cw, 8, C.c_double(sl), 8, cand[c].ctypes.data)
        pred, logits = np.zeros(n * 256, np.uint8), np.zeros(n * 42, np.int32)
        oracle.fho_predict_frame(ws, op.ptr(buf.reshape(-1), org), stride, W, H, 8, qp, pred, C.c_void_p(logits.ctypes.data))
        dmin, dmax = np.zeros((n, 256), np.uint8), np.zeros((n, 256), np.uint8)
        for c in range(n):
            vw, vh = min(64, W - (c % cw) * 64), min(64, H - (c // cw) * 64)
            oracle.fho_depth_range_from_logits(np.ascontiguousarray(logits[c * 42:(c + 1) * 42]), vw, vh, MARGINS[0], MARGINS[1], dmin[c], dmax[c])
        for name, kw in (("first_pass", {"candidates": cand}), ("depth", {"forced_depth": dmin, "forced_depth_max": dmax}),
                         ("depth+first_pass", {"forced_depth": dmin, "forced_depth_max": dmax, "candidates": cand})):
            _, sv = op.rdo_encode(hook, buf, org, stride, W, H, 8, qp, chroma=(u, u), **kw)
            out[(name, qp)] = (sv["coded_bits"], sv["psnr_y"], sv["seconds"])
    return family, k, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", default=os.path.join(ROOT, "fasthevc_amd", "weights", "depthnet_v2.fhw"))
    ap.add_argument("--size", default="1024x576")
    ap.add_argument("--pictures", type=int, default=3)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--json", default=None)
    ap.add_argument("--extra", type=int, default=0, help="FHEVC_FIRST_PASS_EXTRA: that many more candidates than HM's numModesForFullRD go to the full RD check")
    args = ap.parse_args()
    from eval_rd import bd_rate
    W, H = (int(v) for v in args.size.split("x"))
    jobs = [(f, k, (W, H), args.weights, args.extra) for f in FAMILIES for k in range(args.pictures)]
    res = {}
    with Pool(args.workers) as pool:
        for i, (family, k, out) in enumerate(pool.imap_unordered(work, jobs)):
            res[(family, k)] = out
            print(f"{i + 1}/{len(jobs)} {family} #{k}", flush=True)
    variants = ("first_pass", "depth", "depth+first_pass")
    report = {"what": f"{args.pictures} unseen {W}x{H} pictures per family, QP {list(QPS)}: HM's estIntraPredLumaQT fed with the candidate lists of the source-only "
                      f"first pass (PUs of 8x8 and larger; 4x4 PUs keep HM's own pass), alone and with the depth hook ({os.path.basename(args.weights)} at "
                      f"{MARGINS[0]}:{MARGINS[1]}); FHEVC_FIRST_PASS_EXTRA={args.extra}; decision-stage BD-rate and time in compressSlice vs the reference's full RDO", "families": {}, "summary": {}}
    for f in FAMILIES:
        curve = lambda v: [(sum(res[(f, k)][(v, qp)][0] for k in range(args.pictures)), float(np.mean([res[(f, k)][(v, qp)][1] for k in range(args.pictures)])),
                            sum(res[(f, k)][(v, qp)][2] for k in range(args.pictures))) for qp in QPS]
        a = curve("anchor")
        fam = {"anchor": a}
        for v in variants:
            c = curve(v)
            fam[v] = {"points": c, "bd_rate_percent": bd_rate([p[0] for p in a], [p[1] for p in a], [p[0] for p in c], [p[1] for p in c]),
                      "time_ratio": sum(p[2] for p in a) / sum(p[2] for p in c)}
        report["families"][f] = fam
        print(f"{f:11s} " + "  ".join(f"{v} {fam[v]['bd_rate_percent']:+.2f}%/{fam[v]['time_ratio']:.2f}x" for v in variants), flush=True)
    for v in variants:
        report["summary"][v] = {f: f"{report['families'][f][v]['bd_rate_percent']:+.2f} % at {report['families'][f][v]['time_ratio']:.2f}x" for f in FAMILIES}
    if args.json:
        with open(args.json, "w") as fo:
            json.dump(report, fo, indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""TEST/QUALITY INFRASTRUCTURE (uses oracle/ and oracle/_ref).  The <= 1 % BD-rate guard, family by family: for every synthetic
family (the seven of the training set + the held-out zone-plate / checkerboard mix) N pictures with seeds no label ever saw, QP
{22, 27, 32, 37}:
  anchor  = the reference's full-RDO decision path (oracle/_ref/libhmref.so)
  variant = the hm_patch hook (libhmref_hook.so) under depth ranges of the shipped classifier at a (margin_split : margin_stop)
            setting of the soft hook (0:0 = hard decisions; the CPU oracle evaluates the blob, the HIP kernel is bit-exact with it)
Per family: BD-rate over the four QPs with bits and PSNR summed / averaged over its pictures (one RD curve per family), the time
ratio in compressSlice, and the share of 4x4 units left to HM's own search.  Decision-stage figures (bits counted by encodeCtu,
luma PSNR before the in-loop filters) plus the same BD-rate on the PSNR after the reference's own deblocking filter.

Margins: "split:stop", each side one number for all levels or "m64/m32/m16" per split level, "inf" = never.
usage: python tests/quality/eval_families.py --pictures 5 --size 1024x576 --margins 0:0,32000:0,inf:16000/8000/8000 --json profiles/r02_bdrate_generalization.json
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

QPS = (22, 27, 32, 37)
FAMILIES = ("hetero", "texture16", "fractal", "gratings", "polygon", "chirp", "deadleaves", "ood", "glyphs", "waves")   # the last three: never in any training label


def picture(family, k, W, H):
    """picture k of a family with a seed outside every training range (make_labels.py uses 10 000 .. 75 999)"""
    import eval_rd
    seed = 900_000 + 1000 * FAMILIES.index(family) + k
    gen = {"hetero": frames.hetero_luma, "texture16": frames.texture16_luma, "fractal": frames.fractal_luma, "gratings": frames.gratings_luma,
           "polygon": frames.polygon_luma, "chirp": frames.chirp_luma, "deadleaves": frames.deadleaves_luma, "ood": eval_rd.ood_luma,
           "glyphs": frames.glyphs_luma, "waves": frames.waves_luma}[family]
    return gen(W, H, seed=seed)


def work(job):
    family, k, (W, H), margins, blob = job
    os.environ["FHREF_DEBLOCK"] = "1"        # the harness also runs the reference's deblocking filter and reports the PSNR after it
    os.environ["FHREF_ENCODE_SLICE"] = "1"   # ... and the reference's encodeSlice (real arithmetic coder): slice-data bits as written
    os.environ["FHREF_SAO"] = "1"            # ... and the reference's own SAOProcess after the deblocking pass (its syntax is in the written bits)
    from oracle import oracle_py as op
    ref, hook, oracle = op.bind_rdo(op.load_ref()), op.bind_rdo(op.load_ref(hook=True)), op.load_oracle()
    wd = weights.load_any(blob)   # FHW1 (16 / 32 / 64) or FHW3 (a member of the reference's Bayesian-optimisation family)
    fam = op.family_from_arrays(wd) if "widths" in wd else None
    ws = None if fam is not None else op.weights_from_arrays(wd)
    luma = picture(family, k, W, H)
    buf, org, stride = frames.to_pel_plane(luma, 8)
    u = np.full((H // 2, W // 2), 128, np.int16)
    n, cw = ((W + 63) // 64) * ((H + 63) // 64), (W + 63) // 64
    out = {}
    for qp in QPS:
        _, sa = op.rdo_encode(ref, buf, org, stride, W, H, 8, qp, chroma=(u, u))
        out[("anchor", qp)] = (sa["coded_bits"], sa["psnr_y"], sa["seconds"], 1.0, sa.get("psnr_y_deblocked", sa["psnr_y"]), sa.get("slice_data_bits", sa["coded_bits"]), sa.get("psnr_y_filtered", sa.get("psnr_y_deblocked", sa["psnr_y"])))
        pred = np.zeros(n * 256, np.uint8)
        logits = np.zeros(n * 42, np.int32)
        if fam is not None:
            oracle.fho_predict_frame_family(C.byref(fam), op.ptr(buf.reshape(-1), org), stride, W, H, 8, qp, pred.ctypes.data, logits.ctypes.data)
        else:
            oracle.fho_predict_frame(ws, op.ptr(buf.reshape(-1), org), stride, W, H, 8, qp, pred, C.c_void_p(logits.ctypes.data))
        for (name, ms, mt) in margins:
            dmin, dmax = np.zeros((n, 256), np.uint8), np.zeros((n, 256), np.uint8)
            for c in range(n):
                vw, vh = min(64, W - (c % cw) * 64), min(64, H - (c // cw) * 64)
                oracle.fho_depth_range_from_logits_levels(np.ascontiguousarray(logits[c * 42:(c + 1) * 42]), vw, vh, ms, mt, dmin[c], dmax[c])
            _, sv = op.rdo_encode(hook, buf, org, stride, W, H, 8, qp, forced_depth=dmin, forced_depth_max=dmax, chroma=(u, u))
            out[(name, qp)] = (sv["coded_bits"], sv["psnr_y"], sv["seconds"], float((dmin != dmax).mean()), sv.get("psnr_y_deblocked", sv["psnr_y"]), sv.get("slice_data_bits", sv["coded_bits"]), sv.get("psnr_y_filtered", sv.get("psnr_y_deblocked", sv["psnr_y"])))
        for c in (range(4) if not os.environ.get("FHEVC_EVAL_NO_FLOORS") else ()):  # trivial floors
            _, sc = op.rdo_encode(hook, buf, org, stride, W, H, 8, qp, forced_depth=np.full((n, 256), c, np.uint8), chroma=(u, u))
            out[(f"const{c}", qp)] = (sc["coded_bits"], sc["psnr_y"], sc["seconds"], 0.0, sc.get("psnr_y_deblocked", sc["psnr_y"]), sc.get("slice_data_bits", sc["coded_bits"]), sc.get("psnr_y_filtered", sc.get("psnr_y_deblocked", sc["psnr_y"])))
    return family, k, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", default=os.path.join(ROOT, "fasthevc_amd", "weights", "depthnet_v1.fhw"))
    ap.add_argument("--size", default="1024x576")
    ap.add_argument("--pictures", type=int, default=5)
    ap.add_argument("--margins", default="0:0,32000:0")
    ap.add_argument("--families", default=",".join(FAMILIES))
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    from eval_rd import bd_rate
    W, H = (int(v) for v in args.size.split("x"))
    def side(t):  # "m" (all levels) or "m64/m32/m16"; "inf" = never (2^30)
        v = [(1 << 30) if x == "inf" else int(x) for x in t.split("/")]
        return np.array(v * 3 if len(v) == 1 else v, np.int32)
    margins = [(m, side(m.split(":")[0]), side(m.split(":")[1])) for m in args.margins.split(",")]
    fams = args.families.split(",")
    jobs = [(f, k, (W, H), margins, args.weights) for f in fams for k in range(args.pictures)]
    res = {}
    with Pool(args.workers) as pool:
        for i, (family, k, out) in enumerate(pool.imap_unordered(work, jobs)):
            res[(family, k)] = out
            print(f"{i + 1}/{len(jobs)} {family} #{k}", flush=True)
    variants = [m[0] for m in margins] + ([] if os.environ.get("FHEVC_EVAL_NO_FLOORS") else [f"const{c}" for c in range(4)])
    report = {"what": f"{args.pictures} unseen {W}x{H} pictures per family, QP {list(QPS)}, shipped blob {os.path.basename(args.weights)}; per family ONE RD curve "
                      "(bits summed, PSNR averaged over its pictures); decision-stage BD-rate vs the reference's full RDO", "families": {}, "summary": {}}
    for f in fams:
        def curve(v):
            return [(sum(res[(f, k)][(v, qp)][0] for k in range(args.pictures)), float(np.mean([res[(f, k)][(v, qp)][1] for k in range(args.pictures)])),
                     sum(res[(f, k)][(v, qp)][2] for k in range(args.pictures)), float(np.mean([res[(f, k)][(v, qp)][3] for k in range(args.pictures)])),
                     float(np.mean([res[(f, k)][(v, qp)][4] for k in range(args.pictures)])),
                     sum(res[(f, k)][(v, qp)][5] for k in range(args.pictures)),
                     float(np.mean([res[(f, k)][(v, qp)][6] for k in range(args.pictures)]))) for qp in QPS]
        a = curve("anchor")
        fam = {"anchor": a}
        for v in variants:
            c = curve(v)
            fam[v] = {"points": c, "bd_rate_percent": bd_rate([p[0] for p in a], [p[1] for p in a], [p[0] for p in c], [p[1] for p in c]),
                      "bd_rate_percent_after_deblocking": bd_rate([p[0] for p in a], [p[4] for p in a], [p[0] for p in c], [p[4] for p in c]),
                      # rate = slice data written by the reference's encodeSlice, distortion = PSNR after its deblocking filter
                      "bd_rate_percent_true_rate_after_deblocking": bd_rate([p[5] for p in a], [p[4] for p in a], [p[5] for p in c], [p[4] for p in c]),
                      # rate = slice data as written (incl. the SAO syntax), distortion = PSNR after BOTH in-loop filters (deblocking + the reference's own SAO)
                      "bd_rate_percent_written_bits_after_both_filters": bd_rate([p[5] for p in a], [p[6] for p in a], [p[5] for p in c], [p[6] for p in c]),
                      "time_ratio": sum(p[2] for p in a) / sum(p[2] for p in c), "units_left_to_rdo": float(np.mean([p[3] for p in c]))}
        # per picture spread of the first variant pair, to show how much one picture moves the figure
        fam["per_picture_bd_rate"] = {v: [bd_rate([res[(f, k)][("anchor", qp)][0] for qp in QPS], [res[(f, k)][("anchor", qp)][1] for qp in QPS],
                                                   [res[(f, k)][(v, qp)][0] for qp in QPS], [res[(f, k)][(v, qp)][1] for qp in QPS]) for k in range(args.pictures)]
                                      for v in variants[:len(margins)]}
        report["families"][f] = fam
        line = "  ".join(f"{v} {fam[v]['bd_rate_percent']:+.2f}%/{fam[v]['time_ratio']:.1f}x" for v in variants)
        print(f"{f:11s} {line}", flush=True)
    for v in variants:
        report["summary"][v] = {f: f"{report['families'][f][v]['bd_rate_percent']:+.2f} % at {report['families'][f][v]['time_ratio']:.1f}x" for f in fams}
    if args.json:
        with open(args.json, "w") as fo:
            json.dump(report, fo, indent=1)


if __name__ == "__main__":
    main()

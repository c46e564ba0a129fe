#!/usr/bin/env python3
"""Generate tests/golden/ref_vectors.npz by running the REFERENCE's own functions
(oracle/_ref/libhmref.so = objects compiled from /root/reference + oracle/ref_harness.cpp).

Run in the build container only (needs /root/reference):  make -C oracle ref && python oracle/gen_golden.py
The fixture holds inputs and expected outputs (data), never reference source.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import oracle_py as op  # noqa: E402
from fasthevc_amd import frames  # noqa: E402


def main():
    ref = op.load_ref()
    out = {"hm_version": np.frombuffer(ref.href_version(), np.uint8)}
    rng = np.random.default_rng(20261004)

    # --- scan tables (A14)
    r2z = np.zeros(256, np.uint32)
    z2r = np.zeros(256, np.uint32)
    ref.href_scan_tables(r2z, z2r)
    out["raster_to_zscan"], out["zscan_to_raster"] = r2z, z2r

    # --- SATD (A5): calcHAD and xGetHADs, 8- and 10-bit, square and rectangular
    shapes = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (8, 4), (4, 8), (16, 8), (8, 16), (32, 8), (16, 4), (2, 2), (6, 6)]
    satd_a, satd_b, satd_meta, satd_calc, satd_get = [], [], [], [], []
    for bd in (8, 10):
        for (w, h) in shapes:
            for rep in range(3):
                hi = (1 << bd) - 1
                if rep == 2:  # extremes: exercises the largest intermediates
                    a = rng.choice(np.array([0, hi], np.int16), size=(64, 64))
                    b = rng.choice(np.array([0, hi], np.int16), size=(64, 64))
                else:
                    a = rng.integers(0, hi + 1, size=(64, 64)).astype(np.int16)
                    b = np.clip(a + rng.normal(0, 12 * (1 << (bd - 8)), size=(64, 64)), 0, hi).astype(np.int16)
                g = ref.href_get_hads(bd, op.ptr(a), 64, op.ptr(b), 64, w, h)
                c = ref.href_calc_had(bd, op.ptr(a), 64, op.ptr(b), 64, w, h) if (w % 4 == 0 and h % 4 == 0) else 0xFFFFFFFF
                satd_a.append(a); satd_b.append(b); satd_meta.append((bd, w, h)); satd_calc.append(c); satd_get.append(g)
    out["satd_a"] = np.stack(satd_a); out["satd_b"] = np.stack(satd_b)
    out["satd_meta"] = np.array(satd_meta, np.int32)
    out["satd_calchad"] = np.array(satd_calc, np.uint32); out["satd_gethads"] = np.array(satd_get, np.uint32)

    # --- source Hadamard (A6): single blocks + per-CTU sums of the pinned synthetic frames
    blk = rng.integers(0, 1024, size=(16, 8, 8)).astype(np.int16)
    out["islice_blocks"] = blk
    out["islice_expected"] = np.array([ref.href_had8x8_islice(op.ptr(np.ascontiguousarray(b)), 8) for b in blk], np.int32)
    for name, (w, h, bd) in {"t16_416x240_8": (416, 240, 8), "t16_416x240_10": (416, 240, 10), "t16_1920x1080_8": (1920, 1080, 8)}.items():
        luma = frames.texture16_luma(w, h)
        buf, org, stride = frames.to_pel_plane(luma, bd)
        flat = buf.reshape(-1)
        cw, ch = frames.ctu_grid(w, h)
        vals = np.zeros(cw * ch, np.int32)
        for cy in range(ch):
            for cx in range(cw):
                vals[cy * cw + cx] = ref.href_ctu_src_hadamard(op.ptr(flat, org + cy * 64 * stride + cx * 64), stride,
                                                               min(64, w - cx * 64), min(64, h - cy * 64))
        out["ctu_had_" + name] = vals

    # --- reference-sample fill (A7) on random availability patterns
    fr_roi, fr_flags, fr_n, fr_bd, fr_out = [], [], [], [], []
    for n in (4, 8, 16, 32, 64):
        for rep in range(6):
            bd = 8 if rep % 2 == 0 else 10
            pic = rng.integers(0, 1 << bd, size=(2 * n + 8, 2 * n + 8)).astype(np.int16)
            units = 2 * n // 4
            if rep == 0:
                flags = np.ones(2 * units + 1, np.uint8)
            elif rep == 1:
                flags = np.zeros(2 * units + 1, np.uint8)
            else:
                flags = (rng.random(2 * units + 1) < (0.3 + 0.2 * rep / 6)).astype(np.uint8)
                if rep == 5:
                    flags[:units] = 0  # nothing below/left available -> exercises the "search upward" padding
            roi = np.zeros((2 * n + 1) * (2 * n + 1), np.int16)
            stride = pic.shape[1]
            ref.href_fill_ref(bd, op.ptr(pic.reshape(-1), 4 * stride + 4), stride, flags, n, roi)
            full = np.zeros((129, 129), np.int16)
            full[:2 * n + 1, :2 * n + 1] = roi.reshape(2 * n + 1, 2 * n + 1)
            padded = np.zeros((136, 136), np.int16)
            padded[:pic.shape[0], :pic.shape[1]] = pic
            fl = np.zeros(65, np.uint8); fl[:flags.size] = flags
            fr_roi.append(padded); fr_flags.append(fl); fr_n.append(n); fr_bd.append(bd); fr_out.append(full)
    out["fill_pic"] = np.stack(fr_roi); out["fill_flags"] = np.stack(fr_flags)
    out["fill_n"] = np.array(fr_n, np.int32); out["fill_bd"] = np.array(fr_bd, np.int32); out["fill_out"] = np.stack(fr_out)

    # --- filter decision table (A7) and the 35 predictors (A8)
    out["use_filtered"] = np.array([[ref.href_use_filtered(m, n) for m in range(35)] for n in (4, 8, 16, 32, 64)], np.uint8)
    for n in (4, 8, 16, 32, 64):
        reps = 2 if n <= 32 else 1
        for rep in range(reps):
            bd = 8 if rep == 0 else 10
            line = np.clip(rng.integers(0, 1 << bd, size=4 * n + 1) * (0.5 + 0.5 * rng.random()), 0, (1 << bd) - 1).astype(np.int16)
            if rep == 1:
                line[::7] = (1 << bd) - 1  # saturating samples for the edge-filter clip
            roi = np.ascontiguousarray(op.ref_line_to_roi(line, n).reshape(-1))
            preds = np.zeros((35, n, n), np.int16)
            for m in range(35):
                p = np.zeros(n * n, np.int16)
                ref.href_pred_intra(roi, n, m, bd, p)
                preds[m] = p.reshape(n, n)
            out[f"pred_line_n{n}_bd{bd}"] = line
            out[f"pred_out_n{n}_bd{bd}"] = preds

    dst = os.path.join(os.path.dirname(HERE), "tests", "golden", "ref_vectors.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")
    preanalyze_golden(ref)
    intra_lines_golden(ref)
    pattern_search_golden(ref)
    pattern_search_golden(ref, PATTERN_SEARCH_WIDE_CASES, "ref_pattern_search_wide.npz")
    pattern_search_golden(ref, PATTERN_SEARCH_WIDE10_CASES, "ref_pattern_search_wide10.npz")


PREANALYZE_CASES = (("texture16", 416, 240, 8, 3), ("hetero", 1000, 568, 10, 4), ("hetero", 64, 64, 8, 1))


def preanalyze_golden(ref):
    """TEncPreanalyzer::xPreanalyze outputs (N3) on the md5-pinned synthetic pictures: expected doubles only."""
    out = {"cases": np.array([f"{c[0]}:{c[1]}x{c[2]}:bd{c[3]}:d{c[4]}" for c in PREANALYZE_CASES])}
    for k, (content, w, h, bd, depth) in enumerate(PREANALYZE_CASES):
        luma = frames.texture16_luma(w, h) if content == "texture16" else frames.hetero_luma(w, h)
        buf, org, stride = frames.to_pel_plane(luma, bd)
        n = sum(((w + (64 >> d) - 1) // (64 >> d)) * ((h + (64 >> d) - 1) // (64 >> d)) for d in range(depth))
        act = np.zeros(n)
        avg = np.zeros(depth)
        assert ref.href_preanalyze(op.ptr(buf.reshape(-1), org), stride, w, h, bd, depth, act, avg) == n
        out[f"act{k}"], out[f"avg{k}"] = act, avg
        for range_, qp in ((6, 32), (12, 3), (4, 50)):  # TEncCu::xComputeQP on those activities (clips at both ends)
            q = np.zeros(n, np.int32)
            assert ref.href_aq_qp(op.ptr(buf.reshape(-1), org), stride, w, h, bd, depth, range_, qp, q) == n
            out[f"qp{k}_r{range_}_q{qp}"] = q.astype(np.int8)
    dst = os.path.join(os.path.dirname(HERE), "tests", "golden", "ref_preanalyze.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


INTRA_LINE_CASES = (("hetero", 416, 240, 8, 32, (0, 1, 6, 7, 9, 13, 21, 24, 27)), ("texture16", 416, 240, 10, 27, (0, 6, 8, 20, 27)))


def intra_lines_golden(ref):
    """A7: unfiltered + smoothed reference lines from the reference's own initIntraPatternChType on live CUs (availability
    rule, substitution walk, [1 2 1] and strong smoothing), neighbours read from the original picture; and the reference's
    own xModeBitsIntra for the CU at the picture origin.  Expected outputs only."""
    import ctypes as C
    lib = op.bind_rdo(ref)
    lib.href_intra_lines.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p]
    lib.href_mode_bits_origin.argtypes = [C.c_int] * 5 + [C.c_void_p]
    out = {"cases": np.array([f"{c[0]}:{c[1]}x{c[2]}:bd{c[3]}:qp{c[4]}" for c in INTRA_LINE_CASES])}
    for k, (content, w, h, bd, qp, ctus) in enumerate(INTRA_LINE_CASES):
        luma = frames.texture16_luma(w, h) if content == "texture16" else frames.hetero_luma(w, h)
        buf, org, stride = frames.to_pel_plane(luma, bd)
        if bd > 8:  # use the low bits too (the same recipe as the tests)
            m = org % stride
            buf[m:m + h, m:m + w] += np.random.default_rng(bd).integers(0, 1 << (bd - 8), (h, w)).astype(np.int16)
        op.rdo_encode(lib, buf, org, stride, w, h, bd, qp)  # live TComPic / TComDataCU objects for this geometry
        cw = (w + 63) // 64
        meta, unf, flt = [], [], []
        first = True
        for c in ctus:
            for depth in range(4):
                n = 64 >> depth
                for by in range(1 << depth):
                    for bx in range(1 << depth):
                        x0, y0 = (c % cw) * 64 + bx * n, (c // cw) * 64 + by * n
                        if x0 + n > w or y0 + n > h:
                            continue
                        a, b = np.zeros(4 * n + 1, np.int16), np.zeros(4 * n + 1, np.int16)
                        rc = lib.href_intra_lines(w, h, bd, c, depth, (by * n // 4) * 16 + bx * n // 4, 1 if first else 0, a.ctypes.data, b.ctypes.data)
                        assert rc == 4 * n + 1, rc
                        first = False
                        meta.append((c, depth, x0, y0, n)); unf.append(a); flt.append(b)
        out[f"meta{k}"] = np.array(meta, np.int32)
        out[f"unf{k}"] = np.concatenate(unf)
        out[f"flt{k}"] = np.concatenate(flt)
    bits = np.zeros((3, 4, 35), np.uint32)
    for i, qp in enumerate((22, 32, 37)):
        for depth in range(4):
            assert lib.href_mode_bits_origin(416, 240, 8, qp, depth, bits[i, depth].ctypes.data) == 35
    out["mode_bits_qp22_32_37"] = bits
    dst = os.path.join(os.path.dirname(HERE), "tests", "golden", "ref_intra_lines.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    if "--intra-lines-only" in sys.argv:
        intra_lines_golden(op.load_ref())
    elif "--preanalyze-only" in sys.argv:
        preanalyze_golden(op.load_ref())
    else:
        main()


PATTERN_SEARCH_CASES = (  # (bit depth, qp, range, clip seed, CTUs of the 7 x 4 grid of 416x240: corners, edges (32 wide / 48 tall), interior)
    (8, 38, 4, 15, (0, 3, 6, 10, 21, 27)),
    (10, 33, 3, 16, (8, 13, 20)),
    (8, 22, 8, 17, (9, 26)))


PATTERN_SEARCH_WIDE_CASES = (  # the same at HM's own SearchRange and two odd ones, 8 bit, clips with large motion: (.., clip seed, CTUs, speeds)
    (8, 32, 64, 31, (0, 9, 27), (21, -37)),
    (8, 27, 24, 32, (3, 13), (13, 9)),
    (8, 37, 33, 33, (6, 17, 24), (-30, 5)))


PATTERN_SEARCH_WIDE10_CASES = (  # round 4: the wide window ABOVE 8 bit (16-bit SAD kernel laid out for +-64, k_motion.hip): 10 and 12 bit, low bits populated
    (10, 32, 64, 41, (0, 13), (17, -29)),
    (10, 27, 20, 42, (6, 27), (11, 7)),
    (12, 37, 40, 43, (24,), (-26, 12)))


def pattern_search_golden(ref, cases=None, name="ref_pattern_search.npz"):
    """Config 4: what the reference's OWN TEncSearch::xPatternSearch (TEncSearch.cpp:3786-3848: SAD + vector cost, raster order, strict <)
    returns for every in-picture CU node of whole CTUs -- vector, SAD, cost -- on the seeded pan clip (original planes, zero predictor).
    The planes are stored with the expected outputs (data only)."""
    import ctypes as C
    oracle = op.load_oracle()
    ref.href_pattern_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    W, H = 416, 240
    cases = PATTERN_SEARCH_CASES if cases is None else cases
    out = {"cases": np.array([list(c[:4]) for c in cases], np.int32)}
    total = 0
    for k, case in enumerate(cases):
        (bd, qp, rng, seed, ctus), speeds = case[:5], (case[5] if len(case) > 5 else (3, 3))
        ys = frames.pan_clip(W, H, 2, seed=seed, v_structure=speeds[0], v_noise=speeds[1])
        (rb, org, stride), (cb, _, _) = [frames.to_pel_plane(y, bd) for y in ys]
        if bd > 8:  # use the low bits too
            cb = (cb + np.random.default_rng(bd).integers(0, 1 << (bd - 8), size=cb.shape, dtype=np.int16)).astype(np.int16)
            rb = (rb + np.random.default_rng(bd + 1).integers(0, 1 << (bd - 8), size=rb.shape, dtype=np.int16)).astype(np.int16)
        cur = np.ascontiguousarray(cb.reshape(-1)[org:].copy()[: (H - 1) * stride + W])  # from sample (0, 0) on, HM stride
        refp = np.ascontiguousarray(rb.reshape(-1)[org:].copy()[: (H - 1) * stride + W])
        lam = oracle.fho_lambda_intra(qp, bd)
        res = np.full((len(ctus), 85, 4), -1, np.int32)
        for ci, c in enumerate(ctus):
            cx, cy = c % 7, c // 7
            blocks, where = [], []
            idx = 0
            for lvl in range(4):
                n, cnt = 64 >> lvl, 1 << lvl
                for by in range(cnt):
                    for bx in range(cnt):
                        x0, y0 = cx * 64 + bx * n, cy * 64 + by * n
                        if x0 + n <= W and y0 + n <= H:
                            blocks.append((x0, y0, n)); where.append(idx)
                        idx += 1
            b = np.array(blocks, np.int32)
            o = np.zeros((len(blocks), 4), np.int32)
            assert ref.href_pattern_search(cur.ctypes.data, refp.ctypes.data, stride, W, H, bd, C.c_double(lam), rng, len(blocks), b.ctypes.data, o.ctypes.data) == len(blocks)
            res[ci, where] = o
            total += len(blocks)
        out[f"cur{k}"], out[f"ref{k}"], out[f"stride{k}"] = cur, refp, np.int32(stride)
        out[f"ctus{k}"], out[f"nodes{k}"] = np.array(ctus, np.int32), res
    dst = os.path.join(os.path.dirname(HERE), "tests", "golden", name)
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes;", total, "nodes searched by the reference's xPatternSearch")

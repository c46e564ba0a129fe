"""Config 4 (P slices): the source-only motion search per CU node on the MI355X against the CPU oracle (fho_motion_ctu),
bit for bit: SATD at zero motion, cheapest vector, its SATD and cost, for all 85 nodes of every CTU."""
import ctypes as C

import numpy as np
import pytest

from fasthevc_amd import capi, frames

pytestmark = pytest.mark.gpu


def oracle_motion(oracle, cur, ref, origin, stride, W, H, bd, qp, rng, ctus=None, sad=False):
    cw, ch = (W + 63) // 64, (H + 63) // 64
    out = np.zeros((cw * ch, 85), capi.MOTION_DTYPE)
    sl = oracle.fho_lambda_intra(qp, bd) ** 0.5
    cp, rp = cur.reshape(-1).ctypes.data + 2 * origin, ref.reshape(-1).ctypes.data + 2 * origin
    for c in (range(cw * ch) if ctus is None else ctus):
        oracle.fho_motion_ctu_dist(C.c_void_p(cp), stride, C.c_void_p(rp), stride, W, H, c % cw, c // cw, bd, rng, C.c_double(sl), 1 if sad else 0,
                                   C.c_void_p(out[c].ctypes.data))
    return out


def same(a, b):
    return all(np.array_equal(a[k], b[k]) for k in capi.MOTION_DTYPE.names)


@pytest.mark.parametrize("sad", [False, True])
@pytest.mark.parametrize("bd,rng,qp", [(8, 4, 38), (8, 1, 22), (8, 8, 43), (10, 3, 33), (12, 2, 38), (8, 5, 30), (10, 7, 27)])
def test_motion_search_vs_oracle_small(oracle, bd, rng, qp, sad):
    W, H = 416, 240  # ragged: last CTU column 32 wide, last row 48 tall
    ys = frames.pan_clip(W, H, 2, seed=7 + bd)
    planes = [frames.to_pel_plane(y, bd) for y in ys]
    (rb, org, stride), (cb, _, _) = planes
    if bd > 8:  # use the low bits too
        noise = np.random.default_rng(bd).integers(0, 1 << (bd - 8), size=cb.shape, dtype=np.int16)
        cb = (cb + noise).astype(np.int16)
    ctx = capi.Context(W, H, bd)
    if sad:
        ctx.set_motion_distortion("sad")
    got = ctx.motion_search(cb, rb, org, stride, qp=qp, search_range=rng)
    exp = oracle_motion(oracle, cb, rb, org, stride, W, H, bd, qp, rng, sad=sad)
    assert same(got, exp)
    # border nodes are flagged, interior ones are not
    assert got["cost_best"][6, 0] == 0xFFFFFFFF and got["cost_best"][0, 0] != 0xFFFFFFFF
    ctx.close()


def test_motion_search_finds_a_pure_pan(oracle):
    W, H = 256, 192
    base = frames.hetero_luma(W + 16, H + 16, seed=99)
    ref, cur = base[8:8 + H, 8:8 + W], base[6:6 + H, 11:11 + W]   # cur(x, y) = ref(x + 3, y - 2)
    (rb, org, stride), (cb, _, _) = frames.to_pel_plane(np.ascontiguousarray(ref), 8), frames.to_pel_plane(np.ascontiguousarray(cur), 8)
    ctx = capi.Context(W, H, 8)
    got = ctx.motion_search(cb, rb, org, stride, qp=30, search_range=4)
    inner = got[5]  # CTU (1, 1): no border replication inside the window
    assert (inner["mvx"] == 3).all() and (inner["mvy"] == -2).all() and (inner["satd_best"] == 0).all()
    assert (inner["satd_zero"] > 0).any()
    ctx.close()


def test_motion_search_1080p_device_batch(oracle):
    """BASELINE config 4 geometry: the 1080p pan clip I P P P as a device-resident batch (uint8 planes and HM int16 planes),
    whole pictures and a CTU-row band; oracle on a sample of CTUs of every picture pair."""
    import torch
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    W, H, NF, qp, rng = 1920, 1080, 4, 38, 4
    ys = frames.pan_clip(W, H, NF)
    ctx = capi.Context(W, H, 8, max_frames=NF)
    n = ctx.num_ctus
    d8 = torch.from_numpy(np.stack(ys)).to(dev)
    out8 = torch.zeros(((NF - 1) * n * 85, 4), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.motion_search_device(d8.data_ptr(), 1, W, W * H, NF, out8.data_ptr(), qp=qp, search_range=rng)
    planes = [frames.to_pel_plane(y, 8) for y in ys]
    org, stride = planes[0][1], planes[0][2]
    d16 = torch.from_numpy(np.stack([p[0] for p in planes])).to(dev)
    out16 = torch.zeros_like(out8)
    band = torch.zeros(((NF - 1) * 3 * ctx.ctus_x * 85, 4), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    fs = planes[0][0].size
    torch.cuda.synchronize()  # the fills above run on torch's stream: order them explicitly before the library call
    ctx.motion_search_device(d16.data_ptr() + 2 * org, 2, stride, fs, NF, out16.data_ptr(), qp=qp, search_range=rng)
    torch.cuda.synchronize()  # the fills above run on torch's stream: order them explicitly before the library call
    ctx.motion_search_device(d16.data_ptr() + 2 * org, 2, stride, fs, NF, band.data_ptr(), rows=(14, 17), qp=qp, search_range=rng)
    torch.cuda.synchronize()
    g8 = out8.cpu().numpy().view(capi.MOTION_DTYPE).reshape(NF - 1, n, 85)
    g16 = out16.cpu().numpy().view(capi.MOTION_DTYPE).reshape(NF - 1, n, 85)
    gb = band.cpu().numpy().view(capi.MOTION_DTYPE).reshape(NF - 1, 3 * ctx.ctus_x, 85)
    assert same(g8, g16) and same(gb, g16[:, 14 * ctx.ctus_x:17 * ctx.ctus_x])
    sample = sorted(set(np.random.default_rng(5).integers(0, n, 40).tolist()) | {0, 29, n - 30, n - 1})
    for f in range(1, NF):
        exp = oracle_motion(oracle, planes[f][0], planes[f - 1][0], org, stride, W, H, 8, qp, rng, ctus=sample)
        assert same(g16[f - 1][sample], exp[sample]), f
    # The clip's two motions, derived from its generator (frames.pan_clip): the noise of the textured 16x16 blocks is rolled +3 px per
    # picture (cur(x) = ref(x - 3): vector -3), base and edges move the other way (vector +3).  Per 8x8 node (nodes 21.., raster order):
    #   * inside a textured block the noise (sigma 18) dominates: EVERY such node finds -3;
    #   * in an untextured block whose left neighbour is untextured too (no noise spills in) a node crossed by one of the vertical
    #     30-level edges (x = 45 mod 48 in picture 1) sees only the structure: never -3, and +3 wherever the edge outweighs the
    #     vector's cost (about half of them at this QP; the rest keep the free zero vector);
    #   * the other nodes of such blocks see a sinusoid of period 232 px: nothing to gain from any vector, they stay at 0.
    tex = np.random.default_rng(1234).integers(0, 2, size=(H // 16 + 1, W // 16 + 1))   # the generator's first draw
    cwn = ctx.ctus_x
    mvx, valid = g16[0][:, 21:]["mvx"], g16[0][:, 21:]["cost_best"] != 0xFFFFFFFF
    in_tex, on_edge, plain = [], [], []
    for c in range(n):
        for k in range(64):
            x0, y0 = (c % cwn) * 64 + (k % 8) * 8, (c // cwn) * 64 + (k // 8) * 8
            if not valid[c, k] or y0 + 8 > H:
                continue
            if tex[y0 // 16, x0 // 16]:
                in_tex.append(mvx[c, k])
            elif x0 >= 16 and not tex[y0 // 16, x0 // 16 - 1]:
                (on_edge if any((x + 3) % 48 == 0 for x in range(x0 + 1, x0 + 8)) else plain).append(mvx[c, k])
    in_tex, on_edge, plain = np.array(in_tex), np.array(on_edge), np.array(plain)
    assert len(in_tex) > 10000 and len(on_edge) > 500 and len(plain) > 4000
    assert (in_tex == -3).all()
    assert not (on_edge == -3).any() and (on_edge == 3).mean() > 0.25 and np.isin(on_edge, (0, 3)).all()
    assert (plain == 0).all()
    ctx.close()


@pytest.mark.parametrize("name,nodes", [("ref_pattern_search.npz", 706), ("ref_pattern_search_wide.npz", 517), ("ref_pattern_search_wide10.npz", 262)])
def test_sad_mode_reproduces_the_references_xPatternSearch(name, nodes):
    """The HIP kernels in the SAD mode, through the C ABI, against what the reference's OWN TEncSearch::xPatternSearch returned
    (tests/golden/ref_pattern_search.npz: 706 nodes of whole CTUs incl. picture corners and edges, 8 / 10 bit, ranges 3 / 4 / 8;
    ref_pattern_search_wide.npz: 517 nodes at HM's SearchRange 64 and at 24 / 33 on clips moving 5 .. 37 samples per picture --
    k_motion_wide.hip): vector, SAD and cost, directly -- no oracle in between."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))
    W, H = 416, 240
    checked = 0
    for k, (bd, qp, rng, _seed) in enumerate(g["cases"]):
        cur, ref, stride = g[f"cur{k}"], g[f"ref{k}"], int(g[f"stride{k}"])
        ctx = capi.Context(W, H, int(bd))
        ctx.set_motion_distortion("sad")
        got = ctx.motion_search(cur, ref, 0, stride, qp=int(qp), search_range=int(rng))
        for ci, c in enumerate(g[f"ctus{k}"]):
            exp = g[f"nodes{k}"][ci]
            valid = exp[:, 3] >= 0
            n = got[int(c)]
            assert (n["cost_best"][~valid] == 0xFFFFFFFF).all()
            assert np.array_equal(n["mvx"][valid], exp[valid, 0]) and np.array_equal(n["mvy"][valid], exp[valid, 1]), (k, c)
            assert np.array_equal(n["satd_best"][valid], exp[valid, 2].astype(np.uint32)) and np.array_equal(n["cost_best"][valid], exp[valid, 3].astype(np.uint32)), (k, c)
            checked += int(valid.sum())
        ctx.close()
    assert checked == nodes


@pytest.mark.parametrize("rng,qp,speeds,bd", [(64, 32, (21, -37), 8), (16, 22, (9, 14), 8), (33, 40, (-30, 5), 8), (9, 51, (3, 3), 8), (47, 27, (40, -44), 8),
                                              (64, 30, (19, -33), 10), (21, 37, (-12, 16), 10), (40, 25, (28, 9), 12)])
def test_wide_sad_search_vs_oracle(oracle, rng, qp, speeds, bd):
    """Ranges above 8 (8 bit: k_motion_wide.hip, 8 dy x 4 dx vectors per lane on v_qsad_pk_u16_u8, keys merged by LDS atomic minima; above 8 bit:
    k_motion.hip's 16-bit SAD kernel laid out for the +-64 window, round 4) against the oracle's plain loops, every node of every CTU of a ragged
    picture (last column 32 wide, last row 48 tall): zero-vector SAD, vector, SAD, cost.  Above 8 bit the low bits are populated."""
    W, H = 416, 240
    ys = frames.pan_clip(W, H, 2, seed=100 + rng, v_structure=speeds[0], v_noise=speeds[1])
    (rb, org, stride), (cb, _, _) = [frames.to_pel_plane(y, bd) for y in ys]
    if bd > 8:
        cb = (cb + np.random.default_rng(bd).integers(0, 1 << (bd - 8), size=cb.shape, dtype=np.int16)).astype(np.int16)
        rb = (rb + np.random.default_rng(bd + 1).integers(0, 1 << (bd - 8), size=rb.shape, dtype=np.int16)).astype(np.int16)
    ctx = capi.Context(W, H, bd)
    ctx.set_motion_distortion("sad")
    got = ctx.motion_search(cb, rb, org, stride, qp=qp, search_range=rng)
    exp = oracle_motion(oracle, cb, rb, org, stride, W, H, bd, qp, rng, sad=True)
    for k in capi.MOTION_DTYPE.names:
        assert np.array_equal(got[k], exp[k]), (k, np.argwhere(got[k] != exp[k])[:5])
    if min(abs(v) for v in speeds) > 8:
        assert (np.abs(got["mvx"][got["cost_best"] != 0xFFFFFFFF]) > 8).any()   # the clip does move farther than the small kernel reaches
    ctx.close()


def test_wide_sad_search_extremes_and_uint8_batch(oracle):
    """Saturated content (white on black: every packed 16-bit sum at its maximum, ties everywhere -> the FIRST vector in raster order must win),
    a flat picture (all costs equal up to the vector cost), and the device-resident uint8 batch entry point against the int16 one."""
    import torch
    W, H, rng, qp = 192, 128, 64, 30
    white, black = np.full((H, W), 255, np.uint8), np.zeros((H, W), np.uint8)
    half = black.copy(); half[:, W // 2:] = 255
    ctx = capi.Context(W, H, 8)
    ctx.set_motion_distortion("sad")
    for cur, ref in ((white, black), (black, white), (white, white), (half, np.roll(half, 40, axis=1))):
        (rb, org, stride), (cb, _, _) = frames.to_pel_plane(ref, 8), frames.to_pel_plane(cur, 8)
        got = ctx.motion_search(cb, rb, org, stride, qp=qp, search_range=rng)
        exp = oracle_motion(oracle, cb, rb, org, stride, W, H, 8, qp, rng, sad=True)
        assert same(got, exp)
    ys = frames.pan_clip(W, H, 3, seed=5, v_structure=25, v_noise=-18)
    d8 = torch.from_numpy(np.stack(ys)).cuda()
    out8 = torch.zeros((2, 6, 85, 16), dtype=torch.uint8, device="cuda")
    ctx.motion_search_device(d8.data_ptr(), 1, W, W * H, 3, out8.data_ptr(), qp=qp, search_range=rng)
    torch.cuda.synchronize()
    got8 = out8.cpu().numpy().view(capi.MOTION_DTYPE).reshape(2, 6, 85)
    for f in (1, 2):
        (rb, org, stride), (cb, _, _) = frames.to_pel_plane(ys[f - 1], 8), frames.to_pel_plane(ys[f], 8)
        assert same(got8[f - 1], ctx.motion_search(cb, rb, org, stride, qp=qp, search_range=rng))
    # above 8 the search needs the SAD distortion and 8-bit content
    ctx.set_motion_distortion("satd")
    with pytest.raises(capi.FastHevcError):
        ctx.motion_search(cb, rb, org, stride, qp=qp, search_range=9)
    ctx.close()


def test_wide_sad_search_on_a_ctu_row_band():
    """The +-64 search over a CTU-row band of a device-resident batch (fhevc_band partitions) equals the same rows of the whole-picture search."""
    import torch
    W, H, rng, qp = 320, 256, 64, 33
    ys = frames.pan_clip(W, H, 3, seed=11, v_structure=-27, v_noise=14)
    d8 = torch.from_numpy(np.stack(ys)).cuda()
    ctx = capi.Context(W, H, 8)
    ctx.set_motion_distortion("sad")
    n, cw = ctx.num_ctus, 5
    full = torch.zeros((2, n, 85, 16), dtype=torch.uint8, device="cuda")
    ctx.motion_search_device(d8.data_ptr(), 1, W, W * H, 3, full.data_ptr(), qp=qp, search_range=rng)
    band = torch.zeros((2, 2 * cw, 85, 16), dtype=torch.uint8, device="cuda")
    ctx.motion_search_device(d8.data_ptr(), 1, W, W * H, 3, band.data_ptr(), rows=(1, 3), qp=qp, search_range=rng)
    torch.cuda.synchronize()
    f, b = full.cpu().numpy(), band.cpu().numpy()
    assert np.array_equal(b, f[:, cw:3 * cw])
    assert (f.view(capi.MOTION_DTYPE)["mvx"] != 0).any()
    ctx.close()

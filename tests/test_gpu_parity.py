"""GPU parity: every entry point of the C ABI (include/fasthevc.h) against the CPU oracle and the golden vectors
produced by the reference's own functions.  Bit-exact: all of this path is integer arithmetic."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle_py as op
from fasthevc_amd import capi, frames, weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


def _oracle_frame(oracle, w, luma_u8, bd, qp=32):
    h, wd = luma_u8.shape
    buf, org, stride = frames.to_pel_plane(luma_u8, bd)
    cw, ch = frames.ctu_grid(wd, h)
    depth = np.zeros(cw * ch * 256, np.uint8)
    logits = np.zeros(cw * ch * 42, np.int32)
    oracle.fho_predict_frame(op.weights_from_arrays(w), op.ptr(buf.reshape(-1), org), stride, wd, h, bd, qp, depth,
                             C.c_void_p(logits.ctypes.data))
    had = np.zeros(cw * ch, np.int32)
    oracle.fho_frame_src_hadamard(op.ptr(buf.reshape(-1), org), stride, wd, h, had)
    return buf, org, stride, depth.reshape(-1, 256), logits.reshape(-1, 42), had


def test_satd_matches_reference_golden(golden):
    ctx = capi.Context(64, 64, 8)
    a, b, meta = golden["satd_a"], golden["satd_b"], golden["satd_meta"]
    for i in range(len(meta)):
        bd, w, h = (int(v) for v in meta[i])
        got = ctx.satd(a[i], b[i], w, h, bd, 64, 64)
        assert got == int(golden["satd_gethads"][i]), (bd, w, h)
    ctx.close()


def test_satd_random_vs_oracle(oracle):
    ctx = capi.Context(64, 64, 8)
    rng = np.random.default_rng(3)
    for _ in range(40):
        bd = int(rng.choice([8, 10, 12]))
        w, h = int(rng.choice([4, 8, 16, 32, 64])), int(rng.choice([4, 8, 16, 32, 64]))
        a = rng.integers(0, 1 << bd, size=(64, 64)).astype(np.int16)
        b = rng.integers(0, 1 << bd, size=(64, 64)).astype(np.int16)
        assert ctx.satd(a, b, w, h, bd, 64, 64) == oracle.fho_satd(op.ptr(a), 64, op.ptr(b), 64, w, h, bd)
    ctx.close()


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("seed,extreme", [(0, False), (7, True)])
def test_predict_frame_416x240(oracle, golden, bd, seed, extreme, cnn_arith):
    """config 1 geometry (7x4 CTUs, last column 32 px wide, last row 48 px tall), host-buffer entry point."""
    w = weights.random_weights(seed, extreme=extreme)
    luma = frames.texture16_luma(416, 240)
    qp = 22 + 5 * (seed % 4)   # the per-QP prior of the heads is part of the parity
    buf, org, stride, depth_ref, _, had_ref = _oracle_frame(oracle, w, luma, bd, qp)
    ctx = capi.Context(416, 240, bd, w)
    depth, had = ctx.predict_frame(buf, org, stride, qp=qp)
    assert np.array_equal(had, had_ref)
    assert np.array_equal(had, golden[f"ctu_had_t16_416x240_{bd}"])  # == the reference's updateCtuDataISlice
    bad = np.nonzero((depth != depth_ref).any(axis=1))[0]
    assert bad.size == 0, f"CTUs with a differing depth map: {bad[:10]}"
    s = ctx.stats()
    assert s["ctus"] == 28 and s["frames"] == 1
    ctx.close()


@pytest.mark.parametrize("shifts", [(6, 7, 8), (6, 8, 7), (5, 3, 8), (7, 7, 1), (6, 0, 14), (4, 8, 8)])
def test_requant_shifts_of_the_i8_form(oracle, shifts):
    """The i8 form's requant has three forms (k_cnn.hip: requant4_i8) chosen from the blob's shifts and accumulator bounds: shifts
    (6, 7, 8) with small weights take the short ones, every other combination the general one; all must floor and clamp as the oracle does."""
    w = weights.random_weights(11)
    w["shift"] = np.array(shifts, np.int32)
    luma = frames.hetero_luma(416, 240)
    buf, org, stride, depth_ref, logits_ref, had_ref = _oracle_frame(oracle, w, luma, 8, qp=27)
    for arith in ("i8", "f16"):
        ctx = capi.Context(416, 240, 8, w, arith=arith)
        depth, had = ctx.predict_frame(buf, org, stride, qp=27)
        assert np.array_equal(depth, depth_ref), (shifts, arith)
        assert np.array_equal(had, had_ref)
        ctx.close()


@pytest.mark.parametrize("which", ["random", "trained"])
def test_predict_frame_1080p_hetero(oracle, golden, which, cnn_arith):
    """config 2 geometry, the heterogeneous content: 510 CTUs, last row 56 px tall; random-init and shipped weights."""
    w = weights.random_weights(1) if which == "random" else weights.load(
        os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fasthevc_amd", "weights", "depthnet_v1.fhw"))
    luma = frames.hetero_luma(1920, 1080)
    buf, org, stride, depth_ref, _, had_ref = _oracle_frame(oracle, w, luma, 8)
    ctx = capi.Context(1920, 1080, 8, w)
    depth, had = ctx.predict_frame(buf, org, stride)
    assert np.array_equal(had, had_ref)
    assert np.array_equal(depth, depth_ref)
    assert len(np.unique(depth)) >= 3  # several depths occur, so the comparison is not vacuous
    buf2, org2, stride2 = frames.to_pel_plane(frames.texture16_luma(1920, 1080), 8)
    _, had2 = ctx.predict_frame(buf2, org2, stride2)
    assert np.array_equal(had2, golden["ctu_had_t16_1920x1080_8"])
    ctx.close()


def test_device_batch_logits_and_bands(oracle, torch_cuda, cnn_arith):
    """Device-resident batch entry point: 3 frames, uint8 and int16 sample layouts, logits, CTU-row bands."""
    torch = torch_cuda
    w = weights.random_weights(2)
    W, H, NF = 416, 240, 3
    lumas = [frames.texture16_luma(W, H, seed=100 + f) for f in range(NF)]
    refs = [_oracle_frame(oracle, w, y, 8, 37) for y in lumas]
    ctx = capi.Context(W, H, 8, w)
    dev = torch.device("cuda:0")
    n = ctx.num_ctus
    # uint8, tightly packed
    d8 = torch.from_numpy(np.stack(lumas)).to(dev)
    depth = torch.zeros((NF, n, 256), dtype=torch.uint8, device=dev)
    had = torch.zeros((NF, n), dtype=torch.int32, device=dev)
    logits = torch.zeros((NF, n, 42), dtype=torch.int32, device=dev)
    flags = torch.zeros((NF, n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()  # the fills above run on torch's stream: order them explicitly before the library call
    ctx.predict_frames_device(d8.data_ptr(), 1, W, W * H, NF, depth.data_ptr(), had.data_ptr(), logits.data_ptr(), qp=37,
                              d_flags=flags.data_ptr())
    expanded = torch.full((NF, n, 256), 9, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()  # the fills above run on torch's stream: order them explicitly before the library call
    ctx.expand_depth_flags_device(flags.data_ptr(), NF, expanded.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(expanded, depth)  # the 4-byte word per CTU carries the whole map (multi-GPU all-gather payload)
    for f in range(NF):
        lg = refs[f][4].reshape(n, 21, 2)
        for c in range(n):
            vw, vh = min(64, W - (c % ctx.ctus_x) * 64), min(64, H - (c // ctx.ctus_x) * 64)
            assert int(flags[f, c].item()) == oracle.fho_flags_from_logits(np.ascontiguousarray(lg[c].reshape(-1)), vw, vh)
    for f in range(NF):
        assert np.array_equal(depth[f].cpu().numpy(), refs[f][3])
        assert np.array_equal(logits[f].cpu().numpy(), refs[f][4])
        assert np.array_equal(had[f].cpu().numpy(), refs[f][5])
    # int16 Pel planes with HM's stride and margins, band [1, 3) of the 4 CTU rows only
    planes = np.stack([frames.to_pel_plane(y, 8)[0] for y in lumas])
    _, org, stride = frames.to_pel_plane(lumas[0], 8)
    d16 = torch.from_numpy(planes).to(dev)
    band = torch.full((NF, 2 * ctx.ctus_x, 256), 255, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()  # the fills above run on torch's stream: order them explicitly before the library call
    ctx.predict_frames_device(d16.data_ptr() + 2 * org, 2, stride, planes.shape[1] * planes.shape[2], NF,
                              band.data_ptr(), None, None, rows=(1, 3), stream=torch.cuda.current_stream().cuda_stream, qp=37)
    torch.cuda.synchronize()
    for f in range(NF):
        assert np.array_equal(band[f].cpu().numpy(), refs[f][3][ctx.ctus_x:3 * ctx.ctus_x])
    ctx.close()


@pytest.mark.parametrize("W,H", [(200, 136), (72, 72), (64, 8), (8, 200)])
def test_split_flag_words_on_ragged_pictures(oracle, torch_cuda, W, H, cnn_arith):
    """Pictures whose last CTU column/row is cut inside a 16x16 block (width, height = 8 mod 16): the depth maps equal the
    oracle's and the 4-byte split-flag words expand back to them (the expansion masks the units outside the picture)."""
    torch = torch_cuda
    w = weights.random_weights(5)
    NF = 2
    lumas = [frames.texture16_luma(W, H, seed=300 + f) for f in range(NF)]
    refs = [_oracle_frame(oracle, w, y, 8, 27) for y in lumas]
    ctx = capi.Context(W, H, 8, w, max_frames=NF)
    dev = torch.device("cuda:0")
    n = ctx.num_ctus
    d8 = torch.from_numpy(np.stack(lumas)).to(dev)
    depth = torch.full((NF, n, 256), 7, dtype=torch.uint8, device=dev)
    flags = torch.zeros((NF, n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()  # the fills above run on torch's stream: order them explicitly before the library call
    ctx.predict_frames_device(d8.data_ptr(), 1, W, W * H, NF, depth.data_ptr(), None, None, qp=27, d_flags=flags.data_ptr())
    expanded = torch.full((NF, n, 256), 9, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()  # the fills above run on torch's stream: order them explicitly before the library call
    ctx.expand_depth_flags_device(flags.data_ptr(), NF, expanded.data_ptr())
    torch.cuda.synchronize()
    for f in range(NF):
        assert np.array_equal(depth[f].cpu().numpy(), refs[f][3]), (W, H, f)
    assert torch.equal(expanded, depth)
    ctx.close()


@pytest.mark.parametrize("W,H", [(416, 244), (128, 68), (64, 4), (192, 130)])
def test_source_hadamard_on_heights_that_cut_an_8x8_block(oracle, torch_cuda, W, H, cnn_arith):
    """Width a multiple of 16 (the fused form's condition) but height NOT a multiple of 8: updateCtuDataISlice (TEncCu.cpp:1324-1343)
    counts whole 8x8 blocks only, so the bottom block row that the picture cuts must not enter the sum.  The library has to route
    these pictures to the stand-alone kernel (fhevc_cnn_can_fuse_hadamard); host-buffer and device-batch entry points."""
    torch = torch_cuda
    w = weights.random_weights(3)
    NF = 2
    lumas = [frames.texture16_luma(W, H, seed=900 + f) for f in range(NF)]
    refs = [_oracle_frame(oracle, w, y, 8, 32) for y in lumas]
    ctx = capi.Context(W, H, 8, w, max_frames=NF)
    for f in range(NF):
        buf, org, stride, depth_ref, _, had_ref = refs[f]
        depth, had = ctx.predict_frame(buf, org, stride)
        assert np.array_equal(had, had_ref), (W, H, f)
        assert np.array_equal(depth, depth_ref), (W, H, f)
    dev = torch.device("cuda:0")
    d8 = torch.from_numpy(np.stack(lumas)).to(dev)
    depth = torch.zeros((NF, ctx.num_ctus, 256), dtype=torch.uint8, device=dev)
    had = torch.zeros((NF, ctx.num_ctus), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.predict_frames_device(d8.data_ptr(), 1, W, W * H, NF, depth.data_ptr(), had.data_ptr(), None, qp=32)
    torch.cuda.synchronize()
    for f in range(NF):
        assert np.array_equal(had[f].cpu().numpy(), refs[f][5]), (W, H, f)
        assert np.array_equal(depth[f].cpu().numpy(), refs[f][3]), (W, H, f)
    ctx.close()


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_source_hadamard_extremes(oracle, bd):
    """updateCtuDataISlice twin on patterns that maximise single coefficients: at 8/10 bit the kernel runs wrapping packed
    16-bit butterflies (every coefficient but DC stays below 2^15, DC is not summed), at 12 bit the 32-bit path."""
    W, H = 256, 128
    top = (1 << bd) - 1
    yy, xx = np.mgrid[0:H, 0:W]
    rng = np.random.default_rng(bd)
    patterns = [np.full((H, W), top), ((xx + yy) & 1) * top, (xx & 1) * top, ((yy >> 2) & 1) * top, ((xx >> 1) & 1) * top,
                (((xx >> 2) ^ (yy >> 1)) & 1) * top, rng.integers(0, top + 1, (H, W)), np.where(rng.random((H, W)) < 0.5, 0, top)]
    w = weights.random_weights(0)
    ctx = capi.Context(W, H, bd, w)
    m = frames.HM_MARGIN
    for k, pat in enumerate(patterns):
        buf = np.zeros((H + 2 * m, W + 2 * m), np.int16)
        buf[m:m + H, m:m + W] = pat.astype(np.int16)
        org, stride = m * (W + 2 * m) + m, W + 2 * m
        exp = np.zeros(ctx.num_ctus, np.int32)
        oracle.fho_frame_src_hadamard(op.ptr(buf.reshape(-1), org), stride, W, H, exp)
        _, had = ctx.predict_frame(buf, org, stride)
        assert np.array_equal(had, exp), (bd, k)
    ctx.close()


def test_soft_decision_ranges(oracle, torch_cuda, cnn_arith):
    """fhevc_predict_frame_range / _device_range against the oracle's fho_depth_range_from_logits."""
    torch = torch_cuda
    w = weights.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fasthevc_amd", "weights", "depthnet_v1.fhw"))
    W, H, QP = 416, 240, 27
    luma = frames.texture16_luma(W, H)
    buf, org, stride, depth_ref, logits_ref, _ = _oracle_frame(oracle, w, luma, 8, QP)
    ctx = capi.Context(W, H, 8, w)
    n = ctx.num_ctus
    differ = 0
    for margin in (0, 3000, 20000, 1 << 30):
        emin, emax = np.zeros((n, 256), np.uint8), np.zeros((n, 256), np.uint8)
        for c in range(n):
            vw, vh = min(64, W - (c % ctx.ctus_x) * 64), min(64, H - (c // ctx.ctus_x) * 64)
            oracle.fho_depth_range_from_logits(np.ascontiguousarray(logits_ref[c]), vw, vh, margin, margin, emin[c], emax[c])
        gmin, gmax = ctx.predict_frame_range(buf, org, stride, qp=QP, margin=margin)
        assert np.array_equal(gmin, emin) and np.array_equal(gmax, emax), margin
        if margin == 0:
            assert np.array_equal(gmin, depth_ref) and np.array_equal(gmax, depth_ref)
        differ += int((gmin != gmax).sum())
    assert differ > 0
    # device entry point: flags follow depth_min
    dev = torch.device("cuda:0")
    d16 = torch.from_numpy(buf).to(dev)
    dmin = torch.zeros((n, 256), dtype=torch.uint8, device=dev)
    dmax = torch.zeros((n, 256), dtype=torch.uint8, device=dev)
    flags = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx._check(ctx.lib.fhevc_predict_frames_device_range(ctx.h, d16.data_ptr() + 2 * org, 2, stride, 0, 1, 0, ctx.ctus_y, QP, 20000, 5000,
                                                         dmin.data_ptr(), dmax.data_ptr(), None, None, flags.data_ptr(), None))
    expanded = torch.zeros_like(dmin)
    torch.cuda.synchronize()  # the fills above run on torch's stream: order them explicitly before the library call
    ctx.expand_depth_flags_device(flags.data_ptr(), 1, expanded.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(expanded, dmin) and bool((dmin <= dmax).all())
    amin, amax = ctx.predict_frame_range(buf, org, stride, qp=QP, margin=20000, margin_stop=5000)  # asymmetric margins
    assert np.array_equal(amin, dmin.cpu().numpy()) and np.array_equal(amax, dmax.cpu().numpy())
    e5min, e5max = np.zeros((n, 256), np.uint8), np.zeros((n, 256), np.uint8)
    for c in range(n):
        vw, vh = min(64, W - (c % ctx.ctus_x) * 64), min(64, H - (c // ctx.ctus_x) * 64)
        oracle.fho_depth_range_from_logits(np.ascontiguousarray(logits_ref[c]), vw, vh, 20000, 5000, e5min[c], e5max[c])
    assert np.array_equal(amin, e5min) and np.array_equal(amax, e5max)
    with pytest.raises(capi.FastHevcError):
        ctx.predict_frame_range(buf, org, stride, qp=QP, margin=-1)
    ctx.close()


def _oracle_first_pass(oracle, buf, org, stride, W, H, bd, qp):
    sl = oracle.fho_lambda_intra(qp, bd) ** 0.5
    cw, ch = frames.ctu_grid(W, H)
    exp = (op.NodeCost * 85)()
    out = np.zeros((cw * ch, 85), capi.NODE_DTYPE)
    for cy in range(ch):
        for cx in range(cw):
            oracle.fho_first_pass_ctu(op.ptr(buf.reshape(-1), org), stride, W, H, cx, cy, bd, sl, exp)
            out[cy * cw + cx] = np.frombuffer(exp, dtype=capi.NODE_DTYPE)
    return out


def _same_nodes(got, exp, what):
    for k in ("satd", "mode", "cost"):
        bad = np.argwhere(got[k] != exp[k])
        assert bad.size == 0, (what, k, bad[:4].tolist(), got[k][tuple(bad[0])], exp[k][tuple(bad[0])])


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_first_pass_vs_oracle(oracle, bd):
    """config 1 geometry (partial last CTU column and row); 8/10-bit take the packed 16-bit path, 12-bit the wide one."""
    luma = frames.texture16_luma(416, 240)
    buf, org, stride = frames.to_pel_plane(luma, bd)
    if bd > 8:  # populate the low bits so that the arithmetic really runs at that depth
        m = org % stride
        rng = np.random.default_rng(bd)
        buf[m:m + 240, m:m + 416] += rng.integers(0, 1 << (bd - 8), (240, 416)).astype(np.int16)
    ctx = capi.Context(416, 240, bd)
    for qp in (22, 37):
        _same_nodes(ctx.intra_first_pass(buf, org, stride, qp=qp), _oracle_first_pass(oracle, buf, org, stride, 416, 240, bd, qp), (bd, qp))
    ctx.close()


def test_first_pass_1080p_and_device_batch(oracle, torch_cuda):
    """config 2 geometry, all 510 CTUs (availability across CTU borders, above-right CTUs, the 56-px last row);
    then the device-batch entry point: 3 pictures, a CTU-row band, uint8 and int16 layouts."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    W, H, NF = 1920, 1080, 3
    lumas = [frames.hetero_luma(W, H), frames.texture16_luma(W, H), np.random.default_rng(9).integers(0, 256, (H, W)).astype(np.uint8)]
    planes = np.stack([frames.to_pel_plane(y, 8)[0] for y in lumas])
    _, org, stride = frames.to_pel_plane(lumas[0], 8)
    exp = [_oracle_first_pass(oracle, planes[f], org, stride, W, H, 8, 32) for f in range(NF)]
    ctx = capi.Context(W, H, 8)
    _same_nodes(ctx.intra_first_pass(planes[0], org, stride, qp=32), exp[0], "host")
    n = ctx.num_ctus
    d16 = torch.from_numpy(planes).to(dev)
    out = torch.zeros((NF, n * 85, 2), dtype=torch.float64, device=dev)  # 16 bytes per node
    torch.cuda.synchronize()  # the fills above run on torch's stream: order them explicitly before the library call
    ctx.intra_first_pass_device(d16.data_ptr() + 2 * org, 2, stride, planes.shape[1] * planes.shape[2], NF, out.data_ptr(), qp=32)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(capi.NODE_DTYPE).reshape(NF, n, 85)
    for f in range(NF):
        _same_nodes(got[f], exp[f], ("int16 batch", f))
    d8 = torch.from_numpy(np.stack(lumas)).to(dev)
    rb, re = 5, 11
    band = torch.zeros((NF, (re - rb) * ctx.ctus_x * 85, 2), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()  # the fills above run on torch's stream: order them explicitly before the library call
    ctx.intra_first_pass_device(d8.data_ptr(), 1, W, W * H, NF, band.data_ptr(), rows=(rb, re), qp=32)
    torch.cuda.synchronize()
    gb = band.cpu().numpy().view(capi.NODE_DTYPE).reshape(NF, (re - rb) * ctx.ctus_x, 85)
    for f in range(NF):
        _same_nodes(gb[f], exp[f][rb * ctx.ctus_x:re * ctx.ctus_x], ("uint8 band", f))
    ctx.close()


def test_errors_are_status_codes():
    ctx = capi.Context(416, 240, 8)
    buf, org, stride = frames.to_pel_plane(frames.texture16_luma(416, 240), 8)
    with pytest.raises(capi.FastHevcError) as e:
        ctx.predict_frame(buf, org, stride)          # weights not set
    assert e.value.code == capi.E_STATE
    with pytest.raises(capi.FastHevcError) as e:
        ctx.set_weights(b"nope")
    assert e.value.code == capi.E_WEIGHTS
    # device entry points: bad bands, layouts and depths are refused with a status code, nothing is launched
    for call in (lambda: ctx.preanalyze_frames_device(1, 2, stride, 0, 1, 1, 5),                   # max_aq_depth > 4
                 lambda: ctx.preanalyze_frames_device(1, 2, stride, 0, 1, 1, 3, rows=(3, 2)),      # empty/inverted band
                 lambda: ctx.preanalyze_frames_device(1, 3, stride, 0, 1, 1, 3),                   # sample_bytes
                 lambda: ctx.intra_first_pass_device(1, 2, 100, 0, 1, 1),                          # stride < width
                 lambda: ctx.intra_first_pass_device(1, 2, stride, 0, 1, 1, rows=(0, 9)),          # band past the picture
                 lambda: ctx.intra_first_pass_device(1, 2, stride, 0, 1, 1, qp=52),
                 lambda: ctx.intra_first_pass_device(0, 2, stride, 0, 1, 1)):                      # NULL plane
        with pytest.raises(capi.FastHevcError) as e:
            call()
        assert e.value.code == capi.E_INVALID
    ctx.close()
    ctx10 = capi.Context(416, 240, 10)
    with pytest.raises(capi.FastHevcError) as e:   # uint8 samples only make sense at bit depth 8
        ctx10.intra_first_pass_device(1, 1, 416, 0, 1, 1)
    assert e.value.code == capi.E_INVALID
    ctx10.close()
    odd = capi.Context(420, 244, 8)                # not a multiple of 8: HM pads such pictures; the AQ twin refuses them
    with pytest.raises(capi.FastHevcError) as e:
        odd.preanalyze_frames_device(1, 2, 600, 0, 1, 1, 3)
    assert e.value.code == capi.E_INVALID
    odd.close()
    with pytest.raises(capi.FastHevcError):
        capi.Context(416, 240, 7)
    # arithmetic form of the classifier: i8 by default, f16 on request, anything else is a status code
    c2 = capi.Context(416, 240, 8)
    assert c2.cnn_arith == "i8"
    c2.set_cnn_arith("f16")
    assert c2.cnn_arith == "f16"
    assert c2.lib.fhevc_set_cnn_arith(c2.h, 4) == capi.E_INVALID and c2.cnn_arith == "f16"
    c2.close()


def _oracle_first_pass_all(oracle, buf, org, stride, W, H, bd, qp, ctus):
    """every (node, mode) pair of the given CTUs through fho_first_pass_node: satd [len(ctus), 85, 35] (-1 = node outside)"""
    sl = oracle.fho_lambda_intra(qp, bd) ** 0.5
    cw, _ = frames.ctu_grid(W, H)
    out = np.full((len(ctus), 85, 35), -1, np.int64)
    best = op.NodeCost()
    sat = np.zeros(35, np.uint32)
    for i, c in enumerate(ctus):
        idx = 0
        for lvl in range(4):
            n, cnt = 64 >> lvl, 1 << lvl
            for by in range(cnt):
                for bx in range(cnt):
                    x0, y0 = (c % cw) * 64 + bx * n, (c // cw) * 64 + by * n
                    if x0 + n <= W and y0 + n <= H:
                        oracle.fho_first_pass_node(op.ptr(buf.reshape(-1), org), stride, W, H, x0, y0, n, bd, sl, C.byref(best), C.c_void_p(sat.ctypes.data))
                        out[i, idx] = sat
                    idx += 1
    return out, sl


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_first_pass_all_35_modes_per_node(oracle, bd):
    """A7/A8 on the GPU beyond the arg-min: the SATD and the cost of EVERY mode of every node (fhevc_intra_first_pass_all),
    i.e. reference lines, smoothing decision, all 35 predictors and the Hadamard of modes that never win."""
    W, H = 416, 240
    luma = frames.hetero_luma(W, H, seed=31 + bd)
    buf, org, stride = frames.to_pel_plane(luma, bd)
    if bd > 8:
        m = org % stride
        buf[m:m + H, m:m + W] += np.random.default_rng(bd).integers(0, 1 << (bd - 8), (H, W)).astype(np.int16)
    ctx = capi.Context(W, H, bd)
    qp = 27
    best, allm = ctx.intra_first_pass_all(buf, org, stride, qp=qp)
    ctus = list(range(ctx.num_ctus))
    exp, sl = _oracle_first_pass_all(oracle, buf, org, stride, W, H, bd, qp, ctus)
    valid = exp[:, :, 0] >= 0
    assert np.array_equal(allm["satd"][valid].astype(np.int64), exp[valid])
    assert (allm["mode"][valid] == np.arange(35)).all() and (allm["mode"][~valid] == 255).all() and (allm["satd"][~valid] == 0xFFFFFFFF).all()
    bits = np.full(35, 6.0); bits[0] = 2.0; bits[1] = bits[26] = 3.0   # fho_first_pass_node's mode-bit model
    assert np.array_equal(allm["cost"][valid], exp[valid].astype(np.float64) + bits * sl)
    # the winner is the first minimum of those costs
    assert np.array_equal(best["mode"][valid], allm["cost"][valid].argmin(axis=1).astype(np.uint32))
    ctx.close()


def test_first_pass_candidate_lists_vs_oracle(oracle):
    """fhevc_intra_first_pass_candidates (what HM's estIntraPredLumaQT consumes under FHEVC_FIRST_PASS=1): per node the eight modes of smallest
    cost, best first, ties to the earlier mode -- against the oracle's lists, ragged picture, 8 and 10 bit; the head of every list is the
    first pass's best mode."""
    import ctypes as C
    oracle.fho_first_pass_candidates_ctu.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]
    W, H = 416, 240
    for bd, qp in ((8, 32), (10, 22)):
        y = frames.hetero_luma(W, H, seed=40 + bd)
        buf, org, stride = frames.to_pel_plane(y, bd)
        ctx = capi.Context(W, H, bd)
        got = ctx.intra_first_pass_candidates(buf, org, stride, qp=qp, num_candidates=8)
        best = ctx.intra_first_pass(buf, org, stride, qp=qp)
        cw, n = 7, 28
        exp = np.zeros((n, 85, 8), np.uint8)
        sl = oracle.fho_lambda_intra(qp, bd) ** 0.5
        for c in range(n):
            oracle.fho_first_pass_candidates_ctu(C.c_void_p(buf.reshape(-1).ctypes.data + 2 * org), stride, W, H, c % cw, c // cw, bd, C.c_double(sl), 8, exp[c].ctypes.data)
        assert np.array_equal(got, exp)
        inside = best["mode"] != 255
        assert np.array_equal(got[..., 0][inside], best["mode"][inside].astype(np.uint8)) and (got[~inside] == 255).all()
        ctx.close()

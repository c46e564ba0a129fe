"""The reference's Bayesian-optimisation network family on the MI355X against the CPU oracle (fho_cnn_ctu_family), bit for bit: the NetworkDepth-1
member 32 / 64 / 128 (Optimize...Example.m:103-106, 233-259) in its fused kernel (k_cnn_family.inc), the 16 / 32 / 64 widths through the same code
path (must reproduce the tuned base-network kernel's output), and every other member -- NetworkDepth 2 and 3, odd widths -- through the
layer-by-layer path (k_cnn_layers.inc)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle_py as op
from fasthevc_amd import capi, frames, weights

pytestmark = pytest.mark.gpu


def _oracle_family(oracle, fam, luma, bd, qp):
    H, W = luma.shape
    buf, org, stride = frames.to_pel_plane(luma, bd)
    n = ((W + 63) // 64) * ((H + 63) // 64)
    depth, logits = np.zeros(n * 256, np.uint8), np.zeros(n * 42, np.int32)
    f = op.family_from_arrays(fam)
    oracle.fho_predict_frame_family(C.byref(f), op.ptr(buf.reshape(-1), org), stride, W, H, bd, qp, depth.ctypes.data, logits.ctypes.data)
    had = np.zeros(n, np.int32)
    oracle.fho_frame_src_hadamard(op.ptr(buf.reshape(-1), org), stride, W, H, had)
    return buf, org, stride, depth.reshape(n, 256), logits.reshape(n, 42), had


@pytest.mark.parametrize("W,H,bd,seed", [(416, 240, 8, 0), (416, 240, 10, 1), (200, 136, 8, 2), (832, 480, 8, 3)])
def test_family_32_64_128_equals_the_oracle(oracle, W, H, bd, seed):
    fam = weights.random_family((32, 64, 128), 1, seed=seed)
    luma = frames.texture16_luma(W, H, seed=40 + seed)
    buf, org, stride, depth_ref, logits_ref, had_ref = _oracle_family(oracle, fam, luma, bd, 22 + 5 * seed)
    ctx = capi.Context(W, H, bd, fam)
    depth, had = ctx.predict_frame(buf, org, stride, qp=22 + 5 * seed)
    bad = np.nonzero((depth != depth_ref).any(axis=1))[0]
    assert bad.size == 0, f"CTUs with a differing depth map: {bad[:10]}"
    assert np.array_equal(had, had_ref)   # the stand-alone source-Hadamard kernel runs beside the family kernel
    assert len(np.unique(depth)) >= 2
    ctx.close()


def test_family_logits_soft_ranges_and_a_device_batch(oracle):
    import torch
    assert torch.cuda.is_available()
    W, H, NF = 416, 240, 3
    fam = weights.random_family((32, 64, 128), 1, seed=9)
    lumas = [frames.hetero_luma(W, H, seed=60 + f) for f in range(NF)]
    refs = [_oracle_family(oracle, fam, y, 8, 32) for y in lumas]
    ctx = capi.Context(W, H, 8, fam, max_frames=NF)
    dev = torch.device("cuda:0")
    d8 = torch.from_numpy(np.stack(lumas)).to(dev)
    depth = torch.zeros((NF, ctx.num_ctus, 256), dtype=torch.uint8, device=dev)
    logits = torch.zeros((NF, ctx.num_ctus, 42), dtype=torch.int32, device=dev)
    flags = torch.zeros((NF, ctx.num_ctus), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.predict_frames_device(d8.data_ptr(), 1, W, W * H, NF, depth.data_ptr(), None, logits.data_ptr(), qp=32, d_flags=flags.data_ptr())
    expanded = torch.zeros_like(depth)
    ctx.expand_depth_flags_device(flags.data_ptr(), NF, expanded.data_ptr())
    torch.cuda.synchronize()
    for f in range(NF):
        assert np.array_equal(logits[f].cpu().numpy(), refs[f][4]), f
        assert np.array_equal(depth[f].cpu().numpy(), refs[f][3]), f
    assert torch.equal(expanded, depth)
    buf, org, stride = refs[0][0], refs[0][1], refs[0][2]
    dmin, dmax = ctx.predict_frame_range(buf, org, stride, qp=32, margin=30000, margin_stop=10000)
    n = ctx.num_ctus
    emin, emax = np.zeros((n, 256), np.uint8), np.zeros((n, 256), np.uint8)
    cw = (W + 63) // 64
    for c in range(n):
        vw, vh = min(64, W - (c % cw) * 64), min(64, H - (c // cw) * 64)
        oracle.fho_depth_range_from_logits(np.ascontiguousarray(refs[0][4][c]), vw, vh, 30000, 10000, emin[c], emax[c])
    assert np.array_equal(dmin, emin) and np.array_equal(dmax, emax)
    ctx.close()


def test_family_code_path_reproduces_the_base_network(oracle):
    """widths 16 / 32 / 64 through the family kernel == the tuned base-network kernels (both arithmetic forms) == the base oracle"""
    W, H = 416, 240
    base = weights.random_weights(12)
    luma = frames.texture16_luma(W, H, seed=5)
    buf, org, stride = frames.to_pel_plane(luma, 8)
    ref_ctx = capi.Context(W, H, 8, base)
    d_ref, _ = ref_ctx.predict_frame(buf, org, stride, qp=27)
    ref_ctx.close()
    ctx = capi.Context(W, H, 8, weights.family_from_base(base))
    d, _ = ctx.predict_frame(buf, org, stride, qp=27)
    assert np.array_equal(d, d_ref)
    ctx.close()


@pytest.mark.parametrize("widths,depth,W,H,bd,seed,fuse", [((23, 46, 92), 2, 416, 240, 8, 0, True), ((18, 36, 72), 3, 416, 240, 8, 1, True), ((23, 46, 92), 2, 200, 136, 10, 2, False),
                                                           ((32, 64, 128), 1, 416, 240, 8, 3, True), ((20, 44, 100), 2, 256, 192, 8, 4, True), ((12, 24, 48), 3, 320, 256, 12, 5, True),
                                                           ((40, 60, 100), 2, 416, 240, 8, 6, True), ((18, 36, 72), 3, 256, 192, 8, 7, False), ((70, 80, 96), 2, 200, 136, 8, 8, True),
                                                           # every LDS image of the layer kernel: 96 / 128 channels in (three- and four-chunk K, pooled and not), and the
                                                           # single-buffered staging ("1buf": FHEVC_LAYERS_NO_DBUF) next to the default double-buffered LDS-direct one
                                                           ((96, 128, 128), 2, 256, 192, 8, 9, True), ((100, 70, 96), 3, 200, 136, 8, 10, True), ((23, 46, 92), 2, 256, 192, 8, 11, "1buf"),
                                                           ((64, 128, 64), 3, 200, 136, 10, 12, "1buf")])
def test_every_family_member_through_the_layered_path(oracle, monkeypatch, widths, depth, W, H, bd, seed, fuse):
    """The members without a fused kernel -- NetworkDepth 2 (23 / 46 / 92) and 3 (18 / 36 / 72) of the reference's family, and odd widths -- run layer by
    layer through HBM (k_cnn_layers.inc): depth maps, logits, soft ranges and split-flag words against the oracle's plain loops, ragged pictures,
    8 / 10 / 12 bit.  FHEVC_FAMILY_LAYERS sends a fused member (32 / 64 / 128) through the same generic path."""
    import torch
    monkeypatch.setenv("FHEVC_FAMILY_LAYERS", "1")
    if not fuse:   # the first convolution as its own launch instead of inside the second one's LDS staging
        monkeypatch.setenv("FHEVC_LAYERS_NO_FUSE", "1")
    if fuse == "1buf":
        monkeypatch.setenv("FHEVC_LAYERS_NO_DBUF", "1")
    fam = weights.random_family(widths, depth, seed=seed)
    qp = 22 + 5 * seed
    lumas = [frames.hetero_luma(W, H, seed=80 + seed), frames.texture16_luma(W, H, seed=90 + seed)]
    refs = [_oracle_family(oracle, fam, y, bd, qp) for y in lumas]
    ctx = capi.Context(W, H, bd, fam, max_frames=2)
    for y, (buf, org, stride, depth_ref, logits_ref, had_ref) in zip(lumas, refs):
        d, had = ctx.predict_frame(buf, org, stride, qp=qp)
        bad = np.nonzero((d != depth_ref).any(axis=1))[0]
        assert bad.size == 0, f"CTUs with a differing depth map: {bad[:10]}"
        assert np.array_equal(had, had_ref)
    # device batch of both pictures: logits, flag words, soft ranges
    dev = torch.device("cuda:0")
    planes = np.stack([r[0] for r in refs])
    d16 = torch.from_numpy(planes).to(dev)
    org, stride = refs[0][1], refs[0][2]
    n = ctx.num_ctus
    depth = torch.zeros((2, n, 256), dtype=torch.uint8, device=dev)
    logits = torch.zeros((2, n, 42), dtype=torch.int32, device=dev)
    flags = torch.zeros((2, n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.predict_frames_device(d16.data_ptr() + 2 * org, 2, stride, planes.shape[1] * planes.shape[2], 2, depth.data_ptr(), None, logits.data_ptr(), qp=qp, d_flags=flags.data_ptr())
    expanded = torch.zeros_like(depth)
    ctx.expand_depth_flags_device(flags.data_ptr(), 2, expanded.data_ptr())
    torch.cuda.synchronize()
    for f in range(2):
        assert np.array_equal(logits[f].cpu().numpy(), refs[f][4]), f
        assert np.array_equal(depth[f].cpu().numpy(), refs[f][3]), f
        assert np.array_equal(expanded[f].cpu().numpy(), refs[f][3]), f
    ctx.close()
    # a batch larger than the chunk of CTUs whose activations live in HBM at once (max_frames = 1 -> 64 CTUs): several passes over the layers
    small = capi.Context(W, H, bd, fam, max_frames=1)
    order = [0, 1, 0, 1, 1, 0, 0][:max(3, 200 // n + 2)]
    d16b = torch.from_numpy(np.stack([refs[i][0] for i in order])).to(dev)
    depthb = torch.zeros((len(order), n, 256), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    small.predict_frames_device(d16b.data_ptr() + 2 * org, 2, stride, planes.shape[1] * planes.shape[2], len(order), depthb.data_ptr(), None, None, qp=qp)
    torch.cuda.synchronize()
    for k, i in enumerate(order):
        assert np.array_equal(depthb[k].cpu().numpy(), refs[i][3]), (k, i)
    small.close()


def test_family_members_the_build_cannot_run_are_refused_with_a_status():
    fam = weights.random_family((23, 46, 94), 2, seed=1)   # last width not a multiple of 4: the heads' four-byte dot products
    with pytest.raises(capi.FastHevcError) as e:
        capi.Context(416, 240, 8, fam)
    assert e.value.code == capi.E_WEIGHTS


@pytest.mark.parametrize("blob,bd,qp,requant", [("depthnet_family_d1.fhw", 8, 32, None), ("depthnet_family_d2.fhw", 8, 27, None), ("depthnet_family_d2.fhw", 10, 37, None),
                                                ("depthnet_family_d2.fhw", 8, 27, "general"), ("depthnet_family_d3.fhw", 8, 32, None)])
def test_the_shipped_family_blobs_equal_the_oracle(oracle, monkeypatch, blob, bd, qp, requant):
    """The TRAINED members as shipped (fasthevc_amd/weights/): their weight statistics differ from the random members of the other tests (small sums of
    |w| per filter, shifts 6 / 8) -- depth maps and logits of two pictures through the library's default dispatch (fused kernel for 32 / 64 / 128 x 1,
    the fused two-convolution kernel for the x 2 member -- its short-requant instantiation, which this blob qualifies for, and with FHEVC_D2_REQUANT=general
    its general one --, the layer path for the x 3 member) against the oracle, bit for bit."""
    import os
    import torch
    if requant:
        monkeypatch.setenv("FHEVC_D2_REQUANT", requant)
    else:
        monkeypatch.delenv("FHEVC_D2_REQUANT", raising=False)
    fam = weights.load_any(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fasthevc_amd", "weights", blob))
    W, H = 416, 240
    lumas = [frames.hetero_luma(W, H, seed=7), frames.texture16_luma(W, H, seed=8)]
    refs = [_oracle_family(oracle, fam, y, bd, qp) for y in lumas]
    ctx = capi.Context(W, H, bd, fam, max_frames=2)
    dev = torch.device("cuda:0")
    planes = np.stack([r[0] for r in refs])
    d16 = torch.from_numpy(planes).to(dev)
    org, stride, n = refs[0][1], refs[0][2], ctx.num_ctus
    depth = torch.zeros((2, n, 256), dtype=torch.uint8, device=dev)
    logits = torch.zeros((2, n, 42), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.predict_frames_device(d16.data_ptr() + 2 * org, 2, stride, planes.shape[1] * planes.shape[2], 2, depth.data_ptr(), None, logits.data_ptr(), qp=qp)
    torch.cuda.synchronize()
    for f in range(2):
        assert np.array_equal(logits[f].cpu().numpy(), refs[f][4]), f
        assert np.array_equal(depth[f].cpu().numpy(), refs[f][3]), f
    assert len(np.unique(depth.cpu().numpy())) >= 3
    ctx.close()


def test_a_rejected_blob_leaves_the_weights_in_use_untouched(oracle):
    """fhevc_set_weights validates and builds before it flips the dispatch: a blob that is rejected -- of the same kind or of another kind than the
    one in use -- must leave the context predicting exactly as before (round-3 advice: a bad FHW3 after a layered member used to leave a fused-family
    dispatch with null weight pointers)."""
    W, H, QP = 200, 136, 30
    luma = frames.hetero_luma(W, H, seed=77)
    buf, org, stride = frames.to_pel_plane(luma, 8)
    base = weights.random_weights(21)
    lay = weights.random_family((23, 46, 92), 2, seed=5)
    fus = weights.random_family((32, 64, 128), 1, seed=6)

    def fresh(w):
        c = capi.Context(W, H, 8, w)
        d, _ = c.predict_frame(buf, org, stride, qp=QP)
        c.close()
        return d

    want = {"base": fresh(base), "lay": fresh(lay), "fus": fresh(fus)}

    def broken(w, family):
        b = bytearray(weights.pack_family(w) if family else weights.pack(w))
        off = 24 if family else 8          # the first requant shift of either layout
        b[off:off + 4] = np.int32(15).tobytes()
        return bytes(b)

    bad = {"base": broken(base, False), "lay": broken(lay, True), "fus": broken(fus, True)}
    good = {"base": base, "lay": lay, "fus": fus}
    ctx = capi.Context(W, H, 8, base)
    for cur in ("base", "lay", "fus", "lay", "base", "fus", "base"):
        ctx.set_weights(good[cur])
        d, _ = ctx.predict_frame(buf, org, stride, qp=QP)
        assert np.array_equal(d, want[cur]), cur
        for other in ("base", "lay", "fus"):
            with pytest.raises(capi.FastHevcError):
                ctx.set_weights(bad[other])
            with pytest.raises(capi.FastHevcError):
                ctx.set_weights(bad[other][:-3])
            d, _ = ctx.predict_frame(buf, org, stride, qp=QP)
            assert np.array_equal(d, want[cur]), (cur, other)
    ctx.close()


@pytest.mark.parametrize("widths,W,H,bd,seed", [((23, 46, 92), 416, 240, 8, 0), ((23, 46, 92), 200, 136, 10, 1), ((20, 44, 88), 256, 192, 12, 2), ((32, 64, 96), 320, 256, 8, 3),
                                                ((1, 33, 68), 200, 136, 8, 4), ((23, 46, 92), 832, 480, 8, 5)])
def test_two_convolutions_per_block_as_one_kernel(oracle, widths, W, H, bd, seed):
    """The NetworkDepth-2 members at padded widths 32 / 64 / 96 -- the reference's 23 / 46 / 92 x 2 and its neighbours -- through the library's default
    dispatch = k_cnn_d2.inc (a CTU's activations never leave LDS): depth maps, logits, soft ranges and flag words against the oracle's plain loops; ragged
    pictures (the 32-wide last column, 48- and 8-row bottom edges), 8 / 10 / 12 bit; the same context's layer-by-layer twin must agree as well."""
    import torch
    fam = weights.random_family(widths, 2, seed=seed)
    if seed % 2:
        fam["shift"] = np.random.default_rng(seed).integers(4, 11, size=(3, 3)).astype(np.int32)
    qp = 22 + 5 * (seed % 4)
    lumas = [frames.hetero_luma(W, H, seed=180 + seed), frames.texture16_luma(W, H, seed=190 + seed), frames.fractal_luma(W + 8, H + 8, seed=seed)[:H, :W].copy()]
    refs = [_oracle_family(oracle, fam, y, bd, qp) for y in lumas]
    ctx = capi.Context(W, H, bd, fam, max_frames=3)
    st = ctx.stats()
    for y, (buf, org, stride, depth_ref, logits_ref, had_ref) in zip(lumas, refs):
        d, had = ctx.predict_frame(buf, org, stride, qp=qp)
        bad = np.nonzero((d != depth_ref).any(axis=1))[0]
        assert bad.size == 0, f"CTUs with a differing depth map: {bad[:10]}"
        assert np.array_equal(had, had_ref)
    # one launch per picture (+ the stand-alone source Hadamard): the layer path would need nine
    assert ctx.stats()["kernels_launched"] - st["kernels_launched"] == 2 * len(lumas)
    dev = torch.device("cuda:0")
    planes = np.stack([r[0] for r in refs])
    d16 = torch.from_numpy(planes).to(dev)
    org, stride, n = refs[0][1], refs[0][2], ctx.num_ctus
    depth = torch.zeros((3, n, 256), dtype=torch.uint8, device=dev)
    dmax = torch.zeros((3, n, 256), dtype=torch.uint8, device=dev)
    logits = torch.zeros((3, n, 42), dtype=torch.int32, device=dev)
    flags = torch.zeros((3, n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.predict_frames_device(d16.data_ptr() + 2 * org, 2, stride, planes.shape[1] * planes.shape[2], 3, depth.data_ptr(), None, logits.data_ptr(), qp=qp, d_flags=flags.data_ptr())
    expanded = torch.zeros_like(depth)
    ctx.expand_depth_flags_device(flags.data_ptr(), 3, expanded.data_ptr())
    torch.cuda.synchronize()
    for f in range(3):
        assert np.array_equal(logits[f].cpu().numpy(), refs[f][4]), f
        assert np.array_equal(depth[f].cpu().numpy(), refs[f][3]), f
        assert np.array_equal(expanded[f].cpu().numpy(), refs[f][3]), f
    if bd == 8:   # tightly packed uint8 planes: the other sample layout of the LDS-DMA staging (16 bytes per lane instead of 32)
        d8 = torch.from_numpy(np.stack(lumas)).to(dev)
        depth8 = torch.zeros((3, n, 256), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ctx.predict_frames_device(d8.data_ptr(), 1, W, W * H, 3, depth8.data_ptr(), None, None, qp=qp)
        torch.cuda.synchronize()
        for f in range(3):
            assert np.array_equal(depth8[f].cpu().numpy(), refs[f][3]), f
    ctx.close()

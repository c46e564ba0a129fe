"""GPU parity at BASELINE.json's full sizes: config 3 (3840x2160, CTU rows sharded in 8 bands), config 5 (10-bit
1080p: 16-bit pel planes) and the bench workload (a 64-frame 1080p GOP resident in HBM).  Bit-exact against the CPU
oracle where it finishes in seconds, size-independent properties (band assembly, flag-word round trip, determinism,
batch == single-picture calls) over everything else."""
import os

import numpy as np
import pytest

from oracle import oracle_py as op
from fasthevc_amd import capi, frames, weights
from fasthevc_amd import gather as bands

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIPPED = os.path.join(ROOT, "fasthevc_amd", "weights", "depthnet_v2.fhw")


def _oracle(oracle, w, buf, org, stride, W, H, bd, qp):
    cw, ch = frames.ctu_grid(W, H)
    depth = np.zeros(cw * ch * 256, np.uint8)
    oracle.fho_predict_frame(op.weights_from_arrays(w), op.ptr(buf.reshape(-1), org), stride, W, H, bd, qp, depth, None)
    had = np.zeros(cw * ch, np.int32)
    oracle.fho_frame_src_hadamard(op.ptr(buf.reshape(-1), org), stride, W, H, had)
    return depth.reshape(-1, 256), had


def test_config3_4k_eight_bands(oracle, cnn_arith):
    """3840x2160: 60x34 CTUs, last row 48 px tall.  The 8 CTU-row bands of an 8-rank node, launched one after the
    other on this GPU, must assemble (through the 4-byte flag words the ranks all-gather) into the oracle's map."""
    import torch
    dev = torch.device("cuda:0")
    W, H, QP = 3840, 2160, 32
    w = weights.load(SHIPPED)
    buf, org, stride = frames.to_pel_plane(frames.hetero_luma(W, H), 8)
    depth_ref, had_ref = _oracle(oracle, w, buf, org, stride, W, H, 8, QP)
    ctx = capi.Context(W, H, 8, w)
    assert (ctx.ctus_x, ctx.ctus_y) == (60, 34)
    d16 = torch.from_numpy(buf).to(dev)
    flags = torch.zeros(ctx.num_ctus, dtype=torch.int32, device=dev)
    had = torch.zeros(ctx.num_ctus, dtype=torch.int32, device=dev)
    covered = 0
    for rank in range(8):
        rb, re = bands.band(ctx.ctus_y, rank, 8)
        n = (re - rb) * ctx.ctus_x
        part = torch.full((max(n, 1), 256), 7, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()  # the fills above run on torch's stream: order them explicitly before the library call
        ctx.predict_frames_device(d16.data_ptr() + 2 * org, 2, stride, 0, 1, part.data_ptr(),
                                  had[rb * ctx.ctus_x:].data_ptr() if n else None, None, rows=(rb, re), qp=QP,
                                  d_flags=flags[rb * ctx.ctus_x:].data_ptr() if n else None)
        torch.cuda.synchronize()
        assert np.array_equal(part[:n].cpu().numpy(), depth_ref[rb * ctx.ctus_x:re * ctx.ctus_x]), rank
        covered += n
    assert covered == ctx.num_ctus
    full = torch.zeros((ctx.num_ctus, 256), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()  # the fills above run on torch's stream: order them explicitly before the library call
    ctx.expand_depth_flags_device(flags.data_ptr(), 1, full.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(full.cpu().numpy(), depth_ref)
    assert np.array_equal(had.cpu().numpy(), had_ref)
    assert set(np.unique(depth_ref)) == {0, 1, 2, 3}
    ctx.close()


def test_config5_10bit_1080p(oracle, golden, cnn_arith):
    """encoder_intra_main10 geometry: 8-bit content at InternalBitDepth 10 (samples << 2) in 16-bit pel planes."""
    W, H, QP = 1920, 1080, 27
    w = weights.load(SHIPPED)
    buf, org, stride = frames.to_pel_plane(frames.hetero_luma(W, H), 10)
    assert int(buf.max()) > 255
    depth_ref, had_ref = _oracle(oracle, w, buf, org, stride, W, H, 10, QP)
    ctx = capi.Context(W, H, 10, w)
    depth, had = ctx.predict_frame(buf, org, stride, qp=QP)
    assert np.array_equal(depth, depth_ref) and np.array_equal(had, had_ref)
    # native 10-bit content (low bits populated): the classifier sees samples rounded to 8 bits, Hadamard sees all 10
    rng = np.random.default_rng(11)
    buf2 = buf.copy()
    m = org % stride  # HM margin: origin = m * stride + m
    inner = buf2[m:m + H, m:m + W]
    inner[:] = np.clip(inner + rng.integers(0, 4, (H, W)), 0, 1023).astype(np.int16)
    depth_ref2, had_ref2 = _oracle(oracle, w, buf2, org, stride, W, H, 10, QP)
    depth2, had2 = ctx.predict_frame(buf2, org, stride, qp=QP)
    assert np.array_equal(depth2, depth_ref2) and np.array_equal(had2, had_ref2)
    assert not np.array_equal(had2, had_ref)
    ctx.close()


def test_bench_gop_64_frames_properties(oracle, cnn_arith):
    """The bench.py workload: 64 panned 1080p frames as HM-layout int16 planes in HBM, one launch."""
    import torch
    dev = torch.device("cuda:0")
    W, H, NF, QP = 1920, 1080, 64, 32
    w = weights.load(SHIPPED)
    base = frames.hetero_luma(W, H)
    lumas = [np.roll(base, 3 * f, axis=1) for f in range(NF)]
    planes = np.stack([frames.to_pel_plane(y, 8)[0] for y in lumas])
    _, org, stride = frames.to_pel_plane(base, 8)
    ctx = capi.Context(W, H, 8, w, max_frames=NF)
    n = ctx.num_ctus
    d16 = torch.from_numpy(planes).to(dev)
    fs = planes.shape[1] * planes.shape[2]

    def run():
        depth = torch.zeros((NF, n, 256), dtype=torch.uint8, device=dev)
        had = torch.zeros((NF, n), dtype=torch.int32, device=dev)
        flags = torch.zeros((NF, n), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()  # the fills above run on torch's stream: order them explicitly before the library call
        ctx.predict_frames_device(d16.data_ptr() + 2 * org, 2, stride, fs, NF, depth.data_ptr(), had.data_ptr(), None,
                                  qp=QP, d_flags=flags.data_ptr())
        torch.cuda.synchronize()
        return depth, had, flags

    depth, had, flags = run()
    depth_b, had_b, flags_b = run()
    assert torch.equal(depth, depth_b) and torch.equal(had, had_b) and torch.equal(flags, flags_b)  # deterministic
    expanded = torch.zeros_like(depth)
    torch.cuda.synchronize()  # the fills above run on torch's stream: order them explicitly before the library call
    ctx.expand_depth_flags_device(flags.data_ptr(), NF, expanded.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(expanded, depth)  # flag words carry the whole map, all 32 640 CTUs
    dh, hh = depth.cpu().numpy(), had.cpu().numpy()
    for f in (0, 31, 63):  # bit-exact against the oracle on sampled frames
        dref, href = _oracle(oracle, w, planes[f], org, stride, W, H, 8, QP)
        assert np.array_equal(dh[f], dref) and np.array_equal(hh[f], href), f
    for f in (1, 40):  # batch == single-picture host-buffer entry point
        d1, h1 = ctx.predict_frame(planes[f], org, stride, qp=QP)
        assert np.array_equal(dh[f], d1) and np.array_equal(hh[f], h1), f
    # panning permutes columns: the source Hadamard total over interior CTU rows changes, the map is not constant
    assert len({int(hh[f].sum()) for f in range(NF)}) > 1
    assert len(np.unique(dh)) == 4
    ctx.close()


FAMILY_BLOBS = ["depthnet_family_d1.fhw", "depthnet_family_d2.fhw", "depthnet_family_d3.fhw"]


def _oracle_family_ctus(oracle, fam, plane, org, stride, W, H, bd, qp, ctus):
    """the oracle's logits and depth map of the listed CTUs of one picture (per-CTU calls: a whole 1080p picture of a two-convolution member
    would take the plain loops minutes)"""
    import ctypes as C
    f = op.family_from_arrays(fam)
    cw = (W + 63) // 64
    out = {}
    for c in ctus:
        cx, cy = c % cw, c // cw
        ctu = np.zeros(64 * 64, np.int8)
        oracle.fho_load_ctu(op.ptr(plane.reshape(-1), org), stride, W, H, cx, cy, bd, ctu)
        logits = np.zeros(42, np.int32)
        oracle.fho_cnn_ctu_family(C.byref(f), ctu.ctypes.data, qp, logits.ctypes.data)
        depth = np.zeros(256, np.uint8)
        oracle.fho_depth_from_logits(logits, min(64, W - cx * 64), min(64, H - cy * 64), depth)
        out[c] = (logits, depth)
    return out


@pytest.mark.parametrize("blob", FAMILY_BLOBS)
def test_family_blobs_on_the_bench_gop(oracle, blob):
    """The three shipped members of the reference's network family at the size bench.py runs them: 64 panned 1080p pictures as HM-layout
    planes in HBM (32 640 CTUs: four 8 192-CTU chunks on the layer path).  Sampled CTUs of sampled pictures -- corners, the 56-row bottom
    edge, interior, both ends of every chunk -- against the oracle's logits and maps; batch == single picture; flag words carry the map."""
    import torch
    dev = torch.device("cuda:0")
    W, H, NF, QP = 1920, 1080, 64, 32
    fam = weights.load_any(os.path.join(ROOT, "fasthevc_amd", "weights", blob))
    base = frames.hetero_luma(W, H)
    planes = np.stack([frames.to_pel_plane(np.roll(base, 3 * f, axis=1), 8)[0] for f in range(NF)])
    _, org, stride = frames.to_pel_plane(base, 8)
    ctx = capi.Context(W, H, 8, fam, max_frames=NF)
    n = ctx.num_ctus
    d16 = torch.from_numpy(planes).to(dev)
    fs = planes.shape[1] * planes.shape[2]
    depth = torch.zeros((NF, n, 256), dtype=torch.uint8, device=dev)
    logits = torch.zeros((NF, n, 42), dtype=torch.int32, device=dev)
    flags = torch.zeros((NF, n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.predict_frames_device(d16.data_ptr() + 2 * org, 2, stride, fs, NF, depth.data_ptr(), None, logits.data_ptr(), qp=QP, d_flags=flags.data_ptr())
    expanded = torch.zeros_like(depth)
    ctx.expand_depth_flags_device(flags.data_ptr(), NF, expanded.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(expanded, depth)
    dh, lh = depth.cpu().numpy(), logits.cpu().numpy()
    # 8 192-CTU chunks of the layer path end inside pictures 16, 32, 48: sample around those seams as well
    picks = {0: [0, 29, 255, 480, 509], 16: [30, 31, 32, 33, 300], 31: [7, 200, 495], 48: [90, 91, 92, 93], 63: [0, 264, 479, 508, 509]}
    for f, ctus in picks.items():
        ref = _oracle_family_ctus(oracle, fam, planes[f], org, stride, W, H, 8, QP, ctus)
        for c, (lg, dp) in ref.items():
            assert np.array_equal(lh[f, c], lg), (blob, f, c)
            assert np.array_equal(dh[f, c], dp), (blob, f, c)
    for f in (1, 40):   # batch == the single-picture host entry point
        d1, _ = ctx.predict_frame(planes[f], org, stride, qp=QP)
        assert np.array_equal(dh[f], d1), f
    assert len(np.unique(dh)) == 4
    ctx.close()


@pytest.mark.parametrize("blob,max_frames", [("depthnet_family_d2.fhw", 1), ("depthnet_family_d3.fhw", 2), ("depthnet_family_d1.fhw", 2)])
def test_host_batch_with_family_members(oracle, blob, max_frames):
    """fhevc_predict_frames (pinned ring, two streams that alternate per chunk) with members whose activations live in ONE set of scratch tensors
    per context: chunk k + 1 must not overwrite what chunk k's kernels still read.  Five pictures, chunks of max_frames, against the oracle."""
    W, H, NF, QP = 416, 240, 5, 27
    fam = weights.load_any(os.path.join(ROOT, "fasthevc_amd", "weights", blob))
    lumas = [frames.hetero_luma(W, H, seed=300 + f) for f in range(NF)]
    planes = np.stack([frames.to_pel_plane(y, 8)[0] for y in lumas])
    _, org, stride = frames.to_pel_plane(lumas[0], 8)
    ctx = capi.Context(W, H, 8, fam, max_frames=max_frames)
    n = ctx.num_ctus
    for rep in range(2):
        depth, had = ctx.predict_frames(planes, qp=QP, origin=org, stride=stride, frame_stride=planes.shape[1] * planes.shape[2])
        for f in range(NF):
            ref = _oracle_family_ctus(oracle, fam, planes[f], org, stride, W, H, 8, QP, range(n))
            for c in range(n):
                assert np.array_equal(depth[f, c], ref[c][1]), (blob, rep, f, c)
    ctx.close()


def test_config3_4k_bands_through_the_layer_path(oracle):
    """config 3's geometry (3840 x 2160, last CTU row 48 samples tall) with the two-convolutions-per-block member: the 8 CTU-row bands of an
    8-rank node one after the other, flag words assembled, sampled CTUs against the oracle."""
    import torch
    dev = torch.device("cuda:0")
    W, H, QP = 3840, 2160, 32
    fam = weights.load_any(os.path.join(ROOT, "fasthevc_amd", "weights", "depthnet_family_d2.fhw"))
    buf, org, stride = frames.to_pel_plane(frames.hetero_luma(W, H), 8)
    ctx = capi.Context(W, H, 8, fam)
    d16 = torch.from_numpy(buf).to(dev)
    flags = torch.zeros(ctx.num_ctus, dtype=torch.int32, device=dev)
    whole = torch.zeros((ctx.num_ctus, 256), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for rank in range(8):
        rb, re = bands.band(ctx.ctus_y, rank, 8)
        ctx.predict_frames_device(d16.data_ptr() + 2 * org, 2, stride, 0, 1, whole[rb * ctx.ctus_x:].data_ptr(), None, None, rows=(rb, re), qp=QP,
                                  d_flags=flags[rb * ctx.ctus_x:].data_ptr())
    full = torch.zeros_like(whole)
    ctx.expand_depth_flags_device(flags.data_ptr(), 1, full.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(full, whole)
    dh = whole.cpu().numpy()
    picks = [0, 59, 60 * 4 - 1, 60 * 4, 60 * 17 + 30, 60 * 29 + 59, 60 * 30, 60 * 33, 60 * 33 + 31, 60 * 34 - 1]   # band seams, corners, the 48-row bottom edge
    ref = _oracle_family_ctus(oracle, fam, buf, org, stride, W, H, 8, QP, picks)
    for c, (_, dp) in ref.items():
        assert np.array_equal(dh[c], dp), c
    ctx.close()

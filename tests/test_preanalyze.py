"""N3, adaptive-QP pre-analysis (TEncPreanalyzer::xPreanalyze, TEncPreanalyzer.cpp:64-152): the oracle against the
doubles the reference itself produced (tests/golden/ref_preanalyze.npz, oracle/gen_golden.py), and the HIP kernel
against both.  Bit-exact: 64-bit integer sums, then the reference's double operations in the reference's order."""
import os

import numpy as np
import pytest

from oracle import oracle_py as op
from fasthevc_amd import capi, frames

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cases():
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_preanalyze.npz"))
    for k, name in enumerate(g["cases"]):
        content, size, bd, depth = str(name).split(":")
        w, h = (int(v) for v in size.split("x"))
        luma = frames.texture16_luma(w, h) if content == "texture16" else frames.hetero_luma(w, h)
        yield luma, w, h, int(bd[2:]), int(depth[1:]), g[f"act{k}"], g[f"avg{k}"]


def _qp_cases():
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_preanalyze.npz"))
    for k, name in enumerate(g["cases"]):
        _, size, bd, depth = str(name).split(":")
        w, h = (int(v) for v in size.split("x"))
        for key in g.files:
            if key.startswith(f"qp{k}_"):
                _, r, q = key.split("_")
                yield w, h, int(bd[2:]), int(depth[1:]), g[f"act{k}"], g[f"avg{k}"], int(r[1:]), int(q[1:]), g[key]


def _layer_sizes(w, h, depth):
    return [((w + (64 >> d) - 1) // (64 >> d)) * ((h + (64 >> d) - 1) // (64 >> d)) for d in range(depth)]


def _oracle_layers(oracle, luma, bd, depth):
    h, w = luma.shape
    buf, org, stride = frames.to_pel_plane(luma, bd)
    acts, avgs = [], []
    for d, n in enumerate(_layer_sizes(w, h, depth)):
        a = np.zeros(n)
        avgs.append(oracle.fho_preanalyze_layer(op.ptr(buf.reshape(-1), org), stride, w, h, 64 >> d, a))
        acts.append(a)
    return np.concatenate(acts), np.array(avgs)


def test_oracle_matches_reference_doubles(oracle):
    n = 0
    for luma, w, h, bd, depth, act, avg in _cases():
        a, v = _oracle_layers(oracle, luma, bd, depth)
        assert a.tobytes() == act.tobytes(), (w, h, bd)  # bit patterns, not approximate equality
        assert v.tobytes() == avg.tobytes(), (w, h, bd)
        n += 1
    assert n == 3


def test_activity_of_a_flat_picture_is_one(oracle):
    flat = np.full((72, 136), 77, np.uint8)
    a, v = _oracle_layers(oracle, flat, 8, 4)
    assert np.all(a == 1.0) and np.all(v == 1.0)


def test_layout_helper_needs_no_device():
    lib = capi.load_library()
    import ctypes as C
    off = (C.c_longlong * 5)()
    assert lib.fhevc_aq_parts(1000, 568, 4, off) == sum(_layer_sizes(1000, 568, 4))
    assert list(off)[:2] == [0, 16 * 9]
    assert lib.fhevc_aq_parts(1000, 568, 5, None) == capi.E_INVALID


def test_cu_qp_matches_reference_xcomputeqp(oracle):
    """TEncCu::xComputeQP on the reference's activities: the oracle's restatement and the C-ABI host function."""
    import ctypes as C
    lib = capi.load_library()
    n = clipped = 0
    for w, h, bd, depth, act, avg, range_, base_qp, want in _qp_cases():
        sizes = _layer_sizes(w, h, depth)
        layer_of = np.repeat(np.arange(depth), sizes)
        got = np.array([oracle.fho_aq_qp(float(a), float(avg[d]), range_, base_qp, 6 * (bd - 8)) for a, d in zip(act, layer_of)])
        assert np.array_equal(got, want), (w, h, range_, base_qp)
        out = np.zeros(act.size, np.int8)
        assert lib.fhevc_aq_qp(act.ctypes.data, avg.ctypes.data, w, h, depth, range_, base_qp, 6 * (bd - 8), out.ctypes.data) == 0
        assert np.array_equal(out, want), (w, h, range_, base_qp)
        clipped += int(np.sum((want == 51) | (want == -6 * (bd - 8))))
        n += 1
    assert n == 9 and clipped > 0  # both clip ends are exercised by the fixture
    assert lib.fhevc_aq_qp(None, None, 64, 64, 1, 6, 32, 0, None) == capi.E_INVALID


@pytest.mark.gpu
def test_kernel_matches_reference_and_oracle(oracle):
    for luma, w, h, bd, depth, act, avg in _cases():
        buf, org, stride = frames.to_pel_plane(luma, bd)
        ctx = capi.Context(w, h, bd)
        a, v = ctx.preanalyze(buf, org, stride, depth)
        assert a.tobytes() == act.tobytes(), (w, h, bd)
        assert v.tobytes() == avg.tobytes(), (w, h, bd)
        ctx.close()


@pytest.mark.gpu
def test_kernel_device_batch_bands_and_extremes(oracle):
    import torch
    dev = torch.device("cuda:0")
    W, H, NF, D = 1928, 1080, 3, 4
    rng = np.random.default_rng(5)
    lumas = [frames.hetero_luma(W, H, seed=50), rng.integers(0, 256, (H, W)).astype(np.uint8), np.full((H, W), 255, np.uint8)]
    # 12-bit planes: the largest sums of squares the integer path has to carry
    planes = np.stack([frames.to_pel_plane(y, 12)[0] for y in lumas])
    _, org, stride = frames.to_pel_plane(lumas[0], 12)
    ctx = capi.Context(W, H, 12)
    off = ctx.aq_layout(D)
    d16 = torch.from_numpy(planes).to(dev)
    out = torch.full((NF, off[-1]), -1.0, dtype=torch.float64, device=dev)
    rows = ctx.ctus_y
    for rb, re in ((0, 5), (5, rows)):  # two CTU-row bands fill the whole-picture layout between them
        torch.cuda.synchronize()  # the fills above run on torch's stream: order them explicitly before the library call
        ctx.preanalyze_frames_device(d16.data_ptr() + 2 * org, 2, stride, planes.shape[1] * planes.shape[2], NF,
                                     out.data_ptr(), D, rows=(rb, re))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for f in range(NF):
        a, _ = _oracle_layers(oracle, lumas[f], 12, D)
        assert got[f].tobytes() == a.tobytes(), f
    # uint8 sample layout, 8-bit content
    ctx8 = capi.Context(W, H, 8)
    d8 = torch.from_numpy(np.stack(lumas)).to(dev)
    out8 = torch.zeros((NF, off[-1]), dtype=torch.float64, device=dev)
    ctx8.preanalyze_frames_device(d8.data_ptr(), 1, W, W * H, NF, out8.data_ptr(), D)
    torch.cuda.synchronize()
    for f in range(NF):
        a, _ = _oracle_layers(oracle, lumas[f], 8, D)
        assert out8[f].cpu().numpy().tobytes() == a.tobytes(), f
    ctx.close()
    ctx8.close()

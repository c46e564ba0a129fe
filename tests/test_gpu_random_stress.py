"""Randomised GPU parity sweep: picture sizes (multiples of 8, partial CTUs in both directions), bit depths, QPs, weight
blobs (incl. maximum-magnitude ones) and content drawn from a seeded generator; every kernel of the path against the CPU
oracle, bit for bit."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle_py as op
from fasthevc_amd import capi, frames, weights

pytestmark = pytest.mark.gpu


def _content(rng, w, h, bd):
    top = (1 << bd) - 1
    kind = int(rng.integers(0, 5))
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == 0:
        y = rng.integers(0, top + 1, (h, w))
    elif kind == 1:
        y = (frames.fractal_luma(w + 8, h + 8, seed=int(rng.integers(1 << 30)))[:h, :w].astype(np.int64) << (bd - 8)) + rng.integers(0, 1 << (bd - 8), (h, w))
    elif kind == 2:
        y = np.where(((xx // int(rng.integers(1, 9))) + (yy // int(rng.integers(1, 9)))) % 2 == 0, 0, top)
    elif kind == 3:
        y = np.clip((xx * top) // max(w - 1, 1) + rng.integers(-3, 4, (h, w)), 0, top)
    else:
        y = np.full((h, w), int(rng.integers(0, top + 1)))
    return y.astype(np.int16)


@pytest.mark.parametrize("seed", range(40))
def test_random_configuration(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    W, H = int(rng.integers(8, 49)) * 8, int(rng.integers(8, 33)) * 8
    bd = int(rng.choice([8, 8, 10, 12]))
    qp = int(rng.integers(0, 52))
    w = weights.random_weights(seed, extreme=bool(seed % 3 == 0))
    m = frames.HM_MARGIN
    stride = W + 2 * m + 8 * int(rng.integers(0, 3))           # strides other than HM's
    buf = np.zeros((H + 2 * m, stride), np.int16)
    buf[m:m + H, m:m + W] = _content(rng, W, H, bd)
    org = m * stride + m
    cw, ch = frames.ctu_grid(W, H)
    n = cw * ch
    ctx = capi.Context(W, H, bd, w, arith=("i8", "f16")[seed & 1])  # both arithmetic forms of the classifier (the extreme blobs, seed % 3 == 0,
    # depth CNN + source Hadamard                                      # overflow 24 bits: the i8 form then takes its general requant)
    depth_ref, logits = np.zeros(n * 256, np.uint8), np.zeros(n * 42, np.int32)
    oracle.fho_predict_frame(op.weights_from_arrays(w), op.ptr(buf.reshape(-1), org), stride, W, H, bd, qp, depth_ref, C.c_void_p(logits.ctypes.data))
    had_ref = np.zeros(n, np.int32)
    oracle.fho_frame_src_hadamard(op.ptr(buf.reshape(-1), org), stride, W, H, had_ref)
    depth, had = ctx.predict_frame(buf, org, stride, qp=qp)
    assert np.array_equal(depth.reshape(-1), depth_ref), (W, H, bd, qp)
    assert np.array_equal(had, had_ref), (W, H, bd, qp)
    # soft decisions
    ms, mt = int(rng.integers(0, 200000)), int(rng.integers(0, 200000))
    dmin, dmax = ctx.predict_frame_range(buf, org, stride, qp=qp, margin=ms, margin_stop=mt)
    emin, emax = np.zeros((n, 256), np.uint8), np.zeros((n, 256), np.uint8)
    for c in range(n):
        vw, vh = min(64, W - (c % cw) * 64), min(64, H - (c // cw) * 64)
        oracle.fho_depth_range_from_logits(np.ascontiguousarray(logits[c * 42:(c + 1) * 42]), vw, vh, ms, mt, emin[c], emax[c])
    assert np.array_equal(dmin, emin) and np.array_equal(dmax, emax), (W, H, bd, qp, ms, mt)
    # 35-mode first pass on a few CTUs (the oracle takes ~3 ms per CTU)
    nodes = ctx.intra_first_pass(buf, org, stride, qp=qp)
    sl = oracle.fho_lambda_intra(qp, bd) ** 0.5
    exp = (op.NodeCost * 85)()
    for c in rng.choice(n, size=min(n, 6), replace=False):
        oracle.fho_first_pass_ctu(op.ptr(buf.reshape(-1), org), stride, W, H, int(c % cw), int(c // cw), bd, sl, exp)
        e = np.frombuffer(exp, dtype=capi.NODE_DTYPE)
        for k in ("satd", "mode", "cost"):
            assert np.array_equal(nodes[c][k], e[k]), (W, H, bd, qp, int(c), k)
    # AQ pre-analysis, all layers
    depth_layers = int(rng.integers(1, 5))
    act, avg = ctx.preanalyze(buf, org, stride, depth_layers)
    off = ctx.aq_layout(depth_layers)
    for d in range(depth_layers):
        a = np.zeros(off[d + 1] - off[d])
        av = oracle.fho_preanalyze_layer(op.ptr(buf.reshape(-1), org), stride, W, H, 64 >> d, a)
        assert act[off[d]:off[d + 1]].tobytes() == a.tobytes() and avg[d] == av, (W, H, bd, d)
    ctx.close()

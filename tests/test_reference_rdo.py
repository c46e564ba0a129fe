"""The reference's own decision path (TEncSlice::compressSlice -> TEncCu::xCompressCU) behind oracle/ref_rdo_harness.cpp,
and the xCompressCU hook of hm_patch/ applied to it.  Needs oracle/_ref (built where /root/reference exists); elsewhere
only the committed fixture is checked."""
import os

import numpy as np
import pytest

from oracle import oracle_py as op
from fasthevc_amd import frames

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "ref_rdo_hetero_768x512_q32.npz")
need_ref = pytest.mark.skipif(not (op.have_ref() and op.have_hook()), reason="oracle/_ref not built (no /root/reference here)")


def _crop():
    return frames.to_pel_plane(frames.hetero_luma(1920, 1080)[:512, :768].copy(), 8)


def test_fixture_is_a_valid_depth_map(oracle):
    fx = np.load(FIX)
    depth = fx["depth"]
    assert depth.shape == (96, 256) and depth.max() <= 3
    assert len(np.unique(depth)) == 4
    for d in depth:  # every CTU map is a quad-tree: the reference's split-flag codec round-trips it
        flags = np.zeros(85, np.uint8)
        n = oracle.fho_depth_to_split_flags(np.ascontiguousarray(d), flags)
        back = np.zeros(256, np.uint8)
        assert oracle.fho_split_flags_to_depth(flags, n, back) == n
        assert np.array_equal(back, d)


@need_ref
def test_reference_rdo_is_reproducible_and_matches_fixture():
    buf, org, stride = _crop()
    ref = op.bind_rdo(op.load_ref())
    depth, st = op.rdo_encode(ref, buf, org, stride, 768, 512, 8, 32)
    fx = np.load(FIX)
    assert np.array_equal(depth, fx["depth"])
    assert st["bits"] == float(fx["bits"]) and st["dist"] == float(fx["dist"])


@need_ref
def test_hook_with_own_depth_map_reproduces_full_rdo():
    """SURVEY F11 inside the harness: forcing HM's own map through the hm_patch hook gives the identical result."""
    buf, org, stride = _crop()
    hook = op.bind_rdo(op.load_ref(hook=True))
    fx = np.load(FIX)
    d_stock, s_stock = op.rdo_encode(hook, buf, org, stride, 768, 512, 8, 32)          # no map: stock behaviour
    assert np.array_equal(d_stock, fx["depth"]) and s_stock["bits"] == float(fx["bits"])
    d_forced, s_forced = op.rdo_encode(hook, buf, org, stride, 768, 512, 8, 32, forced_depth=fx["depth"])
    assert np.array_equal(d_forced, fx["depth"])
    assert s_forced["bits"] == s_stock["bits"] and s_forced["dist"] == s_stock["dist"]
    assert s_forced["seconds"] < 0.5 * s_stock["seconds"]  # and it skips most of the search
    # a wrong map costs RD: all-8x8 is clearly worse than the reference's choice
    _, s_bad = op.rdo_encode(hook, buf, org, stride, 768, 512, 8, 32, forced_depth=np.full_like(fx["depth"], 3))
    assert s_bad["rdcost"] > 1.05 * s_stock["rdcost"]


def test_trained_weights_follow_the_reference_decisions(oracle):
    """The shipped blob (fasthevc_amd/weights/depthnet_v1.fhw, trained on the reference's full-RDO labels) against the
    committed reference depth map of a picture it never saw (pinned hetero content): most units get HM's depth."""
    from fasthevc_amd import weights
    path = os.path.join(ROOT, "fasthevc_amd", "weights", "depthnet_v1.fhw")
    w = weights.load(path)
    buf, org, stride = _crop()
    pred = np.zeros(96 * 256, np.uint8)
    oracle.fho_predict_frame(op.weights_from_arrays(w), op.ptr(buf.reshape(-1), org), stride, 768, 512, 8, 32, pred, None)
    ref = np.load(FIX)["depth"].reshape(-1)
    agree = float((pred == ref).mean())
    mean_err = float(np.abs(pred.astype(int) - ref.astype(int)).mean())  # compareSplitMode / units (TComSysuCuMDTools.cpp:48-77)
    assert agree > 0.80 and mean_err < 0.30, (agree, mean_err)
    const = max(float((ref == c).mean()) for c in range(4))
    assert agree > const + 0.15  # clearly better than any constant map

"""End to end on the GPU box: the REFERENCE's own TEncSlice::compressSlice / TEncCu::xCompressCU (oracle/_ref, built from
/root/reference in the build container) with the hm_patch hook compiled against include/fasthevc.h and linked to the
in-tree libfasthevc_hip.so.  HM asks the MI355X for the depth maps through the C ABI exactly as a patched encoder
would (TEncFastDepth::predictPicture -> fhevc_predict_frame_range); the result must equal the same encoder driven
by the CPU oracle's maps, and the hook must fall back to stock full RDO when the library is switched off."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle_py as op
from fasthevc_amd import frames, weights

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOB = os.path.join(ROOT, "fasthevc_amd", "weights", "depthnet_v1.fhw")
GPU_SO = os.path.join(ROOT, "oracle", "_ref", "libhmref_hookgpu.so")
KNOBS = ("FHEVC_ENABLE", "FHEVC_WEIGHTS", "FHEVC_MARGIN", "FHEVC_MARGIN_SPLIT", "FHEVC_MARGIN_STOP", "FHEVC_DEVICE")
QP = 32


def _picture(W, H):
    luma = frames.hetero_luma(1920, 1080)[:H, :W].copy()
    cu, cv = frames.chroma_planes("hetero", 1920, 1080)
    chroma = (cu[:H // 2, :W // 2].astype(np.int16), cv[:H // 2, :W // 2].astype(np.int16))
    return frames.to_pel_plane(luma, 8) + (chroma,)


def _oracle_maps(oracle, buf, org, stride, W, H, margin_split, margin_stop):
    n = (W // 64) * (H // 64)
    ws = op.weights_from_arrays(weights.load(BLOB))
    pred = np.zeros(n * 256, np.uint8)
    logits = np.zeros(n * 42, np.int32)
    oracle.fho_predict_frame(ws, op.ptr(buf.reshape(-1), org), stride, W, H, 8, QP, pred, C.c_void_p(logits.ctypes.data))
    dmin, dmax = np.zeros((n, 256), np.uint8), np.zeros((n, 256), np.uint8)
    for c in range(n):
        oracle.fho_depth_range_from_logits(np.ascontiguousarray(logits[c * 42:(c + 1) * 42]), 64, 64, margin_split, margin_stop, dmin[c], dmax[c])
    return dmin, dmax


@pytest.mark.skipif(not os.path.exists(GPU_SO), reason="oracle/_ref/libhmref_hookgpu.so is built where /root/reference exists")
def test_reference_encoder_driven_by_the_gpu_library(oracle):
    hook = op.bind_rdo(op.load_ref(hook=True))      # CPU-only hook build: explicit maps
    gpu = op.bind_rdo(op.load_ref(hook="gpu"))      # hook build whose TEncFastDepth calls libfasthevc_hip.so
    saved = {k: os.environ.get(k) for k in KNOBS}
    try:
        # TEncFastDepth reads its knobs when the harness constructs the encoder of a geometry: one geometry per setting
        for (W, H), env, margins in (((768, 512), {"FHEVC_ENABLE": "1", "FHEVC_WEIGHTS": BLOB, "FHEVC_MARGIN": "0"}, (0, 0)),
                                    ((832, 448), {"FHEVC_ENABLE": "1", "FHEVC_WEIGHTS": BLOB}, (32000, 0)),   # the hook's defaults
                                    ((704, 512), {"FHEVC_ENABLE": "1", "FHEVC_WEIGHTS": BLOB, "FHEVC_MARGIN_SPLIT": "32000"}, (32000, 0)),
                                    ((640, 448), {"FHEVC_ENABLE": "1", "FHEVC_WEIGHTS": BLOB, "FHEVC_MARGIN": "8000"}, (8000, 8000))):
            buf, org, stride, chroma = _picture(W, H)
            dmin, dmax = _oracle_maps(oracle, buf, org, stride, W, H, *margins)
            d_ref, s_ref = op.rdo_encode(hook, buf, org, stride, W, H, 8, QP, forced_depth=dmin, forced_depth_max=dmax, chroma=chroma)
            d_full, s_full = op.rdo_encode(hook, buf, org, stride, W, H, 8, QP, chroma=chroma)
            for k in KNOBS:
                os.environ.pop(k, None)
            os.environ.update(env)
            d_gpu, s_gpu = op.rdo_encode(gpu, buf, org, stride, W, H, 8, QP, chroma=chroma)
            assert np.array_equal(d_gpu, d_ref), (W, H)
            assert s_gpu["bits"] == s_ref["bits"] and s_gpu["dist"] == s_ref["dist"], (W, H)
            assert not np.array_equal(d_gpu, d_full) and s_gpu["seconds"] < 0.8 * s_full["seconds"], (W, H)
        # library switched off (FHEVC_ENABLE unset) / weights missing: stock full RDO, never an abort
        for (W, H), env in (((576, 448), {}), ((512, 448), {"FHEVC_ENABLE": "1", "FHEVC_WEIGHTS": "/nonexistent.fhw"})):
            buf, org, stride, chroma = _picture(W, H)
            d_full, s_full = op.rdo_encode(hook, buf, org, stride, W, H, 8, QP, chroma=chroma)
            for k in KNOBS:
                os.environ.pop(k, None)
            os.environ.update(env)
            d_off, s_off = op.rdo_encode(gpu, buf, org, stride, W, H, 8, QP, chroma=chroma)
            assert np.array_equal(d_off, d_full) and s_off["bits"] == s_full["bits"], (W, H)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

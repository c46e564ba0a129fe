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
BLOB = os.path.join(ROOT, "fasthevc_amd", "weights", "depthnet_v2.fhw")   # the shipped blob
GPU_SO = os.path.join(ROOT, "oracle", "_ref", "libhmref_hookgpu.so")
KNOBS = ("FHEVC_ENABLE", "FHEVC_WEIGHTS", "FHEVC_MARGIN", "FHEVC_MARGIN_SPLIT", "FHEVC_MARGIN_STOP", "FHEVC_DEVICE", "FHEVC_DEVICES", "FHEVC_FIRST_PASS")
QP = 32


def _picture(W, H):
    luma = frames.hetero_luma(1920, 1080)[:H, :W].copy()
    cu, cv = frames.chroma_planes("hetero", 1920, 1080)
    chroma = (cu[:H // 2, :W // 2].astype(np.int16), cv[:H // 2, :W // 2].astype(np.int16))
    return frames.to_pel_plane(luma, 8) + (chroma,)


def _oracle_candidates(oracle, buf, org, stride, W, H):
    """the first-pass candidate lists [numCtus, 85, 8] from the CPU oracle (fhevc_intra_first_pass_candidates is bit-exact with it)"""
    cw, n = W // 64, (W // 64) * (H // 64)
    cand = np.zeros((n, 85, 8), np.uint8)
    oracle.fho_first_pass_candidates_ctu.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]
    sl = oracle.fho_lambda_intra(QP, 8) ** 0.5
    for c in range(n):
        oracle.fho_first_pass_candidates_ctu(C.c_void_p(buf.reshape(-1).ctypes.data + 2 * org), stride, W, H, c % cw, c // cw, 8, C.c_double(sl), 8, cand[c].ctypes.data)
    return cand


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
                                    ((832, 448), {"FHEVC_ENABLE": "1", "FHEVC_WEIGHTS": BLOB}, (100000, 64000)),   # the hook's defaults
                                    ((704, 512), {"FHEVC_ENABLE": "1", "FHEVC_WEIGHTS": BLOB, "FHEVC_MARGIN_SPLIT": "32000"}, (32000, 64000)),   # the other side keeps its default
                                    ((576, 512), {"FHEVC_ENABLE": "1", "FHEVC_WEIGHTS": BLOB, "FHEVC_MARGIN_SPLIT": "32000", "FHEVC_MARGIN_STOP": "0"}, (32000, 0)),
                                    ((640, 448), {"FHEVC_ENABLE": "1", "FHEVC_WEIGHTS": BLOB, "FHEVC_MARGIN": "8000"}, (8000, 8000)),
                                    # a multi-device context behind the hook (FHEVC_DEVICES): CTU-row bands over two queues of the one MI355X here
                                    ((896, 448), {"FHEVC_ENABLE": "1", "FHEVC_WEIGHTS": BLOB, "FHEVC_MARGIN": "8000", "FHEVC_DEVICES": "0,0"}, (8000, 8000)),
                                    # the first pass consumed: estIntraPredLumaQT takes its candidate lists from the GPU (FHEVC_FIRST_PASS)
                                    ((960, 448), {"FHEVC_ENABLE": "1", "FHEVC_WEIGHTS": BLOB, "FHEVC_MARGIN": "8000", "FHEVC_FIRST_PASS": "1"}, (8000, 8000))):
            buf, org, stride, chroma = _picture(W, H)
            dmin, dmax = _oracle_maps(oracle, buf, org, stride, W, H, *margins)
            cand = _oracle_candidates(oracle, buf, org, stride, W, H) if env.get("FHEVC_FIRST_PASS") else None
            d_ref, s_ref = op.rdo_encode(hook, buf, org, stride, W, H, 8, QP, forced_depth=dmin, forced_depth_max=dmax, chroma=chroma, candidates=cand)
            d_full, s_full = op.rdo_encode(hook, buf, org, stride, W, H, 8, QP, chroma=chroma)
            for k in KNOBS:
                os.environ.pop(k, None)
            os.environ.update(env)
            d_gpu, s_gpu = op.rdo_encode(gpu, buf, org, stride, W, H, 8, QP, chroma=chroma)
            assert np.array_equal(d_gpu, d_ref), (W, H)
            assert s_gpu["bits"] == s_ref["bits"] and s_gpu["dist"] == s_ref["dist"], (W, H)
            # the ranges did restrict the search: less time in compressSlice; with narrow margins the decisions differ from full RDO too
            # (at the shipped default the restricted search may well end at full RDO's own map: that is the point of the guard)
            assert s_gpu["seconds"] < (0.95 if margins[1] >= 48000 else 0.8) * s_full["seconds"], (W, H)
            assert margins[1] >= 48000 or not np.array_equal(d_gpu, d_full), (W, H)
        # library switched off (FHEVC_ENABLE unset) / weights missing: stock full RDO, never an abort
        for (W, H), env in (((576, 448), {}), ((512, 448), {"FHEVC_ENABLE": "1", "FHEVC_WEIGHTS": "/nonexistent.fhw"})):  # (geometries not used above)
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


P_SO = os.path.join(ROOT, "oracle", "_ref", "libhmref_p.so")
PGPU_SO = os.path.join(ROOT, "oracle", "_ref", "libhmref_pgpu.so")
P_KNOBS = KNOBS + ("FHEVC_P_MODE", "FHEVC_P_WINDOW", "FHEVC_P_RANGE", "FHEVC_P_THRESH", "FHEVC_P_MC")


def _load_p(path):
    from fasthevc_amd import capi
    capi.load_library()  # one HIP runtime per process: the product loader maps it before the harness library pulls it in
    libdl = C.CDLL(None)
    libdl.dlopen.restype = C.c_void_p
    libdl.dlopen.argtypes = [C.c_char_p, C.c_int]
    h = libdl.dlopen(path.encode(), os.RTLD_LAZY | os.RTLD_LOCAL)
    assert h, path
    lib = op.bind_rdo(C.CDLL(path, handle=h))
    lib.href_rdo_encode_next_p.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_void_p] * 4
    return lib


def _next_p(lib, y, W, H, qp, poc, fmin=None, fmax=None):
    buf, org, stride = frames.to_pel_plane(y, 8)
    u = np.full((H // 2, W // 2), 128, np.int16)
    n = (W // 64) * (H // 64)
    d, s = np.zeros(n * 256, np.uint8), np.zeros(8)
    rc = lib.href_rdo_encode_next_p(buf.reshape(-1).ctypes.data + 2 * org, u.ctypes.data, u.ctypes.data, stride, W, H, 8, qp, poc,
                                    None if fmin is None else fmin.ctypes.data, None if fmax is None else fmax.ctypes.data, d.ctypes.data, s.ctypes.data)
    assert rc == 0
    return d.reshape(n, 256), s


@pytest.mark.skipif(not os.path.exists(GPU_SO), reason="oracle/_ref/libhmref_hookgpu.so is built where /root/reference exists")
def test_reference_encoder_driven_by_a_family_member_on_the_layer_path(oracle):
    """FHEVC_WEIGHTS may name any member of the reference's network family (FHW3): here the trained NetworkDepth-2 member 23 / 46 / 92 x 2, which the
    library runs layer by layer through HBM (k_cnn_layers.inc).  The reference's compressSlice driven by it through the hook == the CPU hook fed with
    the oracle's ranges for the same blob."""
    blob = os.path.join(ROOT, "fasthevc_amd", "weights", "depthnet_family_d2.fhw")
    fam_arrays = weights.load_any(blob)
    assert int(fam_arrays["depth"]) == 2
    hook = op.bind_rdo(op.load_ref(hook=True))
    gpu = op.bind_rdo(op.load_ref(hook="gpu"))
    W, H, margin = 512, 384, 8000
    buf, org, stride, chroma = _picture(W, H)
    n = (W // 64) * (H // 64)
    fam = op.family_from_arrays(fam_arrays)
    pred, logits = np.zeros(n * 256, np.uint8), np.zeros(n * 42, np.int32)
    oracle.fho_predict_frame_family(C.byref(fam), op.ptr(buf.reshape(-1), org), stride, W, H, 8, QP, pred.ctypes.data, logits.ctypes.data)
    dmin, dmax = np.zeros((n, 256), np.uint8), np.zeros((n, 256), np.uint8)
    for c in range(n):
        oracle.fho_depth_range_from_logits(np.ascontiguousarray(logits[c * 42:(c + 1) * 42]), 64, 64, margin, margin, dmin[c], dmax[c])
    d_ref, s_ref = op.rdo_encode(hook, buf, org, stride, W, H, 8, QP, forced_depth=dmin, forced_depth_max=dmax, chroma=chroma)
    d_full, _ = op.rdo_encode(hook, buf, org, stride, W, H, 8, QP, chroma=chroma)
    saved = {k: os.environ.get(k) for k in KNOBS}
    try:
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update({"FHEVC_ENABLE": "1", "FHEVC_WEIGHTS": blob, "FHEVC_MARGIN": str(margin)})
        d_gpu, s_gpu = op.rdo_encode(gpu, buf, org, stride, W, H, 8, QP, chroma=chroma)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert np.array_equal(d_gpu, d_ref) and s_gpu["bits"] == s_ref["bits"] and s_gpu["dist"] == s_ref["dist"]
    assert not np.array_equal(d_gpu, d_full)


@pytest.mark.parametrize("size,rng,speeds,mc", [((640, 448), 4, (3, 3), "0"), ((576, 384), 64, (21, -17), "1"), ((576, 384), 64, (21, -17), "node")])
@pytest.mark.skipif(not os.path.exists(PGPU_SO), reason="oracle/_ref/libhmref_pgpu.so is built where /root/reference exists")
def test_config4_p_pictures_driven_by_the_gpu_motion_search(oracle, size, rng, speeds, mc):
    """BASELINE config 4 end to end: I P P P through the reference's compressSlice with HM-16.14's inter checks restored.  In the
    GPU build TEncFastDepth::predictPicture asks the MI355X for the motion nodes of every P picture whose reference is a P
    picture (fhevc_motion_search) and turns them into depth ranges (fhevc_p_depth_range); the result must equal the CPU-hook
    build fed with the oracle's ranges (fho_motion_ctu + fho_p_depth_range), picture by picture, and differ from full RDO.
    Second case: FHEVC_P_RANGE=64 on a clip moving 21 / 17 samples per picture -- the hook then asks for HM's own integer search
    (SAD, k_motion_wide.hip) and takes the reference picture's depths at the motion-compensated position, per 4x4 unit (FHEVC_P_MC=1:
    fhevc_p_motion_compensated_depth) or per CU node of the current grid (FHEVC_P_MC=node, round 4: fhevc_p_node_depth, the hook's default for wide ranges)."""
    from fasthevc_amd import capi
    (W, H), QPI, QPP = size, 32, 38
    wide = rng > 8
    ys = frames.pan_clip(W, H, 4, seed=77, v_structure=speeds[0], v_noise=speeds[1])
    oracle.fho_p_motion_compensated_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    oracle.fho_p_node_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    cpu, gpu = _load_p(P_SO), _load_p(PGPU_SO)
    n, cw = (W // 64) * (H // 64), W // 64
    u = np.full((H // 2, W // 2), 128, np.int16)
    rule = op.PRule.from_buffer_copy(bytes(capi.p_rule_default_wide() if rng > 8 else capi.p_rule_default()))   # as the hook picks it
    saved = {k: os.environ.get(k) for k in P_KNOBS}
    try:
        for k in P_KNOBS:
            os.environ.pop(k, None)
        buf, org, stride = frames.to_pel_plane(ys[0], 8)

        def run(lib, ranges):
            maps, stats = [], []
            d, s = op.rdo_encode(lib, buf, org, stride, W, H, 8, QPI, chroma=(u, u))  # POC 0: full RDO in both builds
            maps.append(d)
            stats.append(s["coded_bits"])
            for f in range(1, 4):
                fmin = fmax = None
                if ranges and f >= 2:  # the oracle's ranges from the ORIGINAL pictures f, f - 1 and the depths HM chose for f - 1
                    pb, po, ps = frames.to_pel_plane(ys[f], 8)
                    rb, _, _ = frames.to_pel_plane(ys[f - 1], 8)
                    nodes = np.zeros((n, 85), capi.MOTION_DTYPE)
                    sl = oracle.fho_lambda_intra(QPP, 8) ** 0.5
                    fmin, fmax = np.zeros((n, 256), np.uint8), np.zeros((n, 256), np.uint8)
                    prev = np.ascontiguousarray(maps[-1])
                    seen = np.zeros(256, np.uint8)
                    for c in range(n):
                        oracle.fho_motion_ctu_dist(C.c_void_p(pb.reshape(-1).ctypes.data + 2 * po), ps, C.c_void_p(rb.reshape(-1).ctypes.data + 2 * po), ps,
                                                   W, H, c % cw, c // cw, 8, rng, C.c_double(sl), 1 if wide else 0, C.c_void_p(nodes[c].ctypes.data))
                    for c in range(n):
                        pc = prev[c]
                        if wide:
                            (oracle.fho_p_node_depth if mc == "node" else oracle.fho_p_motion_compensated_depth)(nodes[c].ctypes.data, prev.ctypes.data, W, H, c, seen.ctypes.data)
                            pc = seen
                        oracle.fho_p_depth_range(nodes[c].ctypes.data, pc.ctypes.data, 64, 64, QPP, C.byref(rule), fmin[c].ctypes.data, fmax[c].ctypes.data)
                d, s = _next_p(lib, ys[f], W, H, QPP, f, fmin, fmax)
                maps.append(d)
                stats.append(float(s[6]))
            return maps, stats

        full_maps, full_bits = run(cpu, False)
        ref_maps, ref_bits = run(cpu, True)
        # TEncFastDepth reads the I-picture knobs when the encoder object of a geometry is built: build it before FHEVC_ENABLE is
        # set, so that POC 0 runs stock RDO in both builds; the P pictures re-read the knobs (href_rdo_encode_next_p)
        op.rdo_encode(gpu, buf, org, stride, W, H, 8, QPI, chroma=(u, u))
        os.environ.update({"FHEVC_P_MODE": "motion", "FHEVC_ENABLE": "1", "FHEVC_WEIGHTS": BLOB, "FHEVC_P_RANGE": str(rng), "FHEVC_P_MC": mc})
        gpu_maps, gpu_bits = run(gpu, False)
        for f in range(4):
            assert np.array_equal(gpu_maps[f], ref_maps[f]), f
            assert gpu_bits[f] == ref_bits[f], f
        assert np.array_equal(gpu_maps[1], full_maps[1])                       # POC 1 (reference picture is intra): stock RDO
        assert any(not np.array_equal(gpu_maps[f], full_maps[f]) for f in (2, 3))   # the ranges did restrict POC 2 / 3
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

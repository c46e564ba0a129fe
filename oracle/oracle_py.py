"""ctypes bindings for oracle/libfhevc_oracle.so (the CPU restatement) and, when present,
oracle/_ref/libhmref.so (the reference's own objects behind oracle/ref_harness.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under fasthevc_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libfhevc_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libhmref.so")

_i16p = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")


class NodeCost(C.Structure):
    _fields_ = [("satd", C.c_uint32), ("mode", C.c_uint32), ("cost", C.c_double)]


class PRule(C.Structure):
    """Mirror of fho_p_rule"""
    _fields_ = [("w", (C.c_int32 * 10) * 3), ("t_split", C.c_int32 * 3), ("t_stop", C.c_int32 * 3), ("window", C.c_int32)]


class Weights(C.Structure):
    """Mirror of fho_weights (oracle/fhevc_oracle.h)."""
    _fields_ = [
        ("shift", C.c_int32 * 3),
        ("w1", C.c_int8 * (16 * 9)), ("b1", C.c_int32 * 16),
        ("w2", C.c_int8 * (32 * 16 * 9)), ("b2", C.c_int32 * 32),
        ("w3", C.c_int8 * (64 * 32 * 9)), ("b3", C.c_int32 * 64),
        ("wh64", C.c_int8 * (2 * 4096)), ("bh64", C.c_int32 * 2),
        ("wh32", C.c_int8 * (2 * 4096)), ("bh32", C.c_int32 * 2),
        ("wh16", C.c_int8 * (2 * 1024)), ("bh16", C.c_int32 * 2),
        ("qp_bias", C.c_int32 * (3 * 52)),
    ]


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])


def load_oracle():
    if not os.path.exists(ORACLE_SO):
        build_oracle()
    lib = C.CDLL(ORACLE_SO)
    lib.fho_init_scan_tables.argtypes = [_u16p, _u16p]
    lib.fho_depth_to_split_flags.argtypes = [_u8p, _u8p]
    lib.fho_depth_to_split_flags.restype = C.c_int
    lib.fho_split_flags_to_depth.argtypes = [_u8p, C.c_int, _u8p]
    lib.fho_split_flags_to_depth.restype = C.c_int
    lib.fho_compare_split_mode.argtypes = [_u8p, _u8p]
    lib.fho_compare_split_mode.restype = C.c_int
    lib.fho_depth_raster_to_zorder.argtypes = [_u8p, _u8p]
    lib.fho_depth_zorder_to_raster.argtypes = [_u8p, _u8p]
    for name in ("fho_had2x2", "fho_had4x4", "fho_had8x8"):
        f = getattr(lib, name)
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        f.restype = C.c_uint32
    lib.fho_satd.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.fho_satd.restype = C.c_uint32
    lib.fho_had8x8_src.argtypes = [C.c_void_p, C.c_int]
    lib.fho_had8x8_src.restype = C.c_int32
    lib.fho_ctu_src_hadamard.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.fho_ctu_src_hadamard.restype = C.c_int32
    lib.fho_frame_src_hadamard.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _i32p]
    lib.fho_preanalyze_layer.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")]
    lib.fho_preanalyze_layer.restype = C.c_double
    lib.fho_aq_qp.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]
    lib.fho_aq_qp.restype = C.c_int
    lib.fho_lambda_intra.argtypes = [C.c_int, C.c_int]
    lib.fho_lambda_intra.restype = C.c_double
    lib.fho_mv_cost.argtypes = [C.c_int, C.c_int, C.c_double]
    lib.fho_mv_cost.restype = C.c_uint32
    lib.fho_motion_ctu.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_double, C.c_void_p]
    lib.fho_motion_ctu.restype = None
    lib.fho_motion_ctu_dist.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_double, C.c_int, C.c_void_p]
    lib.fho_motion_ctu_dist.restype = None
    lib.fho_cnn_ctu_family.argtypes = [C.POINTER(Family), C.c_void_p, C.c_int, C.c_void_p]
    lib.fho_cnn_ctu_family.restype = None
    lib.fho_predict_frame_family.argtypes = [C.POINTER(Family), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.fho_predict_frame_family.restype = None
    lib.fho_ilog2_q8.argtypes = [C.c_uint32]
    lib.fho_ilog2_q8.restype = C.c_int32
    lib.fho_p_depth_range.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(PRule), C.c_void_p, C.c_void_p]
    lib.fho_p_depth_range.restype = None
    lib.fho_fill_ref.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i16p]
    lib.fho_fill_ref_flags.argtypes = [C.c_void_p, C.c_int, _u8p, C.c_int, C.c_int, _i16p]
    lib.fho_filter_ref.argtypes = [_i16p, C.c_int, C.c_int, C.c_int, _i16p]
    lib.fho_use_filtered_ref.argtypes = [C.c_int, C.c_int]
    lib.fho_use_filtered_ref.restype = C.c_int
    lib.fho_pred_intra.argtypes = [_i16p, _i16p, C.c_int, C.c_int, C.c_int, _i16p]
    lib.fho_first_pass_node.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_double, C.POINTER(NodeCost), C.c_void_p]
    lib.fho_first_pass_ctu.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_double, C.POINTER(NodeCost)]
    lib.fho_cnn_ctu.argtypes = [C.POINTER(Weights), _i8p, C.c_int, _i32p]
    lib.fho_cnn_ctu_debug.argtypes = [C.POINTER(Weights), _i8p, C.c_int, _u8p, _u8p, _u8p, _i32p]
    lib.fho_depth_from_logits.argtypes = [_i32p, C.c_int, C.c_int, _u8p]
    lib.fho_depth_range_from_logits.argtypes = [_i32p, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, _u8p]
    lib.fho_depth_range_from_logits_levels.argtypes = [_i32p, C.c_int, C.c_int, _i32p, _i32p, _u8p, _u8p]
    lib.fho_flags_from_logits.argtypes = [_i32p, C.c_int, C.c_int]
    lib.fho_flags_from_logits.restype = C.c_uint32
    lib.fho_depth_from_flags.argtypes = [C.c_uint32, C.c_int, C.c_int, _u8p]
    lib.fho_load_ctu.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i8p]
    lib.fho_predict_frame.argtypes = [C.POINTER(Weights), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, C.c_void_p]
    return lib


def have_ref():
    return os.path.exists(REF_SO)


HOOK_SO = os.path.join(HERE, "_ref", "libhmref_hook.so")


def have_hook():
    return os.path.exists(HOOK_SO)


def load_ref(hook=False):
    """Open oracle/_ref/libhmref.so (or the hook variant, libhmref_hook.so = the same reference objects with
    hm_patch/ applied; hook="gpu": libhmref_hookgpu.so, whose TEncFastDepth calls the real GPU library) with
    RTLD_LAZY: one never-called reference symbol stays unresolved, see oracle/Makefile."""
    path = HOOK_SO.replace("_hook.so", "_hookgpu.so") if hook == "gpu" else HOOK_SO.replace("_hook.so", "_costs.so") if hook == "costs" else (HOOK_SO if hook else REF_SO)
    if hook == "gpu":
        # this library pulls in fasthevc_amd/lib/libfasthevc_hip.so; in a Python process that also holds torch the HIP runtime
        # must be the one torch ships (same SONAME as /opt/rocm's): let the product loader map it first
        from fasthevc_amd import capi
        capi.load_library()
    libdl = C.CDLL(None)
    libdl.dlopen.restype = C.c_void_p
    libdl.dlopen.argtypes = [C.c_char_p, C.c_int]
    handle = libdl.dlopen(path.encode(), os.RTLD_LAZY | os.RTLD_LOCAL)
    if not handle:
        raise OSError("cannot dlopen " + path)
    lib = C.CDLL(path, handle=handle)
    lib.href_version.restype = C.c_char_p
    for name in ("href_calc_had", "href_get_hads"):
        f = getattr(lib, name)
        f.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        f.restype = C.c_uint32
    lib.href_had8x8_islice.argtypes = [C.c_void_p, C.c_int]
    lib.href_had8x8_islice.restype = C.c_int32
    lib.href_ctu_src_hadamard.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.href_ctu_src_hadamard.restype = C.c_int32
    lib.href_scan_tables.argtypes = [_u32p, _u32p]
    lib.href_fill_ref.argtypes = [C.c_int, C.c_void_p, C.c_int, _u8p, C.c_int, _i16p]
    lib.href_use_filtered.argtypes = [C.c_int, C.c_int]
    lib.href_use_filtered.restype = C.c_int
    lib.href_pred_intra.argtypes = [_i16p, C.c_int, C.c_int, C.c_int, _i16p]
    if hasattr(lib, "href_preanalyze"):
        f64 = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
        lib.href_preanalyze.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f64, f64]
        lib.href_preanalyze.restype = C.c_int
        lib.href_aq_qp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")]
        lib.href_aq_qp.restype = C.c_int
    return lib


def ptr(a, offset_elems=0):
    """Raw pointer into a numpy array (keeps no reference: the caller holds the array)."""
    return C.c_void_p(a.ctypes.data + offset_elems * a.itemsize)


def weights_from_arrays(d):
    """dict of numpy arrays (fasthevc_amd.weights layout) -> Weights struct."""
    w = Weights()
    for k in ("shift", "w1", "b1", "w2", "b2", "w3", "b3", "wh64", "bh64", "wh32", "bh32", "wh16", "bh16", "qp_bias"):
        arr = np.ascontiguousarray(d[k]).reshape(-1)
        field = getattr(w, k)
        assert len(field) == arr.size, (k, len(field), arr.size)
        C.memmove(field, arr.ctypes.data, arr.nbytes)
    return w


class Family(C.Structure):
    _fields_ = [("c", C.c_int32 * 3), ("depth", C.c_int32), ("shift", (C.c_int32 * 3) * 3), ("w", (C.c_void_p * 3) * 3), ("b", (C.c_void_p * 3) * 3),
                ("wh64", C.c_void_p), ("wh32", C.c_void_p), ("wh16", C.c_void_p), ("bh64", C.c_int32 * 2), ("bh32", C.c_int32 * 2), ("bh16", C.c_int32 * 2),
                ("qp_bias", C.c_int32 * 156)]


def family_from_arrays(d):
    """fasthevc_amd.weights family dict (unpack_family / random_family / family_from_base) -> fho_family; the struct keeps the arrays alive"""
    f = Family()
    keep = []
    f.c[:] = [int(v) for v in d["widths"]]
    f.depth = int(d["depth"])
    for b in range(3):
        for j in range(3):
            f.shift[b][j] = int(d["shift"][b][j])
            if j < f.depth:
                wa, ba = np.ascontiguousarray(d[f"w{b}{j}"], np.int8), np.ascontiguousarray(d[f"b{b}{j}"], np.int32)
                keep += [wa, ba]
                f.w[b][j], f.b[b][j] = wa.ctypes.data, ba.ctypes.data
    for k in ("wh64", "wh32", "wh16"):
        a = np.ascontiguousarray(d[k], np.int8)
        keep.append(a)
        setattr(f, k, a.ctypes.data)
    for k in ("bh64", "bh32", "bh16"):
        getattr(f, k)[:] = [int(v) for v in d[k]]
    f.qp_bias[:] = [int(v) for v in np.asarray(d["qp_bias"]).reshape(-1)]
    f._keep = keep
    return f


def ref_line_to_roi(ref, n):
    """our 4N+1 reference line -> HM's (2N+1)x(2N+1) ROI buffer (row 0 = TL+above, col 0 = TL+left)."""
    sw = 2 * n + 1
    roi = np.zeros((sw, sw), np.int16)
    roi[0, :] = ref[2 * n:]
    roi[1:, 0] = ref[:2 * n][::-1]
    return roi


def roi_to_ref_line(roi, n):
    ref = np.zeros(4 * n + 1, np.int16)
    ref[2 * n:] = roi[0, :]
    ref[:2 * n] = roi[1:, 0][::-1]
    return ref


def bind_rdo(lib):
    """href_rdo_encode_frame of oracle/ref_rdo_harness.cpp (the reference's own compressSlice/xCompressCU)."""
    lib.href_rdo_encode_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, _u8p,
                                          np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")]
    lib.href_rdo_encode_frame.restype = C.c_int
    lib.href_rdo_encode_frame_yuv.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, _u8p, np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")]
    lib.href_rdo_encode_frame_yuv.restype = C.c_int
    lib.href_has_hook.restype = C.c_int
    return lib


def split_costs(lib, num_ctus):
    """`make -C oracle costs` build (load_ref(hook="costs")): RD costs of the last rdo_encode's xCompressCU nodes -> [numCtus, 21, 2] float64
    ([..., 0] best non-split mode, [..., 1] four-way split; node 0 the CTU, 1 + q quadrant q, 5 + b 16x16 block b, q / b in z-order;
    NaN where the reference did not evaluate both)."""
    out = np.zeros(num_ctus * 42, np.float64)
    lib.href_split_costs.argtypes = [C.c_void_p, C.c_int]
    if lib.href_split_costs(C.c_void_p(out.ctypes.data), num_ctus) != 0:
        raise RuntimeError("href_split_costs: geometry differs from the last encode")
    return out.reshape(num_ctus, 21, 2)


def rdo_encode(lib, plane, origin, stride, width, height, bit_depth, qp, forced_depth=None, chroma=None, forced_depth_max=None, candidates=None):
    """-> (depth [numCtus,256] uint8, stats dict).  plane: int16 Pel buffer; chroma: optional (cb, cr) int16 arrays
    [H/2, W/2] at the internal bit depth (default: flat mid-grey)."""
    n = ((width + 63) // 64) * ((height + 63) // 64)
    depth = np.zeros(n * 256, np.uint8)
    stats = np.zeros(10, np.float64)
    fd = None
    if forced_depth is not None:
        fd = np.ascontiguousarray(forced_depth, np.uint8).reshape(-1)
        assert fd.size == n * 256
    fdmax = None
    if forced_depth_max is not None:  # soft hook: forced_depth is then the depth_min map
        assert fd is not None
        fdmax = np.ascontiguousarray(forced_depth_max, np.uint8).reshape(-1)
        assert fdmax.size == n * 256
        lib.href_rdo_set_forced_max.argtypes = [C.c_void_p]
        if lib.href_rdo_set_forced_max(C.c_void_p(fdmax.ctypes.data)) != 0:
            raise RuntimeError("soft hook needs the hook build of the reference library")
    if candidates is not None:  # first-pass candidate lists [numCtus, 85, 8] for the hook's estIntraPredLumaQT patch
        cand = np.ascontiguousarray(candidates, np.uint8).reshape(-1)
        assert cand.size == n * 85 * 8
        lib.href_rdo_set_candidates.argtypes = [C.c_void_p]
        if lib.href_rdo_set_candidates(C.c_void_p(cand.ctypes.data)) != 0:
            raise RuntimeError("candidate lists need the hook build of the reference library")
    if chroma is not None:
        cb, cr = (np.ascontiguousarray(c, np.int16) for c in chroma)
        assert cb.shape == (height // 2, width // 2) and cr.shape == cb.shape
        rc = lib.href_rdo_encode_frame_yuv(ptr(plane.reshape(-1), origin), stride, ptr(cb), ptr(cr), width, height, bit_depth, qp,
                                           C.c_void_p(fd.ctypes.data) if fd is not None else None, depth, stats)
    else:
        rc = lib.href_rdo_encode_frame(ptr(plane.reshape(-1), origin), stride, width, height, bit_depth, qp,
                                       C.c_void_p(fd.ctypes.data) if fd is not None else None, depth, stats)
    if rc != 0:
        raise RuntimeError(f"href_rdo_encode_frame failed: {rc}")
    mse = stats[4] / (width * height)
    peak = (1 << bit_depth) - 1
    out = {"bits": stats[0], "dist": stats[1], "rdcost": stats[2], "seconds": stats[3],
           "psnr_y": 10 * np.log10(peak * peak / mse) if mse > 0 else 99.0, "ctus": int(stats[5]), "coded_bits": stats[6]}
    if stats[8] > 0:  # FHREF_ENCODE_SLICE=1: slice-data bits of the reference's own encodeSlice (real arithmetic coder)
        out["slice_data_bits"] = stats[8]
    if stats[7] > 0:  # FHREF_DEBLOCK=1: luma PSNR after the reference's own deblocking filter
        out["psnr_y_deblocked"] = 10 * np.log10(peak * peak / (stats[7] / (width * height)))
    if stats[9] > 0:  # FHREF_SAO=1 (with FHREF_DEBLOCK=1): luma PSNR after both in-loop filters, SAO decided by the reference's own SAOProcess
        out["psnr_y_filtered"] = 10 * np.log10(peak * peak / (stats[9] / (width * height)))
    if stats[8] > 0:  # md5 of the slice-data bytes (SURVEY F11 at the byte level)
        lib.href_slice_md5.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        dg = (C.c_ubyte * 16)()
        if lib.href_slice_md5(width, height, bit_depth, dg) == 0:
            out["slice_data_md5"] = bytes(dg).hex()
    return depth.reshape(n, 256), out

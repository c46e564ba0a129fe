"""ctypes mirror of include/fasthevc.h -- the host-side view of the C ABI used by tests, the trainer and bench.py.

There is no fallback: if the HIP library is missing or no gfx950 device is usable, calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import weights as _weights

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libfasthevc_hip.so")

OK, E_INVALID, E_NO_DEVICE, E_HIP, E_WEIGHTS, E_NOMEM, E_STATE = 0, -1, -2, -3, -4, -5, -6
BACKEND_HIP = 1
NODES_PER_CTU = 85
LOGITS_PER_CTU = 42

# every symbol include/fasthevc.h declares (tests/test_host_logic.py::test_c_abi_exports_every_declared_symbol checks header <-> this list <-> the .so)
SYMBOLS = [
    "fhevc_create", "fhevc_destroy", "fhevc_set_weights", "fhevc_predict_frame", "fhevc_satd",
    "fhevc_intra_first_pass", "fhevc_predict_frames_device", "fhevc_band", "fhevc_kernel_timing",
    "fhevc_enable_kernel_timing", "fhevc_get_stats", "fhevc_last_error", "fhevc_version",
    "fhevc_expand_depth_flags_device", "fhevc_aq_parts", "fhevc_preanalyze", "fhevc_preanalyze_frames_device", "fhevc_aq_qp", "fhevc_intra_first_pass_device",
    "fhevc_predict_frame_range", "fhevc_predict_frames_device_range",
    "fhevc_motion_search", "fhevc_motion_search_device", "fhevc_intra_first_pass_all", "fhevc_intra_first_pass_candidates", "fhevc_p_rule_default", "fhevc_p_rule_default_wide", "fhevc_p_depth_range", "fhevc_p_motion_compensated_depth", "fhevc_p_node_depth",
    "fhevc_predict_frames", "fhevc_alloc_host", "fhevc_free_host", "fhevc_set_cnn_arith", "fhevc_get_cnn_arith", "fhevc_set_motion_distortion", "fhevc_read_yuv_luma",
]
CNN_ARITH = {"i8": 8, "f16": 16}


class Cfg(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("bit_depth", C.c_int), ("ctu_size", C.c_int),
                ("max_depth", C.c_int), ("num_devices", C.c_int), ("device_ids", C.POINTER(C.c_int)),
                ("weights_path", C.c_char_p), ("backend", C.c_int), ("max_frames", C.c_int)]


class PRule(C.Structure):
    _fields_ = [("w", (C.c_int32 * 10) * 3), ("t_split", C.c_int32 * 3), ("t_stop", C.c_int32 * 3), ("window", C.c_int32)]


class NodeCost(C.Structure):
    _fields_ = [("satd", C.c_uint32), ("mode", C.c_uint32), ("cost", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("frames", C.c_uint64), ("ctus", C.c_uint64), ("bytes_h2d", C.c_uint64), ("bytes_d2h", C.c_uint64),
                ("kernels_launched", C.c_uint64), ("ms_h2d", C.c_double), ("ms_kernels", C.c_double),
                ("ms_d2h", C.c_double), ("last_cnn_ms", C.c_double), ("last_hadamard_ms", C.c_double),
                ("last_first_pass_ms", C.c_double), ("devices", C.c_uint64), ("devices_failed", C.c_uint64)]


NODE_DTYPE = np.dtype([("satd", np.uint32), ("mode", np.uint32), ("cost", np.float64)])
MOTION_DTYPE = np.dtype([("satd_zero", np.uint32), ("satd_best", np.uint32), ("cost_best", np.uint32), ("mvx", np.int16), ("mvy", np.int16)])


class FastHevcError(RuntimeError):
    def __init__(self, code, text=""):
        super().__init__(f"fasthevc error {code}: {text}")
        self.code = code


_lib = None


def _share_the_hosts_hip_runtime():
    """A PyTorch wheel bundles its own libamdhip64.so with the SONAME of /opt/rocm's.  Whichever copy a process maps first
    serves every later DT_NEEDED of that SONAME, and torch's other bundled libraries only work with torch's copy: if this
    library came first (binding /opt/rocm's), a later `import torch` finds "No HIP GPUs".  So when torch is installed and not
    yet imported, map ITS runtime first (without importing torch); the in-tree library then shares it, in either order."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return  # torch's runtime is already mapped
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return  # no torch in this interpreter (e.g. a plain C++ host): the system runtime is the only one
    rt = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(rt):
        C.CDLL(rt, mode=C.RTLD_GLOBAL)


def load_library(path=None):
    """dlopen the in-tree HIP library; raises if it has not been built (no silent fallback).  `path`: another build of the same
    library (tools/build_variant.sh), bound separately and not cached -- same-process A/B timing of kernel variants only."""
    global _lib
    if path is None and _lib is not None:
        return _lib
    lib_path = path or LIB_PATH
    if not os.path.exists(lib_path):
        raise FileNotFoundError(f"{lib_path} not built: run `python -m fasthevc_amd.build` (or __graft_entry__.build())")
    _share_the_hosts_hip_runtime()
    lib = C.CDLL(lib_path)
    vp = C.c_void_p
    lib.fhevc_create.argtypes = [C.POINTER(vp), C.POINTER(Cfg)]
    lib.fhevc_destroy.argtypes = [vp]
    lib.fhevc_destroy.restype = None
    lib.fhevc_set_weights.argtypes = [vp, C.c_char_p, C.c_size_t]
    lib.fhevc_predict_frame.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.fhevc_predict_frame_range.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    lib.fhevc_predict_frames_device_range.argtypes = [vp, vp, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int,
                                                      C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
    lib.fhevc_satd.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    lib.fhevc_intra_first_pass.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    lib.fhevc_intra_first_pass_all.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
    lib.fhevc_intra_first_pass_candidates.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
    lib.fhevc_intra_first_pass_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int,
                                                  C.c_int, vp, vp]
    lib.fhevc_predict_frames_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int,
                                                C.c_int, vp, vp, vp, vp, vp]
    lib.fhevc_expand_depth_flags_device.argtypes = [vp, vp, C.c_int, vp, vp]
    lib.fhevc_band.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.fhevc_aq_parts.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong)]
    lib.fhevc_preanalyze.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
    lib.fhevc_aq_qp.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.fhevc_preanalyze_frames_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int,
                                                   C.c_int, vp, vp]
    lib.fhevc_set_cnn_arith.argtypes = [vp, C.c_int]
    lib.fhevc_set_motion_distortion.argtypes = [vp, C.c_int]
    lib.fhevc_read_yuv_luma.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp,
                                        C.c_longlong, C.c_longlong]
    lib.fhevc_get_cnn_arith.argtypes = [vp]
    lib.fhevc_motion_search.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]
    lib.fhevc_motion_search_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.fhevc_p_rule_default.argtypes = [C.POINTER(PRule)]
    lib.fhevc_p_rule_default.restype = None
    lib.fhevc_p_rule_default_wide.argtypes = [C.POINTER(PRule)]
    lib.fhevc_p_rule_default_wide.restype = None
    lib.fhevc_p_depth_range.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(PRule), vp, vp]
    lib.fhevc_p_motion_compensated_depth.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
    lib.fhevc_p_node_depth.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
    lib.fhevc_predict_frames.argtypes = [vp, vp, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, vp, vp]
    lib.fhevc_alloc_host.argtypes = [vp, C.c_size_t]
    lib.fhevc_alloc_host.restype = vp
    lib.fhevc_free_host.argtypes = [vp, vp]
    lib.fhevc_free_host.restype = None
    lib.fhevc_kernel_timing.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    lib.fhevc_enable_kernel_timing.argtypes = [vp, C.c_int]
    lib.fhevc_get_stats.argtypes = [vp, vp, C.c_size_t]
    lib.fhevc_last_error.argtypes = [vp]
    lib.fhevc_last_error.restype = C.c_char_p
    lib.fhevc_version.restype = C.c_char_p
    if path is None:
        _lib = lib
    return lib


def read_yuv_luma(path, file_size, file_bit_depth, out, first=0, internal_bit_depth=None, chroma_format=420):
    """The library's own luma reader (fhevc_read_yuv_luma, host C++): fills out [count, H, W] (uint8, or int16 Pel) -- e.g. pinned memory
    from Context.alloc_host -- with the luma planes of pictures first.. of a planar YUV file, padded by edge replication to out's
    H x W and shifted to the internal bit depth.  -> number of pictures read."""
    fw, fh = file_size
    count, H, W = out.shape
    assert out.dtype in (np.uint8, np.int16) and out.flags["C_CONTIGUOUS"]
    n = load_library().fhevc_read_yuv_luma(os.fsencode(path), fw, fh, file_bit_depth, chroma_format, first, count, W, H,
                                           internal_bit_depth or file_bit_depth, out.dtype.itemsize, out.ctypes.data, W, W * H)
    if n < 0:
        raise FastHevcError(n, "fhevc_read_yuv_luma")
    return n


def p_rule_default():
    r = PRule()
    load_library().fhevc_p_rule_default(C.byref(r))
    return r


def p_rule_default_wide():
    r = PRule()
    load_library().fhevc_p_rule_default_wide(C.byref(r))
    return r


def p_depth_range(nodes, prev_depth, width, height, qp, rule=None):
    """config 4, host side: (depth_min, depth_max) [numCtus, 256] of a P picture from its motion nodes [numCtus, 85] and the
    co-located depths [numCtus, 256] of its reference picture"""
    lib = load_library()
    rule = rule if rule is not None else p_rule_default()
    nodes = np.ascontiguousarray(nodes)
    prev = np.ascontiguousarray(prev_depth, np.uint8)
    cw = (width + 63) // 64
    n = nodes.shape[0]
    dmin, dmax = np.zeros((n, 256), np.uint8), np.zeros((n, 256), np.uint8)
    for c in range(n):
        vw, vh = min(64, width - (c % cw) * 64), min(64, height - (c // cw) * 64)
        rc = lib.fhevc_p_depth_range(nodes[c].ctypes.data, prev[c].ctypes.data, vw, vh, qp, C.byref(rule), dmin[c].ctypes.data, dmax[c].ctypes.data)
        if rc != OK:
            raise FastHevcError(rc, "fhevc_p_depth_range")
    return dmin, dmax


def p_motion_compensated_depth(nodes, prev_map, width, height):
    """config 4, host side: the reference picture's depths [numCtus, 256] seen through the motion nodes [numCtus, 85] of the current picture
    (the prev_depth argument of p_depth_range for content that moves)"""
    lib = load_library()
    nodes = np.ascontiguousarray(nodes)
    prev = np.ascontiguousarray(prev_map, np.uint8)
    out = np.zeros((nodes.shape[0], 256), np.uint8)
    for c in range(nodes.shape[0]):
        rc = lib.fhevc_p_motion_compensated_depth(nodes[c].ctypes.data, prev.ctypes.data, width, height, c, out[c].ctypes.data)
        if rc != OK:
            raise FastHevcError(rc, "fhevc_p_motion_compensated_depth")
    return out


def p_node_depth(nodes, prev_map, width, height):
    """config 4, host side: the reference picture's depths seen through the motion, asked per CU node of the current picture's grid (a partition of
    every CTU: fhevc_p_node_depth)"""
    lib = load_library()
    nodes = np.ascontiguousarray(nodes)
    prev = np.ascontiguousarray(prev_map, np.uint8)
    out = np.zeros((nodes.shape[0], 256), np.uint8)
    for c in range(nodes.shape[0]):
        rc = lib.fhevc_p_node_depth(nodes[c].ctypes.data, prev.ctypes.data, width, height, c, out[c].ctypes.data)
        if rc != OK:
            raise FastHevcError(rc, "fhevc_p_node_depth")
    return out


def band(ctu_rows, rank, world):
    b, e = C.c_int(), C.c_int()
    rc = load_library().fhevc_band(ctu_rows, rank, world, C.byref(b), C.byref(e))
    if rc != OK:
        raise FastHevcError(rc, "fhevc_band")
    return b.value, e.value


class Context:
    """One fhevc_ctx: one picture geometry on one MI355X."""

    def __init__(self, width, height, bit_depth=8, weights=None, device=0, max_frames=1, arith=None, lib_path=None, devices=None):
        """devices: list of HIP ordinals for a multi-device context (the host-buffer entry points shard over them); default [device]"""
        self.lib = load_library(lib_path)
        self.width, self.height, self.bit_depth = width, height, bit_depth
        self.ctus_x, self.ctus_y = (width + 63) // 64, (height + 63) // 64
        self.num_ctus = self.ctus_x * self.ctus_y
        ids = list(devices) if devices else [device]
        dev = (C.c_int * len(ids))(*ids)
        cfg = Cfg(width, height, bit_depth, 64, 3, len(ids), dev, None, BACKEND_HIP, max_frames)
        h = C.c_void_p()
        rc = self.lib.fhevc_create(C.byref(h), C.byref(cfg))
        if rc != OK:
            raise FastHevcError(rc, "fhevc_create (no gfx950 device?)" if rc == E_NO_DEVICE else "fhevc_create")
        self.h = h
        if arith is not None:
            self.set_cnn_arith(arith)
        if weights is not None:
            self.set_weights(weights)

    def _check(self, rc):
        if rc != OK:
            raise FastHevcError(rc, self.lib.fhevc_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.fhevc_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_weights(self, w):
        blob = w if isinstance(w, (bytes, bytearray)) else (_weights.pack_family(w) if "widths" in w else _weights.pack(w))
        self._check(self.lib.fhevc_set_weights(self.h, bytes(blob), len(blob)))

    def predict_frame(self, plane, origin=0, stride=None, qp=32, slice_type=2, want_hadamard=True):
        """plane: int16 numpy buffer holding a Pel plane; origin = element offset of sample (0,0)."""
        flat = np.ascontiguousarray(plane).reshape(-1)
        assert flat.dtype == np.int16
        stride = stride if stride is not None else plane.shape[-1]
        depth = np.zeros(self.num_ctus * 256, np.uint8)
        had = np.zeros(self.num_ctus, np.int32) if want_hadamard else None
        self._check(self.lib.fhevc_predict_frame(self.h, flat.ctypes.data + 2 * origin, stride, qp, slice_type,
                                                 depth.ctypes.data, had.ctypes.data if want_hadamard else None))
        return depth.reshape(self.num_ctus, 256), had

    def alloc_host(self, shape, dtype):
        """numpy array over pinned host memory of the library (fhevc_alloc_host); release with free_host(array)"""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        p = self.lib.fhevc_alloc_host(self.h, nbytes)
        if not p:
            raise FastHevcError(E_NOMEM, "fhevc_alloc_host")
        arr = np.frombuffer((C.c_uint8 * nbytes).from_address(p), dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def free_host(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p:
            self.lib.fhevc_free_host(self.h, p)

    def predict_frames(self, luma, qp=32, want_hadamard=True, depth_out=None, had_out=None, origin=0, stride=None, frame_stride=None):
        """Host batch: luma [F, H, W] uint8 (8-bit content) or an int16 Pel buffer with origin/stride/frame_stride; returns
        (depth [F, numCtus, 256], hadamard [F, numCtus] or None) in host memory."""
        arr = np.asarray(luma)
        if arr.dtype == np.uint8:
            assert arr.ndim == 3 and arr.shape[1:] == (self.height, self.width) and arr.flags.c_contiguous
            nf, sb, st, fst, ptr = arr.shape[0], 1, self.width, self.width * self.height, arr.ctypes.data
        else:
            assert arr.dtype == np.int16 and arr.flags.c_contiguous and stride is not None and frame_stride is not None
            nf, sb, st, fst, ptr = arr.shape[0], 2, stride, frame_stride, arr.ctypes.data + 2 * origin
        depth = depth_out if depth_out is not None else np.zeros((nf, self.num_ctus, 256), np.uint8)
        had = had_out if had_out is not None else (np.zeros((nf, self.num_ctus), np.int32) if want_hadamard else None)
        self._check(self.lib.fhevc_predict_frames(self.h, ptr, sb, st, fst, nf, qp, depth.ctypes.data, had.ctypes.data if had is not None else None))
        return depth, had

    def predict_frame_range(self, plane, origin=0, stride=None, qp=32, margin=0, slice_type=2, margin_stop=None, with_hadamard=False):
        """Soft decisions: (depth_min, depth_max), each [numCtus, 256] (and the per-CTU source Hadamard when asked for)."""
        flat = np.ascontiguousarray(plane).reshape(-1)
        assert flat.dtype == np.int16
        stride = stride if stride is not None else plane.shape[-1]
        dmin = np.zeros(self.num_ctus * 256, np.uint8)
        dmax = np.zeros(self.num_ctus * 256, np.uint8)
        had = np.zeros(self.num_ctus, np.int32) if with_hadamard else None
        self._check(self.lib.fhevc_predict_frame_range(self.h, flat.ctypes.data + 2 * origin, stride, qp, slice_type, margin,
                                                       margin if margin_stop is None else margin_stop,
                                                       dmin.ctypes.data, dmax.ctypes.data, had.ctypes.data if with_hadamard else None))
        out = (dmin.reshape(self.num_ctus, 256), dmax.reshape(self.num_ctus, 256))
        return out + (had,) if with_hadamard else out

    def satd(self, org, cur, w, h, bit_depth=8, org_stride=None, cur_stride=None):
        org = np.ascontiguousarray(org, np.int16)
        cur = np.ascontiguousarray(cur, np.int16)
        out = C.c_uint32()
        self._check(self.lib.fhevc_satd(self.h, org.ctypes.data, org_stride or org.shape[-1], cur.ctypes.data,
                                        cur_stride or cur.shape[-1], w, h, bit_depth, C.byref(out)))
        return out.value

    def intra_first_pass(self, plane, origin=0, stride=None, qp=32):
        flat = np.ascontiguousarray(plane).reshape(-1)
        stride = stride if stride is not None else plane.shape[-1]
        out = np.zeros(self.num_ctus * NODES_PER_CTU, NODE_DTYPE)
        self._check(self.lib.fhevc_intra_first_pass(self.h, flat.ctypes.data + 2 * origin, stride, qp, out.ctypes.data))
        return out.reshape(self.num_ctus, NODES_PER_CTU)

    def motion_search(self, cur_plane, ref_plane, origin=0, stride=None, qp=32, search_range=4):
        """config 4: [numCtus, 85] MOTION_DTYPE nodes of cur searched in ref (two Pel planes of the same layout)."""
        cur = np.ascontiguousarray(cur_plane).reshape(-1)
        ref = np.ascontiguousarray(ref_plane).reshape(-1)
        assert cur.dtype == np.int16 and ref.dtype == np.int16
        stride = stride if stride is not None else cur_plane.shape[-1]
        out = np.zeros(self.num_ctus * NODES_PER_CTU, MOTION_DTYPE)
        self._check(self.lib.fhevc_motion_search(self.h, cur.ctypes.data + 2 * origin, ref.ctypes.data + 2 * origin, stride, qp,
                                                 search_range, out.ctypes.data))
        return out.reshape(self.num_ctus, NODES_PER_CTU)

    def motion_search_device(self, d_luma, sample_bytes, stride, frame_stride, num_frames, d_out, rows=None, stream=None, qp=32,
                             search_range=4):
        """frames 1.. of the batch, each searched in the frame before it; d_out: (num_frames - 1) * band CTUs * 85 nodes (16 B)."""
        rb, re = rows if rows is not None else (0, self.ctus_y)
        self._check(self.lib.fhevc_motion_search_device(self.h, d_luma, sample_bytes, stride, frame_stride, num_frames, rb, re, qp,
                                                        search_range, d_out, stream))

    def intra_first_pass_all(self, plane, origin=0, stride=None, qp=32):
        """(best [numCtus, 85], all [numCtus, 85, 35]): every mode's SATD and cost per node (parity entry point)"""
        flat = np.ascontiguousarray(plane).reshape(-1)
        stride = stride if stride is not None else plane.shape[-1]
        best = np.zeros(self.num_ctus * NODES_PER_CTU, NODE_DTYPE)
        allm = np.zeros(self.num_ctus * NODES_PER_CTU * 35, NODE_DTYPE)
        self._check(self.lib.fhevc_intra_first_pass_all(self.h, flat.ctypes.data + 2 * origin, stride, qp, best.ctypes.data, allm.ctypes.data))
        return best.reshape(self.num_ctus, NODES_PER_CTU), allm.reshape(self.num_ctus, NODES_PER_CTU, 35)

    def intra_first_pass_candidates(self, plane, origin=0, stride=None, qp=32, num_candidates=8):
        """[numCtus, 85, num_candidates] uint8: per node the modes of smallest first-pass cost, best first (HM's candidate list)"""
        flat = np.ascontiguousarray(plane).reshape(-1)
        stride = stride if stride is not None else plane.shape[-1]
        out = np.zeros(self.num_ctus * NODES_PER_CTU * num_candidates, np.uint8)
        self._check(self.lib.fhevc_intra_first_pass_candidates(self.h, flat.ctypes.data + 2 * origin, stride, qp, num_candidates, out.ctypes.data))
        return out.reshape(self.num_ctus, NODES_PER_CTU, num_candidates)

    def aq_layout(self, max_aq_depth):
        """Offsets of the AQ layers in the concatenated activity array (max_aq_depth + 1 entries)."""
        off = (C.c_longlong * (max_aq_depth + 1))()
        n = self.lib.fhevc_aq_parts(self.width, self.height, max_aq_depth, off)
        self._check(min(n, 0))
        return list(off)

    def preanalyze(self, plane, origin=0, stride=None, max_aq_depth=3):
        """TEncPreanalyzer::xPreanalyze: (activity of all layers concatenated, per-layer averages)."""
        flat = np.ascontiguousarray(plane).reshape(-1)
        stride = stride if stride is not None else plane.shape[-1]
        off = self.aq_layout(max_aq_depth)
        act = np.zeros(off[-1], np.float64)
        avg = np.zeros(max_aq_depth, np.float64)
        self._check(self.lib.fhevc_preanalyze(self.h, flat.ctypes.data + 2 * origin, stride, max_aq_depth,
                                              act.ctypes.data, avg.ctypes.data))
        return act, avg

    def aq_qp(self, activity, avg_activity, qp_adaptation_range=6, base_qp=32):
        """TEncCu::xComputeQP for every AQ part (host side; no device work)."""
        activity = np.ascontiguousarray(activity, np.float64)
        avg_activity = np.ascontiguousarray(avg_activity, np.float64)
        out = np.zeros(activity.size, np.int8)
        self._check(self.lib.fhevc_aq_qp(activity.ctypes.data, avg_activity.ctypes.data, self.width, self.height,
                                         avg_activity.size, qp_adaptation_range, base_qp, 6 * (self.bit_depth - 8),
                                         out.ctypes.data))
        return out

    def preanalyze_frames_device(self, d_luma, sample_bytes, stride, frame_stride, num_frames, d_activity,
                                 max_aq_depth=3, rows=None, stream=None):
        rb, re = rows if rows is not None else (0, self.ctus_y)
        self._check(self.lib.fhevc_preanalyze_frames_device(self.h, d_luma, sample_bytes, stride, frame_stride,
                                                            num_frames, rb, re, max_aq_depth, d_activity, stream))

    def intra_first_pass_device(self, d_luma, sample_bytes, stride, frame_stride, num_frames, d_out, rows=None,
                                stream=None, qp=32):
        """d_out: device buffer of num_frames * band CTUs * 85 NODE_DTYPE entries (16 bytes each); asynchronous."""
        rb, re = rows if rows is not None else (0, self.ctus_y)
        self._check(self.lib.fhevc_intra_first_pass_device(self.h, d_luma, sample_bytes, stride, frame_stride, num_frames,
                                                           rb, re, qp, d_out, stream))

    def predict_frames_device(self, d_luma, sample_bytes, stride, frame_stride, num_frames, d_depth, d_hadamard=None,
                              d_logits=None, rows=None, stream=None, qp=32, d_flags=None):
        """All pointers are raw device addresses (e.g. torch.Tensor.data_ptr()); asynchronous."""
        rb, re = rows if rows is not None else (0, self.ctus_y)
        self._check(self.lib.fhevc_predict_frames_device(self.h, d_luma, sample_bytes, stride, frame_stride, num_frames,
                                                         rb, re, qp, d_depth, d_hadamard, d_logits, d_flags, stream))

    def expand_depth_flags_device(self, d_flags, num_frames, d_depth, stream=None):
        self._check(self.lib.fhevc_expand_depth_flags_device(self.h, d_flags, num_frames, d_depth, stream))

    def set_motion_distortion(self, mode):
        """"satd" (default) or "sad" (HM's integer-search distortion: results equal the reference's xPatternSearch)"""
        self._check(self.lib.fhevc_set_motion_distortion(self.h, {"satd": 0, "sad": 1}[mode]))

    def set_cnn_arith(self, arith):
        """"i8" (default) or "f16": the arithmetic of the classifier's conv2 / conv3; the results are the same integers"""
        self._check(self.lib.fhevc_set_cnn_arith(self.h, CNN_ARITH[arith]))

    @property
    def cnn_arith(self):
        v = self.lib.fhevc_get_cnn_arith(self.h)
        return {8: "i8", 16: "f16"}[v]

    def enable_kernel_timing(self, on=True):
        self._check(self.lib.fhevc_enable_kernel_timing(self.h, 1 if on else 0))

    def kernel_timing(self, which, reset=False):
        ms, n = C.c_double(), C.c_uint64()
        self._check(self.lib.fhevc_kernel_timing(self.h, which, 1 if reset else 0, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def stats(self):
        s = Stats()
        self._check(self.lib.fhevc_get_stats(self.h, C.byref(s), C.sizeof(s)))
        return {k: getattr(s, k) for k, _ in Stats._fields_}

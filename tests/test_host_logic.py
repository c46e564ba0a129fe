"""CPU tests of the host-side logic: synthetic frames, weight blob, band arithmetic, C-ABI surface."""
import hashlib
import os
import re
import subprocess

import numpy as np
import pytest

from fasthevc_amd import capi, frames, weights
from fasthevc_amd import gather as bands

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synthetic_frames_are_pinned():
    # md5 of the 4:2:0 files, SURVEY.md section 8(d)
    assert hashlib.md5(frames.texture16_yuv420(416, 240)).hexdigest() == "913977ec4bc414503cefc9e5a8d0bcc3"
    assert hashlib.md5(frames.hetero_yuv420(1920, 1080)).hexdigest() == "d2a7dee5ef67252b527b2f66fcaed0ca"
    # the further training / evaluation families (tests/quality): pinned so that labels and BD-rate numbers stay reproducible
    assert hashlib.md5(frames.fractal_luma(416, 240, seed=424242).tobytes()).hexdigest() == "60d4b75bd42ccf9a10193c3b4a792c8b"
    assert hashlib.md5(frames.gratings_luma(416, 240, seed=424242).tobytes()).hexdigest() == "72c78d42644f4230606f504b0c9d332c"
    assert hashlib.md5(frames.polygon_luma(416, 240, seed=424242).tobytes()).hexdigest() == "05b8769a68e6b1a5c02877cd0ba0e4c5"


def test_pel_plane_layout():
    y = frames.texture16_luma(416, 240)
    buf, org, stride = frames.to_pel_plane(y, 10)
    assert stride == 416 + 160 and org == 80 * stride + 80  # TComPicYuv.cpp:94-103
    assert buf.reshape(-1)[org] == int(y[0, 0]) << 2 and buf[80 + 239, 80 + 415] == int(y[239, 415]) << 2


def test_weight_blob_roundtrip(tmp_path):
    w = weights.random_weights(3)
    blob = weights.pack(w)
    assert len(blob) == weights.BLOB_BYTES
    back = weights.unpack(blob)
    for k in w:
        assert np.array_equal(w[k], back[k])
    p = tmp_path / "w.fhw"
    weights.save(p, w)
    assert weights.pack(weights.load(p)) == blob
    bad = dict(w)
    bad["w2"] = w["w2"].copy()
    bad["w2"][0, 0, 0, 0] = -128
    with pytest.raises(ValueError):
        weights.pack(bad)
    with pytest.raises(ValueError):
        weights.unpack(blob[:-1])


def test_band_partition_covers_rows():
    for rows in (4, 17, 34):
        for world in (1, 2, 3, 4, 8):
            prev = 0
            for r in range(world):
                b, e = bands.band(rows, r, world)
                assert b == prev and e >= b
                prev = e
                assert (b, e) == capi.band(rows, r, world)  # same arithmetic behind the C ABI
            assert prev == rows
    assert bands.max_band_rows(34, 8) == 5 and bands.max_band_rows(17, 8) == 3


def test_c_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "fasthevc.h")).read()
    declared = sorted(set(re.findall(r"\b(fhevc_[a-z_0-9]+)\s*\(", header)))
    assert declared == sorted(capi.SYMBOLS)
    assert os.path.exists(capi.LIB_PATH), "HIP library not built (run __graft_entry__.build())"
    exported = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH]).decode()
    lib = capi.load_library()
    for sym in declared:
        assert re.search(rf"\bT {sym}\b", exported), sym
        assert getattr(lib, sym) is not None
    assert b"gfx950" in lib.fhevc_version()


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_cpu_fallback():
    # the product path must fail loudly without a gfx950 device
    with pytest.raises(capi.FastHevcError) as e:
        capi.Context(416, 240, 8, weights.random_weights(0))
    assert e.value.code == capi.E_NO_DEVICE


def test_struct_sizes_match_header():
    assert capi.NODE_DTYPE.itemsize == 16
    import ctypes
    assert ctypes.sizeof(capi.NodeCost) == 16


def test_p_motion_compensated_depth_equals_the_oracle(oracle):
    """Host-side logic behind the C ABI (no GPU): the reference picture's depths seen through the motion nodes of the current picture,
    on a ragged picture, with vectors up to +-64, nodes flagged as crossing the picture edge (the enclosing node's vector counts), and
    zero motion = the co-located map."""
    import ctypes as C
    W, H = 416, 240
    cw, ch = 7, 4
    rng = np.random.default_rng(11)
    prev = rng.integers(0, 4, size=(cw * ch, 256)).astype(np.uint8)
    nodes = np.zeros((cw * ch, 85), capi.MOTION_DTYPE)
    nodes["mvx"] = rng.integers(-64, 65, size=nodes.shape)
    nodes["mvy"] = rng.integers(-64, 65, size=nodes.shape)
    nodes["cost_best"] = rng.integers(0, 1000, size=nodes.shape)
    flag = rng.random(nodes.shape) < 0.2
    nodes["cost_best"][flag] = 0xFFFFFFFF
    got = capi.p_motion_compensated_depth(nodes, prev, W, H)
    exp = np.zeros_like(got)
    oracle.fho_p_motion_compensated_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    for c in range(cw * ch):
        oracle.fho_p_motion_compensated_depth(nodes[c].ctypes.data, prev.ctypes.data, W, H, c, exp[c].ctypes.data)
    assert np.array_equal(got, exp)
    assert (got != prev).any()
    nodes["mvx"] = 0
    nodes["mvy"] = 0
    still = capi.p_motion_compensated_depth(nodes, prev, W, H)
    inside = np.zeros((cw * ch, 16, 16), bool)   # units whose centre lies inside the picture
    for c in range(cw * ch):
        inside[c, :min(16, (H - (c // cw) * 64) // 4), :min(16, (W - (c % cw) * 64) // 4)] = True
    assert np.array_equal(still.reshape(-1, 16, 16)[inside], prev.reshape(-1, 16, 16)[inside])


def test_p_node_depth_equals_the_oracle_and_is_a_partition(oracle):
    """Host-side logic behind the C ABI (no GPU): the reference picture's depths asked per CU NODE of the current grid (fhevc_p_node_depth): equals the
    oracle's recursive restatement on a ragged picture with vectors up to +-64 and edge-crossing nodes; the result is a quadtree-consistent partition
    (constant over every CU it declares); zero motion over a map that is a partition gives the map back."""
    import ctypes as C
    W, H = 416, 240
    cw, ch = 7, 4
    rng = np.random.default_rng(12)

    def random_partition():   # a valid HM depth map per CTU: top-down random splits
        m = np.zeros((cw * ch, 16, 16), np.uint8)
        for c in range(cw * ch):
            if rng.random() < 0.3:
                continue
            for q in range(4):
                qy, qx = 8 * (q >> 1), 8 * (q & 1)
                if rng.random() < 0.4:
                    m[c, qy:qy + 8, qx:qx + 8] = 1
                    continue
                for b in range(4):
                    by, bx = qy + 4 * (b >> 1), qx + 4 * (b & 1)
                    m[c, by:by + 4, bx:bx + 4] = 2 if rng.random() < 0.5 else 3
        return m.reshape(cw * ch, 256)

    prev = random_partition()
    nodes = np.zeros((cw * ch, 85), capi.MOTION_DTYPE)
    nodes["mvx"] = rng.integers(-64, 65, size=nodes.shape)
    nodes["mvy"] = rng.integers(-64, 65, size=nodes.shape)
    nodes["cost_best"] = rng.integers(0, 1000, size=nodes.shape)
    nodes["cost_best"][rng.random(nodes.shape) < 0.2] = 0xFFFFFFFF
    got = capi.p_node_depth(nodes, prev, W, H)
    exp = np.zeros_like(got)
    oracle.fho_p_node_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    for c in range(cw * ch):
        oracle.fho_p_node_depth(nodes[c].ctypes.data, prev.ctypes.data, W, H, c, exp[c].ctypes.data)
    assert np.array_equal(got, exp)
    assert (got != prev).any()
    g = got.reshape(-1, 16, 16)
    for c in range(cw * ch):   # every unit's depth d sits in a CU of (64 >> d) samples that is constant
        for uy in range(16):
            for ux in range(16):
                d = int(g[c, uy, ux])
                u = max(16 >> d, 2) if d < 3 else 2     # depth 3 = 8x8 CUs: 2x2 units; they come four at a time (a 16x16 node)
                y0, x0 = uy // u * u, ux // u * u
                assert (g[c, y0:y0 + u, x0:x0 + u] == d).all() or d == 3
    nodes["mvx"] = 0
    nodes["mvy"] = 0
    still = capi.p_node_depth(nodes, prev, W, H)
    inside = np.zeros((cw * ch, 16, 16), bool)   # CTUs wholly inside the picture (a node that leaves it asks a clamped position)
    for c in range(cw * ch):
        if (c % cw) * 64 + 64 <= W and (c // cw) * 64 + 64 <= H:
            inside[c] = True
    assert np.array_equal(still.reshape(-1, 16, 16)[inside], prev.reshape(-1, 16, 16)[inside])


def test_cost_sensitive_training_utilities():
    """fasthevc_amd/train/train.py: the z-order -> raster mapping of the recorded node costs, and the tree regret (0 for the cheapest tree, the cost
    difference for a single wrong decision)."""
    from fasthevc_amd.train import train
    rng = np.random.default_rng(5)
    cost = rng.uniform(100, 1000, size=(6, 21, 2)).astype(np.float32)
    g = train.cost_grids(cost)
    # node 5 + b (z-order) = block (by, bx): b = 4 q + s, by = 2 (q >> 1) + (s >> 1), bx = 2 (q & 1) + (s & 1)
    for b in range(16):
        q, s = b >> 2, b & 3
        assert np.array_equal(g["ns16"][:, 2 * (q >> 1) + (s >> 1), 2 * (q & 1) + (s & 1)], cost[:, 5 + b, 0])
    for q in range(4):
        assert np.array_equal(g["sp32"][:, q >> 1, q & 1], cost[:, 1 + q, 1])
    # the cheapest tree under the metric's own approximation, bottom-up
    o16 = g["sp16"] < g["ns16"]
    j16 = np.where(o16, g["sp16"], g["ns16"]).reshape(-1, 2, 2, 2, 2).sum(axis=(2, 4))
    o32 = j16 < g["ns32"]
    o64 = np.where(o32, j16, g["ns32"]).sum(axis=(1, 2)) < g["ns64"]
    chosen, best = train.tree_regret(o64, o32, o16, g)
    assert abs(chosen - best) < 1e-3 * best
    worse, _ = train.tree_regret(~o64, o32, o16, g)
    assert worse > best
    flipped = train.flip_costs(g)
    assert np.array_equal(flipped["ns16"][:, :, ::-1], g["ns16"]) and np.array_equal(flipped["ns64"], g["ns64"])

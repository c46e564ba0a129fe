"""Depth-classifier weight blob ("FHW1"): fixed-point integer weights of the three conv layers and the three
FC heads (HISTORY.md section 4).  The same bytes feed the HIP library (fhevc_cfg.weights_path) and, in tests,
the CPU oracle.

Layout after the 8-byte header (magic "FHW1", uint32 version = 2), all little-endian, no padding:
  int32 shift[3]
  int8 w1[16][3][3]         int32 b1[16]
  int8 w2[32][16][3][3]     int32 b2[32]
  int8 w3[64][32][3][3]     int32 b3[64]
  int8 wh64[2][8][8][64]    int32 bh64[2]      (64-level head, on the 2x2 sum-pooled conv3 map)
  int8 wh32[2][8][8][64]    int32 bh32[2]
  int8 wh16[2][4][4][64]    int32 bh16[2]
  int32 qp_bias[3][52]      added to the "split" logit of the 64-, 32-, 16-level heads for slice QP 0..51
"""
import numpy as np

MAGIC = b"FHW1"
FIELDS = [
    ("shift", np.int32, (3,)),
    ("w1", np.int8, (16, 3, 3)), ("b1", np.int32, (16,)),
    ("w2", np.int8, (32, 16, 3, 3)), ("b2", np.int32, (32,)),
    ("w3", np.int8, (64, 32, 3, 3)), ("b3", np.int32, (64,)),
    ("wh64", np.int8, (2, 8, 8, 64)), ("bh64", np.int32, (2,)),
    ("wh32", np.int8, (2, 8, 8, 64)), ("bh32", np.int32, (2,)),
    ("wh16", np.int8, (2, 4, 4, 64)), ("bh16", np.int32, (2,)),
    ("qp_bias", np.int32, (3, 52)),
]
BIAS_LIMIT = 1 << 22  # keeps |bias + sum(w*a)| < 2^24 so fp32 accumulation of bf16 operands is exact
BLOB_BYTES = 8 + sum(int(np.prod(s)) * np.dtype(t).itemsize for _, t, s in FIELDS)


def validate(w):
    for name, dt, shape in FIELDS:
        a = np.asarray(w[name])
        if a.shape != shape or a.dtype != dt:
            raise ValueError(f"{name}: expected {dt} {shape}, got {a.dtype} {a.shape}")
    for k in ("w1", "w2", "w3", "wh64", "wh32", "wh16"):
        if np.any(w[k] == -128):
            raise ValueError(f"{k}: -128 is not allowed (weights are symmetric int8, |w| <= 127)")
    for k in ("b1", "b2", "b3"):
        if np.any(np.abs(w[k].astype(np.int64)) > BIAS_LIMIT):
            raise ValueError(f"{k}: |bias| must be <= 2^22")
    if np.any(w["shift"] < 0) or np.any(w["shift"] > 14):
        raise ValueError("shift out of range (0..14)")


def pack(w):
    validate(w)
    parts = [MAGIC, np.uint32(2).tobytes()]
    for name, dt, _ in FIELDS:
        parts.append(np.ascontiguousarray(w[name], dtype=dt).tobytes())
    blob = b"".join(parts)
    assert len(blob) == BLOB_BYTES
    return blob


def unpack(blob):
    if len(blob) != BLOB_BYTES or blob[:4] != MAGIC:
        raise ValueError("not an FHW1 weight blob")
    if int(np.frombuffer(blob[4:8], np.uint32)[0]) != 2:
        raise ValueError("unsupported FHW1 version")
    off, out = 8, {}
    for name, dt, shape in FIELDS:
        n = int(np.prod(shape)) * np.dtype(dt).itemsize
        out[name] = np.frombuffer(blob[off:off + n], dt).reshape(shape).copy()
        off += n
    validate(out)
    return out


def save(path, w):
    with open(path, "wb") as f:
        f.write(pack(w))


def load(path):
    with open(path, "rb") as f:
        return unpack(f.read())


def random_weights(seed=0, extreme=False):
    """Random-init weights of the right architecture (bench/test use; no training implied).
    extreme=True: maximum-magnitude weights, to probe the 2^24 exactness bound."""
    rng = np.random.default_rng(seed)
    w = {}
    if extreme:
        for name, dt, shape in FIELDS:
            if dt == np.int8:
                w[name] = rng.choice(np.array([-127, 127], np.int8), size=shape)
            elif name == "shift":
                w[name] = np.array([4, 12, 13], np.int32)
            elif name == "qp_bias":
                w[name] = rng.integers(-100000, 100000, size=shape).astype(np.int32)
            else:
                w[name] = rng.integers(-1000, 1000, size=shape).astype(np.int32)
        return w
    w["shift"] = np.array([6, 7, 8], np.int32)
    w["w1"] = rng.integers(-64, 65, size=(16, 3, 3)).astype(np.int8)
    w["b1"] = rng.integers(-2000, 2000, size=16).astype(np.int32)
    w["w2"] = rng.integers(-32, 33, size=(32, 16, 3, 3)).astype(np.int8)
    w["b2"] = rng.integers(-6000, 6000, size=32).astype(np.int32)
    w["w3"] = rng.integers(-32, 33, size=(64, 32, 3, 3)).astype(np.int8)
    w["b3"] = rng.integers(-9000, 9000, size=64).astype(np.int32)
    for k, shape in (("wh64", (2, 8, 8, 64)), ("wh32", (2, 8, 8, 64)), ("wh16", (2, 4, 4, 64))):
        w[k] = rng.integers(-64, 65, size=shape).astype(np.int8)
    for k in ("bh64", "bh32", "bh16"):
        w[k] = rng.integers(-50000, 50000, size=2).astype(np.int32)
    w["qp_bias"] = rng.integers(-200000, 200000, size=(3, 52)).astype(np.int32)
    return w


# ---- the reference's Bayesian-optimisation network family ("FHW3") -----------------------------------------------------------------
# matlab/dataExtraction/OptimizeDeepNeuralNetworksUsingBayesianOptimizationExample.m:103-106, 233-259, 367-373: three blocks of `depth`
# convolutions (conv3x3 pad 1 + BN + ReLU) with F = round(32 / sqrt(depth)), 2F, 4F filters, max-pool after blocks 1 and 2, FC.
# Layout after the 24-byte header (magic "FHW3", uint32 version = 1, int32 c1, c2, c3, depth), little-endian, no padding:
#   int32 shift[3][3]                      (entries [b][j >= depth] are 0)
#   per block b, per convolution j < depth: int8 w[co][ci][3][3], int32 bias[co]   (ci = 1 | the block's width | the previous block's)
#   int8 wh64[2][8][8][c3] int32 bh64[2]   int8 wh32[2][8][8][c3] int32 bh32[2]   int8 wh16[2][4][4][c3] int32 bh16[2]
#   int32 qp_bias[3][52]
FAMILY_MAGIC = b"FHW3"


def family_widths(depth):
    """channel widths of the reference's family member with `depth` convolutions per block: F = round(32 / sqrt(depth)), 2F, 4F"""
    f = int(round(32.0 / np.sqrt(depth)))
    return (f, 2 * f, 4 * f)


def family_fields(c, depth):
    fields = [("shift", np.int32, (3, 3))]
    ci = 1
    for b in range(3):
        for j in range(depth):
            fields += [(f"w{b}{j}", np.int8, (c[b], ci, 3, 3)), (f"b{b}{j}", np.int32, (c[b],))]
            ci = c[b]
    fields += [("wh64", np.int8, (2, 8, 8, c[2])), ("bh64", np.int32, (2,)), ("wh32", np.int8, (2, 8, 8, c[2])), ("bh32", np.int32, (2,)),
               ("wh16", np.int8, (2, 4, 4, c[2])), ("bh16", np.int32, (2,)), ("qp_bias", np.int32, (3, 52))]
    return fields


def pack_family(w):
    c, depth = tuple(int(v) for v in w["widths"]), int(w["depth"])
    parts = [FAMILY_MAGIC, np.uint32(1).tobytes(), np.array(list(c) + [depth], np.int32).tobytes()]
    for name, dt, shape in family_fields(c, depth):
        a = np.ascontiguousarray(w[name], dtype=dt)
        if a.shape != shape:
            raise ValueError(f"{name}: expected {shape}, got {a.shape}")
        if dt == np.int8 and np.any(a == -128):
            raise ValueError(f"{name}: -128 is not allowed")
        parts.append(a.tobytes())
    return b"".join(parts)


def unpack_family(blob):
    if blob[:4] != FAMILY_MAGIC or int(np.frombuffer(blob[4:8], np.uint32)[0]) != 1:
        raise ValueError("not an FHW3 weight blob")
    hdr = np.frombuffer(blob[8:24], np.int32)
    c, depth = tuple(int(v) for v in hdr[:3]), int(hdr[3])
    off, out = 24, {"widths": np.array(c, np.int32), "depth": depth}
    for name, dt, shape in family_fields(c, depth):
        n = int(np.prod(shape)) * np.dtype(dt).itemsize
        out[name] = np.frombuffer(blob[off:off + n], dt).reshape(shape).copy()
        off += n
    if off != len(blob):
        raise ValueError("FHW3 blob has the wrong size")
    return out


def load_any(path):
    """FHW1 (the 16 / 32 / 64 network) or FHW3 (a member of the reference's Bayesian-optimisation family)"""
    with open(path, "rb") as f:
        blob = f.read()
    return unpack_family(blob) if blob[:4] == FAMILY_MAGIC else unpack(blob)


def family_from_base(w):
    """the 16 / 32 / 64 network as a depth-1 family member (tests: the family code paths must reproduce the base network)"""
    out = {"widths": np.array([16, 32, 64], np.int32), "depth": 1, "shift": np.zeros((3, 3), np.int32)}
    out["shift"][:, 0] = w["shift"]
    out["w00"], out["b00"] = w["w1"].reshape(16, 1, 3, 3), w["b1"]
    out["w10"], out["b10"] = w["w2"], w["b2"]
    out["w20"], out["b20"] = w["w3"], w["b3"]
    for k in ("wh64", "bh64", "wh32", "bh32", "wh16", "bh16", "qp_bias"):
        out[k] = w[k]
    return out


def random_family(widths=(32, 64, 128), depth=1, seed=0):
    rng = np.random.default_rng(seed)
    c = tuple(widths)
    w = {"widths": np.array(c, np.int32), "depth": depth, "shift": np.zeros((3, 3), np.int32)}
    ci = 1
    for b in range(3):
        for j in range(depth):
            fan = 9 * ci
            amp = max(4, min(64, int(round(400.0 / np.sqrt(fan)))))
            w[f"w{b}{j}"] = rng.integers(-amp, amp + 1, size=(c[b], ci, 3, 3)).astype(np.int8)
            w[f"b{b}{j}"] = rng.integers(-4000, 4000, size=c[b]).astype(np.int32)
            w["shift"][b, j] = 6 if (b == 0 and j == 0) else (7 if fan <= 160 else 8)
            ci = c[b]
    for k, shape in (("wh64", (2, 8, 8, c[2])), ("wh32", (2, 8, 8, c[2])), ("wh16", (2, 4, 4, c[2]))):
        w[k] = rng.integers(-64, 65, size=shape).astype(np.int8)
    for k in ("bh64", "bh32", "bh16"):
        w[k] = rng.integers(-50000, 50000, size=2).astype(np.int32)
    w["qp_bias"] = rng.integers(-200000, 200000, size=(3, 52)).astype(np.int32)
    return w

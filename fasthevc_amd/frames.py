"""Seeded synthetic luma frames and CTU-grid helpers.

The two generators are the measurement recipes fixed in SURVEY.md section 8(d) / Appendix C
("texture16", seed 1234, and "hetero", seed 4321); the md5 of the 4:2:0 files they produce is pinned in
tests/test_frames.py.  The reference holds no YUV material, so every benchmark and
parity input in this repository comes from here.
"""
import numpy as np

CTU = 64
HM_MARGIN = 80  # TComPicYuv.cpp:94-95: maxCUWidth + 16 samples of margin on every side


def ctu_grid(width, height):
    return (width + CTU - 1) // CTU, (height + CTU - 1) // CTU


def texture16_luma(width, height, seed=1234, frame=0, rng=None):
    rng = np.random.default_rng(seed) if rng is None else rng
    yy, xx = np.mgrid[0:height, 0:width]
    base = 128 + 60 * np.sin(xx / 37.0 + frame * 0.2) * np.cos(yy / 23.0)
    tex = rng.integers(0, 2, size=(height // 16 + 1, width // 16 + 1)).repeat(16, 0).repeat(16, 1)[:height, :width]
    noise = rng.normal(0, 18, size=(height, width)) * tex
    edges = ((xx // 48 + yy // 40) % 2) * 30
    return np.clip(base + noise + edges, 0, 255).astype(np.uint8)


def texture16_yuv420(width, height, seed=1234, frames=1):
    rng = np.random.default_rng(seed)
    _, xx = np.mgrid[0:height, 0:width]
    out = []
    for f in range(frames):
        y = texture16_luma(width, height, frame=f, rng=rng)
        u = np.full((height // 2, width // 2), 128, np.uint8)
        v = np.clip(128 + 20 * np.sin(xx[::2, ::2] / 50.0), 0, 255).astype(np.uint8)
        out.append(y.tobytes() + u.tobytes() + v.tobytes())
    return b"".join(out)


def _box_blur(a, k):
    if k <= 1:
        return a
    pad = k // 2
    ap = np.pad(a, ((pad, pad), (pad, pad)), mode="reflect")
    c = np.cumsum(ap, axis=0)
    c = np.vstack([np.zeros((1, c.shape[1])), c])
    a1 = (c[k:] - c[:-k]) / k
    c = np.cumsum(a1, axis=1)
    c = np.hstack([np.zeros((c.shape[0], 1)), c])
    return (c[:, k:] - c[:, :-k]) / k


def hetero_luma(width, height, seed=4321):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    base = 110 + 50 * np.sin(xx / 211.0) * np.cos(yy / 157.0) + 0.02 * xx
    cls = rng.integers(0, 4, size=(height // 128 + 1, width // 128 + 1)).repeat(128, 0).repeat(128, 1)[:height, :width]
    white = rng.normal(0, 1, size=(height, width))
    smooth = _box_blur(_box_blur(white, 17), 17)
    smooth /= smooth.std()
    mid = _box_blur(white, 5)
    mid /= mid.std()
    y = base.copy()
    y += (cls == 0) * white * 1.0
    y += (cls == 1) * smooth * 14.0
    y += (cls == 2) * (mid * 10.0 + white * 16.0)
    rect = np.zeros((height, width))
    for _ in range(int(width * height / 3000)):
        rw = int(rng.integers(6, 120))
        rh = int(rng.integers(6, 120))
        x0 = int(rng.integers(0, width - rw))
        y0 = int(rng.integers(0, height - rh))
        rect[y0:y0 + rh, x0:x0 + rw] += rng.choice([-35, 35, -18, 18])
    y += (cls == 3) * (rect + white * 2.0)
    return np.clip(np.rint(y), 0, 255).astype(np.uint8)


def hetero_yuv420(width, height, seed=4321):
    y = hetero_luma(width, height, seed)
    _, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    u = np.full((height // 2, width // 2), 128, np.uint8)
    v = np.clip(128 + 15 * np.sin(xx[::2, ::2] / 90.0), 0, 255).astype(np.uint8)
    return y.tobytes() + u.tobytes() + v.tobytes()


def to_pel_plane(luma_u8, internal_bit_depth=8, margin=HM_MARGIN):
    """uint8 luma -> int16 `Pel` plane laid out like TComPicYuv (TComPicYuv.cpp:81-119): stride = W + 2*margin,
    origin at (margin, margin), samples left-shifted to the internal bit depth (TVideoIOYuv.cpp:70-84,730).
    Returns (buffer, origin_offset_in_samples, stride)."""
    h, w = luma_u8.shape
    stride = w + 2 * margin
    buf = np.zeros((h + 2 * margin, stride), np.int16)
    buf[margin:margin + h, margin:margin + w] = luma_u8.astype(np.int16) << (internal_bit_depth - 8)
    return buf, margin * stride + margin, stride


def chroma_planes(kind, width, height):
    """(U, V) uint8 planes [H/2, W/2] of the pinned generators: U = 128, V a horizontal sinusoid."""
    _, xx = np.mgrid[0:height, 0:width]
    u = np.full((height // 2, width // 2), 128, np.uint8)
    if kind == "hetero":
        v = np.clip(128 + 15 * np.sin(xx[::2, ::2].astype(np.float64) / 90.0), 0, 255).astype(np.uint8)
    else:
        v = np.clip(128 + 20 * np.sin(xx[::2, ::2] / 50.0), 0, 255).astype(np.uint8)
    return u, v

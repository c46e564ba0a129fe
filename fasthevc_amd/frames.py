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


def fractal_luma(width, height, seed=777, beta=None, shapes=None):
    """Third synthetic family (neither of SURVEY 8(d)'s generators): 1/f^beta spectral noise, a soft vignette, a few flat
    rectangles and a thin diagonal line.  beta / shapes default to seeded random draws."""
    rng = np.random.default_rng(seed)
    beta = float(rng.uniform(0.9, 1.9)) if beta is None else beta
    fy, fx = np.fft.fftfreq(height)[:, None], np.fft.rfftfreq(width)[None, :]
    f = np.sqrt(fx * fx + fy * fy)
    f[0, 0] = 1.0
    spec = (rng.normal(size=f.shape) + 1j * rng.normal(size=f.shape)) / f ** beta
    spec[0, 0] = 0
    y = np.fft.irfft2(spec, s=(height, width))
    y = (y - y.mean()) / y.std() * float(rng.uniform(25.0, 55.0)) + float(rng.uniform(100.0, 150.0))
    yy, xx = np.mgrid[0:height, 0:width]
    y -= float(rng.uniform(0.0, 40.0)) * (((xx - width / 2) / width) ** 2 + ((yy - height / 2) / height) ** 2)
    n_shapes = int(rng.integers(0, 5)) if shapes is None else shapes
    for _ in range(n_shapes):
        w, h = int(rng.integers(width // 12, width // 3)), int(rng.integers(height // 10, height // 3))
        x0, y0 = int(rng.integers(0, width - w)), int(rng.integers(0, height - h))
        y[y0:y0 + h, x0:x0 + w] = float(rng.uniform(30, 225))
    if rng.random() < 0.6:
        y[np.abs(yy - (float(rng.uniform(-1.2, 1.2)) * xx + float(rng.uniform(0, height)))) < 1.5] = float(rng.choice([20.0, 235.0]))
    return np.clip(np.rint(y), 0, 255).astype(np.uint8)


def gratings_luma(width, height, seed=777):
    """Fourth synthetic family: sinusoidal gratings whose frequency and orientation change per region, Gaussian blobs,
    step edges and text-like strokes (all seeded)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    y = np.full((height, width), float(rng.uniform(90, 150)))
    reg = int(rng.choice([128, 192, 256, 384]))
    fmax = float(rng.choice([0.08, 0.2, 0.35]))
    for by in range(0, height, reg):
        for bx in range(0, width, reg):
            fr, th, amp = rng.uniform(0.01, fmax), rng.uniform(0, np.pi), rng.uniform(0, 60)
            sl = (slice(by, min(by + reg, height)), slice(bx, min(bx + reg, width)))
            y[sl] += amp * np.sin(2 * np.pi * fr * (np.cos(th) * xx[sl] + np.sin(th) * yy[sl]))
    for _ in range(int(rng.integers(0, 16))):
        cx, cy, sg, a = rng.uniform(0, width), rng.uniform(0, height), rng.uniform(20, 120), rng.uniform(-70, 70)
        y += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sg * sg))
    if rng.random() < 0.7:
        x0 = int(rng.integers(0, width // 2))
        y[:, x0:x0 + width // 6] += float(rng.uniform(-50, 50))
    for _ in range(int(rng.integers(0, 200))):
        x0, y0, ln, horiz = int(rng.integers(0, width - 40)), int(rng.integers(0, height - 40)), int(rng.integers(6, 40)), rng.random() < 0.5
        if horiz:
            y[y0:y0 + 2, x0:x0 + ln] = 25.0
        else:
            y[y0:y0 + ln, x0:x0 + 2] = 25.0
    return np.clip(np.rint(y), 0, 255).astype(np.uint8)


def polygon_luma(width, height, seed=777):
    """Fifth synthetic family: flat or shaded half-planes and discs stacked at random, with a little sensor noise."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    y = np.full((height, width), float(rng.uniform(60, 200)))
    for _ in range(int(rng.integers(6, 40))):
        v = float(rng.uniform(10, 245))
        if rng.random() < 0.5:
            cx, cy, r = rng.uniform(0, width), rng.uniform(0, height), rng.uniform(10, min(width, height) / 3)
            m = (xx - cx) ** 2 + (yy - cy) ** 2 < r * r
        else:
            th, c = rng.uniform(0, 2 * np.pi), rng.uniform(0.15, 0.85)
            m = (np.cos(th) * (xx / width - c) + np.sin(th) * (yy / height - c) > 0) & (rng.random() < 0.5 or True)
            m &= (np.abs(xx - rng.uniform(0, width)) < rng.uniform(40, width / 2)) & (np.abs(yy - rng.uniform(0, height)) < rng.uniform(40, height / 2))
        shade = float(rng.uniform(-0.08, 0.08)) if rng.random() < 0.5 else 0.0
        y = np.where(m, v + shade * (xx - width / 2), y)
    y += rng.normal(0, float(rng.uniform(0.0, 3.0)), size=y.shape)
    return np.clip(np.rint(y), 0, 255).astype(np.uint8)


def chirp_luma(width, height, seed=777):
    """Sixth synthetic family: zone plates (radial chirps), checkerboards of several scales and smooth ramps."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    y = np.full((height, width), float(rng.uniform(100, 150))) + float(rng.uniform(-40, 40)) * (xx / width - 0.5)
    for _ in range(int(rng.integers(1, 4))):
        cx, cy = rng.uniform(0, width), rng.uniform(0, height)
        r2 = (xx - cx) ** 2 + (yy - cy) ** 2
        y += float(rng.uniform(20, 80)) * np.cos(r2 / (2.0 * float(rng.uniform(300, 3000)))) * np.exp(-r2 / (2 * (float(rng.uniform(0.2, 0.6)) * height) ** 2))
    for _ in range(int(rng.integers(0, 5))):
        w, h = int(rng.integers(width // 10, width // 3)), int(rng.integers(height // 8, height // 2))
        x0, y0 = int(rng.integers(0, width - w)), int(rng.integers(0, height - h))
        sc = int(rng.choice([2, 4, 8, 16, 32, 64]))
        chk = ((xx[y0:y0 + h, x0:x0 + w] // sc + yy[y0:y0 + h, x0:x0 + w] // sc) % 2) * 2.0 - 1.0
        y[y0:y0 + h, x0:x0 + w] = float(rng.uniform(90, 160)) + float(rng.uniform(5, 60)) * chk
    y += rng.normal(0, float(rng.uniform(0.0, 2.5)), size=y.shape)
    return np.clip(np.rint(y), 0, 255).astype(np.uint8)


def deadleaves_luma(width, height, seed=777, leaves=900):
    """Dead-leaves picture (occluding discs with a power-law size distribution, each flat, noisy or finely textured):
    the classic natural-image surrogate.  Used as the HELD-OUT evaluation family: never part of the training labels."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    y = np.full((height, width), 128.0)
    for _ in range(leaves):
        r = 6.0 / (rng.uniform(0.02, 1.0) ** 1.0)          # radii 6 .. 300, density ~ r^-2
        cx, cy = rng.uniform(-50, width + 50), rng.uniform(-50, height + 50)
        m = (xx - cx) ** 2 + (yy - cy) ** 2 < r * r
        kind, v = rng.integers(0, 3), rng.uniform(25, 230)
        if kind == 0:
            tex = 0.0
        elif kind == 1:
            tex = rng.normal(0, rng.uniform(2, 14), size=y.shape)
        else:
            th, fr = rng.uniform(0, np.pi), rng.uniform(0.02, 0.3)
            tex = rng.uniform(4, 25) * np.sin(2 * np.pi * fr * (np.cos(th) * xx + np.sin(th) * yy))
        y = np.where(m, v + tex, y)
    return np.clip(np.rint(_box_blur(y, 3)), 0, 255).astype(np.uint8)


def glyphs_luma(width, height, seed=777):
    """HELD-OUT evaluation family (round 4; never part of any training label): screen-like content -- flat panels, rows of small glyph-like marks
    (strokes of 1-3 samples in cells of 6-14), thin rules and boxes, one smooth gradient panel, hardly any noise."""
    rng = np.random.default_rng(seed)
    y = np.full((height, width), float(rng.uniform(180, 240)))
    for _ in range(int(rng.integers(3, 7))):                      # panels
        w, h = int(rng.integers(width // 6, width // 2)), int(rng.integers(height // 6, height // 2))
        x0, y0 = int(rng.integers(0, width - w)), int(rng.integers(0, height - h))
        bg = float(rng.uniform(20, 245))
        y[y0:y0 + h, x0:x0 + w] = bg
        if rng.random() < 0.3:                                    # a gradient panel
            y[y0:y0 + h, x0:x0 + w] += np.linspace(-30, 30, w)[None, :]
            continue
        fg = bg - 120.0 if bg > 128 else bg + 120.0
        cell = int(rng.integers(6, 15))
        for ry in range(y0 + 4, y0 + h - cell, cell + int(rng.integers(2, 6))):
            x = x0 + 4
            while x < x0 + w - cell:
                if rng.random() < 0.85:                           # a glyph: two or three strokes inside its cell
                    for _ in range(int(rng.integers(2, 4))):
                        t = int(rng.integers(1, 3))
                        if rng.random() < 0.5:
                            yy0 = ry + int(rng.integers(0, cell - t))
                            y[yy0:yy0 + t, x:x + int(rng.integers(cell // 2, cell))] = fg
                        else:
                            xx0 = x + int(rng.integers(0, cell - t))
                            y[ry:ry + int(rng.integers(cell // 2, cell)), xx0:xx0 + t] = fg
                x += cell + int(rng.integers(1, 4)) if rng.random() < 0.85 else 3 * cell   # word gaps
    for _ in range(int(rng.integers(4, 12))):                     # rules and boxes
        x0, y0 = int(rng.integers(0, width - 64)), int(rng.integers(0, height - 64))
        w, h, v = int(rng.integers(16, 300)), int(rng.integers(16, 200)), float(rng.uniform(0, 120))
        y[y0:y0 + 1, x0:x0 + w] = v
        if rng.random() < 0.5:
            y[y0:y0 + h, x0:x0 + 1] = v
            y[min(height - 1, y0 + h):min(height, y0 + h + 1), x0:x0 + w] = v
            y[y0:y0 + h, min(width - 1, x0 + w):min(width, x0 + w + 1)] = v
    y += rng.normal(0, float(rng.uniform(0.0, 0.8)), size=y.shape)
    return np.clip(np.rint(y), 0, 255).astype(np.uint8)


def waves_luma(width, height, seed=777):
    """HELD-OUT evaluation family (round 4; never part of any training label): a few superposed plane waves whose amplitude and wavelength drift across the
    picture, soft-edged blobs of other mean level, and grain in patches -- smooth, anisotropic content without hard edges."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    y = np.full((height, width), float(rng.uniform(90, 160)))
    for _ in range(int(rng.integers(2, 6))):
        th, lam = rng.uniform(0, np.pi), rng.uniform(6, 120)
        drift = 1.0 + rng.uniform(-0.5, 0.5) * (xx / width) + rng.uniform(-0.5, 0.5) * (yy / height)
        amp = rng.uniform(4, 45) * (0.3 + 0.7 * np.abs(np.sin(xx / rng.uniform(150, 600) + rng.uniform(0, 6)) * np.cos(yy / rng.uniform(150, 600))))
        y += amp * np.sin(2 * np.pi * (np.cos(th) * xx + np.sin(th) * yy) / (lam * drift) + rng.uniform(0, 6))
    for _ in range(int(rng.integers(3, 14))):
        cx, cy, r = rng.uniform(0, width), rng.uniform(0, height), rng.uniform(15, 140)
        y += rng.uniform(-60, 60) / (1.0 + np.exp((np.sqrt((xx - cx) ** 2 + (yy - cy) ** 2) - r) / rng.uniform(1.5, 12)))
    grain = rng.normal(0, 1.0, size=y.shape)
    patch = _box_blur((rng.random((height, width)) < 0.0006).astype(np.float64), 65) * 65 * 65
    y += grain * np.minimum(patch, 1.0) * rng.uniform(3, 14) + grain * rng.uniform(0.3, 1.5)
    return np.clip(np.rint(y), 0, 255).astype(np.uint8)


def pan_clip(width, height, nframes=4, seed=1234, v_structure=3, v_noise=3):
    """SURVEY Appendix B's config-4 clip (texture16 content): texture mask and noise drawn once; base and edges move
    v_structure px per frame one way, the noise v_noise px per frame the other way (two overlaid motions), flat chroma.
    -> list of uint8 [H, W].  The defaults are the pinned clip."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width]
    tex = rng.integers(0, 2, size=(height // 16 + 1, width // 16 + 1)).repeat(16, 0).repeat(16, 1)[:height, :width]
    noise = rng.normal(0, 18, size=(height, width)) * tex
    out = []
    for f in range(nframes):
        base = 128 + 60 * np.sin((xx + v_structure * f) / 37.0) * np.cos(yy / 23.0)
        edges = (((xx + v_structure * f) // 48 + yy // 40) % 2) * 30
        out.append(np.clip(base + np.roll(noise, v_noise * f, axis=1) + edges, 0, 255).astype(np.uint8))
    return out


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

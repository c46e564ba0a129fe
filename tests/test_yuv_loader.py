"""Input wire format (N2): luma-only reader of planar YUV files, same plane semantics as TVideoIOYuv::read."""
import numpy as np
import pytest

from fasthevc_amd import frames
from fasthevc_amd.yuv import YuvLumaReader


def test_reads_the_pinned_generator_files(tmp_path):
    p = tmp_path / "t16.yuv"
    p.write_bytes(frames.texture16_yuv420(416, 240, frames=2))
    r = YuvLumaReader(str(p), 416, 240)
    assert r.num_frames == 2
    y0 = r.luma(0)
    assert y0.dtype == np.uint8 and np.array_equal(y0, frames.texture16_luma(416, 240))
    pel10 = r.luma(1, internal_bit_depth=10)
    assert pel10.dtype == np.int16 and np.array_equal(pel10 >> 2, r.luma(1).astype(np.int16))  # scalePlane: << (10 - 8)
    with pytest.raises(IndexError):
        r.luma(2)


def test_edge_replication_padding_and_16bit_files(tmp_path):
    w, h = 420, 236  # not multiples of 8 -> padded to 424 x 240 by replicating the last column / row
    rng = np.random.default_rng(1)
    y = rng.integers(0, 1024, size=(h, w)).astype("<u2")
    c = np.full((h // 2) * (w // 2) * 2, 512, "<u2")
    p = tmp_path / "hi.yuv"
    p.write_bytes(y.tobytes() + c.tobytes())
    r = YuvLumaReader(str(p), w, h, file_bit_depth=10)
    assert r.padded_size() == (424, 240)
    out = r.luma(0)
    assert out.shape == (240, 424) and out.dtype == np.int16
    assert np.array_equal(out[:h, :w], y.astype(np.int16))
    assert np.array_equal(out[:h, w:], np.repeat(y[:, -1:].astype(np.int16), 4, axis=1))
    assert np.array_equal(out[h:, :], np.repeat(out[h - 1:h, :], 4, axis=0))

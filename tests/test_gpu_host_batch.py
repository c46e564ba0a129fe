"""The metric as SURVEY 8(d) defines it: depth maps delivered to HOST memory.  From a generated .yuv file through the luma-only
reader (N2) into pinned memory, through fhevc_predict_frames (chunks over two streams), against the CPU oracle."""
import numpy as np
import pytest

from fasthevc_amd import capi, frames, weights
from fasthevc_amd.yuv import YuvLumaReader
from oracle import oracle_py as op

pytestmark = pytest.mark.gpu


def _oracle(oracle, w, luma, bd=8, qp=32):
    H, W = luma.shape
    buf, org, stride = frames.to_pel_plane(luma, bd)
    n = ((W + 63) // 64) * ((H + 63) // 64)
    depth, had = np.zeros(n * 256, np.uint8), np.zeros(n, np.int32)
    oracle.fho_predict_frame(op.weights_from_arrays(w), op.ptr(buf.reshape(-1), org), stride, W, H, bd, qp, depth, None)
    oracle.fho_frame_src_hadamard(op.ptr(buf.reshape(-1), org), stride, W, H, had)
    return depth.reshape(n, 256), had


def test_yuv_file_to_depth_maps_in_host_memory(oracle, tmp_path):
    W, H, NF = 416, 240, 7   # 7 frames, chunks of 3: two full chunks and a ragged one, both ring slots in use
    p = tmp_path / "clip.yuv"
    p.write_bytes(frames.texture16_yuv420(W, H, frames=NF))
    rd = YuvLumaReader(str(p), W, H)
    assert rd.num_frames == NF
    w = weights.random_weights(3)
    ctx = capi.Context(W, H, 8, w, max_frames=3)
    pinned = ctx.alloc_host((NF, H, W), np.uint8)
    rd.read_luma_into(pinned)
    out = ctx.alloc_host((NF, ctx.num_ctus, 256), np.uint8)
    had = ctx.alloc_host((NF, ctx.num_ctus), np.int32)
    out[:] = 9
    d_pin, h_pin = ctx.predict_frames(pinned, qp=30, depth_out=out, had_out=had)            # DMA straight from / to pinned memory
    d_pag, h_pag = ctx.predict_frames(np.array(pinned), qp=30)                                # pageable: through the staging ring
    for f in range(NF):
        ed, eh = _oracle(oracle, w, rd.luma(f), qp=30)
        assert np.array_equal(d_pin[f], ed) and np.array_equal(h_pin[f], eh), f
        assert np.array_equal(d_pag[f], ed) and np.array_equal(h_pag[f], eh), f
    d_nohad, none = ctx.predict_frames(pinned, qp=30, want_hadamard=False)
    assert none is None and np.array_equal(d_nohad, d_pin)
    st = ctx.stats()
    assert st["bytes_h2d"] >= 3 * NF * W * H and st["ctus"] >= 3 * NF * ctx.num_ctus
    for a in (pinned, out, had):
        ctx.free_host(a)
    ctx.close()


def test_host_batch_of_hm_pel_planes_10bit(oracle):
    """int16 planes with HM's margins and stride, 10 bit, one more frame than a chunk"""
    W, H, NF, bd = 832, 480, 5, 10
    w = weights.random_weights(5)
    lumas = [frames.hetero_luma(W, H, seed=40 + f) for f in range(NF)]
    planes = [frames.to_pel_plane(y, bd) for y in lumas]
    org, stride = planes[0][1], planes[0][2]
    buf = np.stack([p[0] for p in planes])
    ctx = capi.Context(W, H, bd, w, max_frames=4)
    depth, had = ctx.predict_frames(buf, qp=27, origin=org, stride=stride, frame_stride=buf[0].size)
    for f in range(NF):
        ed, eh = _oracle(oracle, w, lumas[f], bd=bd, qp=27)
        assert np.array_equal(depth[f], ed) and np.array_equal(had[f], eh), f
    ctx.close()


def test_padded_10bit_file_through_the_librarys_reader(oracle, tmp_path):
    """N2 in C++: a 10-bit 4:2:0 file whose size (410 x 236) is not a multiple of the minimum CU -> fhevc_read_yuv_luma pads it to the
    conformance size 416 x 240 and writes int16 Pel planes straight into fhevc_alloc_host memory -> fhevc_predict_frames (DMA from that
    memory) -> depth maps and source Hadamards in host memory == the CPU oracle on planes read by the Python restatement of
    TVideoIOYuv::read.  Also an 8-bit file read as uint8 planes (the half-traffic layout) the same way."""
    rng = np.random.default_rng(11)
    w = weights.random_weights(8)
    fw, fh, NF = 410, 236, 5
    W, H = 416, 240
    blob = b""
    for f in range(NF):
        y = (frames.texture16_luma(fw, fh, seed=700 + f).astype(np.uint16) << 2) | rng.integers(0, 4, size=(fh, fw)).astype(np.uint16)
        blob += y.astype("<u2").tobytes() + np.full((fw // 2) * (fh // 2) * 2, 512, "<u2").tobytes()
    p10 = tmp_path / "clip10.yuv"
    p10.write_bytes(blob)
    rd = YuvLumaReader(str(p10), fw, fh, file_bit_depth=10)
    ctx = capi.Context(W, H, 10, w, max_frames=2)
    pinned = ctx.alloc_host((NF, H, W), np.int16)
    assert capi.read_yuv_luma(str(p10), (fw, fh), 10, pinned) == NF
    d, hd = ctx.predict_frames(pinned, qp=27, origin=0, stride=W, frame_stride=W * H)
    for f in range(NF):
        pel = rd.luma(f)   # padded int16 plane, 10 bit
        assert np.array_equal(pinned[f], pel)
        buf, org, stride = frames.to_pel_plane(np.zeros((H, W), np.uint8), 8)
        buf = buf.copy(); buf.reshape(-1)[:] = 0
        view = buf[frames.HM_MARGIN:frames.HM_MARGIN + H, frames.HM_MARGIN:frames.HM_MARGIN + W]
        view[:] = pel
        n = ctx.num_ctus
        ed, eh = np.zeros(n * 256, np.uint8), np.zeros(n, np.int32)
        oracle.fho_predict_frame(op.weights_from_arrays(w), op.ptr(buf.reshape(-1), org), stride, W, H, 10, 27, ed, None)
        oracle.fho_frame_src_hadamard(op.ptr(buf.reshape(-1), org), stride, W, H, eh)
        assert np.array_equal(d[f], ed.reshape(n, 256)) and np.array_equal(hd[f], eh), f
    ctx.free_host(pinned)
    ctx.close()
    # 8-bit file, uint8 planes, padded as well
    p8 = tmp_path / "clip8.yuv"
    p8.write_bytes(b"".join(frames.texture16_luma(fw, fh, seed=800 + f).tobytes() + bytes([128]) * ((fw // 2) * (fh // 2) * 2) for f in range(3)))
    rd8 = YuvLumaReader(str(p8), fw, fh)
    ctx = capi.Context(W, H, 8, w, max_frames=2)
    pin8 = ctx.alloc_host((3, H, W), np.uint8)
    assert capi.read_yuv_luma(str(p8), (fw, fh), 8, pin8) == 3
    d8, h8 = ctx.predict_frames(pin8, qp=32)
    for f in range(3):
        ed, eh = _oracle(oracle, w, rd8.luma(f), qp=32)
        assert np.array_equal(d8[f], ed) and np.array_equal(h8[f], eh), f
    ctx.free_host(pin8)
    ctx.close()

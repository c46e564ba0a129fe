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

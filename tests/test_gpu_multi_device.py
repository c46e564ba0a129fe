"""Multi-device contexts behind the C ABI (fhevc_cfg.num_devices > 1): the host-buffer entry points shard CTU-row bands of a picture /
runs of pictures of a batch over the devices and gather the maps into the caller's buffers.  The GPU box has ONE MI355X, so the device
list repeats ordinal 0 (two and three queues on the one device): the sharding, the gathering and the failure path are the code that
runs on an 8-GPU node, the concurrency of real devices is not exercised here (SCALE runs are the driver's)."""
import numpy as np
import pytest

from fasthevc_amd import capi, frames, weights

pytestmark = pytest.mark.gpu


def _single(W, H, bd, w, buf, org, stride, qp):
    ctx = capi.Context(W, H, bd, w)
    d, h = ctx.predict_frame(buf, org, stride, qp=qp)
    dmin, dmax, h2 = ctx.predict_frame_range(buf, org, stride, qp=qp, margin=20000, margin_stop=9000, with_hadamard=True)
    ctx.close()
    return d, h, dmin, dmax, h2


@pytest.mark.parametrize("W,H,bd,devices", [(416, 240, 8, [0, 0]), (1920, 1080, 8, [0, 0, 0]), (416, 240, 10, [0, 0, 0, 0, 0])])
def test_bands_over_devices_equal_the_single_device_maps(W, H, bd, devices, cnn_arith):
    w = weights.random_weights(2)
    luma = frames.texture16_luma(W, H, seed=77)
    buf, org, stride = frames.to_pel_plane(luma, bd)
    d1, h1, dmin1, dmax1, h21 = _single(W, H, bd, w, buf, org, stride, 27)
    ctx = capi.Context(W, H, bd, w, devices=devices)   # 5 devices over 4 CTU rows: one of them gets an empty band
    assert ctx.stats()["devices"] == len(devices)
    d, h = ctx.predict_frame(buf, org, stride, qp=27)
    assert np.array_equal(d, d1) and np.array_equal(h, h1)
    dmin, dmax, h2 = ctx.predict_frame_range(buf, org, stride, qp=27, margin=20000, margin_stop=9000, with_hadamard=True)
    assert np.array_equal(dmin, dmin1) and np.array_equal(dmax, dmax1) and np.array_equal(h2, h21)
    s = ctx.stats()
    assert s["frames"] == 2 and s["ctus"] == 2 * ctx.num_ctus and s["devices_failed"] == 0
    ctx.close()


def test_a_batch_is_dealt_to_the_devices_in_runs_of_pictures():
    W, H, NF = 416, 240, 7
    w = weights.random_weights(6)
    lumas = np.stack([frames.texture16_luma(W, H, seed=500 + f) for f in range(NF)])
    one = capi.Context(W, H, 8, w, max_frames=2)
    d1, h1 = one.predict_frames(lumas, qp=32)
    one.close()
    ctx = capi.Context(W, H, 8, w, max_frames=2, devices=[0, 0, 0])   # runs of 2, 2 and 3 pictures
    d, h = ctx.predict_frames(lumas, qp=32)
    assert np.array_equal(d, d1) and np.array_equal(h, h1)
    assert ctx.stats()["ctus"] == NF * ctx.num_ctus
    ctx.close()


def test_a_failing_device_is_dropped_and_its_share_redone(monkeypatch):
    """'any per-device failure -> status + the remaining devices, never an abort': device index 1 reports a failure on its first share
    (FHEVC_TEST_FAIL_DEVICE); the call still succeeds with the single-device maps, the device is gone from the context afterwards."""
    W, H = 416, 240
    w = weights.random_weights(2)
    luma = frames.texture16_luma(W, H, seed=78)
    buf, org, stride = frames.to_pel_plane(luma, 8)
    d1, h1, _, _, _ = _single(W, H, 8, w, buf, org, stride, 32)
    monkeypatch.setenv("FHEVC_TEST_FAIL_DEVICE", "1")
    ctx = capi.Context(W, H, 8, w, devices=[0, 0, 0])
    d, h = ctx.predict_frame(buf, org, stride, qp=32)
    assert np.array_equal(d, d1) and np.array_equal(h, h1)
    s = ctx.stats()
    assert s["devices"] == 2 and s["devices_failed"] == 1
    assert b"dropped" in ctx.lib.fhevc_last_error(ctx.h)
    d, h = ctx.predict_frame(buf, org, stride, qp=32)   # and the context goes on with the two that are left
    assert np.array_equal(d, d1) and np.array_equal(h, h1)
    ctx.close()


def test_a_device_that_does_not_exist_is_left_out_at_create():
    ctx = capi.Context(416, 240, 8, weights.random_weights(2), devices=[0, 99])
    s = ctx.stats()
    assert s["devices"] == 1 and s["devices_failed"] == 1
    luma = frames.texture16_luma(416, 240, seed=79)
    buf, org, stride = frames.to_pel_plane(luma, 8)
    d, _ = ctx.predict_frame(buf, org, stride)
    ref = capi.Context(416, 240, 8, weights.random_weights(2))
    d1, _ = ref.predict_frame(buf, org, stride)
    assert np.array_equal(d, d1)
    ctx.close(); ref.close()


def test_a_family_member_on_the_layer_path_over_devices():
    """A multi-device context with a member of the reference's network family that runs layer by layer through HBM (23 / 46 / 92 x 2): every device
    builds its own images and activation tensors; the sharded maps equal the single-device ones."""
    W, H = 416, 240
    fam = weights.random_family((23, 46, 92), 2, seed=3)
    luma = frames.hetero_luma(W, H, seed=81)
    buf, org, stride = frames.to_pel_plane(luma, 8)
    one = capi.Context(W, H, 8, fam)
    d1, h1 = one.predict_frame(buf, org, stride, qp=30)
    one.close()
    ctx = capi.Context(W, H, 8, fam, devices=[0, 0, 0])
    d, h = ctx.predict_frame(buf, org, stride, qp=30)
    assert np.array_equal(d, d1) and np.array_equal(h, h1) and len(np.unique(d)) >= 2
    ctx.close()

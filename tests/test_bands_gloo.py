"""world_size-2 `gloo` test of the multi-rank path: each rank computes the depth map of its CTU-row band (with the
CPU oracle standing in for the GPU kernel), one all-gather, every rank ends with the 1-rank map."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fasthevc_amd import frames, weights
from fasthevc_amd import gather as bands


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle_py as op
    oracle = op.load_oracle()
    W, H, NF = 416, 240, 2
    w = weights.random_weights(4)
    ws = op.weights_from_arrays(w)
    cw, ch = frames.ctu_grid(W, H)
    gathered = bands.alloc_gather_buffers(NF, ch, cw, world, "cpu")
    b, e = bands.band(ch, rank, world)
    full = []
    for f in range(NF):
        buf, org, stride = frames.to_pel_plane(frames.texture16_luma(W, H, seed=50 + f), 8)
        depth = np.zeros(cw * ch * 256, np.uint8)
        oracle.fho_predict_frame(ws, op.ptr(buf.reshape(-1), org), stride, W, H, 8, 32, depth, None)
        depth = depth.reshape(ch, cw, 256)
        full.append(depth.reshape(ch * cw, 256))
        gathered[rank, f, : e - b] = torch.from_numpy(depth[b:e])  # this rank only contributes its band
    bands.all_gather_depth(gathered, rank)
    out = bands.assemble(gathered, ch).numpy()
    ok = all(np.array_equal(out[f], full[f]) for f in range(NF))
    # the bench's variant: frames dealt to ranks, 4-byte split-flag words on the wire, expanded on arrival
    flags = bands.alloc_flag_buffers(1, cw * ch, world, "cpu")
    buf, org, stride = frames.to_pel_plane(frames.texture16_luma(W, H, seed=50 + rank), 8)  # rank r owns frame r
    logits = np.zeros(cw * ch * 42, np.int32)
    depth = np.zeros(cw * ch * 256, np.uint8)
    import ctypes as C
    oracle.fho_predict_frame(ws, op.ptr(buf.reshape(-1), org), stride, W, H, 8, 32, depth, C.c_void_p(logits.ctypes.data))
    for c in range(cw * ch):
        vw, vh = min(64, W - (c % cw) * 64), min(64, H - (c // cw) * 64)
        flags[rank, 0, c] = int(oracle.fho_flags_from_logits(np.ascontiguousarray(logits[c * 42:(c + 1) * 42]), vw, vh))
    bands.all_gather_flags(flags, rank)
    for r in range(world):
        for c in range(cw * ch):
            vw, vh = min(64, W - (c % cw) * 64), min(64, H - (c // cw) * 64)
            d = np.zeros(256, np.uint8)
            oracle.fho_depth_from_flags(int(flags[r, 0, c]), vw, vh, d)
            ok = ok and np.array_equal(d, full[r][c])
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_band_gather_matches_single_rank():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_uneven_bands_pad_to_equal_slices():
    g = bands.alloc_gather_buffers(1, 17, 30, 8, "cpu")
    assert g.shape == (8, 1, 3, 30, 256)  # 1080p: 17 rows over 8 ranks -> 3-row slices, 23 040 B each
    for r in range(8):
        b, e = bands.band(17, r, 8)
        g[r, 0, : e - b] = r + 1
    full = bands.assemble(g, 17)
    assert full.shape == (1, 510, 256)
    rows = full.reshape(17, 30, 256)[:, 0, 0].tolist()
    assert rows == [1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 8]

"""Multi-rank path on CPU (gloo): the partition / all-gather / reassembly code bench.py runs (fasthevc_amd/gather.py), at world
sizes 2, 3 and 8 with uneven shares, in both partitions (pictures dealt to ranks; CTU-row bands of every picture).  The CPU
oracle stands in for the GPU kernel; every rank must end with the split-flag words of the single-rank run."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fasthevc_amd import frames, gather, weights

W, H, NF = 416, 240, 5   # 7 x 4 CTUs: 4 CTU rows over 3 or 8 ranks and 5 pictures over 2, 3 or 8 ranks are uneven (some ranks get nothing)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flag_words(oracle, ws, f):
    from oracle import oracle_py as op
    cw, ch = frames.ctu_grid(W, H)
    buf, org, stride = frames.to_pel_plane(frames.texture16_luma(W, H, seed=60 + f), 8)
    logits = np.zeros(cw * ch * 42, np.int32)
    depth = np.zeros(cw * ch * 256, np.uint8)
    oracle.fho_predict_frame(ws, op.ptr(buf.reshape(-1), org), stride, W, H, 8, 32, depth, C.c_void_p(logits.ctypes.data))
    out = np.zeros(cw * ch, np.int32)
    for c in range(cw * ch):
        vw, vh = min(64, W - (c % cw) * 64), min(64, H - (c // cw) * 64)
        out[c] = int(oracle.fho_flags_from_logits(np.ascontiguousarray(logits[c * 42:(c + 1) * 42]), vw, vh))
    return out, depth.reshape(cw * ch, 256)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle_py as op
    oracle = op.load_oracle()
    ws = op.weights_from_arrays(weights.random_weights(4))
    cw, ch = frames.ctu_grid(W, H)
    full = np.stack([_flag_words(oracle, ws, f)[0] for f in range(NF)])  # the single-rank answer, [NF, numCtus]
    ok = True
    for mode in ("frames", "bands"):
        fg = gather.FlagGather(mode, world, rank, NF, cw, ch, "cpu", group=None, host_group=None)
        (f0, f1), (r0, r1) = fg.frames, fg.rows
        for step in range(3):  # both buffers of the double buffering, twice
            b = step & 1
            mine = full[f0:f1].reshape(f1 - f0, ch, cw)[:, r0:r1].reshape(-1)   # what this rank's kernel writes: compact over its share
            assert mine.size == fg.local_ctus
            fg.local_words(b)[:] = torch.from_numpy(mine.copy())
            fg.start(b)
            words = fg.finish().numpy().reshape(NF, cw * ch)
            ok = ok and np.array_equal(words, full)
        # and the words expand to the depth maps of the single-rank run (fho_depth_from_flags = fhevc_expand_depth_flags_device)
        depth0 = _flag_words(oracle, ws, 0)[1]
        for c in range(cw * ch):
            vw, vh = min(64, W - (c % cw) * 64), min(64, H - (c // cw) * 64)
            d = np.zeros(256, np.uint8)
            oracle.fho_depth_from_flags(int(words[0, c]), vw, vh, d)
            ok = ok and np.array_equal(d, depth0[c])
        ok = ok and fg.status == "host-gather"   # on CPU tensors the gloo group IS the host gather
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_every_rank_ends_with_the_single_rank_words(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]


def test_partition_arithmetic_matches_fhevc_band():
    from fasthevc_amd import capi
    for total, world in ((17, 8), (34, 8), (4, 3), (5, 8), (64, 4)):
        got = [gather.span(total, r, world) for r in range(world)]
        assert got == [capi.band(total, r, world) for r in range(world)]
        assert got[0][0] == 0 and got[-1][1] == total and all(a[1] == b[0] for a, b in zip(got, got[1:]))
    fg = gather.FlagGather("bands", 8, 3, 2, 60, 34, "cpu")   # config 3: 4K, 34 CTU rows over 8 ranks -> padded slices of 5 rows
    assert fg.max_rows == 5 and fg.slice_words == 2 * 5 * 60 and fg.rows == (12, 17)


def _agreement_worker(rank, world, port, fail_rank, q):
    """init_groups with a second gloo group standing in for RCCL: the verdict on the fast backend must be the same on all ranks"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if fail_rank is not None:
        os.environ["FHEVC_TEST_FAIL_RCCL_RANK"] = str(fail_rank)   # every rank has it in its environment; only rank == fail_rank fails
    torch.set_num_threads(1)
    group, host_group = gather.init_groups(world, rank, "cpu", backend="gloo-as-rccl", timeout_s=60)
    verdict = "fallback" if group is False else "fast"
    # whichever way the ranks agreed, the gather itself still works and gives the same words everywhere
    fg = gather.FlagGather("frames", world, rank, 4, 2, 2, "cpu", group=group, host_group=host_group)
    f0, f1 = fg.frames
    full = torch.arange(16, dtype=torch.int32) * 7 + 3
    fg.local_words(0)[:] = full[f0 * 4:f1 * 4]
    fg.start(0)
    same = torch.equal(fg.finish(), full)
    # a collective that raises in mid-run is a loud failure on the rank that saw it, never a private switch of backend
    raised = False
    if rank == 0:
        real = dist.all_gather_into_tensor
        dist.all_gather_into_tensor = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("injected"))
        try:
            fg.local_words(1)[:] = full[f0 * 4:f1 * 4]
            fg.start(1)
        except gather.GatherError:
            raised = True
        finally:
            dist.all_gather_into_tensor = real
        fg._inflight = None
    q.put((rank, verdict, bool(same), raised))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [None, 1])
def test_the_fallback_from_rccl_is_a_collective_decision(fail_rank):
    """a rank whose RCCL bring-up fails must not end up in a gloo collective while its peers sit in RCCL.  One rank
    reporting a failed test collective puts EVERY rank on the host gather; with no failure every rank keeps the fast group."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agreement_worker, args=(r, world, port, fail_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    want = "fast" if fail_rank is None else "fallback"
    assert [(r, v, s) for r, v, s, _ in res] == [(r, want, True) for r in range(world)]
    assert res[0][3] is True   # rank 0's injected mid-run failure surfaced as GatherError

#!/usr/bin/env python3
"""bench.py -- CTU depth decisions/s of the MI355X fast-decision path (BASELINE.json metric).

One step = one pass of the hot path (CTU load -> source Hadamard -> depth CNN -> depth map in HBM) over one GOP of
synthetic 1080p luma already resident in HBM.  N > 1: one process per GPU (torch.distributed, "nccl" = RCCL over
xGMI), every rank owns its own GOP (frames dealt to ranks: weak scaling) and the step ends with the single
all-gather of the depth maps.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per CTU: MACs of the three conv layers + the three FC heads
MAC_PER_CTU = 4096 * 9 * 16 + 1024 * 144 * 32 + 256 * 288 * 64 + (4096 + 4 * 4096 + 16 * 1024) * 2
FLOP_PER_CTU = 2 * MAC_PER_CTU
PEAK_BF16_TFLOPS = 2500.0   # dense 16-bit (bf16 = f16) MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
PEAK_INT32_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12  # 256 CUs x 4 SIMD x 16 lanes x 2.4 GHz: one 32-bit integer op per lane-cycle (39.3)


def cpu_baseline(width, height, bit_depth, w, seconds=12.0):
    """CPU baseline on a bounded sample of the same workload, 1 host thread.

    kind "reference": HM's own full-RDO decision path (TEncSlice::compressSlice -> TEncCu::xCompressCU of the reference,
    oracle/_ref/libhmref.so, built from /root/reference in the build container and shipped as a built artefact).
    kind "port": the CPU oracle of the GPU path (depth CNN + source Hadamard) when that library is not present."""
    from oracle import oracle_py as op
    from fasthevc_amd import frames
    luma = frames.hetero_luma(width, height)
    if op.have_ref():
        lib = op.bind_rdo(op.load_ref())
        cu, cv = frames.chroma_planes("hetero", width, height)
        cw_, ch_ = 768, 512  # crops of the picture: 96 CTUs each, about 1.5 s of full RDO
        done, spent, crops = 0, 0.0, 0
        for (ox, oy) in ((0, 0), (768, 0), (1152, 0), (0, 512), (768, 512), (1152, 512), (384, 256), (960, 256),
                         (192, 128), (576, 384), (1088, 64), (128, 448), (640, 192), (1024, 320)):
            if oy + ch_ > height or ox + cw_ > width:
                continue
            buf, org, stride = frames.to_pel_plane(luma[oy:oy + ch_, ox:ox + cw_].copy(), bit_depth)
            chroma = tuple((c[oy // 2:(oy + ch_) // 2, ox // 2:(ox + cw_) // 2].astype(np.int16) << (bit_depth - 8)) for c in (cu, cv))
            _, st = op.rdo_encode(lib, buf, org, stride, cw_, ch_, bit_depth, 32, chroma=chroma)
            done += st["ctus"]
            spent += st["seconds"]
            crops += 1
            if spent > seconds:
                break
        out = {"value": done / spent, "unit": "CTU depth decisions/s", "cores": 1, "kind": "reference",
               "sample": f"{crops} crops of 768x512 ({done} CTUs) of the same 1080p hetero frame at QP32 through the reference's own "
                         f"TEncSlice::compressSlice/TEncCu::xCompressCU full RDO (intra_main settings), {spent:.1f} s of 1 thread"}
        # the same compressSlice with the hm_patch hook linked to the GPU library (oracle/_ref/libhmref_hookgpu.so): HM's
        # decision stage end to end, GPU call (H2D + kernels + D2H) included, on the same crops
        gpu_so = os.path.join(ROOT, "oracle", "_ref", "libhmref_hookgpu.so")
        blob = os.path.join(ROOT, "fasthevc_amd", "weights", "depthnet_v1.fhw")
        if os.path.exists(gpu_so) and os.path.exists(blob):
            knobs = {"FHEVC_ENABLE": "1", "FHEVC_WEIGHTS": blob, "FHEVC_MARGIN_SPLIT": "32000", "FHEVC_MARGIN_STOP": "0"}
            saved = {k: os.environ.get(k) for k in knobs}
            os.environ.update(knobs)
            try:
                glib = op.bind_rdo(op.load_ref(hook="gpu"))
                gdone, gspent = 0, 0.0
                for (ox, oy) in ((0, 0), (768, 0), (1152, 0), (0, 512), (768, 512), (1152, 512), (384, 256), (960, 256))[:crops]:
                    buf, org, stride = frames.to_pel_plane(luma[oy:oy + ch_, ox:ox + cw_].copy(), bit_depth)
                    chroma = tuple((c[oy // 2:(oy + ch_) // 2, ox // 2:(ox + cw_) // 2].astype(np.int16) << (bit_depth - 8)) for c in (cu, cv))
                    _, st = op.rdo_encode(glib, buf, org, stride, cw_, ch_, bit_depth, 32, chroma=chroma)
                    gdone += st["ctus"]
                    gspent += st["seconds"]
                out["with_gpu_hook"] = {"value": gdone / gspent, "unit": "CTUs/s through compressSlice", "speedup": (gdone / gspent) / (done / spent),
                                        "sample": f"{gdone} CTUs, same crops, hm_patch hook -> fhevc_predict_frame_range (margin_split 32000), "
                                                  f"{gspent:.1f} s of 1 thread incl. the GPU calls"}
            finally:
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        return out
    oracle = op.load_oracle()
    ws = op.weights_from_arrays(w)
    buf, org, stride = frames.to_pel_plane(luma, bit_depth)
    cw, ch = frames.ctu_grid(width, height)
    ctu = np.zeros(64 * 64, np.int8)
    logits = np.zeros(42, np.int32)
    depth = np.zeros(256, np.uint8)
    done, t0 = 0, time.perf_counter()
    order = np.random.default_rng(0).permutation(cw * ch)
    for a in order:
        cx, cy = int(a % cw), int(a // cw)
        oracle.fho_load_ctu(op.ptr(buf.reshape(-1), org), stride, width, height, cx, cy, bit_depth, ctu)
        oracle.fho_cnn_ctu(ws, ctu, 32, logits)
        oracle.fho_depth_from_logits(logits, min(64, width - cx * 64), min(64, height - cy * 64), depth)
        oracle.fho_ctu_src_hadamard(op.ptr(buf.reshape(-1), org + cy * 64 * stride + cx * 64), stride,
                                    min(64, width - cx * 64), min(64, height - cy * 64))
        done += 1
        if time.perf_counter() - t0 > seconds:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "CTU depth decisions/s", "cores": 1, "kind": "port",
            "sample": f"{done} CTUs of the same 1080p hetero frame through oracle/fhevc_oracle.c "
                      f"(depth CNN + source Hadamard, 1 thread, {dt:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=64, help="frames per GOP (per rank)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bit-depth", type=int, default=8)
    ap.add_argument("--sample-bytes", type=int, default=2, help="2 = int16 Pel planes as HM holds them, 1 = uint8")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stages", action="store_true", help="skip the first-pass / pre-analysis stage report")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal on a 1-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from fasthevc_amd import bands, capi, frames, weights

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.one_device:
        assert args.backend != "nccl", "RCCL refuses two ranks on one GPU"
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    W, H, NF, bd = args.width, args.height, args.frames, args.bit_depth
    trained = os.path.join(ROOT, "fasthevc_amd", "weights", "depthnet_v1.fhw")
    if os.path.exists(trained):
        w, wdesc = weights.load(trained), "trained weights fasthevc_amd/weights/depthnet_v1.fhw"
    else:
        w, wdesc = weights.random_weights(0), "random-init weights"  # same architecture, same arithmetic
    ctx = capi.Context(W, H, bd, w, device=local, max_frames=NF)
    cw, ch, n_ctus = ctx.ctus_x, ctx.ctus_y, ctx.num_ctus

    # synthetic GOP resident in HBM: the pinned "hetero" frame, panned 3 px per frame, per-rank phase
    base = torch.from_numpy(frames.hetero_luma(W, H).astype(np.int16)).to(dev) << (bd - 8)
    if args.sample_bytes == 2:
        margin = frames.HM_MARGIN
        stride, rows = W + 2 * margin, H + 2 * margin
        gop = torch.zeros((NF, rows, stride), dtype=torch.int16, device=dev)
        for f in range(NF):
            gop[f, margin:margin + H, margin:margin + W] = torch.roll(base, shifts=3 * (f + NF * rank), dims=1)
        origin, frame_stride = margin * stride + margin, rows * stride
        luma_ptr = gop.data_ptr() + 2 * origin
    else:
        assert bd == 8
        stride, frame_stride = W, W * H
        gop = torch.stack([torch.roll(base, shifts=3 * (f + NF * rank), dims=1) for f in range(NF)]).to(torch.uint8).contiguous()
        luma_ptr = gop.data_ptr()

    # every rank ends a step with the depth maps of the WHOLE GOP (all ranks' frames) in `gathered`; on the wire the
    # maps travel as 4-byte split-flag words per CTU (64x less than 256 B) and are expanded on arrival
    gathered = torch.zeros((world, NF, n_ctus, 256), dtype=torch.uint8, device=dev)
    # double-buffered so that the all-gather of step i (RCCL's own stream) overlaps the kernels of step i+1
    flags_all = [bands.alloc_flag_buffers(NF, n_ctus, world, dev) for _ in range(2)]   # receive buffers of the all-gather
    flags_mine = [torch.zeros((NF, n_ctus), dtype=torch.int32, device=dev) for _ in range(2)]  # this rank's words (send)
    had = torch.zeros((NF, n_ctus), dtype=torch.int32, device=dev)
    # A real (non-null) torch stream for everything that follows: the library treats a NULL stream handle as "the context's
    # own stream", which torch's collectives (ordered against torch's CURRENT stream) would not see.  With it the kernels,
    # the all-gather's stream dependencies and the expansion are all expressed on one stream.
    torch.cuda.synchronize()
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0
    inflight = []  # [(work, buffer index)]: at most one collective in flight

    def finish_gather():
        """wait for the collective in flight and expand its words into the depth maps of the whole GOP"""
        while inflight:
            work, b = inflight.pop()
            work.wait()
            ctx.expand_depth_flags_device(flags_all[b].data_ptr(), world * NF, gathered.data_ptr(), stream=stream)

    def step(i):
        b = i & 1
        ctx.predict_frames_device(luma_ptr, args.sample_bytes, stride, frame_stride, NF, gathered[rank].data_ptr(),
                                  had.data_ptr(), None, stream=stream, d_flags=flags_mine[b].data_ptr() if world > 1 else None)
        if world > 1:
            finish_gather()  # step i-1's gather has had this step's kernels to hide under
            inflight.append((dist.all_gather_into_tensor(flags_all[b].view(-1), flags_mine[b].view(-1), async_op=True), b))  # the path's only collective

    def fence():
        finish_gather()  # every step's maps are gathered and expanded before the clock stops
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    ctx.enable_kernel_timing(True)
    ctx.kernel_timing(0, reset=True)
    ctx.kernel_timing(1, reset=True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    cnn_ms, cnn_n = ctx.kernel_timing(0)
    had_ms, had_n = ctx.kernel_timing(1)

    # untimed check of the N > 1 path: what the all-gather + expansion left in `gathered` must be this rank's own maps
    # in its slot, and the same bytes on every rank
    gather_ok = None
    if world > 1:
        own = torch.empty((NF, n_ctus, 256), dtype=torch.uint8, device=dev)
        ctx.predict_frames_device(luma_ptr, args.sample_bytes, stride, frame_stride, NF, own.data_ptr(), None, None, stream=stream)
        torch.cuda.synchronize()
        chk = torch.tensor([float(torch.equal(own, gathered[rank])), float(gathered.to(torch.int64).sum().item())],
                           dtype=torch.float64, device=dev)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        gather_ok = bool(lo[0].item() == 1.0 and lo[1].item() == hi[1].item() and len(torch.unique(gathered)) > 1)

    # the path's other stages (SURVEY 8(d) stage 3 and 8(f) N3), measured AFTER the timed region on the same GOP: the
    # 35-mode SATD first pass over 8 of the frames and the AQ pre-analysis over all of them -- reported, never in `value`
    stages = None
    if rank == 0 and not args.no_stages:
        nb = min(NF, 8)
        nodes = torch.zeros((nb * n_ctus * 85, 2), dtype=torch.float64, device=dev)
        act = torch.zeros((NF, ctx.aq_layout(4)[-1]), dtype=torch.float64, device=dev)
        for timed in (False, True):
            ctx.kernel_timing(2, reset=True)
            ctx.kernel_timing(3, reset=True)
            for _ in range(3):
                ctx.intra_first_pass_device(luma_ptr, args.sample_bytes, stride, frame_stride, nb, nodes.data_ptr(), stream=stream, qp=32)
                ctx.preanalyze_frames_device(luma_ptr, args.sample_bytes, stride, frame_stride, NF, act.data_ptr(), 4, stream=stream)
            torch.cuda.synchronize()
        fp_ms, _ = ctx.kernel_timing(2)
        pre_ms, _ = ctx.kernel_timing(3)
        int_ops = nb * n_ctus * 4 * 35 * 64 * (64 + 575)  # per (level, mode, 8x8 tile) 64 predicted samples + ~575 Hadamard ops
        pre_bytes = NF * (W * H * args.sample_bytes + act.shape[1] * 8)
        stages = {
            "first_pass": {"kernel": "fhevc_first_pass_kernel", "frames": nb, "avg_launch_ms": fp_ms, "ctu_per_s": nb * n_ctus / (fp_ms * 1e-3),
                           "bound": "int VALU/LDS", "achieved_Tintop_s": int_ops / (fp_ms * 1e-3) / 1e12, "peak_Tintop_s": PEAK_INT32_TOPS,
                           "frac": int_ops / (fp_ms * 1e-3) / 1e12 / PEAK_INT32_TOPS,
                           "vs_hadamard_time_per_frame": (fp_ms / nb) / (had_ms / NF) if had_ms else None},
            "preanalyze": {"kernel": "fhevc_preanalyze_kernel", "frames": NF, "avg_launch_ms": pre_ms, "bound": "hbm",
                           "achieved_GB_s": pre_bytes / (pre_ms * 1e-3) / 1e9, "peak_GB_s": PEAK_HBM_GBS,
                           "frac": pre_bytes / (pre_ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
        }
    ctx.enable_kernel_timing(False)

    def measured_traffic(kernel):
        """HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
        FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) -- only for the workload they were taken on."""
        path = os.path.join(ROOT, "profiles", "r01_pmc_bench_frames64_int16.json")
        if not (os.path.exists(path) and NF == 64 and args.sample_bytes == 2 and (W, H) == (1920, 1080)):
            return None
        d = json.load(open(path))
        for k, v in d.items():
            if k.startswith(kernel) and "hbm_read_bytes_corrected" in v and "hbm_write_bytes" in v:
                return v["hbm_read_bytes_corrected"] + v["hbm_write_bytes"]
        return None

    if rank == 0:
        ctus_per_step = world * NF * n_ctus
        value = ctus_per_step * args.steps / dt
        flop_per_launch = FLOP_PER_CTU * NF * n_ctus
        sample_b = args.sample_bytes
        bytes_per_launch_had = NF * (W * H * sample_b + n_ctus * 4)
        roof = {"bound": "mfma", "kernel": "fhevc_cnn_depth_kernel", "achieved": flop_per_launch / (cnn_ms * 1e-3) / 1e12 if cnn_ms else None,
                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "traffic": measured_traffic("fhevc_cnn_depth_kernel"),
                "avg_launch_ms": cnn_ms, "launches": cnn_n, "flop_per_ctu": FLOP_PER_CTU}
        roof["frac"] = roof["achieved"] / roof["peak"] if roof["achieved"] else None
        hbm = {"bound": "hbm", "kernel": "fhevc_src_hadamard_kernel", "achieved": bytes_per_launch_had / (had_ms * 1e-3) / 1e9 if had_ms else None,
               "peak": PEAK_HBM_GBS, "unit": "GB/s", "traffic": measured_traffic("fhevc_src_hadamard_kernel"), "avg_launch_ms": had_ms, "launches": had_n,
               "bytes_per_launch": bytes_per_launch_had}
        hbm["frac"] = hbm["achieved"] / hbm["peak"] if hbm["achieved"] else None
        line = {
            "metric": "CTU depth decisions/sec at 1080p all-intra", "value": value, "unit": "CTU/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 (conv1) and f16 (conv2, conv3) operands / f32 accumulate (fixed-point valued, exact)", "data": "synthetic",
            "config": {"workload": f"BQTerrace geometry {W}x{H} all-intra QP32, GOP of {NF} synthetic 'hetero' frames per GPU "
                                   f"({'int16 Pel planes, HM stride/margins' if sample_b == 2 else 'uint8 planes'}) resident in HBM, "
                                   f"source Hadamard + CTU-batched CNN depth predictor, {wdesc}",
                       "frames_per_gpu": NF, "ctus_per_frame": n_ctus, "bit_depth": bd,
                       "sharding": "frames dealt to ranks + one all-gather of the depth maps (as 4-byte split-flag words per CTU, expanded on every rank)" if world > 1 else "single GPU"},
            "roofline": roof, "roofline_hbm_kernel": hbm,
        }
        if stages:
            line["stages"] = stages
        if gather_ok is not None:
            line["gather_verified"] = gather_ok
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(W, H, bd, w)
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

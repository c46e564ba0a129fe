#!/usr/bin/env python3
"""bench.py -- CTU depth decisions/s of the MI355X fast-decision path (BASELINE.json metric).

One step = one pass of the hot path (CTU load -> source Hadamard -> depth CNN -> depth map in HBM) over one GOP of
synthetic 1080p luma already resident in HBM.  N > 1: one process per GPU (torch.distributed, "nccl" = RCCL over xGMI);
the step ends with the path's single all-gather of the depth decisions (fasthevc_amd/gather.py):
  --scaling weak   (default) every rank owns its own GOP of --frames pictures (frames dealt to ranks, work grows with N)
  --scaling strong           ONE GOP of --frames pictures split over the ranks
  --bands                    CTU-row bands of every picture (BASELINE config 3: 3840x2160 unless --width/--height say otherwise)
Prints ONE JSON line on rank 0.  The timed region (exactly --steps steps between barrier + synchronize) is repeated
--repeats times; `value` is the median region, min / median / max are reported next to it.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per CTU (HISTORY.md section 6): MACs of the three conv layers + the three FC heads
MAC_PER_CTU = 4096 * 9 * 16 + 1024 * 144 * 32 + 256 * 288 * 64 + (4096 + 4 * 4096 + 16 * 1024) * 2
FLOP_PER_CTU = 2 * MAC_PER_CTU
PEAK_BF16_TFLOPS = 2500.0   # dense 16-bit (bf16 = f16) MFMA, MI355X_MICROARCH.md
PEAK_I8_TOPS = 5000.0       # dense i8 MFMA: "2x BF16 per clock" (same table)
ARITH_DTYPE = {
    "i8": "int8 operands / int32 accumulate on v_mfma_i32_32x32x32_i8 (conv2, conv3: 93 % of the MACs); conv1 bf16 operands / f32 accumulate "
          "(fixed-point valued); every result exact",
    "f16": "bf16 (conv1) and f16 (conv2, conv3) operands / f32 accumulate (fixed-point valued, exact)",
}
ARITH_PEAK = {"i8": PEAK_I8_TOPS, "f16": PEAK_BF16_TFLOPS}
PEAK_HBM_GBS = 8000.0
PEAK_PCIE_GBS = 63.0        # PCIe Gen5 x16 (spec), MI355X_MICROARCH.md
PEAK_INT32_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12  # 256 CUs x 4 SIMD x 16 lanes x 2.4 GHz: one 32-bit integer op per lane-cycle (39.3)
CROPS = ((0, 0), (768, 0), (1152, 0), (0, 512), (768, 512), (1152, 512), (384, 256), (960, 256),
         (192, 128), (576, 384), (1088, 64), (128, 448), (640, 192), (1024, 320))
CROP_W, CROP_H = 768, 512   # 96 CTUs, about 1 s of the reference's full RDO


def _rdo_crops(lib, width, height, bit_depth, seconds, first=0):
    """the reference's own compressSlice on crops of the pinned 1080p hetero picture for about `seconds`: (CTUs, seconds)"""
    from oracle import oracle_py as op
    from fasthevc_amd import frames
    luma = frames.hetero_luma(width, height)
    cu, cv = frames.chroma_planes("hetero", width, height)
    done, spent, crops = 0, 0.0, 0
    k = first
    while spent < seconds:
        ox, oy = CROPS[k % len(CROPS)]
        k += 1
        if oy + CROP_H > height or ox + CROP_W > width:
            continue
        buf, org, stride = frames.to_pel_plane(luma[oy:oy + CROP_H, ox:ox + CROP_W].copy(), bit_depth)
        chroma = tuple((c[oy // 2:(oy + CROP_H) // 2, ox // 2:(ox + CROP_W) // 2].astype(np.int16) << (bit_depth - 8)) for c in (cu, cv))
        _, st = op.rdo_encode(lib, buf, org, stride, CROP_W, CROP_H, bit_depth, 32, chroma=chroma)
        done += st["ctus"]
        spent += st["seconds"]
        crops += 1
    return done, spent, crops


def cpu_worker(argv):
    """child process of the all-cores CPU baseline (no GPU, no torch): prints {"ctus", "seconds", "crops"}"""
    width, height, bit_depth, seconds, first = int(argv[0]), int(argv[1]), int(argv[2]), float(argv[3]), int(argv[4])
    from oracle import oracle_py as op
    lib = op.bind_rdo(op.load_ref())
    done, spent, crops = _rdo_crops(lib, width, height, bit_depth, seconds, first)
    print(json.dumps({"ctus": done, "seconds": spent, "crops": crops}), flush=True)


def cpu_baseline_reference(width, height, bit_depth, seconds=10.0, procs=None):
    """HM's own full-RDO decision path (TEncSlice::compressSlice -> TEncCu::xCompressCU of the reference, oracle/_ref/libhmref.so,
    built from /root/reference in the build container and shipped as a built artefact) on the host cores: one process, then one
    process per core (HM is single-threaded).  Runs BEFORE this process touches the GPU: the workers are plain children."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    procs = procs or max(1, min(avail, 16))   # a one-GPU box gives 16 cores to a command

    def run(n):
        t0 = time.perf_counter()
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(width), str(height), str(bit_depth),
                                str(seconds), str(3 * i)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(n)]
        outs = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in ps]
        wall = time.perf_counter() - t0
        return outs, wall

    one, _ = run(1)
    out = {"value": one[0]["ctus"] / one[0]["seconds"], "unit": "CTU depth decisions/s", "cores": 1, "kind": "reference",
           "sample": f"{one[0]['crops']} crops of {CROP_W}x{CROP_H} ({one[0]['ctus']} CTUs) of the same 1080p hetero frame at QP32 through the "
                     f"reference's own TEncSlice::compressSlice/TEncCu::xCompressCU full RDO (intra_main settings), {one[0]['seconds']:.1f} s of 1 thread"}
    if procs > 1:
        many, wall = run(procs)
        out["all_cores"] = {"value": sum(o["ctus"] / o["seconds"] for o in many), "unit": "CTU depth decisions/s", "cores": procs,
                            "per_process": [round(o["ctus"] / o["seconds"], 1) for o in many],
                            "sample": f"{procs} concurrent single-threaded processes (HM has no threads), {sum(o['ctus'] for o in many)} CTUs, "
                                      f"{np.mean([o['seconds'] for o in many]):.1f} s each inside compressSlice, {wall:.1f} s wall incl. start-up"}
    return out


def cpu_baseline_port(width, height, bit_depth, w, seconds=10.0):
    """fallback when the reference library is not present: the CPU oracle of the GPU path (depth CNN + source Hadamard), 1 thread"""
    from oracle import oracle_py as op
    from fasthevc_amd import frames
    oracle = op.load_oracle()
    ws = op.weights_from_arrays(w)
    luma = frames.hetero_luma(width, height)
    buf, org, stride = frames.to_pel_plane(luma, bit_depth)
    cw, ch = frames.ctu_grid(width, height)
    ctu = np.zeros(64 * 64, np.int8)
    logits = np.zeros(42, np.int32)
    depth = np.zeros(256, np.uint8)
    done, t0 = 0, time.perf_counter()
    for a in np.random.default_rng(0).permutation(cw * ch):
        cx, cy = int(a % cw), int(a // cw)
        oracle.fho_load_ctu(op.ptr(buf.reshape(-1), org), stride, width, height, cx, cy, bit_depth, ctu)
        oracle.fho_cnn_ctu(ws, ctu, 32, logits)
        oracle.fho_depth_from_logits(logits, min(64, width - cx * 64), min(64, height - cy * 64), depth)
        oracle.fho_ctu_src_hadamard(op.ptr(buf.reshape(-1), org + cy * 64 * stride + cx * 64), stride, min(64, width - cx * 64), min(64, height - cy * 64))
        done += 1
        if time.perf_counter() - t0 > seconds:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "CTU depth decisions/s", "cores": 1, "kind": "port",
            "sample": f"{done} CTUs of the same 1080p hetero frame through oracle/fhevc_oracle.c (depth CNN + source Hadamard, 1 thread, {dt:.1f} s)"}


def gpu_hook_leg(width, height, bit_depth, crops, margins=None, first_pass=False):
    """the same compressSlice with the hm_patch hook linked to the GPU library (oracle/_ref/libhmref_hookgpu.so): HM's decision
    stage end to end, GPU call (H2D + kernels + D2H) included, on the same crops.  margins None: the hook's SHIPPED defaults
    (depthnet_family_d2.fhw = the 23 / 46 / 92 x 2 member through the fused two-convolution kernel, at 100000 : 64000: every one of
    ten content families, three of them never in a training label, within +0.47 % BD-rate -- profiles/r04_bdrate_family_d2_ten_families.json);
    (split, stop): that setting"""
    from oracle import oracle_py as op
    gpu_so = os.path.join(ROOT, "oracle", "_ref", "libhmref_hookgpu.so")
    blob = os.path.join(ROOT, "fasthevc_amd", "weights", "depthnet_family_d2.fhw")
    if not (os.path.exists(gpu_so) and os.path.exists(blob)):
        return None
    knobs = {"FHEVC_ENABLE": "1", "FHEVC_WEIGHTS": blob}
    if margins is not None:
        knobs.update({"FHEVC_MARGIN_SPLIT": str(margins[0]), "FHEVC_MARGIN_STOP": str(margins[1])})
    if first_pass:  # estIntraPredLumaQT's candidate lists from the GPU's first pass as well (fhevc_intra_first_pass_candidates)
        knobs["FHEVC_FIRST_PASS"] = "1"
    clear = ("FHEVC_MARGIN", "FHEVC_MARGIN_SPLIT", "FHEVC_MARGIN_STOP", "FHEVC_FIRST_PASS")
    saved = {k: os.environ.get(k) for k in set(knobs) | set(clear)}
    for k in clear:
        os.environ.pop(k, None)
    os.environ.update(knobs)
    try:
        glib = op.bind_rdo(op.load_ref(hook="gpu"))
        done, spent = 0, 0.0
        from fasthevc_amd import frames
        luma = frames.hetero_luma(width, height)
        cu, cv = frames.chroma_planes("hetero", width, height)
        # TEncFastDepth reads its knobs when the harness constructs the encoder of a geometry: the second setting runs on crops 8 px narrower
        cw_ = CROP_W - 128 if first_pass else (CROP_W if margins is None else CROP_W - 64)
        for (ox, oy) in [c for c in CROPS if c[1] + CROP_H <= height and c[0] + CROP_W <= width][:crops]:
            buf, org, stride = frames.to_pel_plane(luma[oy:oy + CROP_H, ox:ox + cw_].copy(), bit_depth)
            chroma = tuple((c[oy // 2:(oy + CROP_H) // 2, ox // 2:(ox + cw_) // 2].astype(np.int16) << (bit_depth - 8)) for c in (cu, cv))
            _, st = op.rdo_encode(glib, buf, org, stride, cw_, CROP_H, bit_depth, 32, chroma=chroma)
            done += st["ctus"]
            spent += st["seconds"]
        what = "the hook's shipped defaults 100000:64000 (ten content families, every one within +0.47 % BD-rate)" if margins is None else \
            f"margins {margins[0]}:{margins[1]} (content-matched: +0.04 % BD-rate on this family, <= +0.91 % on nine of ten, +2.25 % on the never-trained mix)"
        if first_pass:
            what += " + FHEVC_FIRST_PASS=1 (candidate lists of estIntraPredLumaQT from the GPU's 35-mode first pass)"
        return {"value": done / spent, "unit": "CTUs/s through compressSlice", "margins": "100000:64000" if margins is None else f"{margins[0]}:{margins[1]}",
                "weights": os.path.basename(blob),
                "sample": f"{done} CTUs, crops of the same picture, hm_patch hook -> fhevc_predict_frame_range at {what}, {spent:.1f} s of 1 thread incl. the GPU calls"}
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def quoted_bd_rate():
    """the quality half of BASELINE's metric, quoted from the committed evaluations under profiles/ (tests/quality/eval_rd.py,
    eval_p.py: minutes of CPU each, not re-run here)"""
    out = {}
    for tag, name in (("intra", "bdrate_generalization"), ("intra_family_32_64_128", "bdrate_family_d1"), ("intra_family_23_46_92_x2", "bdrate_family_d2"), ("hook_default_ten_families", "bdrate_family_d2_ten_families"), ("intra_family_18_36_72_x3", "bdrate_family_d3"), ("p_slices", "p_slice_motion_rule"),
                      ("p_slices_large_motion", "p_slice_motion_speed32_832x480")):
        for rnd in ("r04", "r03", "r02", "r01"):
            path = os.path.join(ROOT, "profiles", f"{rnd}_{name}.json")
            if os.path.exists(path):
                try:
                    d = json.load(open(path))
                    summ = d.get("headline") or d.get("summary")
                    if summ:
                        out[tag] = {"source": f"profiles/{rnd}_{name}.json", "summary": summ}
                except Exception:
                    pass
                break
    return out or None


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)    # 5 regions x 50 steps = 0.1 s of GPU time at the default workload
    ap.add_argument("--warmup", type=int, default=20)   # the clocks of a fresh process take ~30 steps to settle
    ap.add_argument("--prewarm", type=int, default=30, help="untimed steps before the --warmup steps: a fresh process's GPU clocks settle over ~30 steps "
                                                            "(reported as prewarm_steps; 0 = none)")
    ap.add_argument("--repeats", type=int, default=5, help="how many times the timed region of --steps steps is run (value = median)")
    ap.add_argument("--frames", type=int, default=None, help="frames per GOP (default 64; 16 with --bands)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--bit-depth", type=int, default=8)
    ap.add_argument("--sample-bytes", type=int, default=2, help="2 = int16 Pel planes as HM holds them, 1 = uint8")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--bands", action="store_true", help="CTU-row bands of every picture over the ranks (config 3); implies strong scaling")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stages", action="store_true", help="skip the first-pass / pre-analysis / motion-search stage report")
    ap.add_argument("--no-family", action="store_true", help="skip the line of the reference's Bayesian-optimisation family member 32 / 64 / 128")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-to-host (PCIe-inclusive) leg")
    ap.add_argument("--cpu-procs", type=int, default=None, help="processes of the all-cores CPU baseline (default: host cores, at most 16)")
    ap.add_argument("--arith", choices=("i8", "f16"), default=None,
                    help="arithmetic of the classifier's conv2 / conv3 for the headline (default: the library's, i8); the other form is timed "
                         "beside it (\"variants\") unless --no-variants")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal on a 1-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--launch-check", action="store_true",
                    help="only prove the N-rank launch: rendezvous, one barrier and one all-gather of the ranks over the host group, then exit "
                         "(no GPU is touched: the CPU test of the self-launcher uses this with --backend gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: start the N ranks ourselves, one process per GPU, exactly as the driver's launcher would
        # (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...).  This parent has not touched the GPU and never
        # does: it only relays the children's output (rank 0 prints the JSON line) and their exit code -- no exec of a GPU process.
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, "
                         f"or run `python bench.py --gpus {args.gpus}` without WORLD_SIZE set and it starts the ranks itself")
    if args.launch_check:
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
        got = [None] * world
        dist.all_gather_object(got, (rank, local, os.getpid()))
        if rank == 0:
            print(json.dumps({"launch_check": True, "world": world, "ranks": sorted(g[0] for g in got), "distinct_processes": len({g[2] for g in got})}), flush=True)
        dist.destroy_process_group()
        return
    mode = "bands" if args.bands else "frames"
    scaling = "strong" if args.bands else args.scaling
    W = args.width or (3840 if args.bands else 1920)
    H = args.height or (2160 if args.bands else 1080)
    NF = args.frames or (16 if args.bands else 64)
    bd = args.bit_depth

    # ---- CPU baseline first: its workers are children of a process that has not touched the GPU yet ----
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle_py as op
        if op.have_ref():
            cpu = cpu_baseline_reference(1920, 1080, bd, procs=args.cpu_procs)

    import torch
    import torch.distributed as dist
    from fasthevc_amd import capi, frames, gather, weights

    if args.one_device:
        assert args.backend != "nccl", "RCCL refuses two ranks on one GPU"
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    group, host_group = gather.init_groups(world, rank, dev, args.backend)

    trained = os.path.join(ROOT, "fasthevc_amd", "weights", "depthnet_v2.fhw")
    if os.path.exists(trained):
        w, wdesc = weights.load(trained), "trained weights fasthevc_amd/weights/depthnet_v2.fhw"
    else:
        w, wdesc = weights.random_weights(0), "random-init weights"  # same architecture, same arithmetic
    total_frames = world * NF if (mode == "frames" and scaling == "weak") else NF
    ctx = capi.Context(W, H, bd, w, device=local, max_frames=max(1, min(NF, 16)), arith=args.arith)
    arith = ctx.cnn_arith
    cw, ch, n_ctus = ctx.ctus_x, ctx.ctus_y, ctx.num_ctus
    fg = gather.FlagGather(mode, world, rank, total_frames, cw, ch, dev, group=group, host_group=host_group)
    f0, f1 = fg.frames
    r0, r1 = fg.rows
    nf_local = f1 - f0

    # synthetic GOP resident in HBM: the pinned "hetero" frame, panned 3 px per frame; a rank holds the pictures it works on
    base = torch.from_numpy(frames.hetero_luma(W, H).astype(np.int16)).to(dev) << (bd - 8)
    if args.sample_bytes == 2:
        margin = frames.HM_MARGIN
        stride, rows = W + 2 * margin, H + 2 * margin
        gop = torch.zeros((nf_local, rows, stride), dtype=torch.int16, device=dev)
        for f in range(nf_local):
            gop[f, margin:margin + H, margin:margin + W] = torch.roll(base, shifts=3 * (f0 + f), dims=1)
        origin, frame_stride = margin * stride + margin, rows * stride
        luma_ptr = gop.data_ptr() + 2 * origin
    else:
        assert bd == 8
        stride, frame_stride = W, W * H
        gop = torch.stack([torch.roll(base, shifts=3 * (f0 + f), dims=1) for f in range(nf_local)]).to(torch.uint8).contiguous()
        luma_ptr = gop.data_ptr()

    # every rank ends a step with the depth maps of the WHOLE step (all ranks' pictures / bands) in `gathered`
    gathered = torch.zeros((total_frames, n_ctus, 256), dtype=torch.uint8, device=dev)
    own = torch.zeros((max(1, fg.local_ctus), 256), dtype=torch.uint8, device=dev)   # this rank's maps, compact
    had = torch.zeros(max(1, fg.local_ctus), dtype=torch.int32, device=dev)
    # A real (non-null) torch stream: the collective is ordered against torch's CURRENT stream, so the kernels, the gather's
    # dependencies and the expansion are all expressed on one stream
    torch.cuda.synchronize()
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0
    pending = [False]

    def finish_gather():
        """wait for the collective in flight and expand its words into the depth maps of the whole step"""
        if pending[0]:
            words = fg.finish()
            ctx.expand_depth_flags_device(words.data_ptr(), total_frames, gathered.data_ptr(), stream=stream)
            pending[0] = False

    def step(i):
        b = i & 1
        dst = gathered if world == 1 else own
        ctx.predict_frames_device(luma_ptr, args.sample_bytes, stride, frame_stride, nf_local, dst.data_ptr(), had.data_ptr(), None,
                                  rows=(r0, r1), stream=stream, d_flags=fg.local_words(b).data_ptr() if world > 1 else None)
        if world > 1:
            finish_gather()   # step i-1's gather has had this step's kernels to hide under
            fg.start(b)       # the path's only collective
            pending[0] = True

    def fence():
        finish_gather()       # every step's maps are gathered and expanded before the clock stops
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(max(0, args.prewarm) + args.warmup):
        step(i)
    fence()
    ctx.enable_kernel_timing(True)
    ctx.kernel_timing(0, reset=True)
    ctx.kernel_timing(1, reset=True)
    regions = []
    for _ in range(max(1, args.repeats)):
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64)  # the default group is gloo (gather.init_groups): host tensors
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        regions.append(dt)
    cnn_ms, cnn_n = ctx.kernel_timing(0)
    had_ms, had_n = ctx.kernel_timing(1)

    # ---- the other arithmetic form of the classifier, the same steps on the same box (one timed region): a second line, never `value` ----
    variants = None
    if not args.no_variants:
        other = "f16" if arith == "i8" else "i8"
        ctx.set_cnn_arith(other)
        for i in range(args.warmup):
            step(i)
        fence()
        ctx.kernel_timing(0, reset=True)
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        fence()
        dto = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dto], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dto = float(t.item())
        o_ms, o_n = ctx.kernel_timing(0, reset=True)
        ctx.set_cnn_arith(arith)
        variants = {other: {"seconds": dto, "cnn_ms": o_ms, "launches": o_n}}

    # untimed check of the N > 1 path: what gather + expansion left in `gathered` must contain this rank's own maps at their
    # place, and the same bytes on every rank
    gather_ok = None
    if world > 1:
        check = torch.zeros((max(1, fg.local_ctus), 256), dtype=torch.uint8, device=dev)
        ctx.predict_frames_device(luma_ptr, args.sample_bytes, stride, frame_stride, nf_local, check.data_ptr(), None, None, rows=(r0, r1), stream=stream)
        torch.cuda.synchronize()
        g = gathered.view(total_frames, ch, cw, 256)
        mine = g[f0:f1, r0:r1].reshape(-1, 256)
        chk = torch.tensor([float(torch.equal(check, mine)), float(gathered.to(torch.int64).sum().item())], dtype=torch.float64)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        gather_ok = bool(lo[0].item() == 1.0 and lo[1].item() == hi[1].item() and len(torch.unique(gathered)) > 1)

    # ---- host to host ("depth map delivered to host memory"): the same GOP from pinned host memory through
    # fhevc_predict_frames (chunks of 16 pictures over two streams), PCIe included; never `value` ----
    host = None
    if rank == 0 and world == 1 and not args.no_host_path:
        host = {}
        torch.cuda.synchronize()
        for label, sb in (("uint8_planes", 1), ("int16_pel_planes", 2)):
            if sb == 1 and bd != 8:
                continue
            if sb == 1:
                src = ctx.alloc_host((NF, H, W), np.uint8)
                src[:] = torch.stack([torch.roll(base, shifts=3 * f, dims=1) for f in range(NF)]).to(torch.uint8).cpu().numpy()
                kw = {}
            else:
                m = frames.HM_MARGIN
                src = ctx.alloc_host((NF, H + 2 * m, W + 2 * m), np.int16)
                src[:] = 0
                src[:, m:m + H, m:m + W] = torch.stack([torch.roll(base, shifts=3 * f, dims=1) for f in range(NF)]).cpu().numpy()
                kw = {"origin": m * (W + 2 * m) + m, "stride": W + 2 * m, "frame_stride": (H + 2 * m) * (W + 2 * m)}
            dout = ctx.alloc_host((NF, n_ctus, 256), np.uint8)
            hout = ctx.alloc_host((NF, n_ctus), np.int32)
            ctx.predict_frames(src, qp=32, depth_out=dout, had_out=hout, **kw)   # warm-up (allocates the ring)
            s0 = ctx.stats()
            times = []
            for _ in range(5):
                t0 = time.perf_counter()
                ctx.predict_frames(src, qp=32, depth_out=dout, had_out=hout, **kw)
                times.append(time.perf_counter() - t0)
            s1 = ctx.stats()
            moved = (s1["bytes_h2d"] - s0["bytes_h2d"] + s1["bytes_d2h"] - s0["bytes_d2h"]) / 5
            tm = float(np.median(times))
            same = bool(np.array_equal(dout.reshape(-1), gathered.cpu().numpy().reshape(-1))) if (sb == args.sample_bytes and total_frames == NF) else None
            host[label] = {"ctu_per_s": NF * n_ctus / tm, "ms_per_gop": tm * 1e3, "min_ms": min(times) * 1e3, "max_ms": max(times) * 1e3,
                           "pcie_GB_s": moved / tm / 1e9, "pcie_peak_GB_s": PEAK_PCIE_GBS, "pcie_frac": moved / tm / 1e9 / PEAK_PCIE_GBS,
                           "bytes_per_gop": moved, "equals_device_resident_maps": same}
            for a in (src, dout, hout):
                ctx.free_host(a)
        host["how"] = "fhevc_predict_frames: pinned host GOP -> H2D -> source Hadamard + depth CNN -> D2H of maps + Hadamards, 16-picture chunks on two streams, 5 runs (median)"

    # the path's other stages, measured AFTER the timed region on the same GOP -- reported, never in `value`
    stages = None
    if rank == 0 and not args.no_stages:
        nb = min(nf_local, 8)
        nodes = torch.zeros((nb * n_ctus * 85, 2), dtype=torch.float64, device=dev)
        act = torch.zeros((nf_local, ctx.aq_layout(4)[-1]), dtype=torch.float64, device=dev)
        mot = torch.zeros((max(1, nf_local - 1) * n_ctus * 85, 4), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for timed in (False, True):
            for k in (2, 3, 4):
                ctx.kernel_timing(k, reset=True)
            for _ in range(3):
                ctx.intra_first_pass_device(luma_ptr, args.sample_bytes, stride, frame_stride, nb, nodes.data_ptr(), stream=stream, qp=32)
                ctx.preanalyze_frames_device(luma_ptr, args.sample_bytes, stride, frame_stride, nf_local, act.data_ptr(), 4, stream=stream)
                if nf_local > 1:
                    ctx.motion_search_device(luma_ptr, args.sample_bytes, stride, frame_stride, nf_local, mot.data_ptr(), stream=stream, qp=38, search_range=4)
            torch.cuda.synchronize()
        fp_ms, _ = ctx.kernel_timing(2)
        pre_ms, _ = ctx.kernel_timing(3)
        mo_ms, _ = ctx.kernel_timing(4)
        int_ops = nb * n_ctus * 4 * 35 * 64 * (64 + 575)  # per (level, mode, 8x8 tile) 64 predicted samples + ~575 Hadamard ops
        pre_bytes = nf_local * (W * H * args.sample_bytes + act.shape[1] * 8)
        mo_ops = (nf_local - 1) * n_ctus * 81 * 64 * (64 + 575)  # per (vector, 8x8 tile): 64 differences + ~575 Hadamard ops
        stages = {
            "first_pass": {"kernel": "fhevc_first_pass_kernel", "frames": nb, "avg_launch_ms": fp_ms, "ctu_per_s": nb * n_ctus / (fp_ms * 1e-3),
                           "bound": "int VALU/LDS", "achieved_Tintop_s": int_ops / (fp_ms * 1e-3) / 1e12, "peak_Tintop_s": PEAK_INT32_TOPS,
                           "frac": int_ops / (fp_ms * 1e-3) / 1e12 / PEAK_INT32_TOPS},
            "preanalyze": {"kernel": "fhevc_preanalyze_kernel", "frames": nf_local, "avg_launch_ms": pre_ms, "bound": "hbm",
                           "achieved_GB_s": pre_bytes / (pre_ms * 1e-3) / 1e9, "peak_GB_s": PEAK_HBM_GBS,
                           "frac": pre_bytes / (pre_ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
        }
        if nf_local > 1 and mo_ms:
            # the kernel runs on PACKED 16-bit VALU (two sample-ops per lane-op): its peak is twice the 32-bit one
            stages["motion_search"] = {"kernel": "fhevc_motion_kernel", "picture_pairs": nf_local - 1, "search_range": 4, "avg_launch_ms": mo_ms,
                                       "ctu_per_s": (nf_local - 1) * n_ctus / (mo_ms * 1e-3), "bound": "int VALU/LDS (packed 16-bit)",
                                       "achieved_Tintop_s": mo_ops / (mo_ms * 1e-3) / 1e12, "peak_Tintop_s": 2 * PEAK_INT32_TOPS,
                                       "frac": mo_ops / (mo_ms * 1e-3) / 1e12 / (2 * PEAK_INT32_TOPS)}
        if nf_local > 1 and args.bit_depth == 8:
            # the same search at HM's own SearchRange 64 (SAD, k_motion_wide.hip: 16 641 vectors per node instead of 81), on the first 5 pictures
            nw = min(nf_local, 5)
            ctx.set_motion_distortion("sad")
            for timed in (False, True):
                ctx.kernel_timing(4, reset=True)
                for _ in range(2):
                    ctx.motion_search_device(luma_ptr, args.sample_bytes, stride, frame_stride, nw, mot.data_ptr(), stream=stream, qp=38, search_range=64)
                torch.cuda.synchronize()
            mw_ms, _ = ctx.kernel_timing(4)
            ctx.set_motion_distortion("satd")
            qsad_ops = (nw - 1) * n_ctus * 129 * 129 * 4096 / 16  # algorithmic lane-operations: one v_qsad_pk_u16_u8 = 16 sample differences
            stages["motion_search_range64"] = {"kernel": "fhevc_motion_wide_kernel", "picture_pairs": nw - 1, "search_range": 64, "distortion": "SAD",
                                               "avg_launch_ms": mw_ms, "ctu_per_s": (nw - 1) * n_ctus / (mw_ms * 1e-3),
                                               "vectors_per_s": (nw - 1) * n_ctus * 85 * 16641 / (mw_ms * 1e-3),
                                               "bound": "int VALU (v_qsad_pk_u16_u8: 16 sample differences per lane-op)",
                                               "achieved_Tlaneop_s": qsad_ops / (mw_ms * 1e-3) / 1e12, "peak_Tlaneop_s": PEAK_INT32_TOPS,
                                               "frac": qsad_ops / (mw_ms * 1e-3) / 1e12 / PEAK_INT32_TOPS,
                                               "note": "frac counts the SAD instructions only; keys, minima and sums are about as many again"}
    ctx.enable_kernel_timing(False)

    def measured_traffic(kernel):
        """HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
        FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) -- only for the workload they were taken on."""
        if not (NF == 64 and args.sample_bytes == 2 and (W, H) == (1920, 1080) and mode == "frames"):
            return None, None
        for rnd, stem in [(r, st) for r in ("r04", "r03", "r02", "r01") for st in ("pmc_bench_frames64_int16", "pmc_kernels")]:   # newest round first
            path = os.path.join(ROOT, "profiles", f"{rnd}_{stem}.json")
            if os.path.exists(path):
                d = json.load(open(path))
                for k, v in d.items():
                    if kernel == "fhevc_cnn_depth_kernel" and k.startswith(kernel + "<"):
                        # template arguments <STAMPS, HAD, ARITH[, TRIO]>: ARITH 0 = the 16-bit form, 1 / 2 = the i8 form (files of before the i8 form: two arguments)
                        targs = k.split("<", 1)[1].split(">", 1)[0].split(",")
                        if (arith == "i8") != (len(targs) >= 3 and targs[2].strip() in ("1", "2")):
                            continue
                    if k.startswith(kernel) and isinstance(v, dict) and "hbm_read_bytes_corrected" in v and "hbm_write_bytes" in v:
                        return v["hbm_read_bytes_corrected"] + v["hbm_write_bytes"], f"profiles/{rnd}_{stem}.json (commit {d.get('commit', 'of that round')})"
        return None, None

    def measured_mfma_busy():
        """SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel cycles) of the depth kernel from the committed PMC pass: how much of the launch the matrix
        pipes were busy, at the 2.4 GHz the peak is quoted at and at the clock the chip held (GRBM_GUI_ACTIVE / 8 / time, where the pass has it)"""
        if not (NF == 64 and args.sample_bytes == 2 and (W, H) == (1920, 1080) and mode == "frames"):
            return None
        for rnd in ("r04", "r03", "r02"):
            path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_bench_frames64_int16.json")
            if not os.path.exists(path):
                continue
            d = json.load(open(path))
            for k, v in d.items():
                if not (k.startswith("fhevc_cnn_depth_kernel<") and isinstance(v, dict) and "SQ_VALU_MFMA_BUSY_CYCLES" in v):
                    continue
                targs = k.split("<", 1)[1].split(">", 1)[0].split(",")
                if (arith == "i8") != (len(targs) >= 3 and targs[2].strip() in ("1", "2")):
                    continue
                ms = v.get("avg_ms") or v.get("duration_ms") or cnn_ms
                if not ms:
                    continue
                out = {"SQ_VALU_MFMA_BUSY_CYCLES": v["SQ_VALU_MFMA_BUSY_CYCLES"], "launch_ms_used": ms,
                       "busy_frac_at_2.4GHz": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * ms * 2.4e6), "source": f"profiles/{rnd}_pmc_bench_frames64_int16.json"}
                if v.get("GRBM_GUI_ACTIVE"):
                    out["busy_frac_at_held_clock"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * v["GRBM_GUI_ACTIVE"] / 8.0)
                return out
        return None

    # ---- the reference's Bayesian-optimisation network family (NetworkDepth 1: 32 / 64 / 128) on the same GOP: its own line, never `value` ----
    family_line = None
    if rank == 0 and world == 1 and not args.no_family:
        fblob = os.path.join(ROOT, "fasthevc_amd", "weights", "depthnet_family_d1.fhw")
        fw = weights.load_any(fblob) if os.path.exists(fblob) else weights.random_family((32, 64, 128), 1, seed=0)
        fctx = capi.Context(W, H, bd, fw, device=local, max_frames=max(1, min(NF, 16)))
        fctx.enable_kernel_timing(True)
        fdepth = torch.zeros((nf_local, n_ctus, 256), dtype=torch.uint8, device=dev)
        for _ in range(10):
            fctx.predict_frames_device(luma_ptr, args.sample_bytes, stride, frame_stride, nf_local, fdepth.data_ptr(), None, None, stream=stream)
        torch.cuda.synchronize()
        fctx.kernel_timing(0, reset=True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fctx.predict_frames_device(luma_ptr, args.sample_bytes, stride, frame_stride, nf_local, fdepth.data_ptr(), None, None, stream=stream)
        torch.cuda.synchronize()
        fdt = time.perf_counter() - t0
        f_ms, f_n = fctx.kernel_timing(0, reset=True)
        fctx.close()
        c1, c2, c3 = 32, 64, 128
        fmac = 4096 * 9 * c1 + 1024 * 9 * c1 * c2 + 256 * 9 * c2 * c3 + (64 * 4 + 4 * 64 + 16 * 16) * c3 * 2
        fach = 2 * fmac * nf_local * n_ctus / (f_ms * 1e-3) / 1e12 if f_ms else None
        family_line = {"network": "NetworkDepth 1 member of the reference's Bayesian-optimisation family: conv3x3 x 32 / 64 / 128, one convolution per block "
                                  "(Optimize...Example.m:103-106, 233-259)", "weights": "trained (fasthevc_amd/weights/depthnet_family_d1.fhw)" if os.path.exists(fblob) else "random-init",
                       "value": nf_local * n_ctus * args.steps / fdt, "unit": "CTU/s", "ms_per_step": fdt / args.steps * 1e3, "op_per_ctu": 2 * fmac,
                       "note": "depth maps only (the source Hadamard is not fused into this kernel)",
                       "roofline": {"bound": "mfma", "kernel": "fhevc_cnn_family_kernel<32, 64, 128>", "achieved": fach, "peak": PEAK_I8_TOPS, "unit": "TOP/s (2 per MAC)",
                                    "frac": fach / PEAK_I8_TOPS if fach else None, "avg_launch_ms": f_ms, "launches": f_n}}

        # the deeper members: 23 / 46 / 92 x 2 through ONE LDS-resident kernel (k_cnn_d2.inc, round 4) on the whole GOP; 18 / 36 / 72 x 3 layer by layer through
        # HBM (k_cnn_layers.inc) on the first 16 pictures
        deeper = {}
        for widths, depth in (((23, 46, 92), 2), ((18, 36, 72), 3)):
          nfd = nf_local if depth == 2 else min(nf_local, 16)
          try:   # (a secondary line: a failure here is reported in place, it never takes the headline down)
            dblob = os.path.join(ROOT, "fasthevc_amd", "weights", f"depthnet_family_d{depth}.fhw")
            dw = weights.load_any(dblob) if os.path.exists(dblob) else None
            trained = dw is not None and tuple(int(v) for v in dw.get("widths", ())) == widths and int(dw.get("depth", 0)) == depth
            if not trained:
                dw = weights.random_family(widths, depth, seed=0)
            dctx = capi.Context(W, H, bd, dw, device=local, max_frames=nfd)
            dctx.enable_kernel_timing(True)
            for _ in range(2):
                dctx.predict_frames_device(luma_ptr, args.sample_bytes, stride, frame_stride, nfd, fdepth.data_ptr(), None, None, stream=stream)
            torch.cuda.synchronize()
            dctx.kernel_timing(0, reset=True)
            for _ in range(5):
                dctx.predict_frames_device(luma_ptr, args.sample_bytes, stride, frame_stride, nfd, fdepth.data_ptr(), None, None, stream=stream)
            torch.cuda.synchronize()
            d_ms, d_n = dctx.kernel_timing(0, reset=True)
            dctx.close()
            mac, ci, n = 0, 1, 64
            for b in range(3):
                for j in range(depth):
                    mac += n * n * 9 * ci * widths[b]
                    ci = widths[b]
                if b < 2:
                    n //= 2
            mac += (64 * 4 + 4 * 64 + 16 * 16) * widths[2] * 2
            ach = 2 * mac * nfd * n_ctus / (d_ms * 1e-3) / 1e12
            kernels = "fhevc_cnn_d2_kernel<1, 2, 3> (one launch, activations in LDS)" if depth == 2 else \
                f"{3 * depth - 1} x fhevc_layer_conv_kernel (the first convolution inside the second) + stage + heads"
            d_traffic, d_src = measured_traffic("fhevc_cnn_d2_kernel") if depth == 2 else (None, None)
            deeper[f"{widths[0]}/{widths[1]}/{widths[2]} x {depth}"] = {
                "weights": "trained (" + os.path.basename(dblob) + ")" if trained else "random-init",
                "value": nfd * n_ctus / (d_ms * 1e-3), "unit": "CTU/s", "pictures": nfd, "ms_per_launch": d_ms, "op_per_ctu": 2 * mac,
                "roofline": {"bound": "mfma", "kernels": kernels, "achieved": ach, "peak": PEAK_I8_TOPS,
                             "unit": "TOP/s (2 per MAC, unpadded)", "frac": ach / PEAK_I8_TOPS,
                             "algorithmic_bytes": nfd * (W * H * args.sample_bytes + n_ctus * 256), "traffic": d_traffic, "traffic_source": d_src}}
          except Exception as exc:   # noqa: BLE001
            deeper[f"{widths[0]}/{widths[1]}/{widths[2]} x {depth}"] = {"error": repr(exc)}
        family_line["deeper_members"] = deeper

    if rank == 0:
        ctus_per_step = total_frames * n_ctus
        thr = sorted(ctus_per_step * args.steps / t for t in regions)
        value = float(np.median(thr))
        dt = ctus_per_step * args.steps / value
        local_ctus = fg.local_ctus
        flop_per_launch = FLOP_PER_CTU * local_ctus
        sample_b = args.sample_bytes
        band_px = nf_local * W * min(H, (r1 - r0) * 64)
        bytes_per_launch_had = band_px * sample_b + local_ctus * 4
        traffic, traffic_src = measured_traffic("fhevc_cnn_depth_kernel")
        roof = {"bound": "mfma", "kernel": f"fhevc_cnn_depth_kernel ({arith} form)", "achieved": flop_per_launch / (cnn_ms * 1e-3) / 1e12 if cnn_ms else None,
                "peak": ARITH_PEAK[arith], "unit": "TFLOP/s" if arith == "f16" else "TOP/s (2 per MAC)", "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": cnn_ms, "launches": cnn_n, "flop_per_ctu": FLOP_PER_CTU, "ctus_per_launch": local_ctus,
                "peak_note": "dense 16-bit MFMA" if arith == "f16" else "dense i8 MFMA; conv1 (6 % of the MACs) runs on the 16-bit MFMA at half that rate"}
        roof["frac"] = roof["achieved"] / roof["peak"] if roof["achieved"] else None
        busy = measured_mfma_busy()
        if busy:
            roof["mfma_busy"] = busy
        if variants:
            for name, v in variants.items():
                ach = flop_per_launch / (v["cnn_ms"] * 1e-3) / 1e12 if v["cnn_ms"] else None
                variants[name] = {"value": ctus_per_step * args.steps / v["seconds"], "unit": "CTU/s", "ms_per_step": v["seconds"] / args.steps * 1e3,
                                  "dtype": ARITH_DTYPE[name], "regions": 1,
                                  "roofline": {"bound": "mfma", "kernel": f"fhevc_cnn_depth_kernel ({name} form)", "achieved": ach, "peak": ARITH_PEAK[name],
                                               "unit": "TFLOP/s" if name == "f16" else "TOP/s (2 per MAC)", "frac": ach / ARITH_PEAK[name] if ach else None,
                                               "avg_launch_ms": v["cnn_ms"], "launches": v["launches"]}}
        # the source Hadamard rides on the depth kernel since round 2: the stand-alone HBM-bound kernel that remains on the path is the AQ
        # pre-analysis (N3), timed with the other stages after the timed region
        hbm = None
        if stages and stages.get("preanalyze"):
            ptraffic, ptraffic_src = measured_traffic("fhevc_preanalyze_kernel")
            pa = stages["preanalyze"]
            hbm = {"bound": "hbm", "kernel": "fhevc_preanalyze_kernel (four AQ layers)", "achieved": pa["achieved_GB_s"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                   "frac": pa["frac"], "traffic": ptraffic, "traffic_source": ptraffic_src, "avg_launch_ms": pa["avg_launch_ms"], "frames": pa["frames"]}
        geom = "BQTerrace geometry" if (W, H) == (1920, 1080) else "synthetic"
        line = {
            "metric": "CTU depth decisions/sec at 1080p all-intra + BD-rate delta vs full-RDO HM" if (W, H) == (1920, 1080) else f"CTU depth decisions/sec at {W}x{H} all-intra",
            "value": value, "unit": "CTU/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": max(0, args.prewarm), "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": ARITH_DTYPE[arith],
            "dtype_note": ("the classifier is an exact integer network (int8 weights, 8-bit activations): since round 2 the library's default runs conv2 / conv3 on the "
                           "i8 MFMAs instead of the 16-bit ones -- the same integers out, bit for bit (every GPU parity test runs both forms); this is a change of the "
                           "headline's arithmetic type, not of its precision; \"variants\" times the other form in the same process and prices each form against its "
                           "own dense MFMA peak; --arith f16 makes the 16-bit form (round 1's) the headline"),
            "data": "synthetic",
            "config": {"workload": f"{geom} {W}x{H} all-intra QP32, GOP of {total_frames} synthetic 'hetero' frames "
                                   f"({'int16 Pel planes, HM stride/margins' if sample_b == 2 else 'uint8 planes'}) resident in HBM, "
                                   f"source Hadamard + CTU-batched CNN depth predictor, {wdesc}",
                       "frames_per_step": total_frames, "frames_per_gpu": nf_local, "ctu_rows_per_gpu": r1 - r0, "ctus_per_frame": n_ctus, "bit_depth": bd,
                       "sharding": "single GPU" if world == 1 else (
                           ("CTU-row bands of every picture" if mode == "bands" else "pictures dealt to ranks") +
                           " + one all-gather of the depth decisions (4-byte split-flag words per CTU, padded equal slices, expanded on every rank)")},
            "repeats": {"n": len(regions), "region_steps": args.steps, "min": thr[0], "median": value, "max": thr[-1],
                        "region_ms": [round(t * 1e3, 3) for t in regions]},
            "roofline": roof,
        }
        if hbm:
            line["roofline_hbm_kernel"] = hbm
        if family_line:
            line["family"] = family_line
        bdr = quoted_bd_rate()
        if bdr:
            line["bd_rate"] = bdr
        if host:
            line["host_to_host"] = host
        if stages:
            line["stages"] = stages
        if variants:
            line["variants"] = variants
        if world > 1:
            line["gather_verified"] = gather_ok
            line["collective"] = fg.status
            # no SCALE record of an earlier round exists: the builder has no multi-GPU box, RCCL over xGMI runs for the first time in the driver's own run
            line["multi_gpu_note"] = "N > 1 unmeasured on hardware by the builder (gloo rehearsals only); this line is the first RCCL measurement"
        if world == 1 and not args.no_cpu_baseline:
            if cpu is None:
                cpu = cpu_baseline_port(1920, 1080, bd, w)
            else:
                hook = gpu_hook_leg(1920, 1080, bd, crops=8)                       # the SHIPPED hook default
                if hook:
                    hook["speedup"] = hook["value"] / cpu["value"]
                    cpu["with_gpu_hook"] = hook
                    fp = gpu_hook_leg(1920, 1080, bd, crops=6, first_pass=True)   # shipped margins + the first pass consumed
                    if fp:
                        fp["speedup"] = fp["value"] / cpu["value"]
                        cpu["with_gpu_hook_first_pass"] = fp
                    matched = gpu_hook_leg(1920, 1080, bd, crops=6, margins=(64000, 32000))   # the content-matched setting, beside it
                    if matched:
                        matched["speedup"] = matched["value"] / cpu["value"]
                        cpu["with_gpu_hook_content_matched"] = matched
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

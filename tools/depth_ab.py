#!/usr/bin/env python3
"""A/B of the depth kernel alone, interleaved rounds in ONE process on ONE box (cdna_hip_programming.md rule 24).

usage: python tools/depth_ab.py [--rounds 5] [--reps 40] name[:ENV=VAL[,ENV=VAL..]][:nohad] ...
Each variant is a context created under its environment settings (the library reads its knobs in fhevc_create); `nohad` runs
it without the per-CTU source Hadamard output; the pseudo-variable LIB=<name> takes build/ab/<name>.so (tools/build_variant.sh)
instead of the in-tree library.  Workload: bench.py's (64 x 1080p hetero, HM-layout int16 planes, trained
weights).  Prints per variant the median / min of the per-round HIP-event averages of the depth kernel."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fasthevc_amd import capi, frames, weights  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--no-check", action="store_true", help="sensitivity experiments whose variants compute wrong results on purpose")
ap.add_argument("variants", nargs="+")
args = ap.parse_args()

W, H, NF = 1920, 1080, 64
dev = torch.device("cuda:0")
torch.cuda.init()
blob = os.path.join(ROOT, "fasthevc_amd", "weights", "depthnet_v2.fhw")
w = weights.load(blob)
base = frames.hetero_luma(W, H)
planes = np.stack([frames.to_pel_plane(np.roll(base, 3 * f, axis=1), 8)[0] for f in range(NF)])
_, org, stride = frames.to_pel_plane(base, 8)
d16 = torch.from_numpy(planes).to(dev)
fs = planes.shape[1] * planes.shape[2]
ptr = d16.data_ptr() + 2 * org

ctxs = []
for spec in args.variants:
    parts = spec.split(":")
    name, envs, nohad = parts[0], {}, False
    for p in parts[1:]:
        if p == "nohad":
            nohad = True
        elif p:
            for kv in p.split(","):
                k, v = kv.split("=")
                envs[k] = v
    fam = envs.pop("FAMILY", None)   # FAMILY=32x64x128: random-init weights of that member of the reference's Bayesian-optimisation family
    lib_path = envs.pop("LIB", None)
    if lib_path is not None:
        lib_path = os.path.join(ROOT, "build", "ab", lib_path + ".so")
    old = {k: os.environ.get(k) for k in envs}
    os.environ.update(envs)
    ctx = capi.Context(W, H, 8, weights.random_family(tuple(int(v) for v in fam.split("x")), 1, seed=0) if fam else w, max_frames=NF, lib_path=lib_path)
    for k, v in old.items():
        if v is None:
            del os.environ[k]
        else:
            os.environ[k] = v
    ctx.enable_kernel_timing(True)
    ctxs.append((name, ctx, nohad))

depth = torch.zeros((NF, ctxs[0][1].num_ctus, 256), dtype=torch.uint8, device=dev)
had = torch.zeros((NF, ctxs[0][1].num_ctus), dtype=torch.int32, device=dev)
ref = None
res = {n: [] for n, _, _ in ctxs}
for rnd in range(args.rounds + 1):  # round 0 warms the clocks up and is dropped
    for name, ctx, nohad in ctxs:
        for _ in range(5):
            ctx.predict_frames_device(ptr, 2, stride, fs, NF, depth.data_ptr(), None if nohad else had.data_ptr())
        torch.cuda.synchronize()
        ctx.kernel_timing(0, reset=True)
        for _ in range(args.reps):
            ctx.predict_frames_device(ptr, 2, stride, fs, NF, depth.data_ptr(), None if nohad else had.data_ptr())
        torch.cuda.synchronize()
        ms, n = ctx.kernel_timing(0, reset=True)
        if rnd:
            res[name].append(ms)
        if rnd == 1 and not args.no_check:  # every variant must deliver the same maps (and sums)
            cur = (depth.clone(), None if nohad else had.clone())
            if ref is None:
                ref = cur
            else:
                assert torch.equal(cur[0], ref[0]), f"{name}: depth maps differ from the first variant"
                if cur[1] is not None and ref[1] is not None:
                    assert torch.equal(cur[1], ref[1]), f"{name}: Hadamard sums differ from the first variant"
for name, v in res.items():
    v = sorted(v)
    print(f"{name:28s} median {v[len(v) // 2]:.4f} ms  min {v[0]:.4f}  max {v[-1]:.4f}   ({NF * ctxs[0][1].num_ctus / v[len(v) // 2] / 1e3:.1f} M CTU/s)")

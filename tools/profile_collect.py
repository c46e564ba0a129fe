#!/usr/bin/env python3
"""Copy the summaries of gpurun_out/<tag>/ (tools/profile_round.sh) into profiles/ under the round's names."""
import glob
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
pairs = {
    "bench_default.log": f"{tag}_bench_default.log",
    "bench_under_rocprof.json": f"{tag}_bench_under_rocprof.json",
    "bench_kernels.json": f"{tag}_bench_kernels.json",
    "pmc_bench.json": f"{tag}_pmc_bench_frames64_int16.json",
    "pmc_kernels.json": f"{tag}_pmc_kernels.json",
}
for a, b in pairs.items():
    shutil.copyfile(os.path.join(src, a), os.path.join(dst, b))
# stamp the PMC summaries with the commit they were taken at (the GPU box has no .git): bench.py quotes it in roofline.traffic_source
import json
import subprocess
commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "fasthevc_amd/csrc", "bench.py"], capture_output=True, text=True).stdout.strip())
for b in (pairs["pmc_bench.json"], pairs["pmc_kernels.json"]):
    path = os.path.join(dst, b)
    d = json.load(open(path))
    d["commit"] = commit + (" + uncommitted kernel changes" if dirty else "")
    json.dump(d, open(path, "w"), indent=1, sort_keys=True)
stats = sorted(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
assert stats, "no kernel stats"
shutil.copyfile(stats[-1], os.path.join(dst, f"{tag}_kernel_stats_bench_default.csv"))  # gpurun merges runs: newest wins
stats2 = sorted(glob.glob(os.path.join(src, "stats_kernels", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
if stats2:
    shutil.copyfile(stats2[-1], os.path.join(dst, f"{tag}_kernel_stats_bench_kernels.csv"))
print("collected", sorted(os.listdir(dst)))

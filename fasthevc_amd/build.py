"""Build the HIP library (gfx950 only) in-tree: fasthevc_amd/lib/libfasthevc_hip.so.

hipcc cross-compiles without a GPU; the .so travels with the repository snapshot to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libfasthevc_hip.so")
SOURCES = ["fhevc_api.hip", "k_cnn.hip", "k_hadamard.hip", "k_firstpass.hip", "k_preanalyze.hip", "k_motion.hip", "k_motion_wide.hip"]
# per-source extra flags: the CNN kernel holds only finite integers in fp32, so the NaN-canonicalising v_max can go
EXTRA = {"k_cnn.hip": ["-ffinite-math-only", "-fno-signed-zeros"]}
# -ffp-contract=off: the first-pass cost is compared bit-for-bit with the CPU oracle's double arithmetic
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    # every header and every .inc fragment under csrc/ (k_cnn.hip includes k_cnn_family.inc, k_cnn_layers.inc, ...): an edit to any of them rebuilds
    headers = [os.path.join(os.path.dirname(HERE), "include", "fasthevc.h")]
    headers += sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc")))
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _newer(o, [s] + headers):
            jobs.append(["hipcc"] + FLAGS + EXTRA.get(src, []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _newer(LIB, objs):
        run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))

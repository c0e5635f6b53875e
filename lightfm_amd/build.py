"""Builds lightfm_amd/_lib/liblfm_hip.so with hipcc for gfx950 (in-tree).

    python -m lightfm_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off keeps the reference's
separate float32 multiply/add roundings (no FMA contraction);
-munsafe-fp-atomics selects the hardware global_atomic_add_f32.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_lib")
LIB = os.path.join(OUT_DIR, "liblfm_hip.so")
SOURCES = ["fit_kernels.hip", "fit_kernels_wide.hip", "warp_tile.hip", "warp_tile_lpr16.hip", "warp_tile_lpr32.hip",
           "warp_tile_lpr64.hip", "warp_tile_ahead.hip", "warp_tile_bpr.hip", "logistic_tile.hip", "feat_kernels.hip", "feat_kernels_wide.hip", "feat_kernels_hot.hip", "feat_kernels_ada.hip", "hot_slices.hip", "predict_kernels.hip", "csr_build.hip", "session.hip"]
HEADERS = ["device.hpp", "kernels.hpp", "pool.hpp", "warp_tile_kernel.hpp", "warp_tile_ahead.hpp", "warp_tile_narrow.hpp", "feat_kernel.hpp", os.path.join("..", "..", "include", "lfm_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]
if os.environ.get("LFM_BUILD_DEBUG"):  # line tables + symbols for rocgdb (same code generation)
    FLAGS.append("-g")
# A/B builds: LFM_BUILD_DEFINES="-DX -DY" LFM_BUILD_DIR=<dir under lightfm_amd/> python -m lightfm_amd.build
FLAGS += os.environ.get("LFM_BUILD_DEFINES", "").split()
if os.environ.get("LFM_BUILD_DIR"):
    OUT_DIR = os.path.join(HERE, os.environ["LFM_BUILD_DIR"])
    LIB = os.path.join(OUT_DIR, "liblfm_hip.so")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OUT_DIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC] + FLAGS + ["-c", s, "-o", o])
    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

"""Build libmms_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU.  Flags that matter for parity:
  -ffp-contract=off   the reference CPU build has no FMA contraction; mul and
                      add must round separately for bit-exact Euclidean paths.
  -fhip-fp32-correctly-rounded-divide-sqrt   IEEE sqrt / divide (hipcc default,
                      stated explicitly because the parity tests rely on it).
  -mllvm -amdgpu-kernarg-preload-count=16   leading kernel arguments are placed in
                      SGPRs at wave launch (gfx940+), so a latency-bound kernel does not
                      begin with a scalar fetch of its argument block.
"""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmms_hip.so")
LAYER_LIB = os.path.join(HERE, "libmms_caffe.so")

HIP_SOURCES = ["mms_abi.hip", "simcross_elementwise.hip", "bilinear.hip", "pairrank.hip", "ranking.hip", "embed.hip", "f64_paths.hip"]
def _hip_headers():
    """Every header under csrc/ is a dependency of every .hip object (panel_gemm.h is included by bilinear.hip,
    euclid_math.h by three sources, ...): found by glob so that a new header cannot be forgotten."""
    return sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hpp")))


HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-fno-fast-math", "-Wall", "-Wno-unused-function",
    "-mllvm", "-amdgpu-kernarg-preload-count=16",
]
# pairrank.hip: the arrival atomics of the fused step are issued by ONE lane and their return values are consumed
# after the gradient stores; the atomic optimizer's wave scan + readfirstlane would pull the wait to the issue.
HIPCC_FLAGS += os.environ.get("MMS_HIPCC_EXTRA", "").split()   # dev builds (-DMMS_STAMPS, -DMMS_ABLATE=.., ...)
EXTRA_FLAGS = {"pairrank.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP library cannot be built (no fallback exists)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile_one(args):
    cmd, verbose = args
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_hip(force=False, verbose=False):
    """One object per .hip source (compiled in parallel, rebuilt only when stale), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    hdrs = _hip_headers() + [os.path.join(ROOT, "include", "mms.h"), os.path.abspath(__file__)]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"]
    jobs, objs = [], []
    for f in HIP_SOURCES:
        src = os.path.join(CSRC, f)
        obj = os.path.join(objdir, f.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append(([_hipcc()] + cflags + EXTRA_FLAGS.get(f, []) + ["-c", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
                                                src, "-o", obj], verbose))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) // 2))) as ex:
            list(ex.map(_compile_one, jobs))
    if jobs or force or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


def build_layers(force=False, verbose=False):
    """C++ mirror of the Caffe Layer/Blob API on top of the C ABI."""
    src = os.path.join(CSRC, "caffe_layers.cpp")
    io_src = os.path.join(CSRC, "caffemodel_io.cpp")
    h5_src = os.path.join(CSRC, "hdf5_io.cpp")
    if not os.path.exists(src):
        return None
    deps = [src, io_src, h5_src, os.path.join(CSRC, "hdf5_io.hpp"), os.path.join(CSRC, "caffe_api.hpp"), os.path.join(ROOT, "include", "mms.h"),
            os.path.join(ROOT, "include", "mms_layer.h"), LIB, os.path.abspath(__file__)]
    deps = [d for d in deps if os.path.exists(d)]
    if not force and not _stale(LAYER_LIB, deps):
        return LAYER_LIB
    cmd = [_hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "--offload-arch=gfx950",
           "-Wall", "-I", os.path.join(ROOT, "include"), "-I", CSRC, src, io_src, h5_src, "-o", LAYER_LIB,
           "-L", HERE, "-lmms_hip", "-lz", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LAYER_LIB


def build_all(force=False, verbose=False):
    out = [build_hip(force, verbose)]
    l = build_layers(force, verbose)
    if l:
        out.append(l)
    return out


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))

"""Build libshdr.so (hand-written HIP kernels for gfx950) in-tree with hipcc.

    python singlehdr-tf2_amd/build.py [--force]

hipcc cross-compiles for gfx950 without a GPU.  The shared object lands next
to this file (git-ignored, but it travels to the GPU box with the snapshot).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.normpath(os.path.join(HERE, "..", "include"))
OBJ_DIR = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libshdr.so")

# per-file extra flags: the soft-histogram must not contract mul+sub into an FMA
SOURCES = {
    "api.cpp": [],
    "conv.hip": [],
    "conv_plan.hip": [],
    "conv_x3.hip": [],
    "conv_x3n.hip": [],
    "conv_f16.hip": [],
    "conv_f16_patch.hip": [],
    "conv_f16_w3.hip": [],
    "wgrad_x3.hip": [],
    "wgrad_f16.hip": [],
    "wgrad_f16_alltaps.hip": [],
    "elem_f16.hip": ["-ffp-contract=off"],
    "wgrad.hip": [],
    "wgrad_winograd.hip": [],
    "bwd.hip": [],
    "finetune.hip": [],
    "winograd.hip": [],
    "winograd_fused.hip": [],
    "frontend.hip": ["-ffp-contract=off"],
    "pool.hip": [],
    "crf.hip": [],
    "glue.hip": [],
    "imageio.hip": [],
    "camera.hip": ["-ffp-contract=off"],
}
COMMON = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + INCLUDE, "-I" + CSRC,
          "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, "shdr_internal.h"), os.path.join(INCLUDE, "shdr.h"), __file__]
    objs = []
    procs = []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + COMMON + extra + ["-x", "hip", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

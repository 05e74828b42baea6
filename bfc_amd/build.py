"""Build recipe for libbfc_gpu.so (hipcc --offload-arch=gfx950, in-tree, no JIT cache).

    python -m bfc_amd.build            # build if stale
    python -m bfc_amd.build --force
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(ROOT, "include")
OBJ = os.path.join(ROOT, "build", "obj")
SO = os.path.join(HERE, "libbfc_gpu.so")
ARCH = "gfx950"

HIP_SRCS = ["bfcg_kernels.hip", "bfcg_ctx.hip"]
C_SRCS = ["bfc_host.c", "bfc_count.c", "bfc_trim.c"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build step failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INC, "bfc_gpu.h")]
    objs = []
    for f in HIP_SRCS:
        src, obj = os.path.join(CSRC, f), os.path.join(OBJ, f + ".o")
        if force or _stale(obj, [src] + hdrs):
            if verbose:
                print("[build] hipcc", f)
            _run([HIPCC, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-I" + INC, "-I" + CSRC, "-c", src, "-o", obj])
        objs.append(obj)
    for f in C_SRCS:
        src, obj = os.path.join(CSRC, f), os.path.join(OBJ, f + ".o")
        if force or _stale(obj, [src] + hdrs):
            if verbose:
                print("[build] gcc", f)
            _run(["gcc", "-O2", "-g", "-Wall", "-std=gnu99", "-fPIC", "-I" + INC, "-I" + CSRC, "-c", src, "-o", obj])
        objs.append(obj)
    if force or _stale(SO, objs):
        if verbose:
            print("[build] link", SO)
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", SO] + objs + ["-lz", "-lpthread"])
    from . import gen
    gen.build()
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

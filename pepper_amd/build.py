"""In-tree build of the HIP extension: hipcc --offload-arch=gfx950 -> pepper_amd/csrc/libpepper_amd.so.

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels to the GPU box with
the gpurun snapshot.  `python -m pepper_amd.build` rebuilds unconditionally.
"""
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libpepper_amd.so")
SOURCES = ["api.hip", "gemm.hip", "rnn.hip", "head.hip"]
HEADERS = ["common.h", "kernels.h", os.path.join("..", "..", "include", "pepper_amd.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-result", "-o", LIB + ".tmp"] + SOURCES
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, cwd=CSRC, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))

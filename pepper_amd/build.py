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
SOURCES = ["api.hip", "inflate.hip", "gemm.hip", "gemm_h2.hip", "rnn.hip", "rnn_h2.hip", "mlp_h2.hip", "head.hip", "encoder.hip", "encoder_polish.hip", "realign.hip"]
HEADERS = ["common.h", "kernels.h", "encoder_common.h", os.path.join("..", "..", "include", "pepper_amd.h"),
           os.path.join("..", "..", "include", "pepper_amd_encoder.h"),
           os.path.join("..", "..", "include", "pepper_amd_realign.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


IO_LIB = os.path.join(CSRC, "libpepper_amd_io.so")
HDF5_PREFIX = os.environ.get("PEPPER_AMD_HDF5_PREFIX", "/opt/conda")


def _have_libdeflate(inc, lib):
    """Header and a linkable shared library, both under the HDF5 prefix (PEPPER_AMD_LIBDEFLATE=0 turns it off)."""
    if os.environ.get("PEPPER_AMD_LIBDEFLATE", "1") == "0":
        return False
    return os.path.exists(os.path.join(inc, "libdeflate.h")) and os.path.exists(os.path.join(lib, "libdeflate.so"))


def build_io(force=False, verbose=False):
    """g++ build of the HDF5 I/O helper (include/pepper_amd_io.h) against libhdf5 1.10.

    h5py is not installed for this image's torch interpreter; libhdf5 + headers live under
    /opt/conda (override with PEPPER_AMD_HDF5_PREFIX)."""
    src = os.path.join(CSRC, "hdf5io.cpp")
    bam = os.path.join(CSRC, "bamio.cpp")      # BAM reader (zlib) lives in the same host-side library
    bld = os.path.join(CSRC, "h5build.cpp")    # append-only writer of polish prediction files (no libhdf5)
    cnd = os.path.join(CSRC, "candidates.cpp") # candidate selection + VCF record text of a prediction batch
    hdr = os.path.join(CSRC, "..", "..", "include", "pepper_amd_io.h")
    have_src = all(os.path.exists(f) for f in (src, bam, bld, cnd, hdr))
    if os.path.exists(IO_LIB) and (not have_src or (not force and os.path.getmtime(IO_LIB) >= max(
            os.path.getmtime(src), os.path.getmtime(bam), os.path.getmtime(bld), os.path.getmtime(cnd), os.path.getmtime(hdr)))):
        return IO_LIB
    inc, lib = os.path.join(HDF5_PREFIX, "include"), os.path.join(HDF5_PREFIX, "lib")
    if not os.path.exists(os.path.join(inc, "hdf5.h")):
        raise RuntimeError(f"hdf5.h not found under {inc}: set PEPPER_AMD_HDF5_PREFIX")
    # per-process temporary name: loader / writer worker processes that start without a built library may all get
    # here at once; each links its own file and the rename is atomic
    tmp = f"{IO_LIB}.{os.getpid()}.tmp"
    # libraries of the prefix by their full paths (not -L: the prefix also holds an older libstdc++ that must not be the one
    # the link resolves against)
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", tmp, src, bam, bld, cnd, f"-I{inc}",
           os.path.join(lib, "libhdf5.so"), "-lz", "-lpthread", f"-Wl,-rpath,{lib}", "-Wl,--no-undefined"]
    # libdeflate (htslib's own choice for BGZF blocks) is decided HERE, once: the macro and the library go together, and
    # bamio.cpp tests only the macro -- a header found by the compiler without a linkable library (or the other way round)
    # can no longer produce a library that fails at dlopen; --no-undefined makes any such mismatch a build error
    if _have_libdeflate(inc, lib):
        cmd += ["-DPA_HAVE_LIBDEFLATE=1", os.path.join(lib, "libdeflate.so")]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, IO_LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return IO_LIB


TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
SYNTH_BAM = os.path.join(TOOLS, "synth_bam")


def build_tools(force=False):
    """tools/synth_bam: the generator of the synthetic BAM + FASTA that bench.py's image-generation leg reads (bench data, not
    product code).  Returns its path, or None where the tools directory did not travel."""
    src = os.path.join(TOOLS, "synth_bam.cpp")
    if not os.path.exists(src):
        return SYNTH_BAM if os.path.exists(SYNTH_BAM) else None
    if not force and os.path.exists(SYNTH_BAM) and os.path.getmtime(SYNTH_BAM) >= os.path.getmtime(src):
        return SYNTH_BAM
    inc, lib = os.path.join(HDF5_PREFIX, "include"), os.path.join(HDF5_PREFIX, "lib")
    tmp = f"{SYNTH_BAM}.{os.getpid()}.tmp"
    cmd = ["g++", "-O2", "-std=c++17", "-o", tmp, src, "-lz", "-lpthread"]
    if _have_libdeflate(inc, lib):
        cmd += [f"-I{inc}", "-DPA_HAVE_LIBDEFLATE=1", os.path.join(lib, "libdeflate.so"), f"-Wl,-rpath,{lib}"]
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, SYNTH_BAM)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return SYNTH_BAM


OBJ_DIR = os.path.join(CSRC, "_obj")


def _flags():
    # -mf16c on the host side: weight packing converts ~24 M values to f16 hi/lo halves at model creation; without the
    # F16C conversions clang calls a soft-float routine per value (0.4 s per variant model instead of tens of ms)
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Xarch_host", "-mf16c", "-Wno-unused-result"] + \
        os.environ.get("PEPPER_AMD_EXTRA_HIPCC_FLAGS", "").split()


def _same_flags():
    """Were the objects behind the library compiled with the flags of this call?  (A debug build -- e.g. -DPA_ENC_STAMP --
    must not survive the next ordinary build() just because no source changed.)  A library that arrived without its
    objects' stamp (a snapshot on a GPU box) is taken as it is."""
    stamp = os.path.join(OBJ_DIR, "flags.txt")
    return not os.path.exists(stamp) or open(stamp).read() == " ".join(_flags())


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return not _same_flags() or any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    """One object per translation unit (compiled in parallel, re-compiled only when it or a header is newer), then one
    link: an edit of one kernel file costs its own 5-25 s instead of the 45 s of the whole library."""
    if not force and not needs_build():
        return LIB
    flags = _flags()
    os.makedirs(OBJ_DIR, exist_ok=True)
    stamp = os.path.join(OBJ_DIR, "flags.txt")
    same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(flags)
    newest_header = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    todo = []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, src + ".o")
        if force or not same_flags or not os.path.exists(obj) or os.path.getmtime(obj) < max(
                newest_header, os.path.getmtime(os.path.join(CSRC, src))):
            todo.append((src, obj))

    def compile_one(item):
        src, obj = item
        cmd = [_hipcc()] + flags + ["-c", src, "-o", obj + f".{os.getpid()}.tmp"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, cwd=CSRC, check=True)
        os.replace(obj + f".{os.getpid()}.tmp", obj)
    if todo:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4)) as pool:
            list(pool.map(compile_one, todo))
    with open(stamp, "w") as fh:
        fh.write(" ".join(flags))
    tmp = f"{LIB}.{os.getpid()}.tmp"
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + [os.path.join(OBJ_DIR, s + ".o") for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    try:
        subprocess.run(cmd, cwd=CSRC, check=True)
        os.replace(tmp, LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_io(force=True, verbose=True))
    print(build_tools(force=True))

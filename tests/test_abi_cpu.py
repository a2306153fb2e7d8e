"""CPU-side checks of the drop-in boundary: the library builds, loads and exports every symbol
include/pepper_amd.h declares; the product package never touches the oracle."""
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from pepper_amd import build, _lib
    build.build()
    return _lib.load()


def _declared(*headers):
    out = set()
    for h in headers:
        text = open(os.path.join(REPO, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out |= set(re.findall(r"\b(pa_[a-z0-9_]+)\s*\(", text))
    return out


def test_header_symbols_are_exported(lib):
    declared = _declared("pepper_amd.h", "pepper_amd_encoder.h", "pepper_amd_realign.h", "pepper_amd_io_device.h")
    assert len(declared) >= 22
    from pepper_amd import _lib
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(lib, name), name


def test_no_undeclared_exports(lib):
    """Every pa_* symbol the product library exports is declared in a header under include/ (the diagnostic hooks in
    pepper_amd_debug.h, everything else in the drop-in headers)."""
    import subprocess
    from pepper_amd import _lib
    text = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in text.splitlines() if " T " in line and line.split()[-1].startswith("pa_")}
    declared = _declared("pepper_amd.h", "pepper_amd_encoder.h", "pepper_amd_realign.h", "pepper_amd_io_device.h", "pepper_amd_debug.h")
    assert exported <= declared, exported - declared
    assert {"pa_debug_dump_timing", "pa_debug_dump_gru_timing", "pa_debug_gemm_h2_experiment", "pa_debug_gemm_h2"} <= exported


def test_io_header_symbols_are_exported():
    from pepper_amd import h5
    io = h5.load()
    from pepper_amd.variant import bam
    declared = _declared("pepper_amd_io.h")
    bound = {name for name, _, _ in h5.SYMBOLS} | {name for name, _, _ in bam.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(io, name), name


def test_version_and_error_string(lib):
    assert lib.pa_version().decode().startswith("pepper_amd")
    assert lib.pa_last_error() is not None


def test_no_gpu_means_loud_failure(lib):
    """Without a device the product path must raise, never fall back to a CPU implementation."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import ctypes
    from pepper_amd import _lib, synthetic
    cfg = _lib.VariantConfig(26, 33, 1, 3, 0, 0)
    names, data, numel, n, keep = _lib.marshal_state_dict(synthetic.variant_state_dict(seed=1))
    handle = ctypes.c_void_p()
    rc = lib.pa_variant_create(ctypes.byref(cfg), names, data, numel, n, None, ctypes.byref(handle))
    assert rc != 0
    assert b"no HIP device" in lib.pa_last_error() or b"hip" in lib.pa_last_error().lower()
    # the summary encoders and the read re-aligner likewise
    for create in (lib.pa_encoder_create, lib.pa_realigner_create):
        handle = ctypes.c_void_p()
        assert create(0, None, ctypes.byref(handle)) != 0 and not handle.value
        assert b"no cpu fallback" in lib.pa_last_error().lower()
    from pepper_amd.polish.PEPPER import ReadAligner
    with pytest.raises(_lib.PepperAmdError, match="no CPU fallback"):
        ReadAligner(0, 8, "ACGTACGT").align_arrays([0], [0, 4], [65, 67, 71, 84])


def test_diagnostic_accessors_refuse_a_null_handle(lib):
    """pa_variant_overflow_rows / pa_variant_split_fallbacks are plain reads of a handle's counters: a NULL handle or a NULL
    destination is PA_ERR_INVALID with a message, on any machine."""
    import ctypes
    n = ctypes.c_int64(7)
    for fn in (lib.pa_variant_overflow_rows, lib.pa_variant_split_fallbacks):
        assert fn(None, ctypes.byref(n)) != 0 and n.value == 7
        assert lib.pa_last_error()


def test_product_never_imports_oracle():
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "pepper_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M) or "oracle/" in text.replace("oracle/ ", ""):
                    if "never route through oracle/" in text and not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M):
                        continue
                    bad.append(os.path.join(root, f))
    assert not bad, bad

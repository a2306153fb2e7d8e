"""Debug aid: cycle shares of the phases of tile_count_kernel's record loop.  Needs the stamped build:
    PEPPER_AMD_EXTRA_HIPCC_FLAGS=-DPA_ENC_STAMP python -m pepper_amd.build && python tools/enc_phase_cycles.py
(rebuild without the flag afterwards: the stamped kernel is several times slower)."""
import sys, ctypes, json, subprocess
sys.path.insert(0, "/root/repo")
import tools.bench_encoder as be
from pepper_amd import _lib
lib = _lib.load()
sys.argv = ["x", "--regions", "64", "--reps", "5", "--check", "0"]
out = (ctypes.c_ulonglong * 8)()
lib.pa_encoder_debug_cycles.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.pa_encoder_debug_cycles(out, 1)
be.main()
lib.pa_encoder_debug_cycles(out, 0)
tot = sum(out)
names = ["record set-up", "scans + scratch + owner search + byte loads issued", "per-operation section (first wait for memory)", "row phase", "(loop exit)", "wait for the tile's slowest wave", "per-position pass + store + votes"]
for n, v in zip(names, list(out)[:7]):
    print("%-48s %6.1f %%" % (n, 100.0 * v / tot))

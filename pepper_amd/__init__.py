"""pepper_amd: the MI355X-native hot path of PEPPER.

One process-wide setting is made here, before anything can have started the HIP runtime: the runtime multiplexes a process's
streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default), and kernels of different streams that share a queue run one
after the other.  Image generation drives one stream per worker thread (sixteen on a box): with four queues an encoder
launch of one worker waits behind another worker's 10 ms inflate launch -- 105 Mb of reference/s; with a queue per stream
150.  Sixteen, not more: the device runs about two dozen queues of a process at once, and past that the
driver time-slices them -- with 32, the queues a second inference pass added made every later image generation 1.5x slower
(3.3 -> 4.8 s on a 32 Mb polish job, for the rest of the process; 8, 16: 3.3 s throughout; docs/LEDGER_r05.md).  An explicit GPU_MAX_HW_QUEUES in the environment wins, and PEPPER_AMD_KEEP_HW_QUEUES=1 leaves the variable
alone altogether: an embedding process whose other HIP users (torch included) should keep the runtime's default sets that
(the setting is process-wide, and it has no effect once HIP has been initialised; INTEGRATION.md, "Environment").
"""
import os as _os

if _os.environ.get("PEPPER_AMD_KEEP_HW_QUEUES", "0") != "1":
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

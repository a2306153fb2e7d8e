import sys, zlib, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_inflate import member
from pepper_amd.bgzf import DeviceInflater, block_table
inf = DeviceInflater()
for name, data, level in (("abc*20000 L1", b"abc" * 20000, 1), ("abc*100 L1", b"abc" * 100, 1), ("abc*30 L1", b"abc" * 30, 1), ("abc*20000 L6", b"abc" * 20000, 6),
                          ("zeros L1", bytes(65280), 1), ("ab*300", b"ab" * 300, 1)):
    m = member(data, level)
    got = inf.inflate(m, block_table(m)).tobytes()
    bad = [i for i in range(len(data)) if got[i] != data[i]]
    print(name, "comp bytes", len(m) - 26, "mismatches", len(bad), bad[:20], got[:24], flush=True)

"""PackedEncoder.pack_device's error handling without a GPU: a stand-in library whose record walk reports a stale index
(PA_ERR_INVALID) makes the call return None -- the caller then takes the host packer -- and any other error code is raised."""
import ctypes

import numpy as np
import pytest

from pepper_amd import _lib
from pepper_amd.variant.PEPPER_VARIANT import PackedEncoder


def test_error_codes_mirror_the_header():
    import os
    import re
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "pepper_amd.h")).read()
    for name, value in re.findall(r"#define (PA_(?:OK|ERR_\w+)) (\d+)", header):
        assert getattr(_lib, name) == int(value), name


class _FakeLib(object):
    def __init__(self, walk_rc):
        self.walk_rc = walk_rc
        self.calls = []

    def pa_encoder_host_span(self, enc, cap):
        self.buf = np.zeros(cap, np.uint8)
        return self.buf.ctypes.data

    def pa_encoder_inflate_bgzf(self, *a):
        self.calls.append("inflate")
        return 0

    def pa_encoder_last_timing(self, enc, ptr, n):
        return 0

    def pa_encoder_walk_records(self, *a):
        self.calls.append("walk")
        return self.walk_rc


class _FakeBam(object):
    def has_index(self):
        return True

    def region_span(self, contig, start, stop, lookahead):
        return 0, 0, 4096, True

    def read_span(self, begin, end, span, tables, flag):
        return 1, 100, 1000, True, True

    def span_entries(self, contig, first, out_off, n_blocks, entries):
        return 1

    def pack_headers(self, *a):
        raise AssertionError("the host walk must not run after a failed device walk")


def _encoder(walk_rc):
    enc = object.__new__(PackedEncoder)
    enc.lib = _FakeLib(walk_rc)
    enc.enc = ctypes.c_void_p()
    enc.device = 0
    enc.arena = np.zeros(1 << 16, np.uint8)
    enc.reads = enc.pair_read = None
    enc.span = enc.tables = enc.headers = enc.entries = None
    enc.inflate_ms, enc.inflated_bytes = 0.0, 0
    enc.close = lambda: None
    return enc


def test_stale_index_sends_the_batch_to_the_host_packer(monkeypatch):
    monkeypatch.setenv("PEPPER_AMD_DEVICE_WALK", "1")
    enc = _encoder(_lib.PA_ERR_INVALID)
    got = enc.pack_device(_FakeBam(), "chr20", np.array([0]), np.array([1000]), False, 1)
    assert got is None and enc.lib.calls == ["inflate", "walk"]


def test_any_other_walk_error_is_raised(monkeypatch):
    monkeypatch.setenv("PEPPER_AMD_DEVICE_WALK", "1")
    enc = _encoder(_lib.PA_ERR_HIP)
    with pytest.raises(_lib.PepperAmdError) as info:
        enc.pack_device(_FakeBam(), "chr20", np.array([0]), np.array([1000]), False, 1)
    assert info.value.code == _lib.PA_ERR_HIP

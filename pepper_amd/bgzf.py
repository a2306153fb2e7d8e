"""BGZF members inflated on the device (include/pepper_amd_io_device.h).

A BGZF file (SAM/BAM specification 4.1) is a sequence of gzip members of at most 64 KiB of data, each with a 'BC' extra
subfield holding the member's size; `block_table` lists the members of a buffer, `DeviceInflater.inflate` hands them to
csrc/inflate.hip -- one wavefront per member.  No CPU fallback: without the HIP library or a device the calls raise.
"""
import ctypes

import numpy as np

from pepper_amd import _lib


class BgzfError(ValueError):
    pass


def block_table(buf, base=0):
    """-> (comp_off int64[n], comp_len int32[n], out_off int64[n], out_len int32[n]) of the BGZF members in `buf` (bytes or
    uint8 array): comp_* = the raw DEFLATE bytes of each member, out_* = where its ISIZE bytes go when the members are laid
    out back to back from `base`."""
    data = memoryview(buf).cast("B")
    n = len(data)
    comp_off, comp_len, out_off, out_len = [], [], [], []
    p, at = 0, int(base)
    while p < n:
        if n - p < 18 or data[p] != 0x1f or data[p + 1] != 0x8b or data[p + 2] != 8 or not data[p + 3] & 4:
            raise BgzfError("no BGZF member at byte %d" % p)
        xlen = data[p + 10] | data[p + 11] << 8
        q, bsize = p + 12, -1
        while q + 4 <= p + 12 + xlen:
            slen = data[q + 2] | data[q + 3] << 8
            if data[q] == 66 and data[q + 1] == 67 and slen == 2:
                bsize = (data[q + 4] | data[q + 5] << 8) + 1
            q += 4 + slen
        clen = bsize - 12 - xlen - 8
        if bsize < 0 or clen < 0 or p + bsize > n:
            raise BgzfError("truncated or malformed BGZF member at byte %d" % p)
        isize = int.from_bytes(bytes(data[p + bsize - 4:p + bsize]), "little")
        comp_off.append(p + 12 + xlen)
        comp_len.append(clen)
        out_off.append(at)
        out_len.append(isize)
        at += isize
        p += bsize
    return (np.asarray(comp_off, np.int64), np.asarray(comp_len, np.int32), np.asarray(out_off, np.int64),
            np.asarray(out_len, np.int32))


class DeviceInflater(object):
    def __init__(self, device=0):
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        _lib.check(self._lib.pa_inflater_create(int(device), ctypes.byref(self._h)))

    def close(self):
        if self._h:
            self._lib.pa_inflater_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def inflate(self, comp, table, out_bytes=None, repeats=1):
        """comp: bytes / uint8 array; table: block_table(...)'s four arrays -> uint8 array of the inflated members."""
        comp = np.frombuffer(comp, np.uint8) if not isinstance(comp, np.ndarray) else np.ascontiguousarray(comp, np.uint8)
        comp_off, comp_len, out_off, out_len = [np.ascontiguousarray(a, t) for a, t in
                                                zip(table, (np.int64, np.int32, np.int64, np.int32))]
        n = len(comp_off)
        if out_bytes is None:
            out_bytes = int((out_off + out_len).max()) if n else 0
        out = np.empty(max(out_bytes, 1), np.uint8)
        _lib.check(self._lib.pa_inflater_inflate(self._h, comp.ctypes.data, comp.size, n, comp_off.ctypes.data, comp_len.ctypes.data,
                                                  out_off.ctypes.data, out_len.ctypes.data, out.ctypes.data, out_bytes, int(repeats)))
        return out[:out_bytes]

    @property
    def last_kernel_ms(self):
        ms = ctypes.c_double()
        _lib.check(self._lib.pa_inflater_last_kernel_ms(self._h, ctypes.byref(ms)))
        return ms.value


def inflate_host(comp, table, threads=1):
    """The same members through the I/O library on the host (pa_bgzf_inflate_host: libdeflate / zlib on `threads` threads) --
    the CPU baseline beside DeviceInflater.  -> uint8 array."""
    from pepper_amd.variant import bam
    fn = bam._lib().pa_bgzf_inflate_host
    comp = np.frombuffer(comp, np.uint8) if not isinstance(comp, np.ndarray) else np.ascontiguousarray(comp, np.uint8)
    comp_off, comp_len, out_off, out_len = [np.ascontiguousarray(a, t) for a, t in zip(table, (np.int64, np.int32, np.int64, np.int32))]
    n = len(comp_off)
    out_bytes = int((out_off + out_len).max()) if n else 0
    out = np.empty(max(out_bytes, 1), np.uint8)
    rc = fn(comp.ctypes.data, comp.size, n, comp_off.ctypes.data, comp_len.ctypes.data, out_off.ctypes.data, out_len.ctypes.data,
            out.ctypes.data, out_bytes, int(threads))
    if rc != 0:
        raise BgzfError("host inflate failed (%d)" % rc)
    return out[:out_bytes]

"""BGZF writer/reader and a tabix (.tbi) index builder for position-sorted VCF text.

The reference writes its VCFs through pysam (`VariantFile(..., 'w')` on a `.vcf.gz` name, then
`pysam.tabix_index(preset="vcf")`: /root/reference/pepper_variant/modules/python/VcfWriter.py:27-46).
pysam / htslib are not part of this image, so the two container formats are produced here from
their published specifications (SAM spec section 4.1 "The BGZF compression format", and the tabix
index layout of the same document family): gzip members of at most 64 KiB with a `BC` extra field
carrying the compressed block size, terminated by the empty EOF block; `.tbi` = BGZF-compressed
binning (UCSC scheme, 14-bit leaves, 5 levels) + 16 kb linear index over virtual file offsets.
"""
import struct
import zlib

_BLOCK_DATA = 0xff00            # uncompressed payload per block (htslib's BGZF_BLOCK_SIZE)
_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _block(data, level=6):
    comp = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = comp.compress(data) + comp.flush()
    bsize = len(body) + 25      # header 18 + trailer 8 - 1
    head = struct.pack("<4BI2BH2BHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 0x42, 0x43, 2, bsize)
    return head + body + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))


_pool = None


def _compress_pool():
    """Threads for the deferred writers (zlib releases the GIL while it deflates a block)."""
    global _pool
    if _pool is None:
        from concurrent.futures import ThreadPoolExecutor
        from pepper_amd.hostinfo import usable_cpus
        # the deflate of the five files' blocks is what is left when the records are written in bulk: every CPU this
        # process may use (not the host's count: a container's quota is often a fraction of it)
        _pool = ThreadPoolExecutor(max_workers=max(2, min(32, usable_cpus())), thread_name_prefix="bgzf")
    return _pool


class BgzfWriter(object):
    """Buffered BGZF writer; `tell()` returns the virtual offset of the next byte written.

    deferred=True: blocks are deflated on a thread pool and written, in order, at close(); `tell()` then returns
    (block number << 16 | offset in the block) and `resolve()` turns such a value into the real virtual offset once the file
    is closed (the tabix builder keeps the provisional values and resolves them when it writes the index)."""

    def __init__(self, path, deferred=False):
        self._fh = open(path, "wb")
        self._buf = bytearray()
        self._coffset = 0
        self._deferred = deferred
        self._pending = []
        self._block_offsets = [0]

    def tell(self):
        if self._deferred:
            return (len(self._pending) << 16) | len(self._buf)
        return (self._coffset << 16) | len(self._buf)

    def position(self):
        """Uncompressed bytes written so far (deferred writers: every block but the one being filled holds _BLOCK_DATA bytes,
        so byte u of the stream has the provisional offset (u // _BLOCK_DATA) << 16 | u % _BLOCK_DATA)."""
        if not self._deferred:
            raise ValueError("position() needs a deferred writer")
        return len(self._pending) * _BLOCK_DATA + len(self._buf)

    def resolve(self, provisional):
        if not self._deferred:
            return provisional
        return (self._block_offsets[provisional >> 16] << 16) | (provisional & 0xffff)

    def write(self, data):
        if isinstance(data, str):
            data = data.encode()
        if len(self._buf) + len(data) < _BLOCK_DATA:     # the common case: one record, no block boundary
            self._buf += data
            return
        view = memoryview(data)
        while len(view):
            room = _BLOCK_DATA - len(self._buf)
            self._buf += view[:room]
            view = view[room:]
            if len(self._buf) >= _BLOCK_DATA:
                self._flush_block()

    def _flush_block(self):
        if self._buf:
            if self._deferred:
                self._pending.append(_compress_pool().submit(_block, bytes(self._buf)))
            else:
                blk = _block(bytes(self._buf))
                self._fh.write(blk)
                self._coffset += len(blk)
            self._buf = bytearray()

    def close(self):
        if self._fh is None:
            return
        self._flush_block()
        for fut in self._pending:
            blk = fut.result()
            self._fh.write(blk)
            self._coffset += len(blk)
            self._block_offsets.append(self._coffset)
        self._pending = []
        self._fh.write(_EOF)
        self._fh.close()
        self._fh = None


def read_bgzf(path):
    """Whole decompressed content (BGZF is a valid multi-member gzip stream)."""
    out = bytearray()
    with open(path, "rb") as fh:
        raw = fh.read()
    pos = 0
    while pos < len(raw):
        d = zlib.decompressobj(31)
        out += d.decompress(raw[pos:])
        pos = len(raw) - len(d.unused_data)
    return bytes(out)


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


class TabixBuilder(object):
    """Collects (contig, begin0, end0, voffset_begin, voffset_end) of each data line, in file order."""

    def __init__(self):
        self.names = []
        self._bins = {}     # tid -> {bin: [[beg, end], ...]}
        self._lin = {}      # tid -> {window: min voffset}
        self._tid = {}

    def add_many(self, names, contig_code, beg, end, vbeg, vend):
        """add() for records that follow each other in the file (vend[i] == vbeg[i + 1]), as arrays: names[contig_code[i]] is
        record i's contig, the rest integer arrays.  Same bins, chunks and linear index as the calls one by one: a chunk of a bin is a maximal run of
        file-consecutive records in that bin; a 16 kb window points at the first record that overlaps it."""
        import numpy as np
        contig_code = np.asarray(contig_code, dtype=np.int64)
        n = len(contig_code)
        if n == 0:
            return
        beg = np.asarray(beg, dtype=np.int64)
        end = np.maximum(np.asarray(end, dtype=np.int64), beg + 1)
        vbeg = np.asarray(vbeg, dtype=np.int64)
        vend = np.asarray(vend, dtype=np.int64)
        last = end - 1
        bins = np.zeros(n, np.int64)
        done = np.zeros(n, bool)
        for shift, base in ((14, ((1 << 15) - 1) // 7), (17, ((1 << 12) - 1) // 7), (20, ((1 << 9) - 1) // 7),
                            (23, ((1 << 6) - 1) // 7), (26, ((1 << 3) - 1) // 7)):
            hit = ~done & ((beg >> shift) == (last >> shift))
            bins[hit] = base + (beg[hit] >> shift)
            done |= hit
        # runs of one contig, in file order
        cuts = np.concatenate([[0], np.flatnonzero(contig_code[1:] != contig_code[:-1]) + 1, [n]]).tolist()
        for start, stop in zip(cuts[:-1], cuts[1:]):
            name = names[int(contig_code[start])]
            tid = self._tid.get(name)
            if tid is None:
                tid = self._tid[name] = len(self.names)
                self.names.append(name)
                self._bins[tid], self._lin[tid] = {}, {}
            b = bins[start:stop]
            edge = np.flatnonzero(np.concatenate([[True], b[1:] != b[:-1]])) + start
            ends = np.concatenate([edge[1:], [stop]]) - 1
            for s0, s1 in zip(edge.tolist(), ends.tolist()):
                chunks = self._bins[tid].setdefault(int(bins[s0]), [])
                if chunks and chunks[-1][1] == int(vbeg[s0]):
                    chunks[-1][1] = int(vend[s1])
                else:
                    chunks.append([int(vbeg[s0]), int(vend[s1])])
            lin = self._lin[tid]
            w0 = beg[start:stop] >> 14
            w1 = last[start:stop] >> 14
            idx = np.arange(start, stop)
            wide = np.flatnonzero(w1 > w0)
            ws, ids = [w0], [idx]
            for k in wide.tolist():
                span = np.arange(int(w0[k]) + 1, int(w1[k]) + 1)
                ws.append(span)
                ids.append(np.full(len(span), start + k))
            ws, ids = np.concatenate(ws), np.concatenate(ids)
            order = np.lexsort((ids, ws))
            ws, ids = ws[order], ids[order]
            first = np.concatenate([[True], ws[1:] != ws[:-1]])
            for w, k in zip(ws[first].tolist(), ids[first].tolist()):
                if w not in lin:
                    lin[w] = int(vbeg[k])

    def add(self, contig, beg, end, vbeg, vend):
        tid = self._tid.get(contig)
        if tid is None:
            tid = self._tid[contig] = len(self.names)
            self.names.append(contig)
            self._bins[tid], self._lin[tid] = {}, {}
        end = max(end, beg + 1)
        chunks = self._bins[tid].setdefault(reg2bin(beg, end), [])
        if chunks and chunks[-1][1] == vbeg:
            chunks[-1][1] = vend
        else:
            chunks.append([vbeg, vend])
        lin = self._lin[tid]
        for w in range(beg >> 14, ((end - 1) >> 14) + 1):
            if w not in lin:
                lin[w] = vbeg

    def write(self, path, resolve=None):
        """resolve: maps the stored offsets to virtual file offsets (BgzfWriter.resolve of a deferred writer)."""
        if resolve is None:
            resolve = int
        names = b"".join(n.encode() + b"\0" for n in self.names)
        out = bytearray(b"TBI\1")
        # n_ref, format (2 = VCF), col_seq, col_beg, col_end, meta char, skip, l_nm
        out += struct.pack("<8i", len(self.names), 2, 1, 2, 0, ord("#"), 0, len(names))
        out += names
        for tid in range(len(self.names)):
            bins = self._bins[tid]
            out += struct.pack("<i", len(bins))
            for b in sorted(bins):
                out += struct.pack("<Ii", b, len(bins[b]))
                for beg, end in bins[b]:
                    out += struct.pack("<QQ", resolve(beg), resolve(end))
            lin = self._lin[tid]
            n_intv = (max(lin) + 1) if lin else 0
            out += struct.pack("<i", n_intv)
            last = 0
            for w in range(n_intv):
                last = lin.get(w, last)        # empty windows inherit the previous offset, as htslib fills them
                out += struct.pack("<Q", resolve(last) if last else 0)
        w = BgzfWriter(path)
        w.write(bytes(out))
        w.close()


def parse_tbi(path):
    """Decode a .tbi (used by the tests as a structural check of TabixBuilder)."""
    raw = read_bgzf(path)
    if raw[:4] != b"TBI\1":
        raise ValueError("not a tabix index")
    n_ref, fmt, col_seq, col_beg, col_end, meta, skip, l_nm = struct.unpack_from("<8i", raw, 4)
    pos = 36
    names = [n.decode() for n in raw[pos:pos + l_nm].split(b"\0")[:-1]]
    pos += l_nm
    refs = []
    for _ in range(n_ref):
        (n_bin,) = struct.unpack_from("<i", raw, pos)
        pos += 4
        bins = {}
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", raw, pos)
            pos += 8
            bins[b] = [struct.unpack_from("<QQ", raw, pos + 16 * i) for i in range(n_chunk)]
            pos += 16 * n_chunk
        (n_intv,) = struct.unpack_from("<i", raw, pos)
        pos += 4
        ioff = list(struct.unpack_from("<%dQ" % n_intv, raw, pos))
        pos += 8 * n_intv
        refs.append({"bins": bins, "ioff": ioff})
    return {"names": names, "format": fmt, "cols": (col_seq, col_beg, col_end), "meta": chr(meta), "skip": skip,
            "refs": refs}

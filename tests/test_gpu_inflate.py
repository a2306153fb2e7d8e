"""BGZF members inflated on the device (include/pepper_amd_io_device.h, csrc/inflate.hip) against zlib on the same bytes.

The checker here is zlib itself (the library htslib inflates with): every member is built with zlib's raw DEFLATE at a given
level / strategy, so the three block types of RFC 1951, several blocks per member, empty stored blocks (sync flushes),
overlapping matches, incompressible data, the 28-byte end-of-file member and members of the maximum size are all in the
inputs.  Bit-exact or an error: there is no tolerance.
"""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from pepper_amd import _lib
from pepper_amd.bgzf import BgzfError, DeviceInflater, block_table

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EOF_MEMBER = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def member(payload, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_at=(), extra_subfield=False):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    body, last = b"", 0
    for cut in flush_at:
        body += c.compress(payload[last:cut]) + c.flush(zlib.Z_SYNC_FLUSH)
        last = cut
    body += c.compress(payload[last:]) + c.flush()
    extra = b"BC\x02\x00\x00\x00"
    if extra_subfield:                       # another subfield in front of BC: the table must scan for it
        extra = b"XY\x03\x00abc" + extra
    bsize = 12 + len(extra) + len(body) + 8
    assert bsize <= 65536
    extra = extra[:-2] + struct.pack("<H", bsize - 1)
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", len(extra)) + extra + body +
            struct.pack("<II", zlib.crc32(payload), len(payload)))


def payloads(rng):
    text = (b"@read/%d\tchr20\t" * 50) + bytes(rng.integers(33, 74, 3000, dtype=np.uint8))
    quals = bytes(np.clip(rng.normal(20, 6, 60000), 1, 50).astype(np.uint8))
    nibbles = bytes(rng.choice(np.array([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88],
                                        np.uint8), 30000))
    yield "one byte", b"A"
    yield "zeros", bytes(65280)
    yield "period 3", b"abc" * 20000
    yield "period 300", bytes(rng.integers(0, 256, 300, dtype=np.uint8)) * 200
    yield "random", bytes(rng.integers(0, 256, 65280, dtype=np.uint8))
    yield "text", text
    yield "qualities", quals
    yield "4-bit bases", nibbles
    yield "record-like", (struct.pack("<iiIIiiii", 5000, 0, 0x12483c0a, 40 << 16 | 700, 6000, -1, -1, 0) + b"read_000123\0" +
                          bytes(rng.integers(0, 2 ** 31, 700, dtype=np.uint32).view(np.uint8)) + nibbles[:3000] + quals[:6000]) * 4


def inflate_and_compare(members, expect):
    buf = b"".join(members)
    table = block_table(buf)
    assert len(table[0]) == len(members)
    with DeviceInflater() as inf:
        got = inf.inflate(buf, table)
    want = b"".join(expect)
    assert got.size == len(want)
    assert got.tobytes() == want


@pytest.mark.gpu
def test_every_block_type_and_data_kind():
    rng = np.random.default_rng(7)
    members, expect = [], []
    for name, data in payloads(rng):
        data = data[:65280]
        for level, strategy in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY),
                                (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
            if level == 0 and len(data) > 65000:
                data_l = data[:65000]                     # (a stored member carries 5 bytes per block on top of the data)
            else:
                data_l = data
            if strategy == zlib.Z_FIXED and name == "random":
                data_l = data[:50000]                     # (fixed codes expand random bytes beyond a member)
            members.append(member(data_l, level, strategy))
            expect.append(data_l)
    assert zlib.decompress(members[3][18:-8], -15) == expect[3]          # the builder itself
    inflate_and_compare(members, expect)


@pytest.mark.gpu
def test_several_blocks_per_member_and_empty_stored_blocks():
    rng = np.random.default_rng(8)
    data = bytes(np.clip(rng.normal(30, 8, 60000), 0, 93).astype(np.uint8))
    members = [member(data, 6, flush_at=(1, 2, 1000, 1000, 33333)), member(data, 1, flush_at=(59999,)),
               member(data[:40000], 0, flush_at=(7, 8, 9)), member(data, 6, zlib.Z_FIXED, flush_at=(30000,), extra_subfield=True),
               EOF_MEMBER, member(b"", 6), member(b"", 0), member(b"x" * 258 + b"y", 9)]
    expect = [data, data, data[:40000], data, b"", b"", b"", b"x" * 258 + b"y"]
    inflate_and_compare(members, expect)


@pytest.mark.gpu
def test_matches_that_read_what_was_just_written_under_load():
    """2 400 members in one launch (every wavefront slot of the chip busy), each a short-period pattern: every match reads
    bytes the same wavefront stored an instruction or a step earlier -- inside a step (resolved between lanes), across steps
    (through memory, the store still in flight) and through the byte-by-byte path of long matches."""
    rng = np.random.default_rng(10)
    members, expect = [], []
    for k in range(2400):
        period = int(rng.integers(1, 70))
        unit = bytes(rng.integers(0, 256, period, dtype=np.uint8))
        n = int(rng.integers(200, 20000))
        data = bytearray((unit * (n // period + 1))[:n])
        for at in rng.integers(0, n, int(rng.integers(0, 6))):          # a few literals that break the period
            data[int(at)] ^= 0x5a
        data = bytes(data)
        members.append(member(data, (1, 6, 9)[k % 3], (zlib.Z_DEFAULT_STRATEGY, zlib.Z_RLE, zlib.Z_FIXED)[(k // 3) % 3]))
        expect.append(data)
    buf = b"".join(members)
    table = block_table(buf)
    want = b"".join(expect)
    with DeviceInflater() as inf:
        for _ in range(3):
            assert inf.inflate(buf, table).tobytes() == want


def _long_code_payloads(rng):
    """Data whose Huffman codes run past the primary tables (10 bits literal/length, 8 bits distance): a geometric distribution
    over 220 byte values (the rare ones get 11- to 15-bit codes), and fragments copied from a skewed choice of distances (most
    near, a few far: the far distance symbols get 9- to 15-bit codes)."""
    p = 0.93 ** np.arange(220)
    skew = rng.choice(np.arange(220), 60000, p=p / p.sum()).astype(np.uint8)
    yield bytes(skew), zlib.Z_HUFFMAN_ONLY
    yield bytes(skew), zlib.Z_DEFAULT_STRATEGY
    data = bytearray(rng.choice(np.arange(220), 20000, p=p / p.sum()).astype(np.uint8).tobytes())
    while len(data) < 64000:
        far = rng.random() < 0.03
        dist = int(rng.integers(3000, min(len(data), 32000))) if far else int(rng.integers(4, 200))
        n = int(rng.integers(3, 40))
        at = len(data) - dist
        data += data[at:at + n]
        data += bytes(rng.choice(np.arange(220), int(rng.integers(0, 6)), p=p / p.sum()).astype(np.uint8))
    yield bytes(data[:65000]), zlib.Z_DEFAULT_STRATEGY


@pytest.mark.gpu
def test_codes_longer_than_the_primary_tables():
    """Symbols whose codes leave the primary tables are resolved inside the step (long_code in csrc/inflate.hip: all candidate
    lengths at once) -- literals, lengths, the end-of-block code and distances; 600 members in one launch, levels 1 / 6 / 9,
    bit-exact against zlib.  The kernel's own count of such symbols (PA_INFLATE_DEBUG) says the inputs do what they are for."""
    rng = np.random.default_rng(14)
    members, expect = [], []
    for rep in range(200):
        for data, strategy in _long_code_payloads(rng):
            cut = int(rng.integers(1000, len(data)))
            members.append(member(data[:cut], (1, 6, 9)[rep % 3], strategy, flush_at=(cut // 2,) if rep % 5 == 0 else ()))
            expect.append(data[:cut])
    inflate_and_compare(members, expect)
    # the count, from a child process with the diagnostic switched on (it prints to stderr)
    code = ("import sys, zlib, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_inflate as t; "
            "rng = np.random.default_rng(14); ms = [t.member(d, 6, s) for d, s in t._long_code_payloads(rng)]; "
            "buf = b''.join(ms); inf = t.DeviceInflater(); inf.inflate(buf, t.block_table(buf))" % (ROOT, os.path.join(ROOT, "tests")))
    p = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, PA_INFLATE_DEBUG="1"))
    assert p.returncode == 0, p.stderr
    line = [ln for ln in p.stderr.splitlines() if "long-code symbols" in ln][-1]
    assert int(line.split(" matches, ")[1].split()[0]) > 1000, line


@pytest.mark.gpu
def test_crc_of_members_of_every_size_class():
    """The epilogue's CRC deals a member's words over the lanes from its END, 256 bytes apart per lane: lengths around the
    multiples of 4 (head bytes), of 256 (a lane's first word) and tiny members, at output offsets of every alignment -- each
    passes with its true trailer and fails with a flipped one."""
    rng = np.random.default_rng(15)
    sizes = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 63, 64, 65, 252, 253, 254, 255, 256, 257, 258, 259, 260, 261, 511, 512, 513, 515,
             1020, 1021, 1022, 1023, 1024, 1025, 1026, 1027, 16383, 16385, 65277, 65278, 65279, 65280]
    datas = [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in sizes]
    with DeviceInflater() as inf:
        for level in (0, 6):
            members = [member(d, level) for d in datas]
            buf = b"".join(members)
            assert inf.inflate(buf, block_table(buf)).tobytes() == b"".join(datas)
            for k in range(len(members)):
                bad = bytearray(members[k])
                bad[-5] ^= 0x40                                 # the trailer's CRC-32, highest byte
                buf = b"".join(members[:k] + [bytes(bad)] + members[k + 1:])
                with pytest.raises(_lib.PepperAmdError, match="BGZF block %d: CRC32" % k):
                    inf.inflate(buf, block_table(buf))


@pytest.mark.gpu
def test_many_members_of_a_bam_file(tmp_path):
    """A whole synthetic BAM (tools/synth_bam: libdeflate's encoder, not zlib's) member by member."""
    tool = os.path.join(ROOT, "tools", "synth_bam")
    if not os.path.exists(tool):
        pytest.skip("tools/synth_bam is not built")
    subprocess.run([tool, str(tmp_path), "300000", "30", "3", "2"], check=True, capture_output=True)
    bam = str(tmp_path / "reads.bam")
    raw = open(bam, "rb").read()
    table = block_table(raw)
    assert len(table[0]) > 50
    want = b"".join(zlib.decompress(raw[o:o + n], -15) for o, n in zip(table[0].tolist(), table[1].tolist()))
    with DeviceInflater() as inf:
        got = inf.inflate(raw, table, repeats=3)
        assert inf.last_kernel_ms > 0
    assert got.tobytes() == want
    assert got[:4].tobytes() == b"BAM\1"


@pytest.mark.gpu
def test_malformed_members_fail_the_call():
    rng = np.random.default_rng(9)
    data = bytes(np.clip(rng.normal(30, 8, 20000), 0, 93).astype(np.uint8))
    good = member(data, 6)
    table = block_table(good)
    with DeviceInflater() as inf:
        assert inf.inflate(good, table).tobytes() == data
        short = [a.copy() for a in table]
        short[1][0] //= 2                                   # the stream stops half way
        with pytest.raises(_lib.PepperAmdError, match="BGZF block 0"):
            inf.inflate(good, short)
        wrong = [a.copy() for a in table]
        wrong[3][0] -= 1                                    # ISIZE one short of what the stream holds
        with pytest.raises(_lib.PepperAmdError, match="ISIZE"):
            inf.inflate(good, wrong)
        bad = bytearray(good)
        bad[18] = (bad[18] & 0xf9) | 0x06                   # block type 3
        with pytest.raises(_lib.PepperAmdError, match="reserved"):
            inf.inflate(bytes(bad), table)
        stored = member(data[:100], 0)
        bad = bytearray(stored)
        bad[21] ^= 0xff                                     # NLEN no longer the complement of LEN
        with pytest.raises(_lib.PepperAmdError, match="LEN"):
            inf.inflate(bytes(bad), block_table(stored))
        far = [a.copy() for a in table]
        far[0][0] = len(good)                               # outside the buffer: refused before anything is launched
        with pytest.raises(_lib.PepperAmdError, match="outside"):
            inf.inflate(good, far)
        assert inf.inflate(good, table).tobytes() == data   # the handle is fine afterwards


@pytest.mark.gpu
def test_a_flipped_payload_bit_fails_the_members_crc():
    """htslib's inflate_block compares crc32() of the inflated block with the member's trailer; so does the device (the inflate kernel's epilogue):
    a literal flipped inside a stored block, a flipped literal of a fixed-Huffman stream that still decodes, and a wrong
    trailer all inflate structurally and must fail the call -- in members of 1 byte, of 1 024 k +- 1 bytes (the slice edges of
    the fold) and of the maximum size; the untouched members pass."""
    rng = np.random.default_rng(12)
    sizes = [1, 3, 1023, 1024, 1025, 4097, 33000, 65279, 65280]
    datas = [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in sizes]
    with DeviceInflater() as inf:
        for level in (0, 1, 6):
            members = [member(d, level) for d in datas]
            buf = b"".join(members)
            assert inf.inflate(buf, block_table(buf)).tobytes() == b"".join(datas)
        # stored blocks: every payload byte is a literal of the stream; flip one bit in each member in turn
        members = [member(d, 0) for d in datas]
        for k, (m, d) in enumerate(zip(members, datas)):
            bad = bytearray(m)
            at = 18 + 5 + (len(d) * 2) // 3                  # header 18 bytes, stored-block header 5: inside the payload
            bad[at] ^= 0x10
            buf = b"".join(members[:k] + [bytes(bad)] + members[k + 1:])
            with pytest.raises(_lib.PepperAmdError, match="BGZF block %d: CRC32" % k):
                inf.inflate(buf, block_table(buf))
        # a wrong trailer on an otherwise sound member
        m = bytearray(member(datas[4], 6))
        m[-8] ^= 1
        with pytest.raises(_lib.PepperAmdError, match="CRC32"):
            inf.inflate(bytes(m), block_table(bytes(m)))
        # fixed-Huffman literals (text of 8-bit codes 0x30 + c for c < 144): flipping the lowest bit of a code gives another literal
        text = bytes(rng.integers(65, 91, 3000, dtype=np.uint8))
        fixed = bytearray(member(text, 6, zlib.Z_FIXED))
        hit = 0
        for at in range(40, 400):
            trial = bytearray(fixed)
            trial[at] ^= 0x80
            try:
                inf.inflate(bytes(trial), block_table(bytes(trial)))
            except _lib.PepperAmdError as err:
                hit += "CRC32" in str(err)
                continue
            raise AssertionError("a flipped stream bit went unnoticed at byte %d" % at)
        assert hit > 50                                       # (the others break the code structure and fail earlier)
        assert inf.inflate(bytes(fixed), block_table(bytes(fixed))).tobytes() == text


@pytest.mark.gpu
@pytest.mark.parametrize("wide,below", [("0", "48"), ("1", "1"), ("1", "64")])
def test_every_case_in_the_other_step_forms(wide, below):
    """PA_INFLATE_WIDE / PA_INFLATE_WIDE_BELOW pin the kernel's step form for a whole process (they are read once): every case of
    this module through the one-window step, through the two-window step that falls back to one window after every step that wrote
    more than a byte (the switch between the forms at every other step), and through the always-two-window step.  The default
    (two windows while a step writes at most 48 bytes) is what the cases above ran."""
    import sys
    if os.environ.get("PEPPER_AMD_INFLATE_FORMS_CHILD") == "1":
        pytest.skip("the child run itself")
    env = dict(os.environ, PA_INFLATE_WIDE=wide, PA_INFLATE_WIDE_BELOW=below, PEPPER_AMD_INFLATE_FORMS_CHILD="1")
    run = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                         env=env, capture_output=True, text=True, cwd=ROOT)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-2000:]


def test_block_table_reads_the_member_headers():
    a, b = member(b"hello" * 100, 6), member(b"", 6, extra_subfield=True)
    comp_off, comp_len, out_off, out_len = block_table(a + b + EOF_MEMBER, base=10)
    assert comp_off.tolist() == [18, len(a) + 18 + 7, len(a) + len(b) + 18]
    assert out_off.tolist() == [10, 510, 510] and out_len.tolist() == [500, 0, 0]
    assert zlib.decompress((a + b)[comp_off[0]:comp_off[0] + comp_len[0]], -15) == b"hello" * 100
    with pytest.raises(BgzfError):
        block_table(a[:-1])
    with pytest.raises(BgzfError):
        block_table(b"\0" * 40)

"""Test-side BAM tooling: a minimal coordinate-sorted BAM (+ optional BAI) writer built on the package's
BGZF writer, and a Python restatement of the reference's get_reads clipping rules
(/root/reference/pepper_variant/modules/cpp/bam_handler.cpp:115-451) used as the checker for
pepper_amd/csrc/bamio.cpp.  Not product code."""
import struct

import numpy as np

from pepper_amd.variant.bgzf import BgzfWriter, reg2bin

SEQ_CODE = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
REF_OPS = (0, 2, 3, 7, 8)


def ref_length(cigar):
    return sum(n for op, n in cigar if op in REF_OPS)


def encode_record(tid, rec):
    name = rec.get("name", "r").encode() + b"\0"
    cigar = rec["cigar"]
    seq = rec["seq"]
    l_seq = len(seq)
    end = rec["pos"] + max(1, ref_length(cigar))
    body = struct.pack("<iiBBHHHIiii", tid, rec["pos"], len(name), rec.get("mapq", 60), reg2bin(rec["pos"], end),
                       2 if rec.get("long_cigar") else len(cigar), rec.get("flag", 16 if rec.get("reverse") else 0), l_seq, -1, -1, 0)
    body += name
    if rec.get("long_cigar"):
        # what writers do for more than 65535 operations (SAM spec 4.2.2), forced here on short records: the core field
        # holds <l_seq>S<ref_len>N (n_cigar_op = 2) and the real operations travel in the CG:B,I tag
        body += struct.pack("<II", (l_seq << 4) | 4, (ref_length(cigar) << 4) | 3)
    else:
        body += b"".join(struct.pack("<I", (n << 4) | op) for op, n in cigar)
    packed = bytearray((l_seq + 1) // 2)
    for i, c in enumerate(seq):
        packed[i >> 1] |= SEQ_CODE[c] << (4 if i % 2 == 0 else 0)
    body += bytes(packed)
    body += bytes(np.asarray(rec["qual"], dtype=np.uint8))
    body += rec.get("aux", b"")
    if rec.get("long_cigar") and not rec.get("drop_cg"):
        body += b"CGBI" + struct.pack("<I", len(cigar)) + b"".join(struct.pack("<I", (n << 4) | op) for op, n in cigar)
    return struct.pack("<i", len(body)) + body


def write_bam(path, refs, records_by_tid, header_text="@HD\tVN:1.6\tSO:coordinate\n", with_index=True, flush_every=0):
    """refs: [(name, length)]; records_by_tid: {tid: [record dicts sorted by pos]}.  Returns nothing; writes
    `path` and, when with_index, `path + '.bai'`."""
    w = BgzfWriter(path)
    text = header_text + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    w.write(b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs)))
    for name, length in refs:
        w.write(struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", length))
    w._flush_block()                 # records start on a block boundary, as samtools writes them
    index = {}
    count = 0
    for tid in sorted(records_by_tid):
        bins, lin = {}, {}
        for rec in records_by_tid[tid]:
            vbeg = w.tell()
            w.write(encode_record(tid, rec))
            count += 1
            if flush_every and count % flush_every == 0:
                w._flush_block()
            vend = w.tell()
            beg, end = rec["pos"], rec["pos"] + max(1, ref_length(rec["cigar"]))
            chunks = bins.setdefault(reg2bin(beg, end), [])
            if chunks and chunks[-1][1] == vbeg:
                chunks[-1][1] = vend
            else:
                chunks.append([vbeg, vend])
            for win in range(beg >> 14, ((end - 1) >> 14) + 1):
                lin.setdefault(win, vbeg)
        index[tid] = (bins, lin)
    w.close()
    if with_index:
        out = bytearray(b"BAI\1" + struct.pack("<i", len(refs)))
        for tid in range(len(refs)):
            bins, lin = index.get(tid, ({}, {}))
            out += struct.pack("<i", len(bins))
            for b in sorted(bins):
                out += struct.pack("<Ii", b, len(bins[b]))
                for beg, end in bins[b]:
                    out += struct.pack("<QQ", beg, end)
            n_intv = max(lin) + 1 if lin else 0
            out += struct.pack("<i", n_intv)
            for win in range(n_intv):
                out += struct.pack("<Q", lin.get(win, 0))     # 0 = no alignment overlaps this window
        with open(path + ".bai", "wb") as fh:
            fh.write(bytes(out))


def restated_get_reads(records, start, stop, include_supplementary, min_mapq):
    """The reference's filters + clipping on record dicts of ONE contig, in file order."""
    out = []
    for rec in records:
        pos, cigar = rec["pos"], rec["cigar"]
        flag = rec.get("flag", 16 if rec.get("reverse") else 0)
        end = pos + max(1, ref_length(cigar))
        if not (pos < stop and end > start):
            continue
        if flag & (0x200 | 0x400 | 0x100 | 0x4):
            continue
        if not include_supplementary and flag & 0x800:
            continue
        if rec.get("mapq", 60) < min_mapq:
            continue
        seq, qual = rec["seq"], list(np.asarray(rec["qual"]).tolist())
        out_seq, out_qual, out_cigar = [], [], []
        pos_start = pos_end = -1
        rpos, ridx = pos, 0
        for op, n in cigar:
            if rpos > stop:
                break
            kept = 0
            if op in (0, 7, 8):
                skip = 0
                if rpos < start:
                    skip = min(start - rpos, n)
                    ridx += skip
                    rpos += skip
                for _ in range(skip, n):
                    if rpos > stop:
                        break
                    if pos_start == -1:
                        pos_start = pos_end = rpos
                    out_seq.append(seq[ridx].upper())
                    out_qual.append(qual[ridx])
                    kept += 1
                    pos_end += 1
                    ridx += 1
                    rpos += 1
            elif op in (4, 1):
                if start <= rpos <= stop and pos_start != -1:
                    out_seq.extend(c.upper() for c in seq[ridx:ridx + n])
                    out_qual.extend(qual[ridx:ridx + n])
                    kept = n
                ridx += n
            elif op in (3, 2):
                if start <= rpos <= stop and pos_start != -1:
                    for _ in range(n):
                        if rpos > stop:
                            break
                        kept += 1
                        pos_end += 1
                        rpos += 1
                else:
                    rpos += n
            if kept:
                out_cigar.append((op, kept))
        if out_seq:
            out.append(dict(name=rec.get("name", "r"), pos=pos_start, pos_end=pos_end, seq="".join(out_seq), qual=out_qual,
                            cigar=out_cigar, mapq=rec.get("mapq", 60), reverse=bool(flag & 0x10), flag=flag,
                            hp=rec.get("hp", 0)))
    return out


# ---- the packed form (pa_bam_pack_regions) and the closed form of the clipping walk the device runs on it -------------

def unpack_packed_read(arena, rd):
    """One pa_packed_read entry -> dict(pos, cigar, seq, qual, flag, mapq) (arena: uint8 array)."""
    off, n_cig, l_seq = int(rd["data_off"]), int(rd["n_cigar"]), int(rd["l_seq"])
    words = np.frombuffer(arena[off:off + 4 * n_cig].tobytes(), "<u4")
    cigar = [(int(w) & 15, int(w) >> 4) for w in words]
    packed = arena[off + 4 * n_cig:off + 4 * n_cig + (l_seq + 1) // 2]
    codes = np.empty(2 * len(packed), np.uint8)
    codes[0::2] = packed >> 4
    codes[1::2] = packed & 15
    seq = "".join("=ACMGRSVTWYHKDBN"[c] for c in codes[:l_seq])
    q0 = off + 4 * n_cig + (l_seq + 1) // 2
    flags = int(rd["flags"])
    return dict(pos=int(rd["pos"]), cigar=cigar, seq=seq, qual=arena[q0:q0 + l_seq].tolist(), flag=flags & 0xffff,
                mapq=(flags >> 16) & 0xff)


def closed_form_clip(rec, start, stop, chunk=64):
    """The clipping of bam_handler.cpp:176-303 WITHOUT its running state, as unpack_clip_kernel (pepper_amd/csrc/encoder.hip)
    computes it: every operation's reference / read position from prefix sums over `chunk` operations at a time, what is kept
    of it from those alone plus one bit -- has an earlier operation kept an aligned base ("anchored").
    -> None when no base of the read lies inside [start, stop], else dict(pos, cigar, first_idx, written)."""
    rpos, ridx = rec["pos"], 0
    anchored = False
    out, written, pos_start, first_idx = [], 0, -1, -1
    cigar = rec["cigar"]
    for cb in range(0, len(cigar), chunk):
        ops = cigar[cb:cb + chunk]
        rb, qb = [], []
        r, q = rpos, ridx
        for op, n in ops:
            rb.append(r)
            qb.append(q)
            if op in (0, 7, 8, 2, 3):
                r += n
            if op in (0, 7, 8, 1, 4):
                q += n
        kept_m = [max(0, min(b + n - 1, stop) - max(b, start) + 1) if op in (0, 7, 8) else 0 for (op, n), b in zip(ops, rb)]
        first = next((i for i, k in enumerate(kept_m) if k > 0), None)
        for i, ((op, n), b, qq) in enumerate(zip(ops, rb, qb)):
            anch = anchored or (first is not None and i > first)
            kept = 0
            if op in (0, 7, 8):
                kept = kept_m[i]
            elif op in (1, 4):
                kept = n if (start <= b <= stop and anch) else 0
            elif op in (2, 3):
                kept = min(n, stop - b + 1) if (start <= b <= stop and anch) else 0
            if kept > 0:
                out.append((op, kept))
                if op in (0, 7, 8, 1, 4):
                    written += kept
        if not anchored and first is not None:
            anchored = True
            pos_start = max(rb[first], start)
            first_idx = qb[first] + max(0, start - rb[first])
        rpos, ridx = r, q
        if rpos > stop:
            break
    if written == 0:
        return None
    return dict(pos=pos_start, cigar=out, first_idx=first_idx, written=written)

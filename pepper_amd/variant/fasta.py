"""Indexed FASTA access with the reference's `PEPPER_VARIANT.FASTA_handler` surface.

replaces: /root/reference/pepper_variant/modules/cpp/fasta_handler.cpp:7-55 (htslib faidx wrapper:
get_chromosome_names, get_reference_sequence(contig, start, stop) = faidx_fetch_seq(start, stop-1)
upper-cased, get_chromosome_sequence_length).  htslib is not part of this image, so the `.fai`
index is read (or, when missing, rebuilt in memory by one scan of the file) here.

Clamping follows faidx_fetch_seq: start < 0 -> 0, stop beyond the contig -> contig end, an empty
or inverted interval -> ''.  An unknown contig raises (the reference returns NULL into std::string,
which aborts).
"""
import os

_SCANS = {}      # (path, mtime, size) -> (names, index) of the last FASTA file scanned for want of a .fai


_UPPER = bytes(range(256)).upper()


class FASTA_handler(object):
    def __init__(self, path):
        if not os.path.isfile(path):
            raise FileNotFoundError("INVALID FASTA FILE. PLEASE CHECK IF PATH IS CORRECT AND FILE IS INDEXED: " + str(path))
        self.path = path
        self._index = {}        # name -> (length, offset, line_bases, line_width)
        self._names = []
        fai = path + ".fai"
        if os.path.isfile(fai):
            with open(fai) as fh:
                for line in fh:
                    f = line.rstrip("\n").split("\t")
                    if len(f) < 5:
                        continue
                    self._names.append(f[0])
                    self._index[f[0]] = (int(f[1]), int(f[2]), int(f[3]), int(f[4]))
        else:
            # one scan per file and process: candidate finding opens the reference once for the records and once per worker
            st = os.stat(path)
            key = (os.path.abspath(path), st.st_mtime_ns, st.st_size)
            if key not in _SCANS:
                self._scan()
                _SCANS.clear()
                _SCANS[key] = (list(self._names), dict(self._index))
            self._names, self._index = list(_SCANS[key][0]), dict(_SCANS[key][1])
        self._fh = open(path, "rb")

    def _scan(self):
        name, length, offset, line_bases, line_width = None, 0, 0, 0, 0
        pos = 0
        with open(self.path, "rb") as fh:
            for raw in fh:
                if raw.startswith(b">"):
                    if name is not None:
                        self._names.append(name)
                        self._index[name] = (length, offset, line_bases, line_width)
                    name = raw[1:].split()[0].decode()
                    length, offset, line_bases, line_width = 0, pos + len(raw), 0, 0
                else:
                    bases = len(raw.rstrip(b"\r\n"))
                    if line_bases == 0 and bases:
                        line_bases, line_width = bases, len(raw)
                    length += bases
                pos += len(raw)
        if name is not None:
            self._names.append(name)
            self._index[name] = (length, offset, line_bases, line_width)

    def close(self):
        if self._fh is not None:
            self._fh.close()
            self._fh = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_chromosome_names(self):
        return list(self._names)

    def get_chromosome_sequence_length(self, chromosome_name):
        if chromosome_name not in self._index:
            return -1
        return self._index[chromosome_name][0]

    def get_reference_bytes(self, region, start, stop):
        """get_reference_sequence as upper-case bytes (what the encoders take; image generation fetches a whole group of
        intervals with one call and slices it)."""
        if region not in self._index:
            raise KeyError("CHROMOSOME NAME NOT PRESENT IN REFERENCE FASTA FILE: %s %d %d" % (region, start, stop))
        length, offset, line_bases, line_width = self._index[region]
        start = max(0, int(start))
        stop = min(length, int(stop))
        if stop <= start or line_bases <= 0:
            return b""
        first = offset + (start // line_bases) * line_width + start % line_bases
        last = offset + ((stop - 1) // line_bases) * line_width + (stop - 1) % line_bases
        self._fh.seek(first)
        raw = self._fh.read(last - first + 1)
        return raw.translate(_UPPER, b"\n\r")               # line ends out and upper case in one pass

    def get_reference_sequence(self, region, start, stop):
        return self.get_reference_bytes(region, start, stop).decode()

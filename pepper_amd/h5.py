"""h5py-shaped facade over libpepper_amd_io.so (include/pepper_amd_io.h).

Only what PEPPER's inference path uses: open, list a group, whole-dataset reads, and
`file[path] = ndarray`-style writes (numeric, numpy 'S' fixed strings, vlen str), with the byte
layout h5py produces for the reference's DataStore classes.
"""
import ctypes
import functools
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libpepper_amd_io.so")

_CODES = {np.dtype(np.int8): 0, np.dtype(np.uint8): 1, np.dtype(np.int16): 2, np.dtype(np.int32): 3,
          np.dtype(np.int64): 4, np.dtype(np.float32): 5, np.dtype(np.float64): 6, np.dtype(np.uint16): 7,
          np.dtype(np.uint32): 8, np.dtype(np.uint64): 9}
CLASS_INT, CLASS_FLOAT, CLASS_FIXED, CLASS_VLEN = 0, 1, 2, 3

c_void_p, c_int32, c_int64, c_char_p = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_char_p
P64 = ctypes.POINTER(c_int64)
P32 = ctypes.POINTER(c_int32)

SYMBOLS = [
    ("pa_h5_last_error", c_char_p, []),
    ("pa_h5_open", ctypes.c_int, [c_char_p, c_int32, ctypes.POINTER(c_void_p)]),
    ("pa_h5_close", ctypes.c_int, [c_void_p]),
    ("pa_h5_flush", ctypes.c_int, [c_void_p]),
    ("pa_h5_exists", ctypes.c_int, [c_void_p, c_char_p]),
    ("pa_h5_list", ctypes.c_int, [c_void_p, c_char_p, c_void_p, c_int64, P64, P64]),
    ("pa_h5_info", ctypes.c_int, [c_void_p, c_char_p, P32, P64, P32, P32, P32]),
    ("pa_h5_read", ctypes.c_int, [c_void_p, c_char_p, c_int32, c_void_p, c_int64]),
    ("pa_h5_write", ctypes.c_int, [c_void_p, c_char_p, c_int32, c_int32, P64, c_void_p]),
    ("pa_h5_read_strings", ctypes.c_int, [c_void_p, c_char_p, c_void_p, c_int64, P64]),
    ("pa_h5_write_fixed_strings", ctypes.c_int, [c_void_p, c_char_p, c_int32, P64, c_int32, c_void_p]),
    ("pa_h5_write_vlen_strings", ctypes.c_int, [c_void_p, c_char_p, c_int32, P64, ctypes.POINTER(c_char_p)]),
    ("pa_h5_read_polish_chunks", ctypes.c_int, [c_void_p, c_char_p, c_int32, c_int32, c_int32] + [c_void_p] * 7 + [c_int32]),
    ("pa_h5_read_stats", ctypes.c_int, [c_void_p, P64, P64]),
    ("pa_h5_prediction_batch_load", ctypes.c_int, [c_void_p, c_char_p, P64, ctypes.POINTER(c_int32), P64, ctypes.POINTER(c_int32)]),
    ("pa_h5_prediction_batch_take", ctypes.c_int, [c_void_p] * 6),
    ("pa_h5_read_polish_prediction_region", ctypes.c_int, [c_void_p, c_char_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                                           ctypes.POINTER(c_int32)]),
    ("pa_h5_write_polish_image_chunks", ctypes.c_int, [c_void_p, c_char_p, c_int32, c_int32, c_int32, c_char_p, c_int64, c_int64] +
                                                       [c_void_p] * 5),
    ("pa_h5_write_polish_predictions", ctypes.c_int, [c_void_p, c_int32, c_int32, c_void_p, c_int32] + [c_void_p] * 9),
    ("pa_h5_stitch_polish_regions", ctypes.c_int, [c_void_p, c_void_p, c_char_p, c_void_p, c_int32, c_int64, P64, P64, P64, P64]),
    ("pa_h5_stitch_take", ctypes.c_int, [c_void_p, c_int64]),
    ("pa_h5_list_polish_regions", ctypes.c_int, [c_void_p, c_char_p, c_void_p, c_int64, P64, P64, c_void_p, c_void_p, c_int64]),
    ("pa_candidates_reference_flags", ctypes.c_int, [c_char_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    ("pa_candidates_select_format", c_int64, [c_void_p, c_char_p, c_int64] + [c_void_p] * 8 + [c_int32, c_void_p, c_void_p, c_void_p,
                                                                                            c_void_p, c_int64, c_void_p]),
    ("pa_h5_builder_open", ctypes.c_int, [c_char_p, ctypes.POINTER(c_void_p)]),
    ("pa_h5_builder_write_polish_predictions", ctypes.c_int, [c_void_p, c_int32, c_int32, c_void_p, c_int32] + [c_void_p] * 9),
    ("pa_h5_builder_write", ctypes.c_int, [c_void_p, c_char_p, c_int32, c_int32, P64, c_void_p]),
    ("pa_h5_builder_write_string", ctypes.c_int, [c_void_p, c_char_p, c_char_p]),
    ("pa_h5_builder_write_polish_image_chunks", ctypes.c_int, [c_void_p, c_char_p, c_int32, c_int32, c_int32, c_char_p, c_int64, c_int64] +
                                                               [c_void_p] * 5),
    ("pa_h5_builder_write_polish_image_regions", ctypes.c_int, [c_void_p, c_int32, c_char_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32] +
                                                                [c_void_p] * 4),
    ("pa_h5_builder_write_variant_summary", ctypes.c_int, [c_void_p, c_char_p, c_int32, c_char_p] + [c_void_p] * 6 + [c_int32, c_int32]),
    ("pa_h5_builder_write_prediction_batch", ctypes.c_int, [c_void_p, c_char_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p,
                                                            c_void_p, c_void_p, c_void_p, c_void_p, c_int32]),
    ("pa_h5_builder_close", ctypes.c_int, [c_void_p]),
    ("pa_h5_write_prediction_batch", ctypes.c_int, [c_void_p, c_char_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p,
                                                    c_void_p, c_void_p, c_void_p, c_void_p, c_int32]),
]

_lib = None
# libhdf5 is not built thread-safe: every call into it is serialised (a reader thread may prefetch the next image file
# while the main thread writes predictions; the GPU work in between holds no lock)
_LOCK = threading.RLock()


def _locked(fn):
    @functools.wraps(fn)
    def wrapper(*a, **kw):
        with _LOCK:
            return fn(*a, **kw)
    return wrapper


class H5Error(RuntimeError):
    pass


def load():
    global _lib
    if _lib is None:
        # rebuilds when hdf5io.cpp / bamio.cpp / the header are newer than the library (a no-op otherwise, and where the
        # sources or the compiler are absent the shipped library is used as is)
        from pepper_amd import build
        build.build_io()
        lib = ctypes.CDLL(LIB_PATH)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
        _lib = lib
    return _lib


def _check(rc):
    if rc < 0:
        raise H5Error(load().pa_h5_last_error().decode())
    return rc


@_locked
def stitch_polish_regions(files, file_of_region, region_paths, region_starts, buffer_positions):
    """pa_h5_stitch_polish_regions over open File objects: -> (first position, last position, sequence str); (-1, -1, '')
    when no row is left.  A label that is not a base raises KeyError(label), as the reference's label_decoder does."""
    lib = load()
    n = len(region_paths)
    handles = (c_void_p * max(1, len(files)))(*[f._h for f in files])
    which = np.ascontiguousarray(file_of_region, dtype=np.int32)
    starts = np.ascontiguousarray(region_starts, dtype=np.int64)
    blob = b"".join(p.encode() + b"\0" for p in region_paths)
    first, last, length, bad = c_int64(), c_int64(), c_int64(), c_int64()
    rc = lib.pa_h5_stitch_polish_regions(handles, which.ctypes.data, blob, starts.ctypes.data, n, int(buffer_positions),
                                         ctypes.byref(first), ctypes.byref(last), ctypes.byref(length), ctypes.byref(bad))
    if rc < 0 and bad.value >= 0:
        raise KeyError(int(bad.value))
    _check(rc)
    buf = ctypes.create_string_buffer(max(1, length.value))
    _check(lib.pa_h5_stitch_take(buf, length.value))
    return first.value, last.value, buf.raw[:length.value].decode()


class PredictionBuilder(object):
    """Append-only writer of the polish stores -- prediction files and image files -- (pa_h5_builder_*,
    pepper_amd/csrc/h5build.cpp): raw rows go to the file as they come, the HDF5 metadata is written by close().  No libhdf5
    involved, so no lock either."""

    def __init__(self, path):
        # The bytes on disk are not an HDF5 file until close() has written the metadata: they are laid out under
        # `<path>.tmp` and take the final name only when close() has succeeded, so that a worker that is killed or raises mid-run
        # leaves no `*.hdf` behind for the next step's directory listing (perform_stitch, run_inference) to choke on.
        self._lib = load()
        self._h = c_void_p()
        self._tmp = str(path) + ".tmp"
        _check(self._lib.pa_h5_builder_open(os.fsencode(self._tmp), ctypes.byref(self._h)))
        self.filename, self.mode = path, "w"

    def write_polish_predictions(self, contigs, start, end, chunk, new_region, skip, position, index, bases, phred):
        n, seq_len = bases.shape
        _check(self._lib.pa_h5_builder_write_polish_predictions(
            self._h, n, seq_len, contigs.ctypes.data, contigs.dtype.itemsize, start.ctypes.data, end.ctypes.data,
            chunk.ctypes.data, new_region.ctypes.data, skip.ctypes.data, position.ctypes.data, index.ctypes.data,
            bases.ctypes.data, phred.ctypes.data))

    def write_prediction_batch(self, group, contigs, positions, depths, cand_blob, cand_offsets, freqs, probs):
        """File.write_prediction_batch's group (variant predictions/batch_<n>) through the append-only writer; group =
        "predictions/<name>"."""
        head, _, name = group.rpartition("/")
        if head != "predictions":
            raise H5Error("the append-only writer lays prediction batches out under predictions/: " + group)
        n = len(positions)
        blob = np.ascontiguousarray(cand_blob)
        _check(self._lib.pa_h5_builder_write_prediction_batch(
            self._h, name.encode(), n, contigs.ctypes.data, contigs.dtype.itemsize, positions.ctypes.data, depths.ctypes.data,
            blob.ctypes.data, cand_offsets.ctypes.data, freqs.ctypes.data, probs.ctypes.data, probs.shape[1] if probs.ndim == 2 else 1))

    def write_polish_image_chunks(self, names, contig, region_start, region_end, chunk_id, images, labels, position, index):
        n, seq_len, features = images.shape
        blob = b"".join(s.encode() + b"\0" for s in names)
        _check(self._lib.pa_h5_builder_write_polish_image_chunks(
            self._h, blob, n, seq_len, features, contig.encode(), int(region_start), int(region_end), chunk_id.ctypes.data,
            images.ctypes.data, labels.ctypes.data, position.ctypes.data, index.ctypes.data))

    def write_polish_image_regions(self, contig, region_start, region_end, n_chunks, seq_len, features, images, position, index):
        """The chunks of many regions of one contig in one call (pa_h5_builder_write_polish_image_regions): region r has
        n_chunks[r] consecutive chunks in images (address of uint8 [*, seq_len, features]) / position / index (addresses of
        int64 [*, seq_len]); labels are zeros.  The three are raw addresses: the image chain's page-locked blocks."""
        region_start = np.ascontiguousarray(region_start, np.int64)
        region_end = np.ascontiguousarray(region_end, np.int64)
        n_chunks = np.ascontiguousarray(n_chunks, np.int32)
        _check(self._lib.pa_h5_builder_write_polish_image_regions(
            self._h, len(n_chunks), contig.encode(), region_start.ctypes.data, region_end.ctypes.data, n_chunks.ctypes.data,
            int(seq_len), int(features), images, None, position, index))

    def write_variant_summary(self, name, contig, positions, depths, candidates, freqs, images):
        """One summaries/<name> group of a variant image file (pa_h5_builder_write_variant_summary): positions int32 [n], depths
        uint8 [n], candidates: n strings, freqs uint8 [n], images int8 [n, window, features]."""
        n = len(positions)
        blob = b"".join(c.encode("utf-8") + b"\0" for c in candidates)
        offsets = np.zeros(n + 1, np.int64)
        if n:
            np.cumsum([len(c.encode("utf-8")) + 1 for c in candidates], out=offsets[1:])
        _check(self._lib.pa_h5_builder_write_variant_summary(
            self._h, name.encode(), n, contig.encode(), positions.ctypes.data, depths.ctypes.data, blob, offsets.ctypes.data,
            freqs.ctypes.data, images.ctypes.data, images.shape[1], images.shape[2]))

    def write_variant_summary_packed(self, name, contig, positions, depths, blob, offsets, freqs, images):
        """write_variant_summary with the candidate strings as the encoder returns them: `blob` = n NUL-terminated strings back
        to back, `offsets` int64 [n + 1]."""
        offsets = np.ascontiguousarray(offsets, np.int64)
        _check(self._lib.pa_h5_builder_write_variant_summary(
            self._h, name.encode(), len(positions), contig.encode(), positions.ctypes.data, depths.ctypes.data, blob, offsets.ctypes.data,
            freqs.ctypes.data, images.ctypes.data, images.shape[1], images.shape[2]))

    def __setitem__(self, path, value):
        """An integer dataset, or a str as a variable-length string scalar, like h5py's file[path] = value (intermediate
        groups are made as needed)."""
        if isinstance(value, str):
            _check(self._lib.pa_h5_builder_write_string(self._h, path.encode(), value.encode("utf-8")))
            return
        arr = np.asarray(value)
        shape = arr.shape                     # np.ascontiguousarray would promote 0-d to 1-d
        arr = np.ascontiguousarray(arr).reshape(shape)
        if arr.dtype == np.bool_:
            arr = arr.astype(np.uint8).reshape(shape)
        if arr.dtype not in _CODES or arr.dtype.kind not in "iu":
            raise H5Error(f"the prediction builder writes integer datasets only, not {arr.dtype} for '{path}'")
        dims = (c_int64 * max(1, arr.ndim))(*arr.shape)
        _check(self._lib.pa_h5_builder_write(self._h, path.encode(), _CODES[arr.dtype], arr.ndim, dims, arr.ctypes.data))

    def close(self):
        if self._h:
            h, self._h = self._h, None
            try:
                _check(self._lib.pa_h5_builder_close(h))
            except Exception:
                if os.path.exists(self._tmp):
                    os.remove(self._tmp)
                raise
            os.replace(self._tmp, self.filename)

    def __enter__(self):
        return self

    def __exit__(self, exc_type, *a):
        if exc_type is not None:
            # the block raised: whatever was written is not a complete store -- finish the native object, keep nothing
            h, self._h = self._h, None
            if h:
                self._lib.pa_h5_builder_close(h)
                if os.path.exists(self._tmp):
                    os.remove(self._tmp)
            return
        self.close()

    def abort(self):
        """Give the store up: the native object is finished and the temporary file removed -- what the predict loops call
        when a batch raised (a `finally: close()` would publish a partial store under the final name)."""
        h, self._h = self._h, None
        if h:
            self._lib.pa_h5_builder_close(h)
            if os.path.exists(self._tmp):
                os.remove(self._tmp)

    def __del__(self):
        # (a writer that was never closed: the metadata is still written so that the handle is released, and the file keeps
        # its temporary name -- an explicit close() is what publishes a store; say so, a caller that relied on garbage
        # collection would otherwise lose its output silently)
        try:
            h, self._h = self._h, None
            if h:
                self._lib.pa_h5_builder_close(h)
                import warnings
                warnings.warn("PredictionBuilder for %s was never closed: the store stays under %s" % (self.filename, self._tmp),
                              ResourceWarning)
        except Exception:
            pass


class File(object):
    """with File(path, 'r'|'w'|'r+') as f:  f.keys(group), f[path] (read), f[path] = array (create)."""

    @_locked
    def __init__(self, path, mode="r"):
        self._lib = load()
        self._h = c_void_p()
        code = {"r": 0, "w": 1, "w-new": 3, "r+": 2, "a": 2 if os.path.exists(path) else 1}[mode]
        _check(self._lib.pa_h5_open(os.fsencode(path), code, ctypes.byref(self._h)))
        self.filename, self.mode = path, mode

    @_locked
    def close(self):
        if self._h:
            h, self._h = self._h, None
            _check(self._lib.pa_h5_close(h))

    @_locked
    def flush(self):
        _check(self._lib.pa_h5_flush(self._h))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @_locked
    def __contains__(self, path):
        return _check(self._lib.pa_h5_exists(self._h, path.encode())) == 1

    @_locked
    def keys(self, group="/"):
        needed, count = c_int64(), c_int64()
        _check(self._lib.pa_h5_list(self._h, group.encode(), None, 0, ctypes.byref(needed), ctypes.byref(count)))
        if count.value == 0:
            return []
        buf = ctypes.create_string_buffer(needed.value)
        _check(self._lib.pa_h5_list(self._h, group.encode(), buf, needed.value, ctypes.byref(needed), ctypes.byref(count)))
        return [s.decode() for s in buf.raw[:needed.value].split(b"\0")[:count.value]]

    @_locked
    def info(self, path):
        rank, cls, size, sgn = c_int32(), c_int32(), c_int32(), c_int32()
        dims = (c_int64 * 8)()
        _check(self._lib.pa_h5_info(self._h, path.encode(), ctypes.byref(rank), dims, ctypes.byref(cls),
                                    ctypes.byref(size), ctypes.byref(sgn)))
        return tuple(dims[:rank.value]), cls.value, size.value, bool(sgn.value)

    @_locked
    def read_into(self, path, out):
        """Whole numeric dataset into a caller-provided C-contiguous array of the matching size (e.g. a slice of a pinned
        staging buffer): no intermediate array."""
        if out.dtype not in _CODES or not out.flags["C_CONTIGUOUS"]:
            raise H5Error(f"read_into needs a C-contiguous array of a supported dtype for '{path}'")
        _check(self._lib.pa_h5_read(self._h, path.encode(), _CODES[out.dtype], out.ctypes.data, out.nbytes))
        return out

    @_locked
    def read_strings_raw(self, path):
        """A string dataset as one bytes object of NUL-terminated entries (no per-element Python objects)."""
        needed = c_int64()
        _check(self._lib.pa_h5_read_strings(self._h, path.encode(), None, 0, ctypes.byref(needed)))
        buf = ctypes.create_string_buffer(max(1, needed.value))
        _check(self._lib.pa_h5_read_strings(self._h, path.encode(), buf, needed.value, ctypes.byref(needed)))
        return buf.raw[:needed.value]

    @_locked
    def read_polish_chunks(self, names, seq_len, features, contig_width=256, out=None):
        """summaries/<name> groups of a polish image file as bulk arrays (one library call for the block).
        out = (images u8 [n, seq, features], position i64 [n, seq], index i64 [n, seq]) to read into caller memory."""
        n = len(names)
        if out is not None:
            images, position, index = out
            assert images.shape == (n, seq_len, features) and images.dtype == np.uint8 and images.flags.c_contiguous
            assert position.shape == (n, seq_len) and position.dtype == np.int64 and index.shape == (n, seq_len)
        else:
            images = np.empty((n, seq_len, features), np.uint8)
            position, index = np.empty((n, seq_len), np.int64), np.empty((n, seq_len), np.int64)
        start, end, chunk = np.empty(n, np.int64), np.empty(n, np.int64), np.empty(n, np.int64)
        contigs = np.zeros(n, dtype=f"S{contig_width}")
        blob = b"".join(s.encode() + b"\0" for s in names)
        _check(self._lib.pa_h5_read_polish_chunks(self._h, blob, n, seq_len, features, images.ctypes.data, position.ctypes.data,
                                                  index.ctypes.data, start.ctypes.data, end.ctypes.data, chunk.ctypes.data,
                                                  contigs.ctypes.data, contig_width))
        return contigs, start, end, chunk, images, position, index

    def read_stats(self):
        """(polish chunks copied straight from the mapped file, polish chunks read through libhdf5) of this handle"""
        a, b = c_int64(), c_int64()
        _check(self._lib.pa_h5_read_stats(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    @_locked
    def list_polish_regions(self, contig):
        """[(region group name, contig_start, contig_end)] of predictions/<contig>, names in sorted order."""
        needed, count = c_int64(), c_int64()
        _check(self._lib.pa_h5_list_polish_regions(self._h, contig.encode(), None, 0, ctypes.byref(needed), ctypes.byref(count),
                                                   None, None, 0))
        if count.value == 0:
            return []
        buf = ctypes.create_string_buffer(max(1, needed.value))
        starts, ends = np.empty(count.value, np.int64), np.empty(count.value, np.int64)
        _check(self._lib.pa_h5_list_polish_regions(self._h, contig.encode(), buf, needed.value, ctypes.byref(needed),
                                                   ctypes.byref(count), starts.ctypes.data, ends.ctypes.data, count.value))
        names = buf.raw[:needed.value].split(b"\0")[:count.value]
        return [(n.decode(), int(s), int(e)) for n, s, e in zip(names, starts.tolist(), ends.tolist())]

    @_locked
    def read_polish_prediction_region(self, region_path, seq_len, max_chunks=64):
        """(position, index, bases) rows of every chunk under one predictions region group, chunk ids in string order."""
        position, index = np.empty((max_chunks, seq_len), np.int64), np.empty((max_chunks, seq_len), np.int64)
        bases = np.empty((max_chunks, seq_len), np.uint8)
        n = c_int32()
        _check(self._lib.pa_h5_read_polish_prediction_region(self._h, region_path.encode(), seq_len, max_chunks,
                                                             position.ctypes.data, index.ctypes.data, bases.ctypes.data,
                                                             ctypes.byref(n)))
        return position[:n.value], index[:n.value], bases[:n.value]

    @_locked
    def write_polish_image_chunks(self, names, contig, region_start, region_end, chunk_id, images, labels, position, index):
        n, seq_len, features = images.shape
        blob = b"".join(s.encode() + b"\0" for s in names)
        _check(self._lib.pa_h5_write_polish_image_chunks(self._h, blob, n, seq_len, features, contig.encode(), int(region_start),
                                                         int(region_end), chunk_id.ctypes.data, images.ctypes.data,
                                                         labels.ctypes.data, position.ctypes.data, index.ctypes.data))

    @_locked
    def write_polish_predictions(self, contigs, start, end, chunk, new_region, skip, position, index, bases, phred):
        n, seq_len = bases.shape
        _check(self._lib.pa_h5_write_polish_predictions(self._h, n, seq_len, contigs.ctypes.data, contigs.dtype.itemsize,
                                                        start.ctypes.data, end.ctypes.data, chunk.ctypes.data,
                                                        new_region.ctypes.data, skip.ctypes.data, position.ctypes.data,
                                                        index.ctypes.data, bases.ctypes.data, phred.ctypes.data))

    @_locked
    def write_prediction_batch(self, group, contigs, positions, depths, cand_blob, cand_offsets, freqs, probs):
        """One predictions/batch_<n> group in one library call (include/pepper_amd_io.h)."""
        n = len(positions)
        _check(self._lib.pa_h5_write_prediction_batch(self._h, group.encode(), n, contigs.ctypes.data, contigs.dtype.itemsize,
                                                      positions.ctypes.data, depths.ctypes.data, cand_blob.ctypes.data,
                                                      cand_offsets.ctypes.data, freqs.ctypes.data, probs.ctypes.data,
                                                      probs.shape[1]))

    def read_prediction_batch(self, group):        # (no libhdf5 call inside: no lock)
        """One predictions/batch_<n> group in one call (pa_h5_prediction_batch_load / _take): (contigs uint8 [n, width] null
        padded, candidate strings as bytes each followed by a NUL, positions int32 [n], depths uint8 [n], candidate_frequency
        uint8 [n, 1], base_prediction float64 [n, classes]) -- or None when the file is not of a layout the locator reads (the
        caller reads the datasets one by one then)."""
        n, width, cand_bytes, classes = c_int64(), c_int32(), c_int64(), c_int32()
        rc = self._lib.pa_h5_prediction_batch_load(self._h, group.encode(), ctypes.byref(n), ctypes.byref(width),
                                                   ctypes.byref(cand_bytes), ctypes.byref(classes))
        if rc == 1:
            return None
        _check(rc)
        rows = n.value
        contigs = np.empty((rows, max(1, width.value)), np.uint8)
        cand = np.empty(max(1, cand_bytes.value), np.uint8)
        positions, depths = np.empty(rows, np.int32), np.empty(rows, np.uint8)
        freq, probs = np.empty((rows, 1), np.uint8), np.empty((rows, max(1, classes.value)), np.float64)
        _check(self._lib.pa_h5_prediction_batch_take(contigs.ctypes.data, cand.ctypes.data, positions.ctypes.data, depths.ctypes.data,
                                                     freq.ctypes.data, probs.ctypes.data))
        return contigs, cand[:cand_bytes.value].tobytes(), positions, depths, freq, probs

    @_locked
    def read_strings_shaped(self, path):
        """A string dataset as pa_h5_read_strings returns it: (shape, bytes of the elements, each followed by a NUL)."""
        shape, cls, size, sgn = self.info(path)
        if cls not in (CLASS_FIXED, CLASS_VLEN):
            raise H5Error(f"'{path}' is not a string dataset")
        needed = c_int64()
        _check(self._lib.pa_h5_read_strings(self._h, path.encode(), None, 0, ctypes.byref(needed)))
        buf = ctypes.create_string_buffer(max(1, needed.value))
        _check(self._lib.pa_h5_read_strings(self._h, path.encode(), buf, needed.value, ctypes.byref(needed)))
        return shape, buf.raw[:needed.value]

    @_locked
    def __getitem__(self, path):
        """Whole-dataset read (the reference only ever does dataset[()])."""
        shape, cls, size, sgn = self.info(path)
        if cls in (CLASS_FIXED, CLASS_VLEN):
            needed = c_int64()
            _check(self._lib.pa_h5_read_strings(self._h, path.encode(), None, 0, ctypes.byref(needed)))
            buf = ctypes.create_string_buffer(max(1, needed.value))
            _check(self._lib.pa_h5_read_strings(self._h, path.encode(), buf, needed.value, ctypes.byref(needed)))
            n = int(np.prod(shape)) if shape else 1
            parts = buf.raw[:needed.value].split(b"\0")[:n]
            if cls == CLASS_FIXED:
                arr = np.array(parts, dtype=f"S{max(1, size)}")
            else:
                arr = np.empty(n, dtype=object)
                arr[:] = [p.decode("utf-8") for p in parts]   # h5py 2.x returns str for vlen strings
            return arr.reshape(shape) if shape else arr[0]
        if cls == CLASS_FLOAT:
            dt = np.dtype(np.float32 if size == 4 else np.float64)
        elif cls == CLASS_INT:
            dt = np.dtype({1: "i1", 2: "i2", 4: "i4", 8: "i8"}[size] if sgn else {1: "u1", 2: "u2", 4: "u4", 8: "u8"}[size])
        else:
            raise H5Error(f"unsupported dataset class for '{path}'")
        out = np.empty(shape, dtype=dt)
        _check(self._lib.pa_h5_read(self._h, path.encode(), _CODES[dt], out.ctypes.data, out.nbytes))
        return out if shape else out[()]

    @_locked
    def __setitem__(self, path, value):
        """Create a contiguous dataset like h5py's file[path] = value."""
        if isinstance(value, str):
            value = np.array(value, dtype=object)
        if isinstance(value, bytes):
            value = np.array(value, dtype="S")
        arr = np.asarray(value)
        shape = arr.shape                     # np.ascontiguousarray would promote 0-d to 1-d
        arr = np.ascontiguousarray(arr).reshape(shape)
        dims = (c_int64 * max(1, arr.ndim))(*arr.shape)
        if arr.dtype.kind == "S":
            _check(self._lib.pa_h5_write_fixed_strings(self._h, path.encode(), arr.ndim, dims, arr.dtype.itemsize,
                                                       arr.ctypes.data))
        elif arr.dtype.kind in ("O", "U"):
            flat = [(s if isinstance(s, bytes) else str(s).encode("utf-8")) for s in arr.ravel().tolist()]
            ptrs = (c_char_p * max(1, len(flat)))(*flat)
            _check(self._lib.pa_h5_write_vlen_strings(self._h, path.encode(), arr.ndim, dims, ptrs))
        else:
            if arr.dtype == np.bool_:
                arr = arr.astype(np.uint8).reshape(shape)
            if arr.dtype not in _CODES:
                raise H5Error(f"unsupported dtype {arr.dtype} for '{path}'")
            _check(self._lib.pa_h5_write(self._h, path.encode(), _CODES[arr.dtype], arr.ndim, dims, arr.ctypes.data))

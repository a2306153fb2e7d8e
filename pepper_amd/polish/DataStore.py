"""Polish images HDF5 store.  Mirrors /root/reference/pepper/modules/python/DataStore.py:6-67:
summaries/<name>/{image u8 [1000,10], label u8 [1000], position, index, contig, region_start,
region_end, chunk_id} (position = list of (pos, idx) pairs -> int64 [1000,2] through h5py)."""
import os

import numpy as np

from pepper_amd import h5


class DataStore(object):
    _summary_path_ = 'summaries'

    def __init__(self, filename, mode='r'):
        self.filename = filename
        self.mode = mode
        self.file_handler = None
        self._written = set()

    def __enter__(self):
        # 'w' -> the append-only builder (h5.PredictionBuilder, csrc/h5build.cpp): no libhdf5 and so no process-wide lock under
        # the image-generation threads, each of which writes its own file; PEPPER_AMD_H5_BUILDER=0: through libhdf5
        if self.mode == "w" and os.environ.get("PEPPER_AMD_H5_BUILDER", "1") != "0":
            self.file_handler = h5.PredictionBuilder(self.filename)
        else:
            self.file_handler = h5.File(self.filename, self.mode)
        return self

    def __exit__(self, *args):
        self.file_handler.__exit__(*args)      # (the append-only writer publishes the file only when the block did not raise)

    def write_summary(self, region, image, label, position, index, chunk_id, summary_name):
        contig_name, region_start, region_end = region
        if summary_name in self._written:
            return
        self._written.add(summary_name)
        base = '{}/{}/'.format(self._summary_path_, summary_name)
        fh = self.file_handler
        fh[base + 'image'] = np.asarray(image, dtype=np.float64).astype(np.uint8)
        fh[base + 'label'] = np.asarray(label, dtype=np.uint8)
        fh[base + 'position'] = np.asarray(position)
        fh[base + 'index'] = np.asarray(index)
        fh[base + 'contig'] = contig_name
        fh[base + 'region_start'] = region_start
        fh[base + 'region_end'] = region_end
        fh[base + 'chunk_id'] = chunk_id

    def write_summaries(self, region, images, labels, positions, chunk_ids):
        """write_summary for all chunks of one region in one library call: images uint8 [n,1000,10] (or a list of
        [1000,10] arrays), labels [n,1000], positions int64 [n,1000,2] = (position, index) pairs, chunk ids.  Same groups,
        datasets and types; chunks whose group name was written before are skipped, as write_summary does."""
        contig_name, region_start, region_end = region
        names, keep = [], []
        for k, chunk_id in enumerate(chunk_ids):
            name = str(contig_name) + "_" + str(region_start) + "_" + str(region_end) + "_" + str(chunk_id)
            if name in self._written:
                continue
            self._written.add(name)
            names.append(name)
            keep.append(k)
        if not names:
            return
        images = np.ascontiguousarray(np.asarray(images)[keep], dtype=np.uint8)
        labels = np.ascontiguousarray(np.asarray(labels)[keep], dtype=np.uint8)
        positions = np.asarray(positions)[keep]
        self.file_handler.write_polish_image_chunks(
            names, str(contig_name), region_start, region_end, np.asarray(chunk_ids, dtype=np.int64)[keep],
            images, labels, np.ascontiguousarray(positions[:, :, 0], dtype=np.int64),
            np.ascontiguousarray(positions[:, :, 1], dtype=np.int64))

    def write_regions(self, contig_name, region_starts, region_ends, n_chunks, seq_len, features, images, position, index):
        """The chunks of many regions of one contig as the image chain leaves them (pa_polish_chain_chunks: raw addresses of
        uint8 [*, seq_len, features] and int64 [*, seq_len] blocks, region after region, chunk ids 0 .. n_chunks[r] - 1):
        the groups write_summary would make for each of them, in one library call."""
        if hasattr(self.file_handler, "write_polish_image_regions"):
            self.file_handler.write_polish_image_regions(str(contig_name), region_starts, region_ends, n_chunks, seq_len, features,
                                                         images, position, index)
            return
        import ctypes
        total = int(np.sum(n_chunks))
        if total == 0:
            return
        img = np.ctypeslib.as_array(ctypes.cast(images, ctypes.POINTER(ctypes.c_uint8)), shape=(total, seq_len, features))
        pos = np.ctypeslib.as_array(ctypes.cast(position, ctypes.POINTER(ctypes.c_int64)), shape=(total, seq_len))
        idx = np.ctypeslib.as_array(ctypes.cast(index, ctypes.POINTER(ctypes.c_int64)), shape=(total, seq_len))
        at = 0
        for a, b, k in zip(region_starts, region_ends, n_chunks):
            k = int(k)
            if k:
                self.write_summaries((contig_name, int(a), int(b)), img[at:at + k], np.zeros((k, seq_len), np.uint8),
                                     np.stack([pos[at:at + k], idx[at:at + k]], axis=2), list(range(k)))
            at += k

"""Predictions HDF5 store (the output layout downstream candidate finding reads).

Mirrors /root/reference/pepper_variant/modules/python/DataStorePredict.py:6-67 (class DataStore,
write_prediction): predictions/batch_<n>/{contigs 'S', positions i32, depths u8, candidates
vlen-str, candidate_frequency u8, base_prediction float64 [B,3]} -- `np.float` in the reference
is float64; the type_prediction dataset is commented out there and is not written here either.
"""
import os

import numpy as np

from pepper_amd import h5


class DataStore(object):
    _prediction_path_ = 'predictions'

    def __init__(self, filename, mode='r', bulk=False):
        """bulk=True (mode 'w'): the append-only writer (h5.PredictionBuilder: groups laid out as they come, the HDF5 metadata
        written by close(); no libhdf5 call and no process-wide lock per batch) -- for writers that only use
        write_prediction_arrays; PEPPER_AMD_H5_BUILDER=0 keeps libhdf5."""
        self.filename = filename
        self.mode = mode
        if bulk and mode == 'w' and os.environ.get("PEPPER_AMD_H5_BUILDER", "1") != "0":
            self.file_handler = h5.PredictionBuilder(self.filename)
        else:
            self.file_handler = h5.File(self.filename, self.mode)
        self._written = set()

    def abort(self):
        """The run raised: publish nothing (the append-only writer removes its temporary file; a libhdf5 file is closed)."""
        if hasattr(self.file_handler, "abort"):
            self.file_handler.abort()
        else:
            self.file_handler.close()

    def close(self):
        self.file_handler.close()

    def __enter__(self):
        return self

    def __exit__(self, *args):
        self.close()

    def write_prediction(self, batch_no, contigs, positions, depths, candidates, candidate_frequencies,
                         base_predictions):
        name = "batch_" + str(batch_no)
        if name in self._written:
            return
        self._written.add(name)
        base = '{}/{}/'.format(self._prediction_path_, name)
        fh = self.file_handler
        fh[base + "contigs"] = np.array(contigs, dtype='S')
        fh[base + "positions"] = np.asarray(positions, dtype=np.int32)
        fh[base + "depths"] = np.asarray(depths, dtype=np.uint8)
        fh[base + "candidates"] = np.asarray(candidates, dtype=object)
        fh[base + "candidate_frequency"] = np.asarray(candidate_frequencies, dtype=np.uint8)
        fh[base + "base_prediction"] = np.asarray(base_predictions, dtype=np.float64)

    def write_prediction_arrays(self, batch_no, contigs, positions, depths, candidate_blob, candidate_offsets,
                                candidate_frequencies, base_predictions):
        """The same group as write_prediction from bulk arrays in one library call: contigs a numpy 'S' array,
        candidates as NUL-terminated strings at candidate_blob[candidate_offsets[i]:], base_predictions float32 [B,3]
        (stored as float64, as the reference's np.float)."""
        name = "batch_" + str(batch_no)
        if name in self._written:
            return
        self._written.add(name)
        self.file_handler.write_prediction_batch(
            self._prediction_path_ + "/" + name, np.ascontiguousarray(contigs), np.ascontiguousarray(positions, dtype=np.int32),
            np.ascontiguousarray(depths, dtype=np.uint8), candidate_blob, np.ascontiguousarray(candidate_offsets, dtype=np.int64),
            np.ascontiguousarray(candidate_frequencies, dtype=np.uint8), np.ascontiguousarray(base_predictions, dtype=np.float32))


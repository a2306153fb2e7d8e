"""Polish predictions HDF5 store.  Mirrors /root/reference/pepper/modules/python/DataStorePredict.py:6-76:
predictions/<contig>/<contig>-<start>-<end>/{contig_start, contig_end} and
predictions/<contig>/<contig>-<start>-<end>/<chunk_id>/{position, index, bases u8, phred_score u8}."""
import os

import numpy as np

from pepper_amd import h5


def _item(v):
    return v.item() if hasattr(v, "item") else v


class DataStore(object):
    _prediction_path_ = 'predictions'

    def __init__(self, filename, mode='r'):
        self.filename = filename
        self.mode = mode
        # 'w' -> the append-only builder (h5.PredictionBuilder: rows appended as they come, all HDF5 metadata written by close();
        # ~3 us of CPU per chunk where libhdf5 takes 70-150 us for the group and its four datasets).  PEPPER_AMD_H5_BUILDER=0:
        # through libhdf5 with the 1.10 object formats (h5.py 'w-new'), as before.
        if self.mode == "w" and os.environ.get("PEPPER_AMD_H5_BUILDER", "1") != "0":
            self.file_handler = h5.PredictionBuilder(self.filename)
        else:
            self.file_handler = h5.File(self.filename, "w-new" if self.mode == "w" else self.mode)
        self._predictions = set()
        self._contigs = set()

    def close(self):
        self.file_handler.close()

    def abort(self):
        """The run raised: publish nothing (the append-only writer removes its temporary file; a libhdf5 file is closed)."""
        if hasattr(self.file_handler, "abort"):
            self.file_handler.abort()
        else:
            self.file_handler.close()

    def __enter__(self):
        return self

    def __exit__(self, *args):
        self.file_handler.__exit__(*args)      # (the append-only writer publishes the file only when the block did not raise)

    def write_prediction(self, contig, contig_start, contig_end, chunk_id, position, index, predicted_bases,
                         phred_score):
        if isinstance(contig, bytes):
            contig = contig.decode('UTF-8')
        chunk_name_prefix = str(contig) + "-" + str(_item(contig_start)) + "-" + str(_item(contig_end))
        chunk_name_suffix = str(_item(chunk_id))
        name = contig + chunk_name_prefix + chunk_name_suffix
        fh = self.file_handler
        if chunk_name_prefix not in self._contigs:
            self._contigs.add(chunk_name_prefix)
            fh['{}/{}/{}/{}'.format(self._prediction_path_, contig, chunk_name_prefix, 'contig_start')] = _item(contig_start)
            fh['{}/{}/{}/{}'.format(self._prediction_path_, contig, chunk_name_prefix, 'contig_end')] = _item(contig_end)
        if name not in self._predictions:
            self._predictions.add(name)
            base = '{}/{}/{}/{}/'.format(self._prediction_path_, contig, chunk_name_prefix, chunk_name_suffix)
            fh[base + 'position'] = np.asarray(position)
            fh[base + 'index'] = np.asarray(index)
            fh[base + 'bases'] = np.asarray(predicted_bases).astype(np.uint8)
            fh[base + 'phred_score'] = np.asarray(phred_score).astype(np.uint8)

    def write_predictions_block(self, contigs, contig_start, contig_end, chunk_id, position, index, predicted_bases,
                                phred_score):
        """write_prediction for a block of chunks in one library call (contigs: numpy 'S' array; the other arguments
        int64 / uint8 arrays with one row per chunk).  Same groups, datasets and duplicate handling."""
        n = len(contigs)
        new_region, skip = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        for i in range(n):
            contig = contigs[i].decode('UTF-8')
            prefix = contig + "-" + str(int(contig_start[i])) + "-" + str(int(contig_end[i]))
            name = contig + prefix + str(int(chunk_id[i]))
            if prefix not in self._contigs:
                self._contigs.add(prefix)
                new_region[i] = 1
            if name in self._predictions:
                skip[i] = 1
            else:
                self._predictions.add(name)
        self.file_handler.write_polish_predictions(
            np.ascontiguousarray(contigs), np.ascontiguousarray(contig_start, dtype=np.int64),
            np.ascontiguousarray(contig_end, dtype=np.int64), np.ascontiguousarray(chunk_id, dtype=np.int64), new_region, skip,
            np.ascontiguousarray(position, dtype=np.int64), np.ascontiguousarray(index, dtype=np.int64),
            np.ascontiguousarray(predicted_bases, dtype=np.uint8), np.ascontiguousarray(phred_score, dtype=np.uint8))


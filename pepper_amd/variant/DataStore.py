"""Images HDF5 store (the on-disk input of inference).

Mirrors /root/reference/pepper_variant/modules/python/DataStore.py:6-71 (class DataStore,
write_summary): summaries/<chr_start_end>/{contigs 'S', positions i32, depths u8, candidates
vlen-str, candidate_frequency u8, images int8 [N,33,26]} (+ base_labels / type_label u8 in train
mode).  Written through pepper_amd.h5 (libhdf5) instead of h5py.
"""
import numpy as np

from pepper_amd import h5


def wrap_int8(values):
    """numpy 1.22's np.array(list_of_ints, dtype=np.int8) wrapped out-of-range values silently
    (the encoder leaves columns 4, 8-10 and 25 unclamped, so depth > 128 does occur; SURVEY.md
    section 7).  numpy 2 raises instead, so the wrap is spelled out."""
    return np.asarray(values, dtype=np.int64).astype(np.int8)


def wrap_uint8(values):
    return np.asarray(values, dtype=np.int64).astype(np.uint8)


class DataStore(object):
    _summary_path_ = 'summaries'

    def __init__(self, filename, mode='r'):
        self.filename = filename
        self.mode = mode
        self.file_handler = None
        self._written = set()

    def __enter__(self):
        self.file_handler = h5.File(self.filename, self.mode)
        return self

    def __exit__(self, *args):
        self.file_handler.close()

    def write_summary(self, summary_name, contigs, positions, depths, all_candidates, all_candidate_frequency,
                      all_images, all_base_labels, all_type_label, train_mode):
        if summary_name in self._written:
            return
        self._written.add(summary_name)
        base = '{}/{}/'.format(self._summary_path_, summary_name)
        fh = self.file_handler
        fh[base + "contigs"] = np.array(contigs, dtype='S')
        fh[base + "positions"] = np.asarray(positions, dtype=np.int64).astype(np.int32)
        fh[base + "depths"] = wrap_uint8(depths)
        fh[base + "candidates"] = np.asarray(all_candidates, dtype=object)
        fh[base + "candidate_frequency"] = wrap_uint8(all_candidate_frequency)
        img = np.asarray(all_images)
        fh[base + "images"] = img if img.dtype == np.int8 else wrap_int8(all_images)
        if train_mode:
            fh[base + "base_labels"] = wrap_uint8(all_base_labels)
            fh[base + "type_label"] = wrap_uint8(all_type_label)

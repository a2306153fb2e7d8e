"""Images HDF5 store (the on-disk input of inference).

Mirrors /root/reference/pepper_variant/modules/python/DataStore.py:6-71 (class DataStore,
write_summary): summaries/<chr_start_end>/{contigs 'S', positions i32, depths u8, candidates
vlen-str, candidate_frequency u8, images int8 [N,33,26]} (+ base_labels / type_label u8 in train
mode).  Written through pepper_amd.h5 (libhdf5) instead of h5py.
"""
import os

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
        # 'w' -> the append-only builder (h5.PredictionBuilder, csrc/h5build.cpp): no libhdf5 and so no process-wide lock under
        # the image-generation workers, each of which writes its own file; PEPPER_AMD_H5_BUILDER=0: through libhdf5
        if self.mode == "w" and os.environ.get("PEPPER_AMD_H5_BUILDER", "1") != "0":
            self.file_handler = h5.PredictionBuilder(self.filename)
        else:
            self.file_handler = h5.File(self.filename, self.mode)
        return self

    def __exit__(self, *args):
        self.file_handler.__exit__(*args)      # (the append-only writer publishes the file only when the block did not raise)

    def write_summary_packed(self, summary_name, contig, out):
        """One interval's summary straight from the encoder's result dict (PEPPER_VARIANT.StagedBatch.results): the datasets
        of write_summary in inference mode, the candidate strings as the library returned them -- no per-candidate Python
        object on the way (the append-only writer only; returns False when the file is written through libhdf5)."""
        fh = self.file_handler
        if not isinstance(fh, h5.PredictionBuilder) or "candidates_blob" not in out:
            return False
        if summary_name in self._written:
            return True
        self._written.add(summary_name)
        n = len(out["positions"])
        img = out["images"] if n else np.zeros((0, 33, 26), np.int8)
        fh.write_variant_summary_packed(summary_name, contig if n else "",
                                        np.ascontiguousarray(out["positions"].astype(np.int32)),
                                        np.ascontiguousarray(out["depths"].astype(np.uint8)),
                                        out["candidates_blob"], out["candidates_offsets"],
                                        np.ascontiguousarray(out["candidate_frequency"].astype(np.uint8)),
                                        np.ascontiguousarray(img))
        return True

    def write_summary(self, summary_name, contigs, positions, depths, all_candidates, all_candidate_frequency,
                      all_images, all_base_labels, all_type_label, train_mode):
        if summary_name in self._written:
            return
        self._written.add(summary_name)
        base = '{}/{}/'.format(self._summary_path_, summary_name)
        fh = self.file_handler
        if isinstance(fh, h5.PredictionBuilder):
            # one library call per group (pa_h5_builder_write_variant_summary): what image generation writes -- inference
            # mode, one contig per group, one candidate allele per row
            n = len(positions)
            rows = np.asarray(all_candidates, dtype=object).reshape(n, -1) if n else np.zeros((0, 1), object)
            if train_mode or (n and rows.shape[1] != 1) or len(set(contigs)) > 1:
                raise h5.H5Error("the append-only writer takes inference-mode summaries of one contig with one candidate allele "
                                 "per row; set PEPPER_AMD_H5_BUILDER=0 for anything else")
            img = np.asarray(all_images)
            img = np.ascontiguousarray(img if img.dtype == np.int8 else wrap_int8(all_images))
            if n == 0:
                img = np.zeros((0, 33, 26), np.int8)
            fh.write_variant_summary(summary_name, str(contigs[0]) if n else "",
                                     np.ascontiguousarray(np.asarray(positions, dtype=np.int64).astype(np.int32)),
                                     np.ascontiguousarray(wrap_uint8(depths)), [str(c) for c in rows[:, 0]] if n else [],
                                     np.ascontiguousarray(wrap_uint8(all_candidate_frequency)).reshape(n), img)
            return
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

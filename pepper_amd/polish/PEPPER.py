"""Python shim with the names of the reference's pybind11 module for the polish summary encoder.

Mirrors `from pepper.build import PEPPER`'s SummaryGenerator
(/root/reference/pepper/modules/headers/pybind_api.h:18-25; call site
/root/reference/pepper/modules/python/AlignmentSummarizer.py:340-347): after
`generate_summary(reads, start, end)` the object exposes `.image` (uint8 [rows,10]),
`.genomic_pos` ([(position, insert_index)]), `.labels`, `.bad_label_positions`.
Encoded by libpepper_amd.so (include/pepper_amd_encoder.h, pa_polish_encoder_*).
"""
import ctypes

import numpy as np

from pepper_amd import _lib
from pepper_amd.variant.PEPPER_VARIANT import (CigarOp, _Pileup, _encoder, flatten_reads,  # noqa: F401
                                               type_read, type_read_flags)


class SummaryGenerator(object):
    def __init__(self, reference_sequence, chromosome_name, ref_start, ref_end, device=0):
        self.reference_sequence = reference_sequence
        self.chromosome_name = chromosome_name
        self.ref_start = int(ref_start)
        self.ref_end = int(ref_end)
        self.device = device
        self.image = np.zeros((0, 10), np.uint8)
        self.genomic_pos = []
        self.labels = []
        self.bad_label_positions = []

    def generate_summary(self, reads, start_pos, end_pos):
        flat = reads if isinstance(reads, dict) else flatten_reads(reads)
        lib, enc = _encoder(self.device)
        ref = self.reference_sequence.encode("latin-1") if isinstance(self.reference_sequence, str) else bytes(self.reference_sequence)
        p = _Pileup(self.ref_start, self.ref_end, ref, len(ref), flat["n_reads"],
                    flat["read_pos"].ctypes.data, flat["read_reverse"].ctypes.data, flat["read_mapq"].ctypes.data,
                    flat["seq_offset"].ctypes.data, flat["seq"].ctypes.data, flat["qual"].ctypes.data,
                    flat["cigar_offset"].ctypes.data, flat["cigar_op"].ctypes.data, flat["cigar_len"].ctypes.data)
        n = ctypes.c_int64()
        _lib.check(lib.pa_polish_encoder_generate_summary(enc, ctypes.cast(ctypes.pointer(p), ctypes.c_void_p),
                                                          int(start_pos), int(end_pos), ctypes.byref(n)))
        rows = n.value
        image = np.zeros((rows, 10), np.uint8)
        pos = np.zeros((rows, 2), np.int64)
        _lib.check(lib.pa_polish_encoder_get_results(enc, image.ctypes.data, pos.ctypes.data))
        self.image = image
        self.positions_array = pos
        self.genomic_pos = [tuple(x) for x in pos.tolist()]

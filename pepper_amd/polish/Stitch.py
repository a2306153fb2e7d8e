"""Position-keyed merge of the predicted labels into one sequence per contig.

replaces: /root/reference/pepper/modules/python/Stitch.py
    small_chunk_stitch        :36-94    {(position, insert index): label}, later chunks overwrite
    create_consensus_sequence :97-128   pieces ordered by their first position and concatenated
Vectorised with numpy: per piece the (position, index, label) triples of all its 1000-row chunks
are concatenated in the reference's iteration order (chunk ids sorted as strings), filtered
(position/index >= 0; for regions not starting at 0 positions <= start + 2 * MIN_IMAGE_OVERLAP are
overlap and dropped), then the last write of every key wins and keys come out sorted.
Pinned: tests/golden/polish_stitch_ref.fa is the reference's own output on the same prediction arrays.
"""
import concurrent.futures
import os

import numpy as np

from pepper_amd import h5
from pepper_amd.polish.Options import ImageSizeOptions

label_decoder = {1: 'A', 2: 'C', 3: 'G', 4: 'T', 0: ''}
_DECODE = np.frombuffer(b"\0ACGT", dtype=np.uint8)
MIN_SEQUENCE_REQUIRED_FOR_MULTITHREADING = 2


def chunks(file_names, threads):
    return [file_names[i:i + threads] for i in range(0, len(file_names), threads)]


def small_chunk_stitch(contig, small_chunk_keys):
    """One piece of the consensus.  The merge runs inside the I/O library (pa_h5_stitch_polish_regions: chunk rows straight
    from the mapped prediction files, each chunk merged into the tail of the piece; 3 k -> see DESIGN.md chunks/s);
    PEPPER_AMD_STITCH_NUMPY=1 keeps the numpy form below, which tests hold it to."""
    if os.environ.get("PEPPER_AMD_STITCH_NUMPY") == "1":
        return small_chunk_stitch_numpy(contig, small_chunk_keys)
    buffer_positions = ImageSizeOptions.MIN_IMAGE_OVERLAP * 2
    open_files, order = {}, []
    try:
        which, paths, starts = [], [], []
        for file_name, contig_name, _st, _end in small_chunk_keys:
            if file_name not in open_files:
                open_files[file_name] = len(order)
                order.append(h5.File(file_name, 'r'))
            which.append(open_files[file_name])
            paths.append('predictions/' + contig + '/' + contig_name + '-' + str(_st) + '-' + str(_end))
            starts.append(_st)
        return h5.stitch_polish_regions(order, which, paths, starts, buffer_positions)
    finally:
        for f in order:
            f.close()


def small_chunk_stitch_numpy(contig, small_chunk_keys):
    buffer_positions = ImageSizeOptions.MIN_IMAGE_OVERLAP * 2
    pos_parts, idx_parts, base_parts = [], [], []
    open_files = {}                       # each prediction file is opened once per call, not once per region
    try:
        for file_name, contig_name, _st, _end in small_chunk_keys:
            chunk_name = contig_name + '-' + str(_st) + '-' + str(_end)
            prefix = 'predictions/' + contig + '/' + chunk_name
            hdf5_file = open_files.get(file_name)
            if hdf5_file is None:
                hdf5_file = open_files[file_name] = h5.File(file_name, 'r')
            # every chunk of the region in one library call (chunk ids in string order, as sorted() gives them)
            try:
                positions, indices, bases = hdf5_file.read_polish_prediction_region(prefix, ImageSizeOptions.SEQ_LENGTH)
                positions, indices, bases = positions.reshape(-1), indices.reshape(-1), bases.reshape(-1).astype(np.int64)
            except h5.H5Error:                  # chunks of another length (not written by this pipeline): one by one
                parts = [[], [], []]
                for chunk in sorted(set(hdf5_file.keys(prefix)) - {'contig_start', 'contig_end'}):
                    for k, name in enumerate(('position', 'index', 'bases')):
                        parts[k].append(np.asarray(hdf5_file[prefix + '/' + chunk + '/' + name], dtype=np.int64).reshape(-1))
                positions, indices, bases = (np.concatenate(p) if p else np.zeros(0, np.int64) for p in parts)
            keep = (indices >= 0) & (positions >= 0)
            if _st > 0:
                keep &= positions > _st + buffer_positions
            pos_parts.append(positions[keep])
            idx_parts.append(indices[keep])
            base_parts.append(bases[keep])
    finally:
        for f in open_files.values():
            f.close()
    if not pos_parts:
        return -1, -1, ''
    positions = np.concatenate(pos_parts)
    if positions.size == 0:
        return -1, -1, ''
    indices = np.concatenate(idx_parts)
    bases = np.concatenate(base_parts)
    # stable sort on (position, index): the last element of every run is the last write
    order = np.lexsort((indices, positions))
    positions, indices, bases = positions[order], indices[order], bases[order]
    last = np.ones(positions.size, dtype=bool)
    last[:-1] = (positions[1:] != positions[:-1]) | (indices[1:] != indices[:-1])
    labels = bases[last]
    if labels.size and (labels.min() < 0 or labels.max() > 4):
        raise KeyError(int(labels[(labels < 0) | (labels > 4)][0]))      # label_decoder[...] in the reference
    letters = _DECODE[labels]
    sequence = letters[letters != 0].tobytes().decode()
    return int(positions[0]), int(positions[-1]), sequence


def create_consensus_sequence(contig, sequence_chunk_keys, threads):
    key_list = sorted(((file_name, contig, int(contig_start), int(contig_end))
                       for file_name, _, contig_start, contig_end in sorted(sequence_chunk_keys, key=lambda e: e[1])),
                      key=lambda e: (e[2], e[3]))
    file_chunks = chunks(key_list, max(MIN_SEQUENCE_REQUIRED_FOR_MULTITHREADING, int(len(key_list) / max(1, threads)) + 1))
    if threads <= 1 or len(file_chunks) <= 1:
        results = [small_chunk_stitch(contig, chunk) for chunk in file_chunks]
    else:
        with concurrent.futures.ProcessPoolExecutor(max_workers=threads) as executor:
            results = [f.result() for f in [executor.submit(small_chunk_stitch, contig, chunk) for chunk in file_chunks]]
    pieces = sorted((r for r in results if r[0] != -1 and r[1] != -1), key=lambda e: (e[0], e[1]))
    return ''.join(sequence for _, _, sequence in pieces)

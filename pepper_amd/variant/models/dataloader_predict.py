"""Inference dataset over images HDF5 files.

Mirrors /root/reference/pepper_variant/modules/python/models/dataloader_predict.py:13-111
(get_file_paths_from_directory, SequenceDataset, my_collate): a file's summaries are loaded whole
(as the reference does), `dataset[i]` returns (contig, position, depth, candidate,
candidate_frequency, image) and `my_collate` builds the same batch lists.  Internally the groups
are kept as bulk numpy arrays (one int8 [N,33,26] block) so `batches()` can hand the device
contiguous packed windows instead of per-item Python objects.
"""
from os import listdir
from os.path import isfile, join

import numpy as np

from pepper_amd import h5


def get_file_paths_from_directory(directory_path):
    """All files whose name ends in 'hdf5' (RunInference.py:12-21 / dataloader_predict.py:13-21)."""
    return [join(directory_path, file) for file in listdir(directory_path)
            if isfile(join(directory_path, file)) and file[-4:] == 'hdf5']


class SequenceDataset(object):
    def __init__(self, image_directory, input_file=None, summary_names=None, image_alloc=None):
        """image_alloc(n, window, features) -> int8 array [n, window, features] to read the images into (e.g. a reusable
        pinned staging buffer); default: a fresh numpy array.  Every group's image block is read straight into its slice."""
        input_files = get_file_paths_from_directory(image_directory) if input_file is None else [input_file]
        contigs, positions, depths, candidates, freqs = [], [], [], [], []
        self._all_candidates = None
        todo = []                                    # (path, group base, rows)
        shape = None
        for path in input_files:
            with h5.File(path, 'r') as f:
                if 'summaries' not in f:
                    continue
                names = f.keys('summaries') if summary_names is None else summary_names
                for name in names:
                    base = 'summaries/' + name + '/'
                    dims = f.info(base + 'images')[0]
                    if dims[0] == 0:
                        continue
                    if shape is None:
                        shape = tuple(dims[1:])
                    elif tuple(dims[1:]) != shape:
                        raise ValueError("image shapes differ between groups: %r vs %r" % (tuple(dims[1:]), shape))
                    todo.append((path, base, int(dims[0])))
        images = None
        if todo:
            total = sum(rows for _, _, rows in todo)
            images = (image_alloc(total, *shape) if image_alloc is not None else np.empty((total,) + shape, np.int8))
            at, current, f = 0, None, None
            try:
                for path, base, rows in todo:
                    if path != current:
                        if f is not None:
                            f.close()
                        f, current = h5.File(path, 'r'), path
                    f.read_into(base + 'images', images[at:at + rows])
                    at += rows
                    contigs.append(f[base + 'contigs'])
                    positions.append(f[base + 'positions'])
                    depths.append(f[base + 'depths'])
                    candidates.append(f.read_strings_raw(base + 'candidates'))
                    freqs.append(f[base + 'candidate_frequency'])
            finally:
                if f is not None:
                    f.close()
        if images is not None:
            width = max(c.dtype.itemsize for c in contigs)
            self.all_contigs = np.concatenate([c.astype(f'S{width}') for c in contigs])
            self.all_positions = np.concatenate(positions)
            self.all_depths = np.concatenate(depths)
            # candidate strings stay one NUL-separated byte block + start offsets; the object array of the reference's
            # loader is built only if somebody asks for it (`all_candidates`, `dataset[i]`)
            self.candidate_blob = np.frombuffer(b"".join(candidates) + b"\0", np.uint8)
            ends = np.flatnonzero(self.candidate_blob[:-1] == 0)
            self.candidate_offsets = np.concatenate(([0], ends[:-1] + 1)).astype(np.int64) if len(ends) else np.zeros(0, np.int64)
            self.all_candidate_frequency = np.concatenate(freqs)
            self.all_images = images
        else:
            self.all_contigs = np.zeros((0,), 'S1')
            self.all_positions = np.zeros((0,), np.int32)
            self.all_depths = np.zeros((0,), np.uint8)
            self.candidate_blob = np.zeros(1, np.uint8)
            self.candidate_offsets = np.zeros(0, np.int64)
            self.all_candidate_frequency = np.zeros((0, 1), np.uint8)
            self.all_images = np.zeros((0, 33, 26), np.int8)

    @property
    def all_candidates(self):
        if self._all_candidates is None:
            raw = self.candidate_blob.tobytes()
            parts = raw.split(b"\0")[:len(self.candidate_offsets)]
            arr = np.empty((len(parts), 1), dtype=object)
            arr[:, 0] = [p.decode("utf-8") for p in parts]
            self._all_candidates = arr
        return self._all_candidates

    @staticmethod
    def my_collate(batch):
        contig = [item[0] for item in batch]
        position = [item[1] for item in batch]
        depth = [item[2] for item in batch]
        candidate = [item[3] for item in batch]
        candidate_frequency = [item[4] for item in batch]
        import torch        # only here: the lane pipeline's reader processes import this module and must stay torch-free
        image = torch.FloatTensor(np.array([item[5] for item in batch]))
        return [contig, position, depth, candidate, candidate_frequency, image]

    def __getitem__(self, index):
        return (self.all_contigs[index].decode('UTF-8'), self.all_positions[index], self.all_depths[index],
                self.all_candidates[index], self.all_candidate_frequency[index], self.all_images[index])

    def __len__(self):
        return len(self.all_images)

    def batches(self, batch_size):
        """Same batches a DataLoader(shuffle=False, collate_fn=my_collate) yields, but as array
        slices: (contigs 'S'[b], positions, depths, candidates [b,1], frequencies [b,1], images int8)."""
        for s in range(0, len(self), batch_size):
            e = min(len(self), s + batch_size)
            yield (self.all_contigs[s:e], self.all_positions[s:e], self.all_depths[s:e],
                   self.all_candidates[s:e], self.all_candidate_frequency[s:e], self.all_images[s:e])

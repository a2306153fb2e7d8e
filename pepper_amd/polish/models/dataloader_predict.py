"""Polish inference dataset.  Mirrors /root/reference/pepper/modules/python/models/dataloader_predict.py:9-64:
files end in 'hdf'; dataset[i] = (contig, contig_start, contig_end, chunk_id, image, position, index)
of one summaries/<name> group.  `batches()` additionally yields stacked uint8 [B,1000,10] blocks."""
from os import listdir
from os.path import isfile, join

import numpy as np

from pepper_amd import h5


def get_file_paths_from_directory(directory_path):
    return [join(directory_path, file) for file in listdir(directory_path)
            if isfile(join(directory_path, file)) and file[-3:] == 'hdf']


class SequenceDataset(object):
    def __init__(self, image_directory, file_list=None):
        hdf_files = file_list if file_list is not None else get_file_paths_from_directory(image_directory)
        self.all_images = []
        for path in hdf_files:
            with h5.File(path, 'r') as f:
                if 'summaries' in f:
                    for name in f.keys('summaries'):
                        self.all_images.append((path, name))
        self._open = {}

    def _file(self, path):
        if path not in self._open:
            self._open[path] = h5.File(path, 'r')
        return self._open[path]

    def close(self):
        for f in self._open.values():
            f.close()
        self._open = {}

    def __getitem__(self, index):
        path, name = self.all_images[index]
        f = self._file(path)
        base = 'summaries/' + name + '/'
        return (f[base + 'contig'], f[base + 'region_start'], f[base + 'region_end'], f[base + 'chunk_id'],
                f[base + 'image'], f[base + 'position'], f[base + 'index'])

    def __len__(self):
        return len(self.all_images)

    def batches(self, batch_size):
        for s in range(0, len(self), batch_size):
            items = [self[i] for i in range(s, min(len(self), s + batch_size))]
            yield ([it[0] for it in items], [it[1] for it in items], [it[2] for it in items],
                   [it[3] for it in items], np.stack([it[4] for it in items]).astype(np.uint8),
                   [it[5] for it in items], [it[6] for it in items])

    def blocks(self, batch_size, seq_len, features):
        """The same stacked blocks as batches(), read by one library call per (file, block): contigs come back as a
        numpy 'S' array, region bounds / chunk ids / positions / indices as int64 arrays."""
        s = 0
        while s < len(self):
            path = self.all_images[s][0]
            e = s
            while e < len(self) and e - s < batch_size and self.all_images[e][0] == path:
                e += 1
            yield self._file(path).read_polish_chunks([name for _, name in self.all_images[s:e]], seq_len, features)
            s = e


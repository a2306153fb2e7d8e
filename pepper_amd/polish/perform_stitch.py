"""Prediction HDF5 directory -> polished FASTA.

replaces: /root/reference/pepper/modules/python/perform_stitch.py:44-84 (`perform_stitch`): every
`*hdf` of the directory, contigs in natural order, one `>contig` record per non-empty consensus,
written to `<output_path>_pepper_polished.fa`.  (The reference's 5 s sleep is not reproduced.)
"""
import re
import sys
from datetime import datetime
from os import listdir
from os.path import isfile, join
from pathlib import Path

from pepper_amd import h5
from pepper_amd.polish.Stitch import create_consensus_sequence


def natural_key(string_):
    return [int(s) if s.isdigit() else s for s in re.split(r'(\d+)', string_)]


def get_file_paths_from_directory(directory_path):
    return [join(directory_path, file) for file in sorted(listdir(directory_path))
            if isfile(join(directory_path, file)) and file[-3:] == 'hdf']


def _log(message):
    sys.stderr.write("[" + str(datetime.now().strftime('%m-%d-%Y %H:%M:%S')) + "] INFO: " + message + "\n")


def perform_stitch(hdf_file_path, output_path, threads):
    all_prediction_files = get_file_paths_from_directory(hdf_file_path)
    all_contigs = set()
    for prediction_file in all_prediction_files:
        with h5.File(prediction_file, 'r') as hdf5_file:
            if 'predictions' in hdf5_file.keys():
                all_contigs.update(hdf5_file.keys('predictions'))

    output_path = output_path + '_pepper_polished.fa'
    Path(output_path).resolve().parents[0].mkdir(parents=True, exist_ok=True)
    with open(output_path, 'w') as consensus_fasta_file:
        for contig in sorted(all_contigs, key=natural_key):
            _log("PROCESSING CONTIG: " + contig)
            all_chunk_keys = []
            for prediction_file in all_prediction_files:
                with h5.File(prediction_file, 'r') as hdf5_file:
                    if 'predictions' not in hdf5_file.keys() or contig not in hdf5_file.keys('predictions'):
                        continue
                    # every region group with its contig_start / contig_end in one library call (names in sorted order)
                    all_chunk_keys.extend((prediction_file, name, start, end) for name, start, end in hdf5_file.list_polish_regions(contig))
            consensus_sequence = create_consensus_sequence(contig, all_chunk_keys, threads)
            _log("FINISHED PROCESSING " + contig + ", POLISHED SEQUENCE LENGTH: " + str(len(consensus_sequence)) + ".")
            if consensus_sequence is not None and len(consensus_sequence) > 0:
                consensus_fasta_file.write('>' + contig + "\n")
                consensus_fasta_file.write(consensus_sequence + "\n")
    return output_path

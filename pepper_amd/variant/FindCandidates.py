"""predictions HDF5 directory -> candidate VCFs.

replaces: /root/reference/pepper_variant/modules/python/FindCandidates.py
    get_file_paths_from_directory :134-142, candidate_finder :145-186, process_candidates :189-198
(`candidates_to_variants` / `simplify_variants` / `natural_key` of that file are not called by the
live pipeline and are not reproduced.)  Differences: files are listed in sorted order so the
per-site record order is deterministic (the reference uses listdir order and as_completed).
"""
import os
import sys
import time
from datetime import datetime
from os import listdir
from os.path import isfile, join

from pepper_amd import h5
from pepper_amd.variant.CandidateFinder import find_candidates
from pepper_amd.variant.ImageGenerationUI import ImageGenerationUtils
from pepper_amd.variant.VcfWriter import VCFWriter


def get_file_paths_from_directory(directory_path):
    """All *hdf files of a directory (every rank's pepper_prediction_<r>.hdf)."""
    return [join(directory_path, file) for file in sorted(listdir(directory_path))
            if isfile(join(directory_path, file)) and file[-3:] == 'hdf']


def _log(message):
    sys.stderr.write("[" + str(datetime.now().strftime('%m-%d-%Y %H:%M:%S')) + "] INFO: " + message + "\n")


def candidate_finder(options, input_dir, output_path, precomputed=None):
    """precomputed: {(prediction file, batch key): FastCandidates segment} of batches whose selection was done while they were
    written (the fused call_variant); the others are read from the files as always."""
    all_prediction_pair = []
    for prediction_file in get_file_paths_from_directory(input_dir):
        with h5.File(prediction_file, 'r') as hdf5_file:
            if 'predictions' in hdf5_file.keys():
                for batch in hdf5_file.keys('predictions'):
                    all_prediction_pair.append((prediction_file, batch))

    local_start_time = time.time()
    _log("STARTING CANDIDATE FINDING.")
    factory = getattr(options, "fasta_handler_factory", None)

    def writer():
        return VCFWriter([], options.fasta, options.sample_name, output_path, "PEPPER_VARIANT_FULL",
                         "PEPPER_VARIANT_OUTPUT_PEPPER", "PEPPER_VARIANT_OUTPUT_VARIANT_CALLING",
                         fasta_handler=factory(options.fasta) if factory is not None else None)
    if os.environ.get("PEPPER_AMD_CANDIDATES_TUPLES") == "1":
        # the reference-shaped path: one tuple per selected allele, dictionaries of sites, one record at a time
        contigs, selected_candidates_phasing, selected_candidates_variant_calling = find_candidates(options, input_dir, all_prediction_pair)
        end_time = time.time()
        vcf_file_full = writer()
        totals = vcf_file_full.write_vcf_records(selected_candidates_variant_calling, options)
    else:
        # the same rules column-wise (FastCandidates.py; held to the tuple path file by file in tests/test_candidate_finder.py)
        from pepper_amd.variant import FastCandidates
        vcf_file_full = writer()
        contigs, totals = FastCandidates.process(options, all_prediction_pair, vcf_file_full, precomputed=precomputed)
        end_time = time.time()
    vcf_file_full.close()
    total_variants, total_pepper, total_variant_calling, total_variant_calling_snp, total_variant_calling_indel = totals
    _log("FINISHED PROCESSING, TOTAL CANDIDATES FOUND: " + str(total_variants))
    _log("FINISHED PROCESSING, TOTAL VARIANTS IN PEPPER: " + str(total_pepper))
    _log("FINISHED PROCESSING, TOTAL VARIANTS SELECTED FOR RE-GENOTYPING: " + str(total_variant_calling))
    _log("FINISHED PROCESSING, TOTAL SNP VARIANTS SELECTED FOR RE-GENOTYPING: " + str(total_variant_calling_snp))
    _log("FINISHED PROCESSING, TOTAL INDEL VARIANTS SELECTED FOR RE-GENOTYPING: " + str(total_variant_calling_indel))
    elapsed = end_time - local_start_time
    _log("TOTAL TIME SPENT ON CANDIDATE FINDING: " + str(int(elapsed / 60)) + " Min " + str(int(elapsed) % 60) + " Sec")
    return totals


def process_candidates(options, input_dir, output_dir, precomputed=None):
    output_dir = ImageGenerationUtils.handle_output_directory(output_dir)
    return candidate_finder(options, input_dir, output_dir, precomputed=precomputed)

"""Image generation driver (pileup -> images HDF5), inference mode.

Mirrors /root/reference/pepper_variant/modules/python/ImageGenerationUI.py:18-71 (ImageGenerator),
:79-91 (handle_output_directory), :93-96 (natural_key), :141-176 (region string parsing),
:191-274 (generate_image_and_save_to_file), :277-345 (generate_images): the genome is cut into
`options.region_size` intervals, interval i belongs to worker i % options.threads, every worker
writes one `pepper_variants_images_thread_<id>_<time>.hdf5` with one summaries/<chr>_<start>_<end>
group per interval that produced candidates.  BAM/FASTA access goes through `options.bam_handler_factory`
/ `options.fasta_handler_factory` (callables path -> handler object with the reference's handler
methods); reading BAM/FASTA files needs htslib and is the "next" row N3, so without factories this
raises.  Workers run sequentially in-process (one GPU encoder), not in a ProcessPoolExecutor.
"""
import os
import re
import sys
import time
from datetime import datetime

import numpy as np

from pepper_amd.variant.AlignmentSummarizer import AlignmentSummarizer
from pepper_amd.variant.DataStore import DataStore


def _log(msg):
    sys.stderr.write("[" + datetime.now().strftime('%m-%d-%Y %H:%M:%S') + "] " + msg + "\n")
    sys.stderr.flush()


def _handlers(options, bam_path, fasta_path):
    """Injected factories win (tests, other readers); otherwise the package's own BAM (zlib, bamio.cpp) and
    indexed-FASTA readers -- htslib is not needed."""
    bf = getattr(options, "bam_handler_factory", None)
    ff = getattr(options, "fasta_handler_factory", None)
    if bf is None:
        from pepper_amd.variant.bam import BAM_handler as bf
    if ff is None:
        from pepper_amd.variant.fasta import FASTA_handler as ff
    return bf(bam_path), ff(fasta_path)


class ImageGenerator:
    def __init__(self, chromosome_name, bam_file_path, fasta_file_path, options=None):
        self.bam_handler, self.fasta_handler = _handlers(options, bam_file_path, fasta_file_path)
        self.chromosome_name = chromosome_name

    def generate_summary(self, options, start_position, end_position, bed_list, thread_id, as_arrays=False):
        if getattr(options, "use_hp_info", False):
            raise NotImplementedError("--use_hp_info image generation feeds a predictor that is non-functional "
                                      "in the reference at this commit (SURVEY.md 2.1 V13)")
        summarizer = AlignmentSummarizer(self.bam_handler, self.fasta_handler, self.chromosome_name,
                                         start_position, end_position)
        return summarizer.create_summary(options, bed_list, thread_id, as_arrays=as_arrays)

    def prepare(self, options, start_position, end_position):
        """The part of generate_summary in front of the encoder call (reads + reference of the interval), for the callers
        that encode several intervals per call."""
        if getattr(options, "use_hp_info", False):
            raise NotImplementedError("--use_hp_info image generation feeds a predictor that is non-functional "
                                      "in the reference at this commit (SURVEY.md 2.1 V13)")
        return AlignmentSummarizer(self.bam_handler, self.fasta_handler, self.chromosome_name, start_position,
                                   end_position).prepare(options)


class ImageGenerationUtils:
    @staticmethod
    def handle_output_directory(output_dir):
        if not os.path.exists(output_dir):
            os.makedirs(output_dir, exist_ok=True)
        if output_dir[-1] != '/':
            output_dir += '/'
        return output_dir

    @staticmethod
    def natural_key(string_):
        return [int(s) if s.isdigit() else s for s in re.split(r'(\d+)', string_)]

    @staticmethod
    def get_chromosome_list(chromosome_names, fasta_handler, bam_handler):
        """'chr20', 'chr20:1000-2000', 'chr1-3' or a comma list -> [(name, region or None)];
        empty -> contigs common to BAM and FASTA in natural order."""
        if not chromosome_names:
            common = sorted(set(fasta_handler.get_chromosome_names()) & set(bam_handler.get_chromosome_sequence_names()),
                            key=ImageGenerationUtils.natural_key)
            if not common:
                raise RuntimeError("ERROR: NO COMMON CONTIGS FOUND BETWEEN THE BAM FILE AND THE FASTA FILE.")
            return [(c, None) for c in common]
        out = []
        for name in [n.strip() for n in chromosome_names.strip().split(',')]:
            region = None
            if ':' in name:
                parts = name.split(':')
                if len(parts) != 2:
                    raise ValueError("ERROR: --region INVALID value.")
                name, region = parts
                region = [int(p) for p in region.strip().split('-')]
                if len(region) != 2 or not region[0] <= region[1]:
                    raise ValueError("ERROR: --region INVALID value.")
            range_split = name.split('-')
            if len(range_split) > 1:
                prefix = ''
                for ch in name:
                    if ch.isdigit():
                        break
                    prefix += ch
                ints = sorted(int(''.join(c for c in item if c.isdigit())) for item in range_split)
                for k in range(ints[0], ints[-1] + 1):
                    out.append((prefix + str(k), region))
            else:
                out.append((name, region))
        return out

    @staticmethod
    def split_intervals(chr_list, fasta_handler, region_size):
        """generate_images:289-317: [(chr, pos_start, pos_end)] of at most region_size bases."""
        all_intervals, total_bases = [], 0
        for chr_name, region in chr_list:
            last = fasta_handler.get_chromosome_sequence_length(chr_name) - 1
            if not region:
                interval_start, interval_end = 0, last
            else:
                interval_start, interval_end = max(0, region[0]), min(region[1], last)
            for pos in range(interval_start, interval_end, region_size):
                pos_start, pos_end = max(interval_start, pos), min(interval_end, pos + region_size)
                all_intervals.append((chr_name, pos_start, pos_end))
                total_bases += pos_end - pos_start
        return all_intervals, total_bases

    @staticmethod
    def generate_image_and_save_to_file(options, all_intervals, bed_list, process_id):
        timestr = time.strftime("%m%d%Y_%H%M%S")
        file_name = options.image_output_directory + "pepper_variants_images_thread_" + str(process_id) + "_" + str(timestr) + ".hdf5"
        # intervals are encoded ENCODER_BATCH at a time: the reads of a group are fetched (BAM reader, outside the GIL), then one
        # encoder call covers the group; summaries are written per interval under the reference's group names.  A worker takes
        # whole groups of CONSECUTIVE intervals (the reference deals single intervals round robin, ImageGenerationUI.py:262-274;
        # which worker's file an interval lands in is not read by anything downstream): the reads of an interval start up to
        # a read length + 16 kb (the BAM index's window) in front of it, so consecutive fetches through one handle find most of
        # their BGZF blocks already inflated in the handle's cache -- the BAM reader is 93 % of this loop's time.
        batch = max(1, int(getattr(options, "encoder_batch", 0) or os.environ.get("PEPPER_AMD_ENCODER_BATCH", 16)))
        run = max(1, min(batch, -(-len(all_intervals) // max(1, options.threads))))      # (small jobs: every worker gets some)
        intervals = [r for i, r in enumerate(all_intervals) if (i // run) % options.threads == process_id]
        if process_id == 0:
            _log("INFO: STARTING PROCESS: " + str(process_id) + " FOR " + str(len(intervals)) + " INTERVALS")
        from pepper_amd.variant.AlignmentSummarizer import create_summaries
        generators = {}
        with DataStore(file_name, 'w') as output_hdf_file:
            for g0 in range(0, len(intervals), batch):
                group = intervals[g0:g0 + batch]
                prepared = []
                for chr_name, _start, _end in group:
                    if chr_name not in generators:
                        generators.clear()               # one contig's handles at a time per worker
                        generators[chr_name] = ImageGenerator(chr_name, options.bam, options.fasta, options)
                    prepared.append(generators[chr_name].prepare(options, _start, _end))
                for (chr_name, _start, _end), out in zip(group, create_summaries(prepared)):
                    if out is None:
                        continue
                    n = len(out["candidates"])
                    summary_name = chr_name + "_" + str(_start) + "_" + str(_end)
                    output_hdf_file.write_summary(summary_name, [chr_name] * n, out["positions"], out["depths"],
                                                  np.array(out["candidates"], dtype=object).reshape(n, 1),
                                                  out["candidate_frequency"].reshape(n, 1), out["images"],
                                                  [0] * n, [0] * n, False)
        return process_id

    @staticmethod
    def generate_images(options):
        options.image_output_directory = ImageGenerationUtils.handle_output_directory(
            os.path.abspath(options.image_output_directory))
        start_time = time.time()
        bam_handler, fasta_handler = _handlers(options, options.bam, options.fasta)
        chr_list = ImageGenerationUtils.get_chromosome_list(options.region, fasta_handler, bam_handler)
        all_intervals, total_bases = ImageGenerationUtils.split_intervals(chr_list, fasta_handler, options.region_size)
        _log("INFO: TOTAL CONTIGS: " + str(len(chr_list)) + " TOTAL INTERVALS: " + str(len(all_intervals))
             + " TOTAL BASES: " + str(total_bases))
        # the reference forks options.threads processes; here they are threads of one process sharing the GPU: the
        # BAM reader, the encoder's host pass and libhdf5 run outside the GIL, each worker has its own BAM / FASTA
        # handles, encoder workspace and output file (interval i goes to worker i % threads, as in the reference)
        if options.threads <= 1:
            ImageGenerationUtils.generate_image_and_save_to_file(options, all_intervals, None, 0)
        else:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=options.threads) as pool:
                futures = [pool.submit(ImageGenerationUtils.generate_image_and_save_to_file, options, all_intervals, None,
                                       process_id) for process_id in range(options.threads)]
                for fut in futures:
                    fut.result()
        _log("INFO: FINISHED IMAGE GENERATION")
        secs = int(time.time() - start_time)
        _log("INFO: TOTAL ELAPSED TIME FOR GENERATING IMAGES: " + str(secs // 60) + " Min " + str(secs % 60) + " Sec")

"""Polish image generation: contigs -> 1 kb intervals -> re-aligned, summarised, chunked images -> HDF5.

Mirrors /root/reference/pepper/modules/python/ImageGenerationUI.py (UserInterfaceView :13-60,
UserInterfaceSupport.handle_output_directory :67-81, get_chromosome_list :88-166, single_worker / image_generator
:169-221, chromosome_level_parallelization :224-284): same interval grid (1000-base steps widened by
MIN_IMAGE_OVERLAP on both sides), same interval striding over the workers, same file and group names.  The workers
are threads of one process (each with its own BAM / FASTA handles, its own re-aligner and encoder workspaces on the
GPU); errors raise instead of being printed and swallowed; train_mode is outside the inference path.
"""
import os
import re
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from datetime import datetime

from pepper_amd.polish.AlignmentSummarizer import AlignmentSummarizer
from pepper_amd.polish.DataStore import DataStore
from pepper_amd.polish.Options import ImageSizeOptions
from pepper_amd.variant.bam import BAM_handler
from pepper_amd.variant.fasta import FASTA_handler


def _log(message):
    sys.stderr.write("[" + datetime.now().strftime('%m-%d-%Y %H:%M:%S') + "] " + message + "\n")
    sys.stderr.flush()


class UserInterfaceView:
    def __init__(self, chromosome_name, bam_file_path, draft_file_path, truth_bam, train_mode):
        if train_mode:
            raise NotImplementedError("train_mode image generation is outside the inference path")
        self.bam_path = bam_file_path
        self.fasta_path = draft_file_path
        self.bam_handler = BAM_handler(bam_file_path)
        self.fasta_handler = FASTA_handler(draft_file_path)
        self.train_mode = train_mode
        self.downsample_rate = 1.0
        self.truth_bam_handler = None
        self.chromosome_name = chromosome_name

    def parse_region(self, start_position, end_position, downsample_rate):
        summarizer = AlignmentSummarizer(self.bam_handler, self.fasta_handler, self.chromosome_name, start_position,
                                         end_position)
        return summarizer.create_summary(self.truth_bam_handler, self.train_mode, downsample_rate)

    def parse_regions(self, bounds, downsample_rate):
        """Several regions of this contig with one re-alignment call on the GPU; per-region results in order."""
        summarizers = [AlignmentSummarizer(self.bam_handler, self.fasta_handler, self.chromosome_name, a, b) for a, b in bounds]
        return AlignmentSummarizer.create_summaries(summarizers)


class UserInterfaceSupport:
    # intervals whose reads share one re-alignment call on the GPU (PEPPER_AMD_POLISH_REGIONS_PER_CALL)
    REGIONS_PER_CALL = int(os.environ.get("PEPPER_AMD_POLISH_REGIONS_PER_CALL", 32))

    @staticmethod
    def handle_output_directory(output_directory):
        if output_directory[-1] != "/":
            output_directory += "/"
        if not os.path.exists(output_directory):
            os.makedirs(output_directory)
        return output_directory

    @staticmethod
    def natural_key(string_):
        return [int(s) if s.isdigit() else s for s in re.split(r'(\d+)', string_)]

    @staticmethod
    def get_chromosome_list(chromosome_names, ref_file, bam_file, region_bed):
        """`--region` grammar of the reference: empty -> contigs common to BAM and FASTA (natural order);
        a BED file; or comma-separated `name`, `name:start-end`, `chr3-5` (a numeric range of names)."""
        if not chromosome_names and not region_bed:
            bam_contigs = BAM_handler(bam_file).get_chromosome_sequence_names()
            fasta_contigs = FASTA_handler(ref_file).get_chromosome_names()
            common = sorted(set(fasta_contigs) & set(bam_contigs), key=UserInterfaceSupport.natural_key)
            if not common:
                raise RuntimeError("NO COMMON CONTIGS FOUND BETWEEN THE BAM FILE AND THE FASTA FILE.")
            _log("INFO: COMMON CONTIGS FOUND: " + str(common))
            return [(name, None) for name in common]
        if region_bed:
            out = []
            with open(region_bed) as fp:
                for line in fp:
                    fields = line.rstrip().split('\t')
                    if len(fields) < 3:
                        continue
                    out.append((fields[0], sorted([int(fields[1]), int(fields[2])])))
            return out
        out = []
        for name in [n.strip() for n in chromosome_names.strip().split(',')]:
            region = None
            if ':' in name:
                parts = name.strip().split(':')
                if len(parts) != 2:
                    raise ValueError("--region INVALID value.")
                name, region = parts
                region = [int(pos) for pos in region.strip().split('-')]
                if len(region) != 2 or not region[0] <= region[1]:
                    raise ValueError("--region INVALID value.")
            range_split = name.split('-')
            if len(range_split) > 1:
                prefix = ''
                for ch in name:
                    if ch.isdigit():
                        break
                    prefix += ch
                numbers = sorted(int(''.join(c for c in item if c.isdigit())) for item in range_split)
                for seq in range(numbers[0], numbers[-1] + 1):
                    out.append((prefix + str(seq), region))
            else:
                out.append((name, region))
        return out

    @staticmethod
    def single_worker(args, _start, _end, _views=None):
        chr_name, bam_file, draft_file, truth_bam, train_mode, downsample_rate = args
        key = (chr_name, bam_file, draft_file)
        view = _views.get(key) if _views is not None else None
        if view is None:
            view = UserInterfaceView(chr_name, bam_file, draft_file, truth_bam, train_mode)
            if _views is not None:
                _views.clear()          # one contig's handles at a time per worker
                _views[key] = view
        images, labels, positions, image_chunk_ids = view.parse_region(_start, _end, downsample_rate)
        return images, labels, positions, image_chunk_ids, (chr_name, _start, _end)

    @staticmethod
    def image_generator(args, all_intervals, total_threads, thread_id):
        output_path, bam_file, draft_file, truth_bam, train_mode, downsample_rate = args
        timestr = time.strftime("%m%d%Y_%H%M%S")
        file_name = output_path + "pepper_hp_images_thread_" + str(thread_id) + "_" + str(timestr) + ".hdf"
        # a worker takes runs of CONSECUTIVE intervals (the reference deals single intervals round robin,
        # ImageGenerationUI.py:262-274; which worker's file an interval lands in is not read by anything downstream): the reads
        # of an interval start up to a read length + 16 kb in front of it, and consecutive fetches through one BAM handle find
        # those BGZF blocks already inflated in the handle's cache
        run = max(1, min(UserInterfaceSupport.REGIONS_PER_CALL, -(-len(all_intervals) // max(1, total_threads))))
        intervals = [r for i, r in enumerate(all_intervals) if (i // run) % total_threads == thread_id]
        if thread_id == 0:
            _log("INFO: STARTING THREAD: " + str(thread_id) + " FOR " + str(len(intervals)) + " INTERVALS")
        start_time = time.time()
        views = {}
        with DataStore(file_name, 'w') as output_hdf_file:
            counter = 0
            while counter < len(intervals):
                # up to REGIONS_PER_CALL consecutive intervals of one contig share a re-alignment call
                chr_name = intervals[counter][0]
                block = [intervals[counter]]
                while (len(block) < UserInterfaceSupport.REGIONS_PER_CALL and counter + len(block) < len(intervals)
                       and intervals[counter + len(block)][0] == chr_name):
                    block.append(intervals[counter + len(block)])
                key = (chr_name, bam_file, draft_file)
                if key not in views:
                    views.clear()           # one contig's handles at a time per worker
                    views[key] = UserInterfaceView(chr_name, bam_file, draft_file, truth_bam, train_mode)
                results = views[key].parse_regions([(a, b) for _, a, b in block], downsample_rate)
                for region, (images, labels, positions, chunk_ids) in zip(block, results):
                    if len(images):         # group names <contig>_<start>_<end>_<chunk id>, one library call per region
                        output_hdf_file.write_summaries(region, images, labels, positions, chunk_ids)
                before = counter
                counter += len(block)
                if thread_id == 0 and counter // 10 > before // 10:
                    elapsed = int(time.time() - start_time)
                    _log("INFO: [THREAD " + "{:02d}".format(thread_id) + "] " + str(counter) + "/" + str(len(intervals))
                         + " COMPLETE (" + str(int(100 * counter / len(intervals))) + "%) [ELAPSED TIME: "
                         + str(elapsed // 60) + " Min " + str(elapsed % 60) + " Sec]")
        return thread_id

    @staticmethod
    def make_intervals(chr_list, draft_file):
        max_size = 1000
        fasta_handler = FASTA_handler(draft_file)
        contigs, all_intervals = set(), []
        for chr_name, region in chr_list:
            contigs.add(str(chr_name))
            last = fasta_handler.get_chromosome_sequence_length(str(chr_name)) - 1
            if not region:
                interval_start, interval_end = 0, last
            else:
                interval_start, interval_end = tuple(region)
                interval_start = max(0, interval_start)
                interval_end = min(interval_end, last)
            for pos in range(interval_start, interval_end, max_size):
                pos_start = max(interval_start, pos - ImageSizeOptions.MIN_IMAGE_OVERLAP)
                pos_end = min(interval_end, pos + max_size + ImageSizeOptions.MIN_IMAGE_OVERLAP)
                all_intervals.append((chr_name, pos_start, pos_end))
        return contigs, all_intervals

    @staticmethod
    def chromosome_level_parallelization(chr_list, bam_file, draft_file, truth_bam, output_path, total_threads, train_mode,
                                         downsample_rate=1.0):
        if train_mode:
            raise NotImplementedError("train_mode image generation is outside the inference path")
        contigs, all_intervals = UserInterfaceSupport.make_intervals(chr_list, draft_file)
        _log("INFO: TOTAL CONTIGS: " + str(len(contigs)) + " TOTAL INTERVALS: " + str(len(all_intervals)))
        args = (output_path, bam_file, draft_file, truth_bam, train_mode, downsample_rate)
        if total_threads <= 1:
            UserInterfaceSupport.image_generator(args, all_intervals, 1, 0)
            return
        with ThreadPoolExecutor(max_workers=total_threads) as executor:
            futures = [executor.submit(UserInterfaceSupport.image_generator, args, all_intervals, total_threads, thread_id)
                       for thread_id in range(total_threads)]
            for fut in futures:
                fut.result()            # a worker's exception stops the run (the reference logs it and carries on)

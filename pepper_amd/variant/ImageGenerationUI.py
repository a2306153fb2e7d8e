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


_STATS_LOCK = __import__("threading").Lock()


def _log(msg):
    sys.stderr.write("[" + datetime.now().strftime('%m-%d-%Y %H:%M:%S') + "] " + msg + "\n")
    sys.stderr.flush()


def _on_device(options, device):
    """`options` as the host-clipped form reads them, with .device = this worker's device."""
    if int(getattr(options, "device", 0) or 0) == device:
        return options
    import copy
    clone = copy.copy(options)
    clone.device = device
    return clone


def worker_device(options, process_id):
    """The device of image-generation worker `process_id`: options.image_device_ids ("0,1,...", a list) or -- what call_variant's
    callers already give for the inference step -- options.device_ids, dealt round robin; options.device (default 0) without them.
    The reference's image generation is CPU work (ImageGenerationUI.py:326-339 starts one process per thread); its run_inference
    deals callers over device_ids the same way (RunInference.py:101-116)."""
    ids = getattr(options, "image_device_ids", None)
    if ids in (None, ""):
        ids = getattr(options, "device_ids", None)
    if ids in (None, ""):
        return int(getattr(options, "device", 0) or 0)
    if isinstance(ids, str):
        ids = [int(d) for d in ids.split(",") if d.strip() != ""]
    elif isinstance(ids, int):
        ids = [ids]
    ids = [int(d) for d in ids]
    return ids[process_id % len(ids)] if ids else 0


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
        device = worker_device(options, process_id)
        wopts = _on_device(options, device)          # (what the host-clipped form reads .device from)
        batch = max(1, int(getattr(options, "encoder_batch", 0) or os.environ.get("PEPPER_AMD_ENCODER_BATCH", 16)))
        # a worker takes whole runs of consecutive intervals; large jobs are cut so that every worker gets several runs
        run = max(1, min(batch, -(-len(all_intervals) // max(1, options.threads * 4))))
        intervals = [r for i, r in enumerate(all_intervals) if (i // run) % options.threads == process_id]
        if process_id == 0:
            _log("INFO: STARTING PROCESS: " + str(process_id) + " FOR " + str(len(intervals)) + " INTERVALS")
        from pepper_amd.variant.AlignmentSummarizer import create_summaries
        generators = {}
        stats = getattr(options, "stage_seconds", None)      # a dict the caller wants the stage times of this worker added to
        mine = {}

        def lap(key, t0):
            now = time.perf_counter()
            mine[key] = mine.get(key, 0.0) + now - t0
            return now

        def report():
            if stats is not None:
                with _STATS_LOCK:
                    for key, v in mine.items():
                        stats[key] = stats.get(key, 0.0) + v

        sink = getattr(options, "fused_sink", None)         # call_variant's fused form: predictions straight from the encoder's windows

        def write(output_hdf_file, chr_name, _start, _end, out, probs=None):
            n = len(out["candidates"])
            if sink is not None:
                sink.submit(chr_name, out, probs if probs is not None else sink.forward_host(device, out["images"]))
            summary_name = chr_name + "_" + str(_start) + "_" + str(_end)
            if output_hdf_file.write_summary_packed(summary_name, chr_name, out):
                return
            output_hdf_file.write_summary(summary_name, [chr_name] * n, out["positions"], out["depths"],
                                          np.array(out["candidates"], dtype=object).reshape(n, 1),
                                          out["candidate_frequency"].reshape(n, 1), out["images"],
                                          [0] * n, [0] * n, False)

        def host_clipped(output_hdf_file, group):
            """The form in which the host clips every read to its interval (BAM_handler.get_reads): injected handlers,
            intervals whose reads are sampled down, operations the device-side clip refuses."""
            prepared = []
            for chr_name, _start, _end in group:
                if chr_name not in generators:
                    generators.clear()               # one contig's handles at a time per worker
                    generators[chr_name] = ImageGenerator(chr_name, options.bam, options.fasta, wopts)
                prepared.append(generators[chr_name].prepare(wopts, _start, _end))
            for (chr_name, _start, _end), out in zip(group, create_summaries(prepared)):
                if out is not None:
                    write(output_hdf_file, chr_name, _start, _end, out)

        packed = (getattr(options, "bam_handler_factory", None) is None and getattr(options, "fasta_handler_factory", None) is None
                  and os.environ.get("PEPPER_AMD_PACKED_READS", "1") != "0" and not getattr(options, "train_mode", False))
        if getattr(options, "use_hp_info", False):
            packed = False                           # (host_clipped raises the reference's message for it)
        with DataStore(file_name, 'w') as output_hdf_file:
            if not packed:
                for g0 in range(0, len(intervals), batch):
                    host_clipped(output_hdf_file, intervals[g0:g0 + batch])
                report()
                return process_id
            # The packed form: per group of consecutive intervals ONE call of the BAM reader (inflate, header walk, filters; no
            # clipping, no decoding: the reads cross PCIe as BAM stores them, once per group) and ONE of the encoder (clip +
            # decode + summary + windows on the device); both run outside the GIL, each worker on its own handles.
            from pepper_amd import _lib
            from pepper_amd.variant.AlignmentSummarizer import AlingerOptions, ConsensCandidateFinder
            from pepper_amd.variant.Options import ImageSizeOptions
            from pepper_amd.variant.PEPPER_VARIANT import PackedEncoder
            t_setup = time.perf_counter()
            # (every CPU the process has is inflating BGZF blocks in some worker: the encoder's host part runs on this thread)
            try:
                enc = PackedEncoder.acquire(device, int(os.environ.get("PEPPER_AMD_ARENA_MB", 256)) << 20, host_threads=1)
            except _lib.PepperAmdError:
                # no page-locked arena to be had (memlock / cgroup limit): the host-clipped form needs none
                for g0 in range(0, len(intervals), batch):
                    host_clipped(output_hdf_file, intervals[g0:g0 + batch])
                report()
                return process_id
            bam_handler, fasta_handler = _handlers(options, options.bam, options.fasta)
            lap("setup", t_setup)
            safe = ConsensCandidateFinder.REGION_SAFE_BASES
            params = (options.min_snp_baseq, options.min_indel_baseq, options.snp_frequency, options.insert_frequency,
                      options.delete_frequency, options.min_coverage_threshold, options.snp_candidate_frequency_threshold,
                      options.indel_candidate_frequency_threshold, options.candidate_support_threshold, options.skip_indels)
            device_inflate = os.environ.get("PEPPER_AMD_DEVICE_INFLATE", "1") != "0"
            g0 = 0
            while g0 < len(intervals):
                # ADJACENT intervals of one contig, ascending (the packer walks every record between the first and the last
                # region of a call: a group must not bridge the gap to this worker's next run of intervals)
                g1 = g0 + 1
                while (g1 < len(intervals) and g1 - g0 < batch and intervals[g1][0] == intervals[g0][0]
                       and intervals[g1 - 1][1] <= intervals[g1][1] <= intervals[g1 - 1][2] + 2 * safe
                       and intervals[g1][2] >= intervals[g1 - 1][2]):
                    g1 += 1
                group = intervals[g0:g1]
                chr_name = group[0][0]
                regions = [(max(0, s - safe), e + safe) for _, s, e in group]
                t0 = time.perf_counter()
                # the BGZF members inflated on the device where the BAM has an index (PEPPER_AMD_DEVICE_INFLATE=0: on the host);
                # a batch the device form cannot take (pack_device's docstring) goes through the host packer
                on_device = enc.pack_device(bam_handler, chr_name, [r[0] for r in regions], [r[1] for r in regions],
                                            options.include_supplementary, options.min_mapq, laps=mine) if device_inflate else None
                resident = on_device is not None
                if resident:
                    n_done, region_pairs, counts = on_device
                    t0 = time.perf_counter()
                else:
                    try:
                        n_done, region_pairs, counts = enc.pack(bam_handler, chr_name, [r[0] for r in regions], [r[1] for r in regions],
                                                                options.include_supplementary, options.min_mapq)
                    except Exception as err:
                        if getattr(err, "code", 0) != -7:
                            raise
                        n_done = 0                       # (-7: one interval's reads outgrow the arena)
                    t0 = lap("bam_pack", t0)
                if n_done == 0:
                    host_clipped(output_hdf_file, group[:1])
                    g0 += 1
                    continue
                group, regions = group[:n_done], regions[:n_done]
                per_region = np.diff(region_pairs[:n_done + 1])
                # the reference samples an interval's reads down to min(MAX_READS_IN_REGION, downsample_rate * n)
                # (AlignmentSummarizer.py:192-199) in read order: such intervals take the host-clipped form
                if options.downsample_rate < 1.0 or int(per_region.max(initial=0)) > AlingerOptions.MAX_READS_IN_REGION:
                    host_clipped(output_hdf_file, group)
                    g0 += n_done
                    continue
                # one fetch for the group's whole stretch of the contig (adjacent intervals), sliced per interval
                lo, hi = regions[0][0], max(b for _, b in regions) + 1
                whole = fasta_handler.get_reference_bytes(chr_name, lo, hi)
                references = [whole[a - lo:b + 1 - lo] for a, b in regions]
                t0 = lap("fasta", t0)
                try:
                    outs, live = enc.encode(regions, references, region_pairs, counts, params, [(s, e) for _, s, e in group],
                                            ImageSizeOptions.CANDIDATE_WINDOW_SIZE, ImageSizeOptions.IMAGE_HEIGHT, resident=resident)
                except _lib.PepperAmdError as err:
                    if getattr(err, "code", 0) != _lib.PA_ERR_UNSUPPORTED:
                        raise
                    host_clipped(output_hdf_file, group)
                    g0 += n_done
                    continue
                t0 = lap("encode", t0)
                probs, at = None, 0
                if sink is not None:
                    # the group's windows are still where the encoder left them on the device: the model reads them there
                    total = sum(len(o["positions"]) for o in outs)
                    probs = sink.forward_device(device, enc.lib.pa_encoder_device_images(enc.enc), total)
                    t0 = lap("fused_forward", t0)
                for (chr_name, _start, _end), out, n_reads in zip(group, outs, live):
                    k = len(out["positions"])
                    if n_reads > 0:                  # (no read with a base inside: create_summary returns None, nothing is written)
                        write(output_hdf_file, chr_name, _start, _end, out, None if probs is None else probs[at:at + k])
                    at += k
                lap("hdf5", t0)
                g0 += n_done
            if enc.inflated_bytes:
                mine["inflate_kernel"] = mine.get("inflate_kernel", 0.0) + enc.inflate_ms / 1e3
                mine["inflated_bytes"] = mine.get("inflated_bytes", 0.0) + enc.inflated_bytes
            enc.inflate_ms, enc.inflated_bytes = 0.0, 0
            enc.release()
            t_close = time.perf_counter()
        lap("close", t_close)
        report()
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

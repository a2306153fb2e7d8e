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
import threading
from concurrent.futures import ThreadPoolExecutor
from datetime import datetime

from pepper_amd.polish.AlignmentSummarizer import AlignmentSummarizer
from pepper_amd.polish.DataStore import DataStore
from pepper_amd.polish.Options import ImageSizeOptions
from pepper_amd.variant.bam import BAM_handler
from pepper_amd.variant.fasta import FASTA_handler


_STATS_LOCK = threading.Lock()


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


def parse_device_ids(device_ids):
    """`device_ids` as the reference's callers give it ("0,1,2", call_consensus.py:60-66), a list, or None -> [0]."""
    if device_ids is None or device_ids == "":
        return [0]
    if isinstance(device_ids, str):
        return [int(d) for d in device_ids.split(",") if d.strip() != ""] or [0]
    if isinstance(device_ids, int):
        return [device_ids]
    return [int(d) for d in device_ids] or [0]


class UserInterfaceSupport:
    # intervals whose reads share one re-alignment call on the GPU (PEPPER_AMD_POLISH_REGIONS_PER_CALL)
    REGIONS_PER_CALL = int(os.environ.get("PEPPER_AMD_POLISH_REGIONS_PER_CALL", 32))
    # intervals per call of the device-resident chain (PEPPER_AMD_POLISH_CHAIN_REGIONS): ~64 reads each at 60x, so 128 of them
    # are ~8 000 re-alignments per launch -- several wavefronts per SIMD
    CHAIN_REGIONS = int(os.environ.get("PEPPER_AMD_POLISH_CHAIN_REGIONS", 128))
    # intervals per call of the BAM reader's packed form (PEPPER_AMD_POLISH_PACK_REGIONS): the file span of that many intervals is
    # inflated by ONE launch (a member takes a wavefront ~6 ms whatever the launch holds: 128 intervals are ~220 members, a
    # launch that leaves most of the chip idle) and stays on the device for the chain calls over its stretches (one worker
    # thread on a 16 Mb draft at 60x: 4.7 Mb/s with 128 intervals per span, 5.7 with 512, 6.1 with 1 024 -- ~110 MB of arena)
    PACK_REGIONS = int(os.environ.get("PEPPER_AMD_POLISH_PACK_REGIONS", 1024))

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
    def chain_generator(args, all_intervals, total_threads, thread_id, device, stats=None, fused=None):
        """image_generator through the device-resident chain (PEPPER.PolishChain): per run of consecutive intervals ONE call of
        the BAM reader's packed form (the file's BGZF members inflated and walked on the device) and ONE of the chain (clip,
        re-align, summarise, cut into chunks on the device), then one call of the image writer.  Intervals the chain does not
        take -- more than MAX_READS_IN_REGION reads (the reservoir sample is drawn in read order on the host), a record with
        its CIGAR in the CG tag, a read that keeps more bases than a pair's slot -- go through parse_regions as before."""
        import numpy as np
        from pepper_amd import _lib
        from pepper_amd.polish import PEPPER
        from pepper_amd.polish.AlignmentSummarizer import AlingerOptions
        from pepper_amd.variant.PEPPER_VARIANT import PackedEncoder
        output_path, bam_file, draft_file, truth_bam, train_mode, downsample_rate = args
        timestr = time.strftime("%m%d%Y_%H%M%S")
        file_name = output_path + "pepper_hp_images_thread_" + str(thread_id) + "_" + str(timestr) + ".hdf"
        per_call = max(1, UserInterfaceSupport.CHAIN_REGIONS)
        batch = max(per_call, UserInterfaceSupport.PACK_REGIONS)
        run = max(1, min(batch, -(-len(all_intervals) // max(1, total_threads * 2))))
        intervals = [r for i, r in enumerate(all_intervals) if (i // run) % total_threads == thread_id]
        if thread_id == 0:
            _log("INFO: STARTING THREAD: " + str(thread_id) + " FOR " + str(len(intervals)) + " INTERVALS")
        start_time = time.time()
        mine = {}

        def lap(key, t0):
            now = time.perf_counter()
            mine[key] = mine.get(key, 0.0) + now - t0
            return now

        try:
            enc = PackedEncoder.acquire(device, int(os.environ.get("PEPPER_AMD_ARENA_MB", 256)) << 20, host_threads=1,
                                        torch_stream=fused is not None)
        except _lib.PepperAmdError as err:
            # no page-locked arena to be had (memlock / cgroup limit): the host form needs none.  Not under the fused form:
            # image_generator writes images only, and polish(fused_inference=True) has no call_consensus step that would read
            # them -- this worker's intervals would silently be missing from the stitched FASTA
            if fused is not None:
                raise RuntimeError("fused polish: worker %d got no page-locked arena for the device chain (%s); run without "
                                   "fused_inference, or raise the memlock limit" % (thread_id, err)) from err
            return UserInterfaceSupport.image_generator(args, all_intervals, total_threads, thread_id)
        chain = PEPPER.PolishChain(enc)
        consensus = fused.worker(thread_id, device, stream=enc.stream) if fused is not None else None      # polish(fused_inference=True): fused.py
        device_inflate = os.environ.get("PEPPER_AMD_DEVICE_INFLATE", "1") != "0"
        safe = AlingerOptions.ALIGNMENT_SAFE_BASES
        seq_len, features = ImageSizeOptions.SEQ_LENGTH, ImageSizeOptions.IMAGE_HEIGHT
        views = {}
        handles = {}
        try:
            with DataStore(file_name, 'w') as output_hdf_file:
                def host_form(chr_name, block):
                    """parse_regions (host-clipped reads, host arrays between the stages) for the intervals of `block`."""
                    key = (chr_name, bam_file, draft_file)
                    if key not in views:
                        views.clear()
                        views[key] = UserInterfaceView(chr_name, bam_file, draft_file, truth_bam, train_mode)
                    for g0 in range(0, len(block), UserInterfaceSupport.REGIONS_PER_CALL):
                        part = block[g0:g0 + UserInterfaceSupport.REGIONS_PER_CALL]
                        results = views[key].parse_regions([(a, b) for _, a, b in part], downsample_rate)
                        for region, (images, labels, positions, chunk_ids) in zip(part, results):
                            if len(images):
                                output_hdf_file.write_summaries(region, images, labels, positions, chunk_ids)
                                if consensus is not None:
                                    consensus.add_host(region, images, positions, chunk_ids)

                counter = 0
                while counter < len(intervals):
                    chr_name = intervals[counter][0]
                    g1 = counter + 1
                    # ADJACENT intervals of one contig, ascending (the packer walks every record between the first and the last one)
                    while (g1 < len(intervals) and g1 - counter < batch and intervals[g1][0] == chr_name
                           and intervals[g1 - 1][1] <= intervals[g1][1] <= intervals[g1 - 1][2] + 1
                           and intervals[g1][2] >= intervals[g1 - 1][2]):
                        g1 += 1
                    block = intervals[counter:g1]
                    if chr_name not in handles:
                        handles.clear()
                        handles[chr_name] = (BAM_handler(bam_file), FASTA_handler(draft_file))
                    bam_handler, fasta_handler = handles[chr_name]
                    starts, stops = [a for _, a, _ in block], [b for _, _, b in block]
                    t0 = time.perf_counter()
                    on_device = enc.pack_device(bam_handler, chr_name, starts, stops, False, 0, laps=mine) if device_inflate else None
                    resident = on_device is not None
                    n_done = 0
                    if resident:
                        n_done, region_pairs, counts = on_device
                    else:
                        try:
                            n_done, region_pairs, counts = enc.pack(bam_handler, chr_name, starts, stops, False, 0)
                        except Exception as err:
                            if getattr(err, "code", 0) != -7:      # (-7: one interval's reads outgrow the arena)
                                raise
                    t0 = lap("bam_pack", t0)
                    if n_done == 0:
                        host_form(chr_name, block[:1])
                        counter += 1
                        continue
                    block, starts, stops = block[:n_done], starts[:n_done], stops[:n_done]
                    region_pairs = np.asarray(region_pairs[:n_done + 1], np.int32)
                    per_region = np.diff(region_pairs)
                    deep = np.flatnonzero(per_region > AlingerOptions.MAX_READS_IN_REGION)
                    if len(deep):
                        # a pile beyond the reference's cap is sampled down in read order (AlignmentSummarizer.py:314-326): on the
                        # host; the chain sees those intervals without reads and their chunks come from host_form below
                        keep = np.ones(int(region_pairs[-1]), bool)
                        for r in deep:
                            keep[region_pairs[r]:region_pairs[r + 1]] = False
                        kept = enc.pair_read[:int(region_pairs[-1])][keep]
                        enc.pair_read[:len(kept)] = kept
                        per_region = per_region.copy()
                        per_region[deep] = 0
                        region_pairs = np.concatenate([[0], np.cumsum(per_region)]).astype(np.int32)
                        counts = (counts[0], int(region_pairs[-1]), counts[2])
                    lo = starts[0]
                    whole = fasta_handler.get_reference_bytes(chr_name, lo, max(stops) + safe + 1)
                    windows = [whole[a - lo:b + safe + 1 - lo] for a, b in zip(starts, stops)]
                    t0 = lap("fasta", t0)
                    refused_at = None
                    for r0 in range(0, n_done, per_call):           # the chain over stretches of the resident span
                        r1 = min(n_done, r0 + per_call)
                        if consensus is not None:
                            consensus.settle()                       # (the copies of the last run's chunks: this run overwrites them)
                            t0 = lap("fused_consensus", t0)
                        try:
                            _rows, _live, chunks = chain.run(list(zip(starts[r0:r1], stops[r0:r1])), windows[r0:r1], region_pairs[r0:r1 + 1],
                                                             counts, realign=True, resident=resident, chunk_size=seq_len,
                                                             chunk_overlap=ImageSizeOptions.SEQ_OVERLAP)
                        except _lib.PepperAmdError as err:
                            if getattr(err, "code", 0) != _lib.PA_ERR_UNSUPPORTED:
                                raise
                            host_form(chr_name, block[r0:])          # (this stretch and what follows it in the span)
                            refused_at = r0
                            break
                        t0 = lap("chain", t0)
                        for key, v in chain.timing().items():
                            if key.endswith("_ms"):
                                mine["chain_" + key[:-3]] = mine.get("chain_" + key[:-3], 0.0) + v / 1e3
                            else:
                                mine[key] = mine.get(key, 0) + v
                        if consensus is not None and chain.n_chunks:
                            # (first: the gather copy then runs on the device while this thread writes the image file)
                            _img, pos_v, idx_v = chain.chunk_arrays()
                            consensus.add(chr_name, starts[r0:r1], stops[r0:r1], chunks, chain.device_chunks(), pos_v, idx_v)
                            t0 = lap("fused_consensus", t0)
                        img, pos, idx = chain.chunk_pointers()
                        output_hdf_file.write_regions(chr_name, starts[r0:r1], stops[r0:r1], chunks, seq_len, features, img, pos, idx)
                        t0 = lap("hdf5", t0)
                    if refused_at is not None:
                        deep = [r for r in deep if r < refused_at]
                    if len(deep):
                        host_form(chr_name, [block[r] for r in deep])
                        lap("deep_host_form", t0)
                    before = counter
                    counter += n_done
                    if thread_id == 0 and counter // 1000 > before // 1000:
                        elapsed = int(time.time() - start_time)
                        _log("INFO: [THREAD " + "{:02d}".format(thread_id) + "] " + str(counter) + "/" + str(len(intervals))
                             + " COMPLETE (" + str(int(100 * counter / len(intervals))) + "%) [ELAPSED TIME: "
                             + str(elapsed // 60) + " Min " + str(elapsed % 60) + " Sec]")
                t_close = time.perf_counter()
            lap("close", t_close)
            if consensus is not None:
                t0 = time.perf_counter()
                consensus.close()
                lap("fused_consensus", t0)
        except BaseException:
            if consensus is not None:
                consensus.close(failed=True)
            raise
        finally:
            enc.inflate_ms, enc.inflated_bytes = 0.0, 0
            enc.release()
        if stats is not None:
            with _STATS_LOCK:
                for key, v in mine.items():
                    stats[key] = stats.get(key, 0.0) + v
        return thread_id

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
    def worker_device(device_ids, thread_id):
        """Worker t works on device_ids[t % n] (the reference's callers give device_ids to call_consensus only,
        call_consensus.py:60-66; its image generation is CPU work)."""
        ids = parse_device_ids(device_ids)
        return ids[thread_id % len(ids)]

    @staticmethod
    def chromosome_level_parallelization(chr_list, bam_file, draft_file, truth_bam, output_path, total_threads, train_mode,
                                         downsample_rate=1.0, device_ids=None, stats=None, fused=None):
        if train_mode:
            raise NotImplementedError("train_mode image generation is outside the inference path")
        contigs, all_intervals = UserInterfaceSupport.make_intervals(chr_list, draft_file)
        _log("INFO: TOTAL CONTIGS: " + str(len(contigs)) + " TOTAL INTERVALS: " + str(len(all_intervals)))
        args = (output_path, bam_file, draft_file, truth_bam, train_mode, downsample_rate)
        # the device-resident chain is the default; PEPPER_AMD_POLISH_CHAIN=0: host arrays between the stages (round 4's form,
        # device 0 only)
        chain = os.environ.get("PEPPER_AMD_POLISH_CHAIN", "1") != "0" and downsample_rate >= 1.0

        def work(thread_id, n):
            if chain:
                return UserInterfaceSupport.chain_generator(args, all_intervals, n, thread_id,
                                                            UserInterfaceSupport.worker_device(device_ids, thread_id), stats, fused)
            if fused is not None:
                raise RuntimeError("the fused polish() needs the device-resident chain (PEPPER_AMD_POLISH_CHAIN=0 or a downsample rate switch it off)")
            return UserInterfaceSupport.image_generator(args, all_intervals, n, thread_id)
        if total_threads <= 1:
            work(0, 1)
            return
        with ThreadPoolExecutor(max_workers=total_threads) as executor:
            futures = [executor.submit(work, thread_id, total_threads) for thread_id in range(total_threads)]
            for fut in futures:
                fut.result()            # a worker's exception stops the run (the reference logs it and carries on)

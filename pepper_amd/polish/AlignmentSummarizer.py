"""Polish region -> 1000-row image chunks (inference branch).

Mirrors /root/reference/pepper/modules/python/AlignmentSummarizer.py:18-56 (chunk_images) and
:296-358 (read fetch, reservoir sampling to 1500 reads, reference fetch, SummaryGenerator,
chunking) and :159-177 (reads_to_reference_realignment: every read re-aligned to the draft before it is
summarised, on by default; the alignments run on the GPU, include/pepper_amd_realign.h).
"""
import os

import numpy as np

from pepper_amd.polish import PEPPER
from pepper_amd.polish.Options import ImageSizeOptions


class AlingerOptions(object):
    ALIGNMENT_SAFE_BASES = 20      # pepper Options.py:23-29
    MAX_READS_IN_REGION = 1500
    # reads of consecutive regions handed to the GPU re-aligner together (PEPPER_AMD_POLISH_READS_PER_CALL)
    MAX_READS_PER_CALL = int(os.environ.get("PEPPER_AMD_POLISH_READS_PER_CALL", 3000))
    RANDOM_SEED = 2719747673


class AlignmentSummarizer:
    def __init__(self, bam_handler, fasta_handler, chromosome_name, region_start, region_end):
        self.bam_handler = bam_handler
        self.fasta_handler = fasta_handler
        self.chromosome_name = chromosome_name
        self.region_start_position = region_start
        self.region_end_position = region_end

    @staticmethod
    def chunk_images(summary, chunk_size, chunk_overlap):
        """Rows -> chunks of chunk_size, the next chunk starting chunk_overlap rows before the end
        of the previous one; the last chunk is padded with zero rows and (-1, -1) positions."""
        image = np.asarray(summary.image, dtype=np.uint8).reshape(-1, ImageSizeOptions.IMAGE_HEIGHT)
        pos = getattr(summary, "positions_array", None)        # the encoder's bulk form of genomic_pos
        if pos is None:
            pos = np.asarray(summary.genomic_pos, dtype=np.int64).reshape(-1, 2)
        total = len(pos)
        images, labels, positions, chunk_ids = [], [], [], []
        chunk_start, chunk_id = 0, 0
        chunk_end = min(total, chunk_size)
        while True:
            img = np.zeros((chunk_size, ImageSizeOptions.IMAGE_HEIGHT), np.uint8)
            p = np.full((chunk_size, 2), -1, np.int64)
            n = chunk_end - chunk_start
            img[:n] = image[chunk_start:chunk_end]
            p[:n] = pos[chunk_start:chunk_end]
            images.append(img)
            labels.append(np.zeros(chunk_size, np.uint8))
            positions.append(p)
            chunk_ids.append(chunk_id)
            chunk_id += 1
            if chunk_end == total:
                break
            chunk_start = chunk_end - chunk_overlap
            chunk_end = min(total, chunk_start + chunk_size)
        return images, labels, positions, chunk_ids

    def reads_to_reference_realignment(self, region_start, region_end, reads):
        """Local re-alignment of the reads to the reference (AlignmentSummarizer.py:159-177)."""
        if len(reads) == 0:
            return reads
        ref_start, ref_end = region_start, region_end + AlingerOptions.ALIGNMENT_SAFE_BASES
        ref_sequence = self.fasta_handler.get_reference_sequence(self.chromosome_name, ref_start, ref_end)
        aligner = PEPPER.ReadAligner(ref_start, ref_end, ref_sequence)
        return aligner.align_reads_to_reference(reads)

    def fetch_reads(self):
        """Reads of the region as the reference fetches them: region-clipped, at most MAX_READS_IN_REGION of them by
        reservoir sampling with the reference's seed (AlignmentSummarizer.py:296-326)."""
        read_start = max(0, self.region_start_position)
        read_end = self.region_end_position
        all_reads = self.bam_handler.get_reads(self.chromosome_name, read_start, read_end, False, 0, 0)
        total_reads = len(all_reads)
        if total_reads > AlingerOptions.MAX_READS_IN_REGION:
            flat = hasattr(all_reads, "as_pileup")
            random = np.random.RandomState(AlingerOptions.RANDOM_SEED)
            sample = []
            for i in range(total_reads):
                if len(sample) < AlingerOptions.MAX_READS_IN_REGION:
                    sample.append(i)
                else:
                    j = random.randint(0, i + 1)
                    if j < AlingerOptions.MAX_READS_IN_REGION:
                        sample[j] = i
            all_reads = all_reads.take(sample) if flat else [all_reads[i] for i in sample]
        return all_reads

    def summarise(self, all_reads):
        """Reads (already re-aligned if that stage is on) -> chunked summary (AlignmentSummarizer.py:334-358)."""
        if len(all_reads) == 0:
            return [], [], [], []
        if hasattr(all_reads, "as_pileup"):          # pepper_amd.variant.bam.ReadSet (the same get_reads serves both)
            all_reads = all_reads.as_pileup()
        ref_seq = self.fasta_handler.get_reference_sequence(self.chromosome_name, self.region_start_position,
                                                            self.region_end_position + 1)
        summary_generator = PEPPER.SummaryGenerator(ref_seq, self.chromosome_name, self.region_start_position,
                                                    self.region_end_position)
        summary_generator.generate_summary(all_reads, self.region_start_position, self.region_end_position)
        return self.chunk_images(summary_generator, chunk_size=ImageSizeOptions.SEQ_LENGTH,
                                 chunk_overlap=ImageSizeOptions.SEQ_OVERLAP)

    def create_summary(self, truth_bam_handler=None, train_mode=False, downsample_rate=1.0, realignment_flag=True):
        """Argument order and defaults of the reference (AlignmentSummarizer.py:179): with realignment_flag (the
        default) every read is re-aligned to the draft before encoding; realignment_flag=False encodes the reads as
        aligned in the BAM."""
        if train_mode:
            raise NotImplementedError("train_mode image generation is outside the inference path")
        all_reads = self.fetch_reads()
        if len(all_reads) == 0:
            return [], [], [], []
        if realignment_flag:
            all_reads = self.reads_to_reference_realignment(self.region_start_position, self.region_end_position, all_reads)
        return self.summarise(all_reads)

    @staticmethod
    def create_summaries(summarizers, realignment_flag=True):
        """create_summary for several regions with ONE re-alignment call on the GPU (a region at ordinary coverage is a
        few dozen wavefronts; the job table takes the reads of all regions).  Per-region results, in order."""
        from pepper_amd.variant.bam import ReadSet
        reads = [s.fetch_reads() for s in summarizers]
        batchable = realignment_flag and all(isinstance(r, ReadSet) for r in reads)
        if realignment_flag and not batchable:
            reads = [s.reads_to_reference_realignment(s.region_start_position, s.region_end_position, r) if len(r) else r
                     for s, r in zip(summarizers, reads)]
        elif batchable and any(len(r) for r in reads):
            # one call per run of regions holding up to MAX_READS_PER_CALL reads (the band stage keeps ~0.5 MB of
            # direction bytes per read on the device; regions at the 1500-read cap go through in smaller groups)
            done, lo = [], 0
            while lo < len(reads):
                hi, total = lo + 1, len(reads[lo])
                while hi < len(reads) and total + len(reads[hi]) <= AlingerOptions.MAX_READS_PER_CALL:
                    total += len(reads[hi])
                    hi += 1
                group, group_s = reads[lo:hi], summarizers[lo:hi]
                if total:
                    windows = []
                    for s in group_s:
                        ref_end = s.region_end_position + AlingerOptions.ALIGNMENT_SAFE_BASES
                        windows.append((s.region_start_position, s.fasta_handler.get_reference_sequence(
                            s.chromosome_name, s.region_start_position, ref_end)))
                    counts = [len(r) for r in group]
                    seq_lens = np.concatenate([r.seq_offset[1:] - r.seq_offset[:-1] for r in group])
                    seq_offset = np.zeros(total + 1, np.int64)
                    np.cumsum(seq_lens, out=seq_offset[1:])
                    out = PEPPER.align_windows(windows, np.repeat(np.arange(len(group), dtype=np.int32), counts),
                                               np.concatenate([r.pos for r in group]), seq_offset,
                                               np.concatenate([r.seq[:int(r.seq_offset[-1])] for r in group]))
                    first = 0
                    for r in group:
                        done.append(PEPPER.apply_alignment(r, out, first))
                        first += len(r)
                else:
                    done.extend(group)
                lo = hi
            reads = done
        # ... and ONE summary-encoder call for the regions that have reads (pa_polish_encoder_generate_summary_batch)
        gens, flats, spans, live = [], [], [], []
        for k, (s, r) in enumerate(zip(summarizers, reads)):
            if len(r) == 0:
                continue
            ref_seq = s.fasta_handler.get_reference_sequence(s.chromosome_name, s.region_start_position, s.region_end_position + 1)
            gens.append(PEPPER.SummaryGenerator(ref_seq, s.chromosome_name, s.region_start_position, s.region_end_position))
            flats.append(r.as_pileup() if hasattr(r, "as_pileup") else r)
            spans.append((s.region_start_position, s.region_end_position))
            live.append(k)
        PEPPER.generate_summaries(gens, flats, spans)
        out = [([], [], [], [])] * len(summarizers)
        for k, g in zip(live, gens):
            out[k] = summarizers[k].chunk_images(g, chunk_size=ImageSizeOptions.SEQ_LENGTH, chunk_overlap=ImageSizeOptions.SEQ_OVERLAP)
        return out

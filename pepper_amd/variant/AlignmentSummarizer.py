"""Region -> candidate summaries (inference branch).

Mirrors /root/reference/pepper_variant/modules/python/AlignmentSummarizer.py:8-16,180-240:
fetch reads +-REGION_SAFE_BASES around the interval, reservoir-sample down to
min(MAX_READS_IN_REGION, downsample_rate * n) with numpy RandomState(2719747673), fetch the
reference for [region_start, region_end + 1), run the summary encoder.  The BAM / FASTA handlers
are injected objects with the reference's handler methods (`get_reads`, `get_reference_sequence`);
htslib ingestion itself is out of scope here.  train_mode (truth-VCF labels)
is not provided.
"""
import numpy as np

from pepper_amd.variant import PEPPER_VARIANT
from pepper_amd.variant.Options import ImageSizeOptions


class ConsensCandidateFinder(object):
    REGION_SAFE_BASES = 100          # pepper_variant Options.py:1-2


class AlingerOptions(object):
    MAX_READS_IN_REGION = 5000       # pepper_variant Options.py:96-100
    RANDOM_SEED = 2719747673


class AlignmentSummarizer:
    def __init__(self, bam_handler, fasta_handler, chromosome_name, region_start, region_end):
        self.bam_handler = bam_handler
        self.fasta_handler = fasta_handler
        self.chromosome_name = chromosome_name
        self.region_start_position = region_start
        self.region_end_position = region_end

    def prepare(self, options):
        """Everything of create_summary up to the encoder call: reads of the padded region (sampled down as the reference
        does), its reference.  -> (generator, reads, encoder arguments) or None when the region has no reads."""
        if getattr(options, "train_mode", False):
            raise NotImplementedError("train_mode image generation is outside the inference path")
        region_start = max(0, self.region_start_position - ConsensCandidateFinder.REGION_SAFE_BASES)
        region_end = self.region_end_position + ConsensCandidateFinder.REGION_SAFE_BASES
        all_reads = self.bam_handler.get_reads(self.chromosome_name, region_start, region_end,
                                               options.include_supplementary, options.min_mapq,
                                               options.min_snp_baseq)
        total_reads = len(all_reads)
        total_allowed_reads = int(min(AlingerOptions.MAX_READS_IN_REGION, options.downsample_rate * total_reads))
        flat = hasattr(all_reads, "as_pileup")          # pepper_amd.variant.bam.ReadSet: structure of arrays
        if total_reads > total_allowed_reads:
            # reservoir sampling exactly as the reference (nucleus utils.reservoir_sample); on read indices
            random = np.random.RandomState(AlingerOptions.RANDOM_SEED)
            sample = []
            for i in range(total_reads):
                if len(sample) < total_allowed_reads:
                    sample.append(i)
                else:
                    j = random.randint(0, i + 1)
                    if j < total_allowed_reads:
                        sample[j] = i
            all_reads = all_reads.take(sample) if flat else [all_reads[i] for i in sample]
        if len(all_reads) == 0:
            return None
        # ref_seq should contain region_end_position base
        ref_seq = self.fasta_handler.get_reference_sequence(self.chromosome_name, region_start, region_end + 1)
        regional_summary = PEPPER_VARIANT.RegionalSummaryGenerator(self.chromosome_name, region_start, region_end,
                                                                   ref_seq, device=getattr(options, "device", 0))
        regional_summary.generate_max_insert_summary(all_reads)
        if flat:
            all_reads = all_reads.as_pileup()
        args = (options.min_snp_baseq, options.min_indel_baseq, options.snp_frequency,
                options.insert_frequency, options.delete_frequency, options.min_coverage_threshold,
                options.snp_candidate_frequency_threshold, options.indel_candidate_frequency_threshold,
                options.candidate_support_threshold, options.skip_indels, self.region_start_position,
                self.region_end_position, ImageSizeOptions.CANDIDATE_WINDOW_SIZE, ImageSizeOptions.IMAGE_HEIGHT,
                False)
        return regional_summary, all_reads, args

    def create_summary(self, options, bed_list, thread_id, as_arrays=False):
        prepared = self.prepare(options)
        if prepared is None:
            return None
        regional_summary, all_reads, args = prepared
        if as_arrays:
            return regional_summary.generate_summary_arrays(all_reads, *args)
        return regional_summary.generate_summary(all_reads, *args)


def create_summaries(prepared):
    """The encoder call of several regions at once (pa_encoder_generate_summary_batch: one workgroup per 512-position tile, a
    batch fills the chip where one region does not).  prepared: results of AlignmentSummarizer.prepare (None entries are
    regions without reads) -> the arrays of generate_summary_arrays per region, None where the region had no reads."""
    live = [p for p in prepared if p is not None]
    out = iter(PEPPER_VARIANT.generate_summary_arrays_batch(
        [p[0] for p in live], [p[1] for p in live], *live[0][2][:10], [(p[2][10], p[2][11]) for p in live],
        live[0][2][12], live[0][2][13], live[0][2][14])) if live else iter(())
    return [None if p is None else next(out) for p in prepared]

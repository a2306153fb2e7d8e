"""VCF output of the candidate finder.

replaces: /root/reference/pepper_variant/modules/python/VcfWriter.py
    VCFWriter.__init__ / __del__   :13-46    five bgzipped VCFs + tabix indices
    candidate_list_to_variant      :48-138   per-site merge of the allele records
    write_vcf_records              :140-218  QUAL, filters, routing into the five files
    get_vcf_header                 :220-289
The reference renders records through pysam/htslib, which this image does not have; the text
below is produced directly: same header lines in the same order (htslib drops the repeated GT and
the explicit PASS definition, and puts `##fileformat=VCFv4.2` + the PASS filter first), FORMAT keys
in the keyword order of `new_record` (GT:AP:GQ:DP:AD:VAF:REP), floats stored as float32 and printed
with six significant digits as htslib's kputd does.  Text rendering is UNPINNED against pysam
(absent here); the selection logic is covered by hand-derived cases in tests/test_candidate_finder.py.
"""
import math
import struct

import numpy as np

from pepper_amd.variant.bgzf import BgzfWriter, TabixBuilder
from pepper_amd.variant.fasta import FASTA_handler


_F32 = struct.Struct("f")


def _fmt_float(value):
    """BCF keeps floats as float32; htslib prints them with %g-like 6 significant digits."""
    try:
        v = _F32.unpack(_F32.pack(value))[0]           # round to float32 (what float(np.float32(value)) gives, 5x cheaper)
    except OverflowError:
        v = float(np.float32(value))
    if v != v:
        return "."
    if v == int(v) and abs(v) < 1e6:
        return str(int(v))
    return "%g" % v


def _picklable_options(options):
    """The option fields the formatter reads, as a plain namespace (the caller's object may hold handles / factories)."""
    from types import SimpleNamespace
    names = ("allowed_multiallelics", "snp_q_cutoff", "snp_q_cutoff_in_lc", "indel_q_cutoff", "indel_q_cutoff_in_lc")
    return SimpleNamespace(**{n: getattr(options, n) for n in names})


class _VcfFile(object):
    def __init__(self, path, header_text):
        self.path = path
        self._out = BgzfWriter(path, deferred=True)
        self._out.write(header_text)
        self._index = TabixBuilder()

    def write(self, record):
        contig, start, ref_len, line = record
        out = self._out
        vbeg = out.tell()
        out.write(line)
        self._index.add(contig, start, start + ref_len, vbeg, out.tell())

    def write_columns(self, names, contig_code, start, ref_len, lengths, lines):
        """write() for records in file order given as columns (names[contig_code[i]], start, len(REF), len(line), line): one
        join, offsets by arithmetic (bgzf.BgzfWriter.position), one vectorised index update."""
        from pepper_amd.variant.bgzf import _BLOCK_DATA
        out = self._out
        ends = out.position() + np.cumsum(lengths)
        begs = ends - lengths
        out.write(b"".join(lines))
        self._index.add_many(names, contig_code, start, start + ref_len, ((begs // _BLOCK_DATA) << 16) | (begs % _BLOCK_DATA),
                             ((ends // _BLOCK_DATA) << 16) | (ends % _BLOCK_DATA))

    def close(self):
        if self._out is not None:
            out = self._out
            out.close()
            self._out = None
            self._index.write(self.path + ".tbi", resolve=out.resolve)


class VCFWriter:
    def __init__(self, all_contigs, reference_file_path, sample_name, output_dir, filename_full, filename_pepper,
                 filename_variant_calling, fasta_handler=None):
        self.fasta_handler = fasta_handler if fasta_handler is not None else FASTA_handler(reference_file_path)
        contigs = self.fasta_handler.get_chromosome_names()
        self.contigs = contigs
        self.vcf_header = self.get_vcf_header(sample_name, contigs)
        self.output_dir = output_dir

        self.full_vcf_file_name = self.output_dir + filename_full + '.vcf.gz'
        self.pepper_vcf_file_name = self.output_dir + filename_pepper + '.vcf.gz'
        self.variant_vcf_file_name = self.output_dir + filename_variant_calling + '.vcf.gz'
        self.snp_variant_vcf_file_name = self.output_dir + filename_variant_calling + '_SNPs.vcf.gz'
        self.indel_variant_vcf_file_name = self.output_dir + filename_variant_calling + '_INDEL.vcf.gz'

        self.vcf_file_full = _VcfFile(self.full_vcf_file_name, self.vcf_header)
        self.vcf_file_pepper = _VcfFile(self.pepper_vcf_file_name, self.vcf_header)
        self.vcf_file_variant_calling = _VcfFile(self.variant_vcf_file_name, self.vcf_header)
        self.vcf_file_variant_calling_snp = _VcfFile(self.snp_variant_vcf_file_name, self.vcf_header)
        self.vcf_file_variant_calling_indel = _VcfFile(self.indel_variant_vcf_file_name, self.vcf_header)

    def close(self):
        for f in (self.vcf_file_full, self.vcf_file_pepper, self.vcf_file_variant_calling,
                  self.vcf_file_variant_calling_snp, self.vcf_file_variant_calling_indel):
            f.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def candidate_list_to_variant(self, candidates, options):
        # records: (contig, ref_start, ref_end, ref_allele, alt_alleles, genotype, depth, supports,
        #           genotype_probability, predictions, non_alt_predictions, in_repeat)
        candidates = sorted(candidates, key=lambda x: (x[5], x[8]), reverse=True)
        if len(candidates) > options.allowed_multiallelics:
            candidates = candidates[:options.allowed_multiallelics]

        # the longest REF of the site; shorter records are padded with its tail (:55-75)
        max_ref_allele = ''
        for c in candidates:
            if len(c[3]) > len(max_ref_allele):
                max_ref_allele = c[3]

        gt_qual = -1.0
        genotype_hp1, genotype_hp2 = [], []
        site = None
        site_depth = 0
        site_alts, site_supports, site_non_alt_predictions = [], [], []
        site_in_repeat = False
        for i, c in enumerate(candidates):
            contig, ref_start, _, ref_allele, alt_allele, _, depth, support, _, predictions, non_alt_predictions, in_repeat = c
            pad = len(max_ref_allele) - len(ref_allele)
            if pad > 0:
                suffix = max_ref_allele[-pad:]
                ref_allele = ref_allele + suffix
                alt_allele = [alt + suffix for alt in alt_allele]

            site_in_repeat = in_repeat or site_in_repeat
            predicted_genotype = max(range(len(predictions)), key=predictions.__getitem__)   # first maximum, as numpy.argmax
            if predicted_genotype != 0:
                gt_qual = predictions[predicted_genotype] if gt_qual < 0 else min(gt_qual, predictions[predicted_genotype])
            elif gt_qual < 0:
                gt_qual = max(predictions[1], predictions[2])

            if site is None:
                site = (contig, ref_start, ref_start + len(ref_allele), ref_allele)
                site_depth = depth
            site_depth = min(site_depth, depth)
            site_alts.append(alt_allele[0])
            site_supports.append(support[0])
            site_non_alt_predictions.extend(non_alt_predictions)

            if predicted_genotype == 1:
                genotype_hp1.append(i + 1)
            elif predicted_genotype == 2:
                genotype_hp1.append(i + 1)
                genotype_hp2.append(i + 1)

        if 0 < len(genotype_hp1) + len(genotype_hp2) <= 2:
            gt = genotype_hp1 + genotype_hp2
            if len(gt) == 1:
                gt = [0, gt[0]]
        else:
            gt = [0, 0]
        if site is None:
            site = ('', 0, 0, '')
        return site[0], site[1], site[2], site[3], site_alts, gt, site_depth, site_supports, gt_qual, \
            site_non_alt_predictions, site_in_repeat

    @staticmethod
    def _record(contig, ref_start, alleles, qual, filt, genotype, ap, gq, depth, ad, vafs, rep):
        sample = ":".join([
            "/".join(str(g) for g in genotype),
            ",".join(_fmt_float(v) for v in ap) if len(ap) else ".",
            _fmt_float(gq),
            str(int(depth)),
            ",".join(str(int(v)) for v in ad) if len(ad) else ".",
            ",".join(_fmt_float(v) for v in vafs) if len(vafs) else ".",
            rep,
        ])
        line = "\t".join([str(contig), str(ref_start + 1), ".", alleles[0], ",".join(alleles[1:]), _fmt_float(qual),
                          filt, ".", "GT:AP:GQ:DP:AD:VAF:REP", sample]) + "\n"
        return (str(contig), ref_start, len(alleles[0]), line.encode())     # encoded once, written to up to three files

    @staticmethod
    def format_sites(site_candidates, options):
        """The per-site part of write_vcf_records (:150-218) for a list of sites in order: (record, is_snp,
        selected_for_variant_calling) per site, or None where the site has no allele.  No state: runs in worker processes."""
        out = []
        for candidates in site_candidates:
            contig, ref_start, ref_end, ref_seq, alleles, genotype, depth, variant_allele_support, genotype_probability, \
                non_alt_predictions, site_in_repeat = VCFWriter.candidate_list_to_variant(None, candidates, options)
            if len(alleles) <= 0:
                out.append(None)
                continue
            max_alt_len = max(len(ref_seq), max(len(x) for x in alleles))
            alleles = (ref_seq,) + tuple(alleles)
            qual = max(1, int(-10 * math.log10(max(0.000000001, 1.0 - genotype_probability))))
            is_snp = max_alt_len == 1
            if is_snp:
                cutoff = options.snp_q_cutoff_in_lc if site_in_repeat else options.snp_q_cutoff
            else:
                cutoff = options.indel_q_cutoff_in_lc if site_in_repeat else options.indel_q_cutoff
            failed_variant = qual <= cutoff
            # everything not confidently genotyped goes to the re-genotyping set (:178-182)
            selected_for_variant_calling = genotype == [0, 0] or failed_variant
            vafs = [round(ad / max(1, depth), 3) for ad in variant_allele_support]
            rep = "1" if site_in_repeat else "0"
            record = VCFWriter._record(contig, ref_start, alleles, qual, 'refCall' if genotype == [0, 0] else 'PASS',
                                       genotype, non_alt_predictions, qual, depth, variant_allele_support, vafs, rep)
            out.append((record, is_snp, selected_for_variant_calling))
        return out

    def write_vcf_records(self, variants_list, options):
        total_variants, total_pepper, total_calling, total_calling_snp, total_calling_indel = 0, 0, 0, 0, 0
        last_position = -1
        keys = sorted(variants_list)
        sites = [variants_list[k] for k in keys]
        # records are formatted in `threads` worker processes (a pure function of the site); the sequential part -- the
        # duplicate-start rule and the five compressed, indexed files -- stays here
        threads = max(1, int(getattr(options, "threads", 1) or 1))
        block = 4096
        blocks = [sites[i:i + block] for i in range(0, len(sites), block)]
        # (worker processes pay for pickling the site tuples both ways: below a few hundred thousand sites one process is faster)
        if threads > 1 and len(sites) >= 400000:
            import concurrent.futures
            plain = _picklable_options(options)
            with concurrent.futures.ProcessPoolExecutor(max_workers=threads) as executor:
                formatted_blocks = [f.result() for f in [executor.submit(VCFWriter.format_sites, b, plain) for b in blocks]]
        else:
            formatted_blocks = [VCFWriter.format_sites(b, options) for b in blocks]
        for formatted in formatted_blocks:
            for item in formatted:
                if item is None:
                    continue
                record, is_snp, selected_for_variant_calling = item
                ref_start = record[1]
                if ref_start == last_position:     # (sic: compared across contigs too, :150-151)
                    continue
                last_position = ref_start
                self.vcf_file_full.write(record)
                total_variants += 1
                if selected_for_variant_calling:
                    if is_snp:
                        self.vcf_file_variant_calling_snp.write(record)
                        total_calling_snp += 1
                    else:
                        self.vcf_file_variant_calling_indel.write(record)
                        total_calling_indel += 1
                    self.vcf_file_variant_calling.write(record)
                    total_calling += 1
                else:
                    self.vcf_file_pepper.write(record)
                    total_pepper += 1
        return total_variants, total_pepper, total_calling, total_calling_snp, total_calling_indel

    def get_vcf_header(self, sample_name, contigs):
        lines = [
            '##fileformat=VCFv4.2',
            '##FILTER=<ID=PASS,Description="All filters passed">',
            '##FILTER=<ID=refCall,Description="Call is homozygous">',
            '##FILTER=<ID=lowGQ,Description="Low genotype quality">',
            '##FILTER=<ID=lowQUAL,Description="Low variant call quality">',
            '##FILTER=<ID=conflictPos,Description="Overlapping record">',
            '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
            '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Depth">',
            '##FORMAT=<ID=AD,Number=A,Type=Integer,Description="Allele depth">',
            '##FORMAT=<ID=VAF,Number=A,Type=Float,Description="Variant allele fractions.">',
            '##FORMAT=<ID=AP,Number=A,Type=Float,Description="Maximum variant allele probability for each allele.">',
            '##FORMAT=<ID=GQ,Number=1,Type=Float,Description="Genotype Quality">',
            '##FORMAT=<ID=REP,Number=1,Type=String,Description="If set to 1 then variant site is considered to be ina LowCompexity repeat region">',
        ]
        for sq in self.fasta_handler.get_chromosome_names():
            if sq not in contigs:
                continue
            lines.append('##contig=<ID=%s,length=%d>' % (sq, self.fasta_handler.get_chromosome_sequence_length(sq)))
        lines.append('#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t' + str(sample_name))
        return "\n".join(lines) + "\n"

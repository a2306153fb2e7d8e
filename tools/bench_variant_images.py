"""Variant image generation rate (BAM + reference -> candidate image HDF5: BAM reader, GPU summary encoder, HDF5 writer)
on the data set of tools/bench_polish_images.py (python tools/bench_polish_images.py make <dir>), with cProfile of the
single-thread run.   python tools/bench_variant_images.py <dir>"""
import cProfile
import json
import os
import pstats
import shutil
import sys
import time
from types import SimpleNamespace

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pepper_amd.variant.ImageGenerationUI import ImageGenerationUtils  # noqa: E402


def options(data, out, threads):
    return SimpleNamespace(
        bam=os.path.join(data, "reads.bam"), fasta=os.path.join(data, "draft.fa"), region=None, region_size=10000,
        threads=threads, train_mode=False, use_hp_info=False, include_supplementary=False, image_output_directory=out,
        min_mapq=1, min_snp_baseq=1, min_indel_baseq=1, snp_frequency=0.10, insert_frequency=0.15, delete_frequency=0.15,
        min_coverage_threshold=3, snp_candidate_frequency_threshold=0.10, indel_candidate_frequency_threshold=0.12,
        candidate_support_threshold=2, skip_indels=False, downsample_rate=1.0)


def main(data):
    out = []
    kb = os.path.getsize(os.path.join(data, "draft.fa")) // 1000
    for threads in (1, 4, 8):
        tmp = os.path.join(data, "vimages_t%d" % threads)
        shutil.rmtree(tmp, ignore_errors=True)
        if threads == 1:
            ImageGenerationUtils.generate_images(options(data, tmp + "_warm", 1))
            shutil.rmtree(tmp + "_warm", ignore_errors=True)
            pr = cProfile.Profile()
            pr.enable()
        t0 = time.perf_counter()
        ImageGenerationUtils.generate_images(options(data, tmp, threads))
        dt = time.perf_counter() - t0
        if threads == 1:
            pr.disable()
            pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
        out.append({"threads": threads, "seconds": round(dt, 3), "kb_per_s": round(kb / dt, 1)})
        shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps({"metric": "variant make_images, %d kb at ~60x in 10 kb intervals, one GPU" % kb, "runs": out}))


if __name__ == "__main__":
    main(sys.argv[1])

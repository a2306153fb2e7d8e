"""Variant image generation rate: BAM + reference -> candidate image HDF5 (BAM reader, GPU summary encoder, HDF5 writer)
through pepper_amd.variant.ImageGenerationUI.generate_images, the entry point of pepper_variant make_images / call_variant.

  python tools/bench_variant_images.py make_fast <dir> [genome_bases=64000000] [coverage=60] [seed] [level=1] [tags=0] [quals=0]    synthetic data set
  python tools/bench_variant_images.py run <dir> [threads,threads,...] [region_size=100000]     GPU; one JSON line

`run` reports, per thread count, the wall time, Mb of reference per second, aligned bases per second and the stage times
summed over the workers (bam_pack: inflate + header walk + slice copies; fasta; encode: upload + clip/decode + kernels +
candidate enumeration + result copy; hdf5).  PEPPER_AMD_PACKED_READS=0 times the host-clipped form."""
import json
import os
import shutil
import subprocess
import sys
import time
from types import SimpleNamespace

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def options(data, out, threads, region_size=100000, **over):
    o = SimpleNamespace(
        bam=os.path.join(data, "reads.bam"), fasta=os.path.join(data, "draft.fa"), region=None, region_size=region_size,
        threads=threads, train_mode=False, use_hp_info=False, include_supplementary=False, image_output_directory=out,
        min_mapq=1, min_snp_baseq=1, min_indel_baseq=1, snp_frequency=0.10, insert_frequency=0.15, delete_frequency=0.15,
        min_coverage_threshold=3, snp_candidate_frequency_threshold=0.10, indel_candidate_frequency_threshold=0.12,
        candidate_support_threshold=2, skip_indels=False, downsample_rate=1.0)
    for k, v in over.items():
        setattr(o, k, v)
    return o


def make_fast(out, bases=64000000, coverage=60, seed=2027, level=1, tags=0, quals=0):
    """level 6, tags 1: zlib level 6 members and NM / MD / RG aux data in every record, as samtools writes a BAM; quals 1: quality
    strings with run-length structure (binned plateaus), so that members compress > 3 x as a binning basecaller's BAM does."""
    from pepper_amd import build
    tool = build.build_tools()
    os.makedirs(out, exist_ok=True)
    t0 = time.perf_counter()
    info = json.loads(subprocess.run([tool, out, str(int(bases)), str(coverage), str(int(seed)), "0", "1", str(int(level)), str(int(tags)), str(int(quals))],
                                     check=True, capture_output=True, text=True).stdout)
    info["seconds"] = round(time.perf_counter() - t0, 2)
    with open(os.path.join(out, "synth.json"), "w") as fh:
        json.dump(info, fh)
    return info


def run(data, thread_counts=(16,), region_size=100000, warm=True):
    from pepper_amd.variant.ImageGenerationUI import ImageGenerationUtils
    info = json.load(open(os.path.join(data, "synth.json")))
    mb = info["genome_bases"] / 1e6
    if warm:      # library load, the workers' buffers grown to the size of the job's groups, the file in the page cache -- not part of the rate
        ImageGenerationUtils.generate_images(options(data, os.path.join(data, "vimages_warm"), thread_counts[0], region_size))
        shutil.rmtree(os.path.join(data, "vimages_warm"), ignore_errors=True)
    runs = []
    for threads in thread_counts:
        tmp = os.path.join(data, "vimages_t%d" % threads)
        shutil.rmtree(tmp, ignore_errors=True)
        stages = {}
        t0 = time.perf_counter()
        ImageGenerationUtils.generate_images(options(data, tmp, threads, region_size, stage_seconds=stages))
        dt = time.perf_counter() - t0
        size = sum(os.path.getsize(os.path.join(tmp, f)) for f in os.listdir(tmp))
        runs.append({"threads": threads, "seconds": round(dt, 3), "mb_reference_per_s": round(mb / dt, 2),
                     "aligned_gbases_per_s": round(info["read_bases"] / dt / 1e9, 3),
                     "stage_seconds_summed_over_workers": {k: round(v, 2) for k, v in sorted(stages.items())},
                     "image_file_mb": round(size / 1e6, 1)})
        shutil.rmtree(tmp, ignore_errors=True)
    return {"metric": "variant make_images (generate_images): Mb of reference per second",
            "data": "synthetic BAM %.1f Mb at %.0fx, %d records, %.2f GB (tools/synth_bam, %s%s), intervals of %d" % (
                mb, info["coverage"], info["records"], info["bam_bytes"] / 1e9, info.get("deflate", "level 1"),
                (", NM/MD/RG tags" if info.get("aux_tags") else "") + (", run-length qualities" if "run-length" in info.get("quals", "") else ""),
                region_size),
            "packed_reads": os.environ.get("PEPPER_AMD_PACKED_READS", "1") != "0", "runs": runs}


if __name__ == "__main__":
    if sys.argv[1] == "make_fast":
        print(json.dumps(make_fast(sys.argv[2], *(float(a) for a in sys.argv[3:9]))))
    else:
        counts = tuple(int(t) for t in sys.argv[3].split(",")) if len(sys.argv) > 3 else (16,)
        print(json.dumps(run(sys.argv[2], counts, int(sys.argv[4]) if len(sys.argv) > 4 else 100000)))

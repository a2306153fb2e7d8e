"""Polish image generation rate through the device-resident chain: BAM + draft -> image HDF5 files through
pepper_amd.polish.make_images.make_images (pepper polish's first step, /root/reference/pepper/modules/python/make_images.py).

  python tools/bench_polish_chain.py make_fast <dir> [draft_bases=16000000] [coverage=60]     synthetic data set (tools/synth_bam)
  python tools/bench_polish_chain.py run <dir> [threads,threads,...] [regions_per_call]       GPU; one JSON line

`run` reports, per thread count, wall time, Mb of draft per second, intervals per second, reads re-aligned per second and the stage
times summed over the workers (bam_*: file span read, device inflate, device record walk, host pair lists; fasta; chain: the
whole device chain on the host clock, with its parts chain_unpack / chain_realign / chain_encode / chain_chunk and the kernels'
event times chain_score_kernel / chain_band_kernel; hdf5).  PEPPER_AMD_POLISH_CHAIN=0 times round 4's host form."""
import json
import os
import shutil
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def make_fast(out, bases=16000000, coverage=60, seed=2027):
    from bench_variant_images import make_fast as mk
    return mk(out, bases, coverage, seed)


def run(data, thread_counts=(8,), regions_per_call=None, warm=True):
    from pepper_amd.polish import ImageGenerationUI as ui
    from pepper_amd.polish.make_images import make_images
    if regions_per_call:
        ui.UserInterfaceSupport.CHAIN_REGIONS = int(regions_per_call)
    info = json.load(open(os.path.join(data, "synth.json")))
    mb = info["genome_bases"] / 1e6
    bam, fa = os.path.join(data, "reads.bam"), os.path.join(data, "draft.fa")
    if warm:      # library load, workspaces grown to the job's call size, the file in the page cache -- not part of the rate
        make_images(bam, fa, None, os.path.join(data, "pimages_warm"), thread_counts[0])
        shutil.rmtree(os.path.join(data, "pimages_warm"), ignore_errors=True)
    runs = []
    for threads in thread_counts:
        tmp = os.path.join(data, "pimages_t%d" % threads)
        shutil.rmtree(tmp, ignore_errors=True)
        stages = {}
        t0 = time.perf_counter()
        make_images(bam, fa, None, tmp, threads, stats=stages)
        dt = time.perf_counter() - t0
        size = sum(os.path.getsize(os.path.join(tmp, f)) for f in os.listdir(tmp))
        counts = {k: int(stages.pop(k)) for k in ("pairs", "realigned", "cigar_ops", "rows", "proven_overflows") if k in stages}
        n_intervals = -(-info["genome_bases"] // 1000)
        runs.append({"threads": threads, "seconds": round(dt, 3), "mb_draft_per_s": round(mb / dt, 2),
                     "intervals_per_s": round(n_intervals / dt, 1), "reads_realigned_per_s": round(counts.get("realigned", 0) / dt, 1),
                     "counts": counts, "stage_seconds_summed_over_workers": {k: round(v, 2) for k, v in sorted(stages.items())},
                     "image_file_mb": round(size / 1e6, 1)})
        shutil.rmtree(tmp, ignore_errors=True)
    return {"metric": "polish make_images: Mb of draft per second",
            "data": "synthetic BAM %.1f Mb at %.0fx, %d records, %.2f GB (tools/synth_bam), intervals of 1 kb + 2 x 100" % (
                mb, info["coverage"], info["records"], info["bam_bytes"] / 1e9),
            "chain": os.environ.get("PEPPER_AMD_POLISH_CHAIN", "1") != "0", "regions_per_call": ui.UserInterfaceSupport.CHAIN_REGIONS,
            "runs": runs}


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(REPO, "tools"))
    if sys.argv[1] == "make_fast":
        print(json.dumps(make_fast(sys.argv[2], *(float(a) for a in sys.argv[3:5]))))
    else:
        counts = tuple(int(t) for t in sys.argv[3].split(",")) if len(sys.argv) > 3 else (8,)
        print(json.dumps(run(sys.argv[2], counts, int(sys.argv[4]) if len(sys.argv) > 4 else None)))

"""The two top entry points end to end on synthetic inputs (tools/synth_bam), one job each with the three steps' wall times:

  python tools/bench_e2e.py call_variant <dir> [genome_bases=256000000] [coverage=30] [runs=3]
      pepper_amd.variant.CallVariant.call_variant (pepper_variant call_variant, CallVariant.py:74-104):
      BAM + FASTA + checkpoint -> image HDF5 -> predictions HDF5 -> five VCFs
  python tools/bench_e2e.py polish <dir> [draft_bases=64000000] [coverage=60] [runs=3]
      pepper_amd.polish.polish.polish (pepper polish, polish.py:94-117): BAM + draft + checkpoint -> images -> predictions -> FASTA

Each prints one JSON line: the run with the median wall of `runs` (after one untimed run that loads the libraries, grows the
workspaces and leaves the input files in the page cache), every run's wall, the stage walls and the units per second."""
import glob
import json
import os
import shutil
import subprocess
import sys
import time
from types import SimpleNamespace

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))


def synth(work, bases, coverage, seed=2027):
    from pepper_amd import build
    tool = build.build_tools()
    os.makedirs(work, exist_ok=True)
    t0 = time.perf_counter()
    info = json.loads(subprocess.run([tool, work, str(int(bases)), str(coverage), str(seed)], check=True, capture_output=True, text=True).stdout)
    info["seconds"] = round(time.perf_counter() - t0, 2)
    return info


def checkpoint(path, kind):
    import torch
    from pepper_amd import synthetic
    sd = synthetic.variant_state_dict(seed=0) if kind == "variant" else synthetic.polish_state_dict(seed=0)
    torch.save(synthetic.checkpoint_dict({k: torch.from_numpy(v) for k, v in sd.items()}, hidden_size=128), path)


def median_run(runs, key="seconds"):
    order = sorted(runs, key=lambda r: r[key])
    return order[len(order) // 2]


def call_variant_job(work, bases, coverage, n_runs):
    from pepper_amd.hostinfo import usable_cpus
    from pepper_amd.variant.CallVariant import call_variant
    info = synth(work, bases, coverage)
    model = os.path.join(work, "variant.pkl")
    checkpoint(model, "variant")
    threads = max(1, usable_cpus())
    runs = []
    for k in range(n_runs + 1):
        out = os.path.join(work, "cv_out_%d" % k)
        shutil.rmtree(out, ignore_errors=True)
        walls, stages = {}, {}
        options = SimpleNamespace(
            bam=os.path.join(work, "reads.bam"), fasta=os.path.join(work, "draft.fa"), region=None, region_size=100000, threads=threads,
            train_mode=False, use_hp_info=False, include_supplementary=False, output_dir=out, min_mapq=1, min_snp_baseq=1, min_indel_baseq=1,
            snp_frequency=0.10, insert_frequency=0.15, delete_frequency=0.15, min_coverage_threshold=3, snp_candidate_frequency_threshold=0.10,
            indel_candidate_frequency_threshold=0.12, candidate_support_threshold=2, skip_indels=False, downsample_rate=1.0,
            model_path=model, batch_size=512, num_workers=0, gpu=True, device_ids="0", callers_per_gpu=1, quantized=False, dry=False,
            sample_name="SYN", allowed_multiallelics=4, snp_p_value=0.1, insert_p_value=0.25, delete_p_value=0.25, snp_p_value_in_lc=0.1,
            insert_p_value_in_lc=0.3, delete_p_value_in_lc=0.3, snp_q_cutoff=20, indel_q_cutoff=15, snp_q_cutoff_in_lc=20,
            indel_q_cutoff_in_lc=10, report_snp_above_freq=0, report_indel_above_freq=0, stage_walls=walls, stage_seconds=stages)
        t0 = time.perf_counter()
        image_dir, pred_dir, totals = call_variant(options)
        dt = time.perf_counter() - t0
        windows = 0
        from pepper_amd import h5
        for path in glob.glob(os.path.join(pred_dir, "*.hdf")):
            with h5.File(path) as f:
                for g in f.keys("predictions"):
                    windows += f.info("predictions/" + g + "/positions")[0][0]
        if k > 0:
            runs.append({"seconds": round(dt, 3), "stage_walls": {n: round(v, 3) for n, v in walls.items()}, "windows": windows,
                         "candidates_written": [int(t) for t in totals] if hasattr(totals, "__iter__") else int(totals)})
        shutil.rmtree(out, ignore_errors=True)
    mid = median_run(runs)
    longest = max(mid["stage_walls"].values())
    return {"metric": "call_variant end to end (BAM + FASTA + checkpoint -> 5 VCFs)", "value": round(info["genome_bases"] / 1e6 / mid["seconds"], 2),
            "unit": "Mb of reference/s", "seconds": mid["seconds"], "runs_seconds": [r["seconds"] for r in runs], "stage_walls": mid["stage_walls"],
            "wall_over_longest_stage": round(mid["seconds"] / longest, 3), "windows": mid["windows"],
            "windows_per_s": round(mid["windows"] / mid["seconds"], 1),
            "candidates_per_s_in_find_candidates": round(mid["windows"] / max(1e-9, mid["stage_walls"]["find_candidates"]), 1),
            "candidates_written": mid["candidates_written"], "threads": threads,
            "data": "synthetic BAM %.0f Mb at %.0fx, %d records, %.2f GB (tools/synth_bam), seeded random-init checkpoint" % (
                info["genome_bases"] / 1e6, info["coverage"], info["records"], info["bam_bytes"] / 1e9), "synth_seconds": info["seconds"]}


def polish_job(work, bases, coverage, n_runs):
    from pepper_amd.hostinfo import usable_cpus
    from pepper_amd.polish.polish import polish
    info = synth(work, bases, coverage)
    model = os.path.join(work, "polish.pkl")
    checkpoint(model, "polish")
    threads = max(1, usable_cpus())
    runs = []
    for k in range(n_runs + 1):
        out = os.path.join(work, "polish_out_%d" % k) + "/"
        shutil.rmtree(out, ignore_errors=True)
        walls = {}
        t0 = time.perf_counter()
        polish(os.path.join(work, "reads.bam"), os.path.join(work, "draft.fa"), out, threads, None, model, 512, True, "0", 0, stage_walls=walls)
        dt = time.perf_counter() - t0
        fasta = glob.glob(out + "*.fa")
        size = os.path.getsize(fasta[0]) if fasta else 0
        images = sum(os.path.getsize(p) for p in glob.glob(out + "images_*/*.hdf"))
        if k > 0:
            runs.append({"seconds": round(dt, 3), "stage_walls": {n: round(v, 3) for n, v in walls.items()}, "polished_fasta_bytes": size,
                         "image_file_mb": round(images / 1e6, 1)})
        shutil.rmtree(out, ignore_errors=True)
    mid = median_run(runs)
    longest = max(mid["stage_walls"].values())
    return {"metric": "polish end to end (BAM + draft + checkpoint -> polished FASTA)", "value": round(info["genome_bases"] / 1e6 / mid["seconds"], 2),
            "unit": "Mb of draft/s", "seconds": mid["seconds"], "runs_seconds": [r["seconds"] for r in runs], "stage_walls": mid["stage_walls"],
            "wall_over_longest_stage": round(mid["seconds"] / longest, 3), "polished_fasta_bytes": mid["polished_fasta_bytes"],
            "image_file_mb": mid["image_file_mb"], "threads": threads,
            "data": "synthetic BAM %.0f Mb at %.0fx, %d records, %.2f GB (tools/synth_bam), seeded random-init checkpoint" % (
                info["genome_bases"] / 1e6, info["coverage"], info["records"], info["bam_bytes"] / 1e9), "synth_seconds": info["seconds"]}


if __name__ == "__main__":
    kind, work = sys.argv[1], sys.argv[2]
    bases = float(sys.argv[3]) if len(sys.argv) > 3 else (256e6 if kind == "call_variant" else 64e6)
    coverage = float(sys.argv[4]) if len(sys.argv) > 4 else (30 if kind == "call_variant" else 60)
    n_runs = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    print(json.dumps((call_variant_job if kind == "call_variant" else polish_job)(work, bases, coverage, n_runs)))

"""The two top entry points end to end on synthetic inputs (tools/synth_bam), one job each with the three steps' wall times:

  python tools/bench_e2e.py call_variant <dir> [genome_bases=256000000] [coverage=30] [runs=3]
      pepper_amd.variant.CallVariant.call_variant (pepper_variant call_variant, CallVariant.py:74-104):
      BAM + FASTA + checkpoint -> image HDF5 -> predictions HDF5 -> five VCFs
  python tools/bench_e2e.py call_variant_fused <dir> ...     the same with options.fused_inference (pepper_amd/variant/fused.py)
  python tools/bench_e2e.py polish <dir> [draft_bases=64000000] [coverage=60] [runs=3]
  python tools/bench_e2e.py polish_fused <dir> ...           the same with fused_inference=True (pepper_amd/polish/fused.py)
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


def checkpoint(path, kind, reference_bias=0.0):
    """Seeded random-init weights; reference_bias leans the variant model's output layer towards the reference class (a random-init
    head calls every window a variant -- 2.3 M VCF records per 128 Mb --, a trained model a few per hundred: calibrate_bias)."""
    import torch
    from pepper_amd import synthetic
    sd = synthetic.variant_state_dict(seed=0, gain=2.0) if kind == "variant" else synthetic.polish_state_dict(seed=0)
    if kind == "variant":
        sd["output_layer_type.bias"] = sd["output_layer_type.bias"] + __import__("numpy").array([reference_bias, 0.0, 0.0], "float32")
    torch.save(synthetic.checkpoint_dict({k: torch.from_numpy(v) for k, v in sd.items()}, hidden_size=128), path)


REFERENCE_BIAS = 6.1      # calibrate_bias() on a 128 Mb / 30x job with the seed-0, gain-2 weights: ~2 % of the windows are called


def calibrate_bias(pred_dir, fraction=0.02, threshold=0.1):
    """The reference-class bias at which `fraction` of the warm-up run's windows keep a non-reference probability above the
    candidate finder's p-value threshold: softmax(z0 + b, z1, z2) for every window from the probabilities written with b = 0."""
    import numpy as np
    from pepper_amd import h5
    rows = []
    for path in sorted(glob.glob(os.path.join(pred_dir, "*.hdf"))):
        with h5.File(path) as f:
            for g in f.keys("predictions")[:400]:
                rows.append(np.asarray(f["predictions/" + g + "/base_prediction"], np.float64))
    p = np.clip(np.concatenate(rows), 1e-300, 1.0)
    r1, r2 = p[:, 1] / p[:, 0], p[:, 2] / p[:, 0]
    lo, hi = -10.0, 40.0
    for _ in range(60):
        b = 0.5 * (lo + hi)
        a1, a2 = r1 * np.exp(-b), r2 * np.exp(-b)
        called = (np.maximum(a1, a2) / (1.0 + a1 + a2) > threshold).mean()
        lo, hi = (b, hi) if called > fraction else (lo, b)
    return 0.5 * (lo + hi)


def median_run(runs, key="seconds"):
    order = sorted(runs, key=lambda r: r[key])
    return order[len(order) // 2]


def call_variant_job(work, bases, coverage, n_runs, fused=False):
    from pepper_amd.hostinfo import usable_cpus
    from pepper_amd.variant.CallVariant import call_variant
    info = synth(work, bases, coverage)
    model = os.path.join(work, "variant.pkl")
    bias = REFERENCE_BIAS
    checkpoint(model, "variant", reference_bias=bias)
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
            indel_q_cutoff_in_lc=10, report_snp_above_freq=0, report_indel_above_freq=0, stage_walls=walls, stage_seconds=stages,
            fused_inference=fused)
        t0 = time.perf_counter()
        image_dir, pred_dir, totals = call_variant(options)
        dt = time.perf_counter() - t0
        if k == 0 and os.environ.get("PEPPER_AMD_E2E_CALIBRATE") == "1":
            # what REFERENCE_BIAS was found with: the bias at which ~2 % of the warm-up run's windows are called
            print(json.dumps({"calibrated_reference_bias": calibrate_bias(pred_dir)}), file=sys.stderr)
        windows = 0
        from pepper_amd import h5
        for path in glob.glob(os.path.join(pred_dir, "*.hdf")):
            with h5.File(path) as f:
                for g in f.keys("predictions"):
                    windows += f.info("predictions/" + g + "/positions")[0][0]
        if k > 0:
            runs.append({"seconds": round(dt, 3), "stage_walls": {n: round(v, 3) for n, v in walls.items()}, "windows": windows,
                         "image_stage_seconds_summed_over_workers": {n: round(v, 2) for n, v in sorted(stages.items()) if n != "inflated_bytes"},
                         "candidates_written": [int(t) for t in totals] if hasattr(totals, "__iter__") else int(totals)})
        shutil.rmtree(out, ignore_errors=True)
    mid = median_run(runs)
    longest = max(mid["stage_walls"].get(n, 0.0) for n in ("make_images", "run_inference", "find_candidates"))
    return {"metric": "call_variant end to end (BAM + FASTA + checkpoint -> 5 VCFs)" + (", images and inference fused" if fused else ""), "value": round(info["genome_bases"] / 1e6 / mid["seconds"], 2),
            "unit": "Mb of reference/s", "seconds": mid["seconds"], "runs_seconds": [r["seconds"] for r in runs], "stage_walls": mid["stage_walls"],
            "wall_over_longest_stage": round(mid["seconds"] / longest, 3), "windows": mid["windows"],
            "runs_stage_walls": [r["stage_walls"] for r in runs],
            "image_stage_seconds_summed_over_workers": mid["image_stage_seconds_summed_over_workers"],
            "windows_per_s": round(mid["windows"] / mid["seconds"], 1),
            "candidates_per_s_in_find_candidates": round(mid["windows"] / max(1e-9, mid["stage_walls"]["find_candidates"]), 1),
            "candidates_written": mid["candidates_written"], "threads": threads,
            "model": "seeded random init (gain 2), output layer leaned towards the reference class by %.2f so that ~2 %% of the windows are called" % bias,
            "data": "synthetic BAM %.0f Mb at %.0fx, %d records, %.2f GB (tools/synth_bam), seeded random-init checkpoint" % (
                info["genome_bases"] / 1e6, info["coverage"], info["records"], info["bam_bytes"] / 1e9), "synth_seconds": info["seconds"]}


def polish_job(work, bases, coverage, n_runs, fused=False):
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
        polish(os.path.join(work, "reads.bam"), os.path.join(work, "draft.fa"), out, threads, None, model, 512, True, "0", 0, stage_walls=walls,
               fused_inference=fused)
        dt = time.perf_counter() - t0
        fasta = glob.glob(out + "*.fa")
        size = os.path.getsize(fasta[0]) if fasta else 0
        images = sum(os.path.getsize(p) for p in glob.glob(out + "images_*/*.hdf"))
        if k > 0:
            stages = walls.pop("image_stage_seconds_summed_over_workers", None) or {}
            runs.append({"seconds": round(dt, 3), "stage_walls": {n: round(v, 3) for n, v in walls.items()}, "polished_fasta_bytes": size,
                         "image_stage_seconds_summed_over_workers": {n: round(v, 2) for n, v in sorted(stages.items()) if isinstance(v, float)},
                         "image_file_mb": round(images / 1e6, 1)})
        shutil.rmtree(out, ignore_errors=True)
    mid = median_run(runs)
    longest = max(mid["stage_walls"].values())
    return {"metric": "polish end to end (BAM + draft + checkpoint -> polished FASTA)" + (", images and inference fused" if fused else ""), "value": round(info["genome_bases"] / 1e6 / mid["seconds"], 2),
            "unit": "Mb of draft/s", "seconds": mid["seconds"], "runs_seconds": [r["seconds"] for r in runs], "stage_walls": mid["stage_walls"],
            "wall_over_longest_stage": round(mid["seconds"] / longest, 3), "polished_fasta_bytes": mid["polished_fasta_bytes"],
            "image_stage_seconds_summed_over_workers": mid["image_stage_seconds_summed_over_workers"],
            "runs_stage_walls": [r["stage_walls"] for r in runs],
            "image_file_mb": mid["image_file_mb"], "threads": threads,
            "data": "synthetic BAM %.0f Mb at %.0fx, %d records, %.2f GB (tools/synth_bam), seeded random-init checkpoint" % (
                info["genome_bases"] / 1e6, info["coverage"], info["records"], info["bam_bytes"] / 1e9), "synth_seconds": info["seconds"]}


if __name__ == "__main__":
    kind, work = sys.argv[1], sys.argv[2]
    bases = float(sys.argv[3]) if len(sys.argv) > 3 else (256e6 if kind.startswith("call_variant") else 64e6)
    coverage = float(sys.argv[4]) if len(sys.argv) > 4 else (30 if kind.startswith("call_variant") else 60)
    n_runs = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    if kind == "call_variant_fused":
        print(json.dumps(call_variant_job(work, bases, coverage, n_runs, fused=True)))
    elif kind == "polish_fused":
        print(json.dumps(polish_job(work, bases, coverage, n_runs, fused=True)))
    else:
        print(json.dumps((call_variant_job if kind == "call_variant" else polish_job)(work, bases, coverage, n_runs)))

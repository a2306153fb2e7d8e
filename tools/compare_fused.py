"""Probabilities of the fused and the three-step call_variant on the same synthetic job, window by window (a diagnostic:
python tools/compare_fused.py <dir> [bases] [coverage])."""
import glob
import json
import os
import sys
from types import SimpleNamespace

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import bench_e2e  # noqa: E402


def preds(directory):
    from pepper_amd import h5
    out = {}
    for path in sorted(glob.glob(os.path.join(directory, "*.hdf"))):
        with h5.File(path) as f:
            for g in f.keys("predictions"):
                base = "predictions/" + g + "/"
                pos, cand, p = f[base + "positions"].tolist(), f[base + "candidates"].tolist(), f[base + "base_prediction"]
                for k in range(len(pos)):
                    out[(pos[k], cand[k][0])] = p[k]
    return out


def main():
    from pepper_amd.variant.CallVariant import call_variant
    work = sys.argv[1]
    bases, cov = float(sys.argv[2]) if len(sys.argv) > 2 else 8e6, float(sys.argv[3]) if len(sys.argv) > 3 else 30
    bench_e2e.synth(work, bases, cov)
    model = os.path.join(work, "variant.pkl")
    bench_e2e.checkpoint(model, "variant", reference_bias=float(sys.argv[4]) if len(sys.argv) > 4 else 6.0)
    res = {}
    for fused in (False, True):
        out = os.path.join(work, "out_%d" % fused)
        o = SimpleNamespace(
            bam=os.path.join(work, "reads.bam"), fasta=os.path.join(work, "draft.fa"), region=None, region_size=100000, threads=16,
            train_mode=False, use_hp_info=False, include_supplementary=False, output_dir=out, min_mapq=1, min_snp_baseq=1, min_indel_baseq=1,
            snp_frequency=0.10, insert_frequency=0.15, delete_frequency=0.15, min_coverage_threshold=3, snp_candidate_frequency_threshold=0.10,
            indel_candidate_frequency_threshold=0.12, candidate_support_threshold=2, skip_indels=False, downsample_rate=1.0,
            model_path=model, batch_size=512, num_workers=0, gpu=True, device_ids="0", callers_per_gpu=1, quantized=False, dry=False,
            sample_name="SYN", allowed_multiallelics=4, snp_p_value=0.1, insert_p_value=0.25, delete_p_value=0.25, snp_p_value_in_lc=0.1,
            insert_p_value_in_lc=0.3, delete_p_value_in_lc=0.3, snp_q_cutoff=20, indel_q_cutoff=15, snp_q_cutoff_in_lc=20,
            indel_q_cutoff_in_lc=10, report_snp_above_freq=0, report_indel_above_freq=0, fused_inference=fused)
        _, pred_dir, totals = call_variant(o)
        res[fused] = (preds(pred_dir), totals)
    a, b = res[False][0], res[True][0]
    assert a.keys() == b.keys(), (len(a), len(b))
    keys = list(a)
    pa, pb = np.array([a[k] for k in keys]), np.array([b[k] for k in keys])
    d = np.abs(pa - pb).max(axis=1)
    la, lb = np.log(np.clip(pa, 1e-300, 1)), np.log(np.clip(pb, 1e-300, 1))
    dl = np.abs((la - la[:, :1]) - (lb - lb[:, :1])).max(axis=1)
    print(json.dumps({"windows": len(keys), "max_abs_prob_diff": float(d.max()), "windows_differing": int((d > 0).sum()),
                      "max_logit_diff": float(dl.max()), "p99.9_logit_diff": float(np.quantile(dl, 0.999)),
                      "totals_three_step": list(map(int, res[False][1])), "totals_fused": list(map(int, res[True][1]))}))


if __name__ == "__main__":
    main()

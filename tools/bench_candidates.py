"""Candidate finding + VCF writing rate (host Python, as in the reference): predictions HDF5 -> five VCFs.
python tools/bench_candidates.py [n_candidates]   (CPU only)"""
import json
import os
import shutil
import sys
import tempfile
import time
from types import SimpleNamespace

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pepper_amd.variant.DataStorePredict import DataStore  # noqa: E402
from pepper_amd.variant.FindCandidates import process_candidates  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    rng = np.random.default_rng(1)
    tmp = tempfile.mkdtemp()
    try:
        length = 20 * n + 1000
        ref = "".join("ACGT"[k] for k in rng.integers(0, 4, length))
        fa = os.path.join(tmp, "ref.fa")
        with open(fa, "w") as fh:
            fh.write(">chr20\n" + "\n".join(ref[i:i + 60] for i in range(0, length, 60)) + "\n")
        pred_dir = os.path.join(tmp, "pred")
        os.makedirs(pred_dir)
        store = DataStore(os.path.join(pred_dir, "pepper_prediction_0.hdf"), "w")
        positions = np.sort(rng.choice(np.arange(100, length - 100), n, replace=False)).astype(np.int32)
        for b, s in enumerate(range(0, n, 512)):
            e = min(n, s + 512)
            m = e - s
            kind = rng.integers(0, 3, m)
            cands = []
            for k, p in zip(kind, positions[s:e]):
                r = ref[int(p)]
                if k == 0:
                    cands.append(["1" + "ACGT"[("ACGT".index(r) + 1) % 4]])
                elif k == 1:
                    cands.append(["2" + r + "AC"])
                else:
                    cands.append(["3" + ref[int(p):int(p) + 3]])
            probs = rng.dirichlet([1.0, 1.0, 1.0], m)
            store.write_prediction(b, ["chr20"] * m, positions[s:e], rng.integers(20, 80, m).astype(np.uint8),
                                   np.array(cands, dtype=object), rng.integers(5, 40, (m, 1)).astype(np.uint8), probs)
        store.close()
        options = SimpleNamespace(fasta=fa, threads=4, sample_name="SYN", allowed_multiallelics=4, snp_p_value=0.1,
                                  insert_p_value=0.25, delete_p_value=0.25, snp_p_value_in_lc=0.1, insert_p_value_in_lc=0.3,
                                  delete_p_value_in_lc=0.3, snp_q_cutoff=20, indel_q_cutoff=15, snp_q_cutoff_in_lc=20,
                                  indel_q_cutoff_in_lc=10, report_snp_above_freq=0, report_indel_above_freq=0)
        t0 = time.perf_counter()
        process_candidates(options, pred_dir, os.path.join(tmp, "out"))
        dt = time.perf_counter() - t0
        print(json.dumps({"metric": "process_candidates: predictions HDF5 -> 5 VCFs (host Python)", "candidates": n,
                          "seconds": round(dt, 2), "candidates_per_s": round(n / dt)}))
    finally:
        shutil.rmtree(tmp)


if __name__ == "__main__":
    main()

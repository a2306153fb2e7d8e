"""End-to-end inference step on the GPU through the reference's entry points: images HDF5 files +
checkpoint -> run_inference / call_consensus -> predictions HDF5, compared with the oracle."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import models_np
from pepper_amd import h5, synthetic

pytestmark = pytest.mark.gpu


def _variant_images(dirpath, sizes_per_file):
    from pepper_amd.variant.DataStore import DataStore
    all_images = []
    seed = 500
    for fi, groups in enumerate(sizes_per_file):
        with DataStore(os.path.join(dirpath, f"pepper_variant_images_thread_{fi}.hdf5"), "w") as ds:
            for gi, n in enumerate(groups):
                x = synthetic.variant_windows(n, seed=seed)
                seed += 1
                start = 100000 * gi
                ds.write_summary(f"chr20_{start}_{start + 100000}", ["chr20"] * n, list(range(start, start + n)),
                                 [30] * n, [[f"1{'ACGT'[i % 4]}"] for i in range(n)], [[7]] * n, x.tolist(),
                                 [0] * n, [0] * n, False)
                all_images.append((fi, gi, x))
    return all_images


def test_run_inference_end_to_end(tmp_path):
    from pepper_amd.variant.RunInference import run_inference
    img_dir, out_dir = tmp_path / "images", tmp_path / "pred"
    img_dir.mkdir()
    groups = _variant_images(str(img_dir), [[700, 0, 13], [5]])     # ragged, one empty region
    (img_dir / "notes.txt").write_text("not an image file")          # ignored: suffix rule
    sd = synthetic.variant_state_dict(seed=41, gain=2.0)
    ckpt = synthetic.checkpoint_dict({k: torch.from_numpy(v) for k, v in sd.items()}, hidden_size=128)
    model_path = str(tmp_path / "model.pkl")
    torch.save(ckpt, model_path)
    options = SimpleNamespace(model_path=model_path, batch_size=512, num_workers=0, use_hp_info=False, gpu=True,
                              device_ids="0", callers_per_gpu=4, threads=8, quantized=False, dry=False)
    run_inference(options, str(img_dir), str(out_dir))
    out_file = out_dir / "pepper_prediction.hdf"
    assert out_file.exists()
    x_all = np.concatenate([g[2] for g in groups])
    ref = models_np.variant_forward(sd, x_all)
    with h5.File(str(out_file)) as f:
        batches = sorted(f.keys("predictions"), key=lambda s: int(s.split("_")[1]))
        # file 0: 713 windows -> batch_0 (512), batch_1 (201); file 1: 5 windows -> batch_2
        assert batches == ["batch_0", "batch_1", "batch_2"]
        got = np.concatenate([f[f"predictions/{b}/base_prediction"] for b in batches])
        pos = np.concatenate([f[f"predictions/{b}/positions"] for b in batches])
        cand = np.concatenate([f[f"predictions/{b}/candidates"] for b in batches])
        assert f["predictions/batch_0/base_prediction"].dtype == np.float64
        assert f["predictions/batch_0/contigs"][0] == b"chr20"
    assert got.shape == (718, 3) and np.abs(got - ref).max() < 1e-4
    assert (got.argmax(1) == ref.argmax(1)).mean() > 0.995
    assert pos[:3].tolist() == [0, 1, 2] and pos[700] == 200000 and cand[1, 0] == "1C"

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        options.gpu = False
        run_inference(options, str(img_dir), str(out_dir))


def test_call_consensus_end_to_end(tmp_path):
    from pepper_amd.polish.DataStore import DataStore
    from pepper_amd.polish.call_consensus import call_consensus
    img_dir, out_dir = tmp_path / "images", tmp_path / "pred"
    img_dir.mkdir()
    chunks = synthetic.polish_chunks(5, seed=900)
    with DataStore(str(img_dir / "pepper_images_thread_0.hdf"), "w") as ds:
        for cid in range(5):
            pos = [(2000 + i, 0) for i in range(1000)]
            ds.write_summary(("contig_7", 2000, 3000), chunks[cid].tolist(), [0] * 1000, pos, list(range(1000)), cid,
                             f"contig_7_2000_3000_{cid}")
    sd = synthetic.polish_state_dict(seed=42, gain=2.0)
    model_path = str(tmp_path / "polish.pkl")
    torch.save(synthetic.checkpoint_dict({k: torch.from_numpy(v) for k, v in sd.items()}, hidden_size=128), model_path)
    call_consensus(str(img_dir), model_path, 3, 0, str(out_dir), "0", True, 4)
    labels, phred, inter = models_np.polish_predict_chunks(sd, chunks, 128, return_intermediates=True)
    with h5.File(str(out_dir / "pepper_prediction_0.hdf")) as f:
        base = "predictions/contig_7/contig_7-2000-3000/"
        assert f[base + "contig_start"] == 2000 and f[base + "contig_end"] == 3000
        for cid in range(5):
            got_b, got_p = f[base + f"{cid}/bases"], f[base + f"{cid}/phred_score"]
            assert got_b.dtype == np.uint8 and got_b.shape == (1000,)
            assert (got_b == labels[cid]).mean() > 0.999
            assert (got_p == phred[cid]).mean() > 0.99
            assert f[base + f"{cid}/position"].shape == (1000, 2)

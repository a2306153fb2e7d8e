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


def _variant_images(dirpath, sizes_per_file, file_stride=0):
    from pepper_amd.variant.DataStore import DataStore
    all_images = []
    seed = 500
    for fi, groups in enumerate(sizes_per_file):
        with DataStore(os.path.join(dirpath, f"pepper_variant_images_thread_{fi}.hdf5"), "w") as ds:
            for gi, n in enumerate(groups):
                x = synthetic.variant_windows(n, seed=seed)
                seed += 1
                start = file_stride * fi + 100000 * gi
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
                              device_ids="0", callers_per_gpu=1, threads=8, quantized=False, dry=False)
    out_dir.mkdir()
    (out_dir / "pepper_prediction_3.hdf").write_bytes(b"left over from a run with four callers")
    run_inference(options, str(img_dir), str(out_dir))
    assert sorted(os.listdir(out_dir)) == ["pepper_prediction.hdf"]      # stale per-rank files are removed
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


def _read_predictions(path):
    """{(contig, position, candidate): probabilities} of one predictions file, plus the batch names in order."""
    out = {}
    with h5.File(path) as f:
        batches = sorted(f.keys("predictions"), key=lambda s: int(s.split("_")[1]))
        for b in batches:
            probs = f[f"predictions/{b}/base_prediction"]
            pos = f[f"predictions/{b}/positions"]
            cand = f[f"predictions/{b}/candidates"]
            contigs = f[f"predictions/{b}/contigs"]
            for i in range(len(pos)):
                out[(bytes(contigs[i]), int(pos[i]), str(cand[i, 0]))] = np.array(probs[i])
    return out, batches


def test_run_inference_multi_process_and_loader_workers(tmp_path):
    """The N-caller leg of RunInference.distributed_gpu (mp.spawn, one process per caller, one weight broadcast, file
    shards, pepper_prediction_<rank>.hdf) with two callers sharing GPU 0 -- --device_ids "0,0", which the reference's
    list semantics allow (RunInference.py:41-60); the broadcast then runs over gloo because RCCL refuses two ranks on
    one device --
    and options.num_workers = 2 (two reader / writer process lanes): both must reproduce the single-process records
    exactly."""
    from pepper_amd.variant.RunInference import run_inference
    img_dir = tmp_path / "images"
    img_dir.mkdir()
    _variant_images(str(img_dir), [[600, 40], [300], [77, 3], [150]], file_stride=10_000_000)    # unique positions
    sd = synthetic.variant_state_dict(seed=43, gain=2.0)
    model_path = str(tmp_path / "model.pkl")
    torch.save(synthetic.checkpoint_dict({k: torch.from_numpy(v) for k, v in sd.items()}, hidden_size=128, module_prefix=True),
               model_path)

    def run(out, **over):
        opts = dict(model_path=model_path, batch_size=256, num_workers=0, use_hp_info=False, gpu=True, device_ids="0",
                    callers_per_gpu=1, threads=4, quantized=False, dry=False)
        opts.update(over)
        run_inference(SimpleNamespace(**opts), str(img_dir), str(tmp_path / out))
        return tmp_path / out

    single = run("single")
    want, want_batches = _read_predictions(str(single / "pepper_prediction.hdf"))
    assert len(want) == 600 + 40 + 300 + 77 + 3 + 150

    two = run("two", device_ids="0,0", callers_per_gpu=4)        # callers_per_gpu is not multiplied in
    assert sorted(os.listdir(two)) == ["pepper_prediction_0.hdf", "pepper_prediction_1.hdf"]
    got = {}
    for r in (0, 1):
        part, batches = _read_predictions(str(two / f"pepper_prediction_{r}.hdf"))
        assert batches == [f"batch_{i}" for i in range(len(batches))] and len(part) > 0
        assert not (set(part) & set(got))
        got.update(part)
    assert set(got) == set(want)
    assert all(np.array_equal(got[k], want[k]) for k in want)

    # num_workers = 2: two lanes (reader + writer process each, pepper_amd/hostpipe.py), one prediction file per lane with
    # its own batch numbering; the GPU loop stays in this process
    lanes = run("lanes", num_workers=2)
    assert sorted(os.listdir(lanes)) == ["pepper_prediction_0.hdf", "pepper_prediction_1.hdf"]
    got2 = {}
    for k in (0, 1):
        part, batches = _read_predictions(str(lanes / f"pepper_prediction_{k}.hdf"))
        assert batches == [f"batch_{i}" for i in range(len(batches))] and len(part) > 0
        assert not (set(part) & set(got2))
        got2.update(part)
    assert set(got2) == set(want)
    assert all(np.array_equal(got2[k], want[k]) for k in want)


def test_call_consensus_two_callers_share_gpu(tmp_path):
    """Polish counterpart: device_ids "0,0" = two processes on GPU 0 (predict_distributed_gpu's mp.spawn leg), files
    sharded between them; per-chunk outputs equal the one-process run."""
    from pepper_amd.polish.DataStore import DataStore
    from pepper_amd.polish.call_consensus import call_consensus
    img_dir = tmp_path / "images"
    img_dir.mkdir()
    chunks = synthetic.polish_chunks(7, seed=901)
    for fi, ids in enumerate(([0, 1, 2, 3], [4, 5, 6])):
        with DataStore(str(img_dir / f"pepper_images_thread_{fi}.hdf"), "w") as ds:
            for cid in ids:
                ds.write_summary((f"contig_{fi}", 2000, 3000), chunks[cid].tolist(), [0] * 1000, list(range(2000, 3000)),
                                 [0] * 1000, cid, f"contig_{fi}_2000_3000_{cid}")
    sd = synthetic.polish_state_dict(seed=44, gain=2.0)
    model_path = str(tmp_path / "polish.pkl")
    torch.save(synthetic.checkpoint_dict({k: torch.from_numpy(v) for k, v in sd.items()}, hidden_size=128), model_path)
    call_consensus(str(img_dir), model_path, 128, 0, str(tmp_path / "one"), "0", True, 4)
    call_consensus(str(img_dir), model_path, 128, 0, str(tmp_path / "two"), "0,0", True, 4)
    assert sorted(os.listdir(tmp_path / "two")) == ["pepper_prediction_0.hdf", "pepper_prediction_1.hdf"]

    def read(dirpath):
        out = {}
        for name in sorted(os.listdir(dirpath)):
            with h5.File(str(dirpath / name)) as f:
                for contig in f.keys("predictions"):
                    base = f"predictions/{contig}/{contig}-2000-3000"
                    for cid in f.keys(base):
                        if cid.isdigit():
                            out[(contig, int(cid))] = (np.array(f[f"{base}/{cid}/bases"]), np.array(f[f"{base}/{cid}/phred_score"]))
        return out
    one, two = read(tmp_path / "one"), read(tmp_path / "two")
    assert len(one) == 7 and set(one) == set(two)
    for k in one:
        assert np.array_equal(one[k][0], two[k][0]) and np.array_equal(one[k][1], two[k][1])
    # num_workers = 2: two lanes around one caller's GPU loop
    call_consensus(str(img_dir), model_path, 128, 2, str(tmp_path / "lanes"), "0", True, 4)
    assert sorted(os.listdir(tmp_path / "lanes")) == ["pepper_prediction_0_0.hdf", "pepper_prediction_0_1.hdf"]
    lanes = read(tmp_path / "lanes")
    assert set(lanes) == set(one)
    for k in one:
        assert np.array_equal(one[k][0], lanes[k][0]) and np.array_equal(one[k][1], lanes[k][1])


def test_call_consensus_end_to_end(tmp_path):
    from pepper_amd.polish.DataStore import DataStore
    from pepper_amd.polish.call_consensus import call_consensus
    img_dir, out_dir = tmp_path / "images", tmp_path / "pred"
    img_dir.mkdir()
    chunks = synthetic.polish_chunks(5, seed=900)
    with DataStore(str(img_dir / "pepper_images_thread_0.hdf"), "w") as ds:
        for cid in range(5):
            pos = [2000 + i for i in range(1000)]        # position, index = zip(*positions[i]) (ImageGenerationUI.py:198)
            ds.write_summary(("contig_7", 2000, 3000), chunks[cid].tolist(), [0] * 1000, pos, [i % 2 for i in range(1000)], cid,
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
            assert f[base + f"{cid}/position"].tolist() == list(range(2000, 3000))
            assert f[base + f"{cid}/index"].tolist() == [i % 2 for i in range(1000)]


class _FakeFasta(object):
    def __init__(self, contigs):
        self.contigs = contigs

    def get_chromosome_names(self):
        return list(self.contigs)

    def get_chromosome_sequence_length(self, name):
        return len(self.contigs[name])

    def get_reference_sequence(self, name, start, end):
        return self.contigs[name][start:end]


class _FakeBam(object):
    """get_reads without htslib's region clipping (that belongs to the out-of-scope BAM reader)."""

    def __init__(self, reads_by_contig):
        self.reads = reads_by_contig

    def get_chromosome_sequence_names(self):
        return list(self.reads)

    def get_reads(self, name, start, end, include_supplementary, min_mapq, min_baseq):
        out = []
        for r in self.reads[name]:
            ref_len = sum(n for o, n in r["cigar"] if o in (0, 2, 3, 6, 7, 8))
            if r["pos"] <= end and r["pos"] + ref_len >= start and r["mapq"] >= min_mapq:
                out.append(_as_read(r))
        return out


def test_images_to_predictions_plumbing(tmp_path):
    """BASELINE configs[0] plumbing on synthetic reads: pileup -> GPU encoder -> images HDF5 ->
    GPU inference -> predictions HDF5, against the oracle encoder + oracle model."""
    import pileup_utils as pu
    from test_gpu_encoder import R
    from pepper_amd.variant.ImageGenerationUI import ImageGenerationUtils
    from pepper_amd.variant.RunInference import run_inference
    rng = np.random.default_rng(2024)
    ref = pu.random_reference(rng, 6000)
    sites = {int(p): ("ACGT"[(("ACGT".index(ref[p]) + 1) % 4)], 0.5) for p in rng.choice(np.arange(300, 5700), 25, replace=False)}
    indels = {1111: ("I", "ACG", 0.6), 2222: ("D", 4, 0.7), 3333: ("D", 40, 0.5)}
    reads = pu.simulate_reads(rng, ref, 0, n_reads=500, read_len=(300, 900), snp_sites=sites, indel_sites=indels)
    options = SimpleNamespace(
        bam="fake.bam", fasta="fake.fa", region="chr20:0-5999", region_size=2000, threads=2, train_mode=False,
        use_hp_info=False, image_output_directory=str(tmp_path / "images"), include_supplementary=False,
        min_mapq=1, min_snp_baseq=1, min_indel_baseq=1, snp_frequency=0.10, insert_frequency=0.15,
        delete_frequency=0.15, min_coverage_threshold=3, snp_candidate_frequency_threshold=0.10,
        indel_candidate_frequency_threshold=0.12, candidate_support_threshold=2, skip_indels=False,
        downsample_rate=1.0,
        bam_handler_factory=lambda path: _FakeBam({"chr20": reads}),
        fasta_handler_factory=lambda path: _FakeFasta({"chr20": ref}))
    ImageGenerationUtils.generate_images(options)
    files = sorted(os.listdir(options.image_output_directory))
    assert len(files) == 2 and all(f.endswith(".hdf5") for f in files)

    # oracle side: same intervals, same read selection, oracle encoder
    oracle = pu.load_restatement()
    want = {}
    for (start, end) in ((0, 2000), (2000, 4000), (4000, 5999)):
        rs, re_ = max(0, start - 100), end + 100
        sel = [d for d in reads if d["pos"] <= re_ and d["pos"] + sum(n for o, n in d["cigar"] if o in (0, 2, 3, 6, 7, 8)) >= rs and d["mapq"] >= 1]
        pile = pu.FlatPileup(rs, re_, ref[rs:re_ + 1], sel)
        want[f"chr20_{start}_{end}"] = pu.run_variant(oracle, pile, pu.make_params(start, end))
    got_imgs = {}
    for fn in files:
        with h5.File(os.path.join(options.image_output_directory, fn)) as f:
            for name in f.keys("summaries"):
                got_imgs[name] = (f[f"summaries/{name}/images"], f[f"summaries/{name}/positions"],
                                  f[f"summaries/{name}/candidates"][:, 0].tolist())
    assert sorted(got_imgs) == sorted(k for k, v in want.items() if len(v["candidates"]))
    total = 0
    for name, (img, pos, cand) in got_imgs.items():
        w = want[name]
        assert cand == w["candidates"] and pos.tolist() == w["positions"].tolist()
        assert np.array_equal(img, w["images"].astype(np.int64).astype(np.int8))
        total += len(cand)
    assert total > 20

    sd = synthetic.variant_state_dict(seed=43, gain=2.0)
    model_path = str(tmp_path / "model.pkl")
    torch.save(synthetic.checkpoint_dict({k: torch.from_numpy(v) for k, v in sd.items()}, hidden_size=128), model_path)
    opts = SimpleNamespace(model_path=model_path, batch_size=512, num_workers=0, use_hp_info=False, gpu=True,
                           device_ids="0", callers_per_gpu=4, threads=8, quantized=False, dry=False)
    run_inference(opts, options.image_output_directory, str(tmp_path / "pred"))
    with h5.File(str(tmp_path / "pred" / "pepper_prediction.hdf")) as f:
        got = np.concatenate([f[f"predictions/{b}/base_prediction"] for b in
                              sorted(f.keys("predictions"), key=lambda s: int(s.split("_")[1]))])
    # run_inference reads the image files in sorted order and their groups in name order
    ordered = []
    for fn in files:
        with h5.File(os.path.join(options.image_output_directory, fn)) as f:
            for name in f.keys("summaries"):
                ordered.append(f[f"summaries/{name}/images"])
    ref_probs = models_np.variant_forward(sd, np.concatenate(ordered))
    assert got.shape == ref_probs.shape and np.abs(got - ref_probs).max() < 1e-4


def _as_read(d):
    from test_gpu_encoder import R
    return R(d)


def test_call_variant_vcf_identity(tmp_path):
    """BASELINE success criterion ("identical candidate VCF"): pileup -> GPU encoder -> GPU inference
    -> candidate finder -> VCF through call_variant, against the VCF the candidate finder writes from
    the ORACLE model's predictions on the oracle encoder's images."""
    import pileup_utils as pu
    from pepper_amd.variant import bgzf
    from pepper_amd.variant.CallVariant import call_variant
    from pepper_amd.variant.DataStorePredict import DataStore as PredStore
    from pepper_amd.variant.FindCandidates import process_candidates
    rng = np.random.default_rng(77)
    ref = pu.random_reference(rng, 5000)
    ref = ref[:2500] + "AAAAAAAA" + ref[2508:]            # a homopolymer for the low-complexity rules
    sites = {int(p): ("ACGT"[(("ACGT".index(ref[p]) + 1) % 4)], 0.5) for p in rng.choice(np.arange(200, 4800), 30, replace=False)}
    indels = {900: ("I", "TG", 0.6), 1800: ("D", 3, 0.7), 2503: ("I", "A", 0.5), 3700: ("D", 12, 0.5)}
    reads = pu.simulate_reads(rng, ref, 0, n_reads=450, read_len=(300, 900), snp_sites=sites, indel_sites=indels)
    sd = synthetic.variant_state_dict(seed=91, gain=2.5)
    model_path = str(tmp_path / "model.pkl")
    torch.save(synthetic.checkpoint_dict({k: torch.from_numpy(v) for k, v in sd.items()}, hidden_size=128), model_path)
    fasta = _FakeFasta({"chr20": ref})
    options = SimpleNamespace(
        bam="fake.bam", fasta="fake.fa", region="chr20:0-4999", region_size=2500, threads=1, train_mode=False,
        use_hp_info=False, include_supplementary=False, output_dir=str(tmp_path / "out"),
        min_mapq=1, min_snp_baseq=1, min_indel_baseq=1, snp_frequency=0.10, insert_frequency=0.15,
        delete_frequency=0.15, min_coverage_threshold=3, snp_candidate_frequency_threshold=0.10,
        indel_candidate_frequency_threshold=0.12, candidate_support_threshold=2, skip_indels=False,
        downsample_rate=1.0,
        model_path=model_path, batch_size=256, num_workers=0, gpu=True, device_ids="0", callers_per_gpu=1,
        quantized=False, dry=False, sample_name="SYN", allowed_multiallelics=4,
        snp_p_value=0.1, insert_p_value=0.25, delete_p_value=0.25, snp_p_value_in_lc=0.1,
        insert_p_value_in_lc=0.3, delete_p_value_in_lc=0.3, snp_q_cutoff=20, indel_q_cutoff=15,
        snp_q_cutoff_in_lc=20, indel_q_cutoff_in_lc=10, report_snp_above_freq=0, report_indel_above_freq=0,
        bam_handler_factory=lambda path: _FakeBam({"chr20": reads}),
        fasta_handler_factory=lambda path: fasta)
    image_dir, pred_dir, totals = call_variant(options)
    assert totals[0] > 10

    # oracle leg: oracle encoder images -> oracle model -> same candidate finder
    oracle = pu.load_restatement()
    contigs, positions, depths, cands, freqs, images = [], [], [], [], [], []
    for (start, end) in ((0, 2500), (2500, 4999)):
        rs, re_ = max(0, start - 100), end + 100
        sel = [d for d in reads if d["pos"] <= re_ and d["pos"] + sum(n for o, n in d["cigar"] if o in (0, 2, 3, 6, 7, 8)) >= rs and d["mapq"] >= 1]
        res = pu.run_variant(oracle, pu.FlatPileup(rs, re_, ref[rs:re_ + 1], sel), pu.make_params(start, end))
        n = len(res["candidates"])
        contigs += ["chr20"] * n
        positions += res["positions"].tolist()
        depths += res["depths"].tolist()
        cands += [[c] for c in res["candidates"]]
        freqs += [[f] for f in res["candidate_frequency"].tolist()]
        images.append(res["images"].astype(np.int64).astype(np.int8))
    probs = models_np.variant_forward(sd, np.concatenate(images))
    os.makedirs(str(tmp_path / "opred"))
    with PredStore(str(tmp_path / "opred" / "pepper_prediction.hdf"), "w") as store:
        store.write_prediction(0, contigs, positions, np.asarray(depths).astype(np.uint8), np.array(cands, dtype=object),
                               np.asarray(freqs).astype(np.uint8), probs)
    oracle_totals = process_candidates(options, str(tmp_path / "opred"), str(tmp_path / "ovcf"))
    assert oracle_totals == totals
    _assert_same_vcfs(options.output_dir, str(tmp_path / "ovcf"))


def test_fused_polish_equals_the_three_step_run(tmp_path):
    """polish(..., fused_inference=True): the image workers hand the chain's chunks to the model on the device; the image files are
    those of the three-step run, every chunk has the same prediction (bases; phred up to one unit where two classes tie), and the
    polished FASTA is identical -- with a pile beyond the reservoir cap in one interval (host form) and several workers."""
    import glob
    import bam_utils as bu
    import pileup_utils as pu
    from pepper_amd.polish.polish import polish
    rng = np.random.default_rng(92)
    draft = pu.random_reference(rng, 6300)
    reads = pu.simulate_reads(rng, draft, 0, n_reads=300, read_len=(600, 2500), ins_rate=0.02, del_rate=0.02)
    deep = pu.simulate_reads(rng, draft[2100:2900], 2100, n_reads=1700, read_len=(150, 300), ins_rate=0.02, del_rate=0.02)
    reads = sorted([r for r in reads + deep if not any(op in (3, 6) for op, _ in r["cigar"])], key=lambda r: r["pos"])
    for i, r in enumerate(reads):
        r["name"] = "q%d" % i
    bam_path, fa_path = str(tmp_path / "reads.bam"), str(tmp_path / "draft.fa")
    bu.write_bam(bam_path, [("ctg1", len(draft))], {0: reads})
    with open(fa_path, "w") as fh:
        fh.write(">ctg1\n" + draft + "\n")
    sd = synthetic.polish_state_dict(seed=18, gain=2.0)
    model_path = str(tmp_path / "polish.pkl")
    torch.save(synthetic.checkpoint_dict({k: torch.from_numpy(v) for k, v in sd.items()}, hidden_size=128), model_path)
    outs = {}
    for name, threads, fused in (("plain", 1, False), ("fused", 2, True)):
        out_dir = str(tmp_path / name) + "/"
        walls = {}
        polish(bam_path, fa_path, out_dir, threads, None, model_path, 64, True, "0", 0, stage_walls=walls, fused_inference=fused)
        assert (walls["call_consensus"] == 0) == fused
        fasta = glob.glob(out_dir + "*.fa")
        assert len(fasta) == 1
        preds = {}
        for path in glob.glob(out_dir + "predictions_*/*.hdf"):
            with h5.File(path) as f:
                for region, a, b in f.list_polish_regions("ctg1"):
                    base = "predictions/ctg1/" + region + "/"
                    assert int(f[base + "contig_start"]) == a and int(f[base + "contig_end"]) == b
                    for chunk in f.keys(base.rstrip("/")):
                        if chunk in ("contig_start", "contig_end"):
                            continue
                        preds[(region, chunk)] = tuple(np.asarray(f[base + chunk + "/" + k]) for k in ("position", "index", "bases", "phred_score"))
        images = {}
        for path in glob.glob(out_dir + "images_*/*.hdf"):
            with h5.File(path) as f:
                for g in f.keys("summaries"):
                    images[g] = np.asarray(f["summaries/" + g + "/image"])
        outs[name] = (open(fasta[0]).read(), preds, images)
    (fa_a, pa, ia), (fa_b, pb, ib) = outs["plain"], outs["fused"]
    assert sorted(ia) == sorted(ib) and all(np.array_equal(ia[k], ib[k]) for k in ia)
    assert sorted(pa) == sorted(pb) and len(pa) >= 10
    for key in pa:
        assert np.array_equal(pa[key][0], pb[key][0]) and np.array_equal(pa[key][1], pb[key][1])
        assert np.array_equal(pa[key][2], pb[key][2]), key
        assert np.abs(pa[key][3].astype(int) - pb[key][3].astype(int)).max() <= 1
    assert fa_a == fa_b
    # ... and ONE hop from the oracle: the fused run's bases / phred against models_np on the chunks that run wrote
    names = sorted(ib)
    labels, phred = models_np.polish_predict_chunks(sd, np.stack([ib[g] for g in names]), 128)
    same_base = same_phred = total = 0
    for g, want_b, want_p in zip(names, labels, phred):
        contig, start, end, chunk = g.rsplit("_", 3)
        got = pb[("%s-%s-%s" % (contig, start, end), chunk)]
        same_base += int((got[2] == want_b).sum())
        same_phred += int((np.abs(got[3].astype(int) - want_p.astype(int)) <= 1).sum())
        total += len(want_b)
    assert total >= 10000 and same_base > 0.999 * total and same_phred > 0.99 * total, (same_base, same_phred, total)


def test_fused_call_variant_equals_the_three_step_run(tmp_path):
    """options.fused_inference: the encoder's windows go to the model on the device, both HDF5 stores are still written -- the
    image files are those of the unfused run, every candidate has the same prediction (batch membership differs: arrival order),
    and the five VCFs are byte-identical."""
    import bam_utils as bu
    import pileup_utils as pu
    from pepper_amd.variant.CallVariant import call_variant
    rng = np.random.default_rng(404)
    ref = pu.random_reference(rng, 9000)
    sites = {int(p): ("ACGT"[(("ACGT".index(ref[p]) + 1) % 4)], 0.5) for p in rng.choice(np.arange(200, 8800), 50, replace=False)}
    indels = {900: ("I", "CA", 0.6), 2500: ("D", 2, 0.7), 6100: ("D", 9, 0.5), 7000: ("I", "ACGTACGTTTGACA", 0.4)}
    reads = pu.simulate_reads(rng, ref, 0, n_reads=700, read_len=(400, 1800), snp_sites=sites, indel_sites=indels)
    reads = [r for r in reads if not any(op in (3, 6) for op, _ in r["cigar"])]
    # a pile beyond MAX_READS_IN_REGION on one interval: that interval takes the host-clipped form (forward from host buffers)
    deep = pu.simulate_reads(rng, ref[3000:3900], 3000, n_reads=5300, read_len=(150, 300), snp_sites=sites)
    reads = sorted(reads + [r for r in deep if not any(op in (3, 6) for op, _ in r["cigar"])], key=lambda r: r["pos"])
    for i, r in enumerate(reads):
        r["name"] = "q%d" % i
    bam_path, fa_path = str(tmp_path / "in.bam"), str(tmp_path / "ref.fa")
    bu.write_bam(bam_path, [("chr20", len(ref))], {0: reads}, flush_every=50)
    with open(fa_path, "w") as fh:
        fh.write(">chr20\n" + "\n".join(ref[i:i + 60] for i in range(0, len(ref), 60)) + "\n")
    sd = synthetic.variant_state_dict(seed=93, gain=2.5)
    model_path = str(tmp_path / "model.pkl")
    torch.save(synthetic.checkpoint_dict({k: torch.from_numpy(v) for k, v in sd.items()}, hidden_size=128), model_path)

    def options(out, **over):
        o = SimpleNamespace(
            bam=bam_path, fasta=fa_path, region=None, region_size=1500, threads=3, train_mode=False,
            use_hp_info=False, include_supplementary=False, output_dir=out,
            min_mapq=1, min_snp_baseq=1, min_indel_baseq=1, snp_frequency=0.10, insert_frequency=0.15,
            delete_frequency=0.15, min_coverage_threshold=3, snp_candidate_frequency_threshold=0.10,
            indel_candidate_frequency_threshold=0.12, candidate_support_threshold=2, skip_indels=False,
            downsample_rate=1.0,
            model_path=model_path, batch_size=128, num_workers=0, gpu=True, device_ids="0", callers_per_gpu=1,
            quantized=False, dry=False, sample_name="SYN", allowed_multiallelics=4,
            snp_p_value=0.1, insert_p_value=0.25, delete_p_value=0.25, snp_p_value_in_lc=0.1,
            insert_p_value_in_lc=0.3, delete_p_value_in_lc=0.3, snp_q_cutoff=20, indel_q_cutoff=15,
            snp_q_cutoff_in_lc=20, indel_q_cutoff_in_lc=10, report_snp_above_freq=0, report_indel_above_freq=0)
        for k, v in over.items():
            setattr(o, k, v)
        return o
    img_a, pred_a, totals_a = call_variant(options(str(tmp_path / "plain")))
    img_b, pred_b, totals_b = call_variant(options(str(tmp_path / "fused"), fused_inference=True))
    assert totals_a == totals_b and totals_a[0] > 30

    def predictions(directory):
        out = {}
        for fn in sorted(os.listdir(directory)):
            with h5.File(os.path.join(directory, fn)) as f:
                for batch in f.keys("predictions"):
                    base = "predictions/" + batch + "/"
                    contigs, pos = f[base + "contigs"].tolist(), f[base + "positions"].tolist()
                    cand, probs = f[base + "candidates"].tolist(), f[base + "base_prediction"]
                    depth, freq = f[base + "depths"].tolist(), f[base + "candidate_frequency"].tolist()
                    assert probs.dtype == np.float64 and len(pos) <= 128
                    for k in range(len(pos)):
                        key = (contigs[k], pos[k], cand[k][0])
                        assert key not in out
                        out[key] = (depth[k], freq[k][0], tuple(probs[k].tolist()))
        return out
    a, b = predictions(pred_a), predictions(pred_b)
    assert len(a) > 60 and a == b                              # bit-identical probabilities, whatever batch a window rode in

    def images(directory):
        out = {}
        for fn in sorted(os.listdir(directory)):
            with h5.File(os.path.join(directory, fn)) as f:
                for name in (f.keys("summaries") if "summaries" in f else []):
                    out[name] = (f["summaries/" + name + "/images"], f["summaries/" + name + "/positions"])
        return out
    ia, ib = images(img_a), images(img_b)
    assert sorted(ia) == sorted(ib) and all(np.array_equal(ia[k][0], ib[k][0]) and np.array_equal(ia[k][1], ib[k][1]) for k in ia)
    _assert_same_vcfs(str(tmp_path / "fused"), str(tmp_path / "plain"))
    # ... and ONE hop from the oracle: the fused run's base_prediction against models_np.variant_forward on the windows that run
    # wrote (north_star: within 1e-4), candidate by candidate
    checked = 0
    for fn in sorted(os.listdir(img_b)):
        with h5.File(os.path.join(img_b, fn)) as f:
            for name in (f.keys("summaries") if "summaries" in f else []):
                base = "summaries/" + name + "/"
                windows = np.asarray(f[base + "images"])
                if len(windows) == 0:
                    continue
                want = models_np.variant_forward(sd, windows)
                contigs, pos, cand = f[base + "contigs"].tolist(), f[base + "positions"].tolist(), f[base + "candidates"].tolist()
                for k in range(len(pos)):
                    got = b[(contigs[k], pos[k], cand[k][0])][2]
                    assert np.abs(np.asarray(got) - want[k]).max() < 1e-4, (name, k)
                    checked += 1
    assert checked == len(b)


def _assert_same_vcfs(got_dir, want_dir):
    from pepper_amd.variant import bgzf
    for name in ("PEPPER_VARIANT_FULL", "PEPPER_VARIANT_OUTPUT_PEPPER", "PEPPER_VARIANT_OUTPUT_VARIANT_CALLING",
                 "PEPPER_VARIANT_OUTPUT_VARIANT_CALLING_SNPs", "PEPPER_VARIANT_OUTPUT_VARIANT_CALLING_INDEL"):
        got = bgzf.read_bgzf(os.path.join(got_dir, name + ".vcf.gz")).decode().splitlines()
        want = bgzf.read_bgzf(os.path.join(want_dir, name + ".vcf.gz")).decode().splitlines()
        assert len(got) == len(want)
        for g, w in zip(got, want):
            if g == w:
                continue
            # identical call, alleles, genotype, depths; probabilities may differ in the 6th digit
            gf, wf = g.split("\t"), w.split("\t")
            assert gf[:5] == wf[:5] and gf[6:9] == wf[6:9], (g, w)
            gs, ws = gf[9].split(":"), wf[9].split(":")
            assert gs[0] == ws[0] and gs[3:] == ws[3:], (g, w)
            assert abs(float(gf[5]) - float(wf[5])) <= 1, (g, w)
            assert np.allclose([float(v) for v in gs[1].split(",")], [float(v) for v in ws[1].split(",")], atol=2e-5), (g, w)


def test_call_variant_from_bam_and_fasta_files(tmp_path):
    """The whole variant path from files on disk with the package's own readers (no injected handlers):
    BAM (+ .bai) + FASTA -> clipped reads -> GPU encoder -> GPU inference -> candidate finder -> VCF, against
    the oracle leg fed with the Python restatement of the reference's read clipping."""
    import bam_utils as bu
    import pileup_utils as pu
    from pepper_amd.variant.CallVariant import call_variant
    from pepper_amd.variant.DataStorePredict import DataStore as PredStore
    from pepper_amd.variant.FindCandidates import process_candidates
    rng = np.random.default_rng(303)
    ref = pu.random_reference(rng, 4200)
    sites = {int(p): ("ACGT"[(("ACGT".index(ref[p]) + 2) % 4)], 0.5) for p in rng.choice(np.arange(200, 4000), 24, replace=False)}
    indels = {700: ("I", "CA", 0.6), 1500: ("D", 2, 0.7), 3100: ("D", 9, 0.5)}
    reads = pu.simulate_reads(rng, ref, 0, n_reads=380, read_len=(400, 1500), snp_sites=sites, indel_sites=indels)
    # the simulator follows the encoder's N/P quirk (bases for skipped positions), which no BAM record has
    reads = [r for r in reads if not any(op in (3, 6) for op, _ in r["cigar"])]
    for i, r in enumerate(reads):
        r["name"] = "q%d" % i
    bam_path, fa_path = str(tmp_path / "in.bam"), str(tmp_path / "ref.fa")
    bu.write_bam(bam_path, [("chr20", len(ref))], {0: reads}, flush_every=50)
    with open(fa_path, "w") as fh:
        fh.write(">chr20\n" + "\n".join(ref[i:i + 60] for i in range(0, len(ref), 60)) + "\n")
    sd = synthetic.variant_state_dict(seed=92, gain=2.5)
    model_path = str(tmp_path / "model.pkl")
    torch.save(synthetic.checkpoint_dict({k: torch.from_numpy(v) for k, v in sd.items()}, hidden_size=128), model_path)
    options = SimpleNamespace(
        bam=bam_path, fasta=fa_path, region="chr20:0-4199", region_size=2100, threads=1, train_mode=False,
        use_hp_info=False, include_supplementary=False, output_dir=str(tmp_path / "out"),
        min_mapq=1, min_snp_baseq=1, min_indel_baseq=1, snp_frequency=0.10, insert_frequency=0.15,
        delete_frequency=0.15, min_coverage_threshold=3, snp_candidate_frequency_threshold=0.10,
        indel_candidate_frequency_threshold=0.12, candidate_support_threshold=2, skip_indels=False,
        downsample_rate=1.0,
        model_path=model_path, batch_size=256, num_workers=0, gpu=True, device_ids="0", callers_per_gpu=1,
        quantized=False, dry=False, sample_name="SYN", allowed_multiallelics=4,
        snp_p_value=0.1, insert_p_value=0.25, delete_p_value=0.25, snp_p_value_in_lc=0.1,
        insert_p_value_in_lc=0.3, delete_p_value_in_lc=0.3, snp_q_cutoff=20, indel_q_cutoff=15,
        snp_q_cutoff_in_lc=20, indel_q_cutoff_in_lc=10, report_snp_above_freq=0, report_indel_above_freq=0)
    image_dir, pred_dir, totals = call_variant(options)
    assert totals[0] > 8

    oracle = pu.load_restatement()
    contigs, positions, depths, cands, freqs, images = [], [], [], [], [], []
    for (start, end) in ((0, 2100), (2100, 4199)):
        rs, re_ = max(0, start - 100), end + 100
        clipped = bu.restated_get_reads(reads, rs, re_, False, 1)
        res = pu.run_variant(oracle, pu.FlatPileup(rs, re_, ref[rs:re_ + 1], clipped), pu.make_params(start, end))
        n = len(res["candidates"])
        contigs += ["chr20"] * n
        positions += res["positions"].tolist()
        depths += res["depths"].tolist()
        cands += [[c] for c in res["candidates"]]
        freqs += [[f] for f in res["candidate_frequency"].tolist()]
        images.append(res["images"].astype(np.int64).astype(np.int8))
    probs = models_np.variant_forward(sd, np.concatenate(images))
    os.makedirs(str(tmp_path / "opred"))
    with PredStore(str(tmp_path / "opred" / "pepper_prediction.hdf"), "w") as store:
        store.write_prediction(0, contigs, positions, np.asarray(depths).astype(np.uint8), np.array(cands, dtype=object),
                               np.asarray(freqs).astype(np.uint8), probs)
    assert process_candidates(options, str(tmp_path / "opred"), str(tmp_path / "ovcf")) == totals
    _assert_same_vcfs(options.output_dir, str(tmp_path / "ovcf"))


def test_polish_create_summary_from_bam(tmp_path):
    """Polish image generation for one region straight from a BAM + FASTA on disk (the same native reader; the
    reference's get_reads is identical for both tools) against the oracle legs on restated clipped reads: without
    the re-alignment stage, and with it (the reference's default; oracle = SSW restatement + oracle encoder)."""
    import bam_utils as bu
    import pileup_utils as pu
    from pepper_amd.polish.AlignmentSummarizer import AlignmentSummarizer
    from pepper_amd.variant.bam import BAM_handler
    from pepper_amd.variant.fasta import FASTA_handler
    rng = np.random.default_rng(77)
    ref = pu.random_reference(rng, 3000)
    reads = pu.simulate_reads(rng, ref, 0, n_reads=260, read_len=(300, 1200), ins_rate=0.02, del_rate=0.02)
    reads = [r for r in reads if not any(op in (3, 6) for op, _ in r["cigar"])]
    for i, r in enumerate(reads):
        r["name"] = "p%d" % i
    bam_path, fa_path = str(tmp_path / "p.bam"), str(tmp_path / "draft.fa")
    bu.write_bam(bam_path, [("contig_1", len(ref))], {0: reads})
    with open(fa_path, "w") as fh:
        fh.write(">contig_1\n" + ref + "\n")
    start, end = 900, 2100
    summ = AlignmentSummarizer(BAM_handler(bam_path), FASTA_handler(fa_path), "contig_1", start, end)
    images, labels, positions, chunk_ids = summ.create_summary(None, False, 1.0, realignment_flag=False)
    assert len(images) >= 2 and chunk_ids == list(range(len(images)))

    oracle = pu.load_restatement()
    clipped = bu.restated_get_reads(reads, start, end, False, 0)
    img, pos = pu.run_polish_oracle(oracle, pu.FlatPileup(start, end, ref[start:end + 1], clipped), start, end)
    want = AlignmentSummarizer.chunk_images(SimpleNamespace(image=img, genomic_pos=pos), 1000, 50)
    assert len(want[0]) == len(images)
    for got_i, want_i, got_p, want_p in zip(images, want[0], positions, want[2]):
        assert np.array_equal(got_i, want_i) and np.array_equal(got_p, want_p)

    # default call = with re-alignment (AlignmentSummarizer.py:179,328-332)
    from oracle import ssw
    images_r, _, positions_r, _ = summ.create_summary(None, False, 1.0)
    ref_win = ref[start:end + 20]
    res = ssw.realign_reads(ref_win, start, [r["pos"] for r in clipped], [r["seq"] for r in clipped])
    realigned = []
    for r, (st, score, p, pe, ops) in zip(clipped, res):
        assert st >= 0
        if st == 1:
            r = dict(r, pos=p, cigar=[(0 if o in (7, 8) else o, n) for o, n in ops])
        realigned.append(r)
    assert sum(st == 1 for st, *_ in res) > 0.9 * len(clipped)
    img, pos = pu.run_polish_oracle(oracle, pu.FlatPileup(start, end, ref[start:end + 1], realigned), start, end)
    want = AlignmentSummarizer.chunk_images(SimpleNamespace(image=img, genomic_pos=pos), 1000, 50)
    assert len(want[0]) == len(images_r)
    for got_i, want_i, got_p, want_p in zip(images_r, want[0], positions_r, want[2]):
        assert np.array_equal(got_i, want_i) and np.array_equal(got_p, want_p)


def test_polish_end_to_end_from_bam(tmp_path):
    """`polish(bam, draft, out, threads, region, model, ...)`: make_images (BAM reader, GPU re-aligner, GPU encoder) ->
    call_consensus -> perform_stitch.  Images are compared with the oracle legs region by region (restated clipped reads
    -> SSW restatement -> oracle encoder -> chunking); the polished FASTA must not depend on the number of workers."""
    import glob
    import bam_utils as bu
    import pileup_utils as pu
    from oracle import ssw
    from pepper_amd.polish.AlignmentSummarizer import AlignmentSummarizer
    from pepper_amd.polish.ImageGenerationUI import UserInterfaceSupport
    from pepper_amd.polish.polish import polish
    rng = np.random.default_rng(91)
    draft = pu.random_reference(rng, 4300)
    reads = pu.simulate_reads(rng, draft, 0, n_reads=220, read_len=(600, 2500), ins_rate=0.02, del_rate=0.02)
    reads = [r for r in reads if not any(op in (3, 6) for op, _ in r["cigar"])]
    for i, r in enumerate(reads):
        r["name"] = "q%d" % i
    bam_path, fa_path = str(tmp_path / "reads.bam"), str(tmp_path / "draft.fa")
    bu.write_bam(bam_path, [("ctg1", len(draft))], {0: reads})
    with open(fa_path, "w") as fh:
        fh.write(">ctg1\n" + draft + "\n")
    sd = synthetic.polish_state_dict(seed=17, gain=2.0)
    model_path = str(tmp_path / "polish.pkl")
    torch.save(synthetic.checkpoint_dict({k: torch.from_numpy(v) for k, v in sd.items()}, hidden_size=128), model_path)

    outs = []
    for threads in (1, 3):
        out_dir = str(tmp_path / ("run%d" % threads)) + "/"
        polish(bam_path, fa_path, out_dir, threads, None, model_path, 64, True, "0", 0)
        fasta = glob.glob(out_dir + "*.fa")
        assert len(fasta) == 1
        text = open(fasta[0]).read()
        assert text.startswith(">ctg1") and 0.5 * len(draft) < len(text.split("\n", 1)[1].replace("\n", "")) < 2 * len(draft)
        outs.append(text)
    assert outs[0] == outs[1]

    # images of the single-worker run against the oracle legs
    _, intervals = UserInterfaceSupport.make_intervals([("ctg1", None)], fa_path)
    assert intervals[0] == ("ctg1", 0, 1100) and intervals[1] == ("ctg1", 900, 2100) and len(intervals) == 5
    img_file = glob.glob(str(tmp_path / "run1") + "/images_*/*.hdf")[0]
    oracle = pu.load_restatement()
    checked = 0
    with h5.File(img_file) as f:
        for (_, start, end) in intervals[:3]:
            clipped = bu.restated_get_reads(reads, start, end, False, 0)
            res = ssw.realign_reads(draft[start:end + 20], start, [r["pos"] for r in clipped], [r["seq"] for r in clipped])
            realigned = [dict(r, pos=p, cigar=[(0 if o in (7, 8) else o, n) for o, n in ops]) if st == 1 else r
                         for r, (st, score, p, pe, ops) in zip(clipped, res)]
            img, pos = pu.run_polish_oracle(oracle, pu.FlatPileup(start, end, draft[start:end + 1], realigned), start, end)
            want = AlignmentSummarizer.chunk_images(SimpleNamespace(image=img, genomic_pos=pos), 1000, 50)
            for cid, (wi, wp) in enumerate(zip(want[0], want[2])):
                base = "summaries/ctg1_%d_%d_%d/" % (start, end, cid)
                assert np.array_equal(f[base + "image"], wi)
                assert np.array_equal(f[base + "position"], wp[:, 0]) and np.array_equal(f[base + "index"], wp[:, 1])
                checked += 1
    assert checked >= 4



def test_two_ranks_on_one_device_give_the_single_device_outputs(tmp_path):
    """device_ids "0,0": what an N-GPU run does with its image workers (dealt over device_ids) and its inference callers (one
    process per entry, predict_distributed_gpu's spawn leg; RunInference.py:101-116, ImageGenerationUI.py:326-339), on the one
    device a test box has.  call_variant's five VCFs and polish's FASTA must be those of device_ids "0"."""
    import glob
    import bam_utils as bu
    import pileup_utils as pu
    from pepper_amd.polish.polish import polish
    from pepper_amd.variant.CallVariant import call_variant
    rng = np.random.default_rng(606)
    ref = pu.random_reference(rng, 7000)
    sites = {int(p): ("ACGT"[(("ACGT".index(ref[p]) + 1) % 4)], 0.5) for p in rng.choice(np.arange(200, 6800), 40, replace=False)}
    indels = {900: ("I", "CA", 0.6), 2500: ("D", 2, 0.7), 5100: ("D", 9, 0.5)}
    reads = pu.simulate_reads(rng, ref, 0, n_reads=520, read_len=(400, 1800), snp_sites=sites, indel_sites=indels)
    reads = sorted([r for r in reads if not any(op in (3, 6) for op, _ in r["cigar"])], key=lambda r: r["pos"])
    for i, r in enumerate(reads):
        r["name"] = "q%d" % i
    bam_path, fa_path = str(tmp_path / "in.bam"), str(tmp_path / "ref.fa")
    bu.write_bam(bam_path, [("chr20", len(ref))], {0: reads}, flush_every=50)
    with open(fa_path, "w") as fh:
        fh.write(">chr20\n" + "\n".join(ref[i:i + 60] for i in range(0, len(ref), 60)) + "\n")
    sd = synthetic.variant_state_dict(seed=94, gain=2.5)
    model_path = str(tmp_path / "model.pkl")
    torch.save(synthetic.checkpoint_dict({k: torch.from_numpy(v) for k, v in sd.items()}, hidden_size=128), model_path)

    def options(out, device_ids):
        return SimpleNamespace(
            bam=bam_path, fasta=fa_path, region=None, region_size=1500, threads=4, train_mode=False,
            use_hp_info=False, include_supplementary=False, output_dir=out,
            min_mapq=1, min_snp_baseq=1, min_indel_baseq=1, snp_frequency=0.10, insert_frequency=0.15,
            delete_frequency=0.15, min_coverage_threshold=3, snp_candidate_frequency_threshold=0.10,
            indel_candidate_frequency_threshold=0.12, candidate_support_threshold=2, skip_indels=False,
            downsample_rate=1.0,
            model_path=model_path, batch_size=64, num_workers=0, gpu=True, device_ids=device_ids, callers_per_gpu=1,
            quantized=False, dry=False, sample_name="SYN", allowed_multiallelics=4,
            snp_p_value=0.1, insert_p_value=0.25, delete_p_value=0.25, snp_p_value_in_lc=0.1,
            insert_p_value_in_lc=0.3, delete_p_value_in_lc=0.3, snp_q_cutoff=20, indel_q_cutoff=15,
            snp_q_cutoff_in_lc=20, indel_q_cutoff_in_lc=10, report_snp_above_freq=0, report_indel_above_freq=0)
    _, pred_one, totals_one = call_variant(options(str(tmp_path / "v_one"), "0"))
    _, pred_two, totals_two = call_variant(options(str(tmp_path / "v_two"), "0,0"))
    assert totals_one == totals_two and totals_one[0] > 20
    assert sorted(os.listdir(pred_two)) == ["pepper_prediction_0.hdf", "pepper_prediction_1.hdf"]      # one file per caller
    _assert_same_vcfs(str(tmp_path / "v_two"), str(tmp_path / "v_one"))

    # polish: the same reads as a draft's pile
    psd = synthetic.polish_state_dict(seed=19, gain=2.0)
    pmodel = str(tmp_path / "polish.pkl")
    torch.save(synthetic.checkpoint_dict({k: torch.from_numpy(v) for k, v in psd.items()}, hidden_size=128), pmodel)
    texts = []
    for name, device_ids in (("p_one", "0"), ("p_two", "0,0")):
        out_dir = str(tmp_path / name) + "/"
        polish(bam_path, fa_path, out_dir, 4, None, pmodel, 64, True, device_ids, 0)
        fasta = glob.glob(out_dir + "*.fa")
        assert len(fasta) == 1
        texts.append(open(fasta[0]).read())
        n_pred = len(glob.glob(out_dir + "predictions_*/*.hdf"))
        assert n_pred == (1 if device_ids == "0" else 2), n_pred
    assert texts[0] == texts[1] and texts[0].startswith(">chr20")

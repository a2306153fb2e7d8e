"""The oracle's variant- and polish-encoder restatements (oracle/pileup_oracle.cpp) against the REFERENCE's
own C++ (oracle/_ref, built from /root/reference where present) and against committed golden vectors
that the reference builds produced.  CPU only."""
import os

import numpy as np
import pytest

import pileup_utils as pu
from conftest import need_reference_build


def _same(a, b):
    assert a["candidates"] == b["candidates"]
    for k in ("positions", "depths", "candidate_frequency", "images"):
        assert np.array_equal(a[k], b[k]), k


def _case(seed, depth=40, region=2500, flank=100, **kw):
    rng = np.random.default_rng(seed)
    ref_offset = 10_000
    L = region + 2 * flank
    ref = pu.random_reference(rng, L + 1, n_frac=kw.pop("n_frac", 0.0), lower_frac=kw.pop("lower_frac", 0.0))
    sites = {}
    for p in rng.choice(np.arange(ref_offset + flank + 20, ref_offset + flank + region - 20), size=12, replace=False):
        rb = ref[p - ref_offset].upper()
        sites[int(p)] = (rng.choice([c for c in "ACGT" if c != rb]), float(rng.choice([0.15, 0.5, 1.0])))
    indels = {}
    lo, hi = ref_offset + flank + 20, ref_offset + flank + region - 90
    for p in rng.choice(np.arange(lo, hi), size=10, replace=False):
        if int(p) in sites:
            continue
        frac = float(rng.choice([0.2, 0.5, 0.9]))
        if rng.random() < 0.5:
            n = int(rng.choice([1, 2, 5, 17, 59, 60, 61, 64]))
            indels[int(p)] = ("I", "".join(rng.choice(list("ACGT"), size=n)), frac)
        else:
            indels[int(p)] = ("D", int(rng.choice([1, 2, 4, 12, 20, 40, 59, 60, 61])), frac)
    if region < 1000:
        indels = dict(list(indels.items())[:3])
    reads = pu.simulate_reads(rng, ref, ref_offset, n_reads=int(depth * L / 500), snp_sites=sites,
                              indel_sites=indels, **kw)
    pile = pu.FlatPileup(ref_offset, ref_offset + L - 1, ref, reads)
    params = pu.make_params(ref_offset + flank, ref_offset + flank + region)
    return pile, params


CASES = {
    "plain": dict(seed=1),
    "eqx_cigars": dict(seed=2, eqx=True),
    "ref_skip_pad_fallthrough": dict(seed=3, skip_rate=0.004),
    "long_indels_61_cap": dict(seed=4, long_indel_rate=0.15, ins_rate=0.02, del_rate=0.02),
    "deep_over_125": dict(seed=5, depth=330, region=600),
    "lowercase_reference": dict(seed=6, lower_frac=0.2),
    "low_quality_heavy": dict(seed=7, low_q_rate=0.5),
    "indel_heavy": dict(seed=8, ins_rate=0.05, del_rate=0.05),
}


@pytest.fixture(scope="module")
def libs():
    return pu.load_restatement(), pu.load_reference_encoder()


@pytest.mark.parametrize("name", sorted(CASES))
def test_restatement_equals_reference_build(libs, name):
    oracle, ref = libs
    if ref is None:
        need_reference_build("oracle/_ref")
    pile, params = _case(**CASES[name])
    a = pu.run_variant(oracle, pile, params)
    b = pu.run_variant(ref, pile, params, reference_impl=True)
    assert len(b["candidates"]) > 0
    _same(a, b)


def test_restatement_threshold_variants(libs):
    oracle, ref = libs
    if ref is None:
        need_reference_build("oracle/_ref")
    pile, _ = _case(seed=11, ins_rate=0.03, del_rate=0.03)
    lo, hi = pile.region_start + 100, pile.region_end - 100
    for over in (dict(skip_indels=1), dict(min_snp_baseq=10, min_indel_baseq=10), dict(candidate_support_threshold=5),
                 dict(snp_freq_threshold=0.4, snp_candidate_freq_threshold=0.4), dict(min_coverage_threshold=60)):
        params = pu.make_params(lo, hi, **over)
        _same(pu.run_variant(oracle, pile, params), pu.run_variant(ref, pile, params, reference_impl=True))
    # candidate region narrower than the pileup, and touching the region edges (zero-padded windows)
    params = pu.make_params(pile.region_start, pile.region_end)
    _same(pu.run_variant(oracle, pile, params), pu.run_variant(ref, pile, params, reference_impl=True))


# pileups on which the base-quality branches of region_summary.cpp:366-454 decide a lot: qualities 2..39 with half of the
# bases (low_quality_heavy) or a fifth (the others) below the HiFi cut-off of 10, inserts whose quality sum straddles
# min_indel_baseq * length
PRESET_CASES = {
    "low_quality_heavy": dict(seed=31, low_q_rate=0.5, ins_rate=0.03, del_rate=0.03),
    "indel_heavy": dict(seed=32, ins_rate=0.05, del_rate=0.05, low_q_rate=0.2),
    "deep": dict(seed=33, depth=200, region=700, low_q_rate=0.3),
}


def preset_case(preset, name):
    pile, _ = _case(**PRESET_CASES[name])
    params = pu.make_params(pile.region_start + 100, pile.region_end - 100, **pu.PRESET_PARAMS[preset])
    return pile, params


@pytest.mark.parametrize("preset", sorted(pu.PRESET_PARAMS))
@pytest.mark.parametrize("name", sorted(PRESET_CASES))
def test_restatement_under_reference_presets(libs, preset, name):
    """HiFi / CLR / R10 image-generation thresholds (SetParameters.py:122-256) on quality-sensitive pileups."""
    oracle, ref = libs
    if ref is None:
        need_reference_build("oracle/_ref")
    pile, params = preset_case(preset, name)
    a = pu.run_variant(oracle, pile, params)
    assert len(a["candidates"]) > 0
    _same(a, pu.run_variant(ref, pile, params, reference_impl=True))


def test_presets_change_the_result(libs):
    """The quality cut-offs are live on these pileups: HiFi (baseq 10) and CLR (baseq 0) give different candidates."""
    oracle, _ = libs
    pile, hifi = preset_case("hifi", "low_quality_heavy")
    _, clr = preset_case("clr", "low_quality_heavy")
    a, b = pu.run_variant(oracle, pile, hifi), pu.run_variant(oracle, pile, clr)
    assert a["candidates"] != b["candidates"] or not np.array_equal(a["images"], b["images"])
    assert not np.array_equal(a["depths"][:20], b["depths"][:20]) or len(a["depths"]) != len(b["depths"])


def test_restatement_against_committed_golden(libs, golden_dir):
    """Golden vectors made by the reference build (tests/golden/make_golden_encoder.py): this is the
    check that still runs where /root/reference does not exist."""
    oracle, _ = libs
    for name in sorted(CASES)[:4]:
        g = np.load(os.path.join(golden_dir, f"encoder_variant_{name}.npz"), allow_pickle=False)
        pile, params = _case(**CASES[name])
        a = pu.run_variant(oracle, pile, params)
        assert a["candidates"] == [s for s in str(g["candidates"]).split("\n") if s]
        assert np.array_equal(a["positions"], g["positions"])
        assert np.array_equal(a["depths"], g["depths"])
        assert np.array_equal(a["candidate_frequency"], g["candidate_frequency"])
        assert np.array_equal(a["images"].astype(np.int16), g["images"])


def test_empty_and_uncovered_region(libs):
    oracle, ref = libs
    rng = np.random.default_rng(0)
    refseq = pu.random_reference(rng, 501)
    pile = pu.FlatPileup(1000, 1499, refseq, [])
    params = pu.make_params(1000, 1499)
    a = pu.run_variant(oracle, pile, params)
    assert a["candidates"] == [] and a["images"].shape == (0, 33, 26)
    if ref is not None:
        assert pu.run_variant(ref, pile, params, reference_impl=True)["candidates"] == []


# ---- polish SummaryGenerator ------------------------------------------------------------------------
POLISH_CASES = {
    "plain": dict(seed=21),
    "eqx_cigars": dict(seed=22, eqx=True),
    "indel_heavy": dict(seed=23, ins_rate=0.06, del_rate=0.05),
    "deep": dict(seed=24, depth=300, region=700),
    "lowercase_and_n": dict(seed=25, lower_frac=0.2, n_frac=0.02),
    "long_inserts": dict(seed=26, long_indel_rate=0.2, ins_rate=0.03),
}


def _polish_case(seed, depth=45, region=1800, **kw):
    """Region [start, end] of a draft contig; reads clipped to it as get_reads would deliver them."""
    import bam_utils as bu
    rng = np.random.default_rng(seed)
    ref_offset = 5_000
    ref = pu.random_reference(rng, region + 1, n_frac=kw.pop("n_frac", 0.0), lower_frac=kw.pop("lower_frac", 0.0))
    reads = pu.simulate_reads(rng, ref, ref_offset, n_reads=int(depth * region / 500), **kw)
    reads = [r for r in reads if not any(op in (3, 6) for op, _ in r["cigar"])]
    start, end = ref_offset, ref_offset + region
    clipped = bu.restated_get_reads(reads, start, end, False, 0)
    return pu.FlatPileup(start, end, ref, clipped), start, end


@pytest.mark.parametrize("name", sorted(POLISH_CASES))
def test_polish_restatement_equals_reference_build(name):
    ref = pu.load_reference_polish_encoder()
    if ref is None:
        need_reference_build("oracle/_ref")
    oracle = pu.load_restatement()
    pile, start, end = _polish_case(**POLISH_CASES[name])
    img_o, pos_o = pu.run_polish_oracle(oracle, pile, start, end)
    img_r, pos_r = pu.run_polish_reference(ref, pile, start, end)
    assert len(img_r) > 500
    assert np.array_equal(pos_o, pos_r) and np.array_equal(img_o, img_r)


def test_polish_restatement_against_committed_golden(golden_dir):
    oracle = pu.load_restatement()
    for name in sorted(POLISH_CASES)[:3]:
        g = np.load(os.path.join(golden_dir, f"encoder_polish_{name}.npz"), allow_pickle=False)
        pile, start, end = _polish_case(**POLISH_CASES[name])
        img, pos = pu.run_polish_oracle(oracle, pile, start, end)
        assert np.array_equal(img, g["image"]) and np.array_equal(pos, g["positions"])

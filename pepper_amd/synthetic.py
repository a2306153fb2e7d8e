"""Deterministic synthetic checkpoints and summary tensors (numpy only, no torch RNG).

There is no network in the build environment, so neither the released PEPPER
checkpoints nor real HG003 summaries exist here.  Everything that needs weights or
inputs (tests, golden-vector generation, ``bench.py``, ``smoke()``) draws them from
the recipes below so that the container that generated ``tests/golden/*`` and the GPU
box see bit-identical tensors without shipping 47 MB of weights.

Weight layout follows the reference state_dicts:
  variant: /root/reference/pepper_variant/modules/python/models/simple_model.py:23-46
  polish : /root/reference/pepper/modules/python/models/simple_model.py:12-21
Checkpoint dict keys follow
  /root/reference/pepper_variant/modules/python/models/train_distributed.py:36-42
  ({'model_state_dict', 'hidden_size', 'gru_layers', 'epochs', ...}).
Synthetic summary distributions follow SURVEY.md section 8(d) (V-syn / P-syn).
"""
import os
from collections import OrderedDict

import numpy as np

# ---- shapes of the two model families -------------------------------------------------
VARIANT_FEATURES = 26      # ImageSizeOptions.IMAGE_HEIGHT   (pepper_variant Options.py:6)
VARIANT_WINDOW = 33        # CANDIDATE_WINDOW_SIZE + 1        (pepper_variant Options.py:8)
VARIANT_LSTM_HIDDEN = 256  # hard-coded lstm_1/2_hidden_size  (simple_model.py:15-16)
VARIANT_LINEAR = 512
VARIANT_CLASSES = 3        # TOTAL_TYPE_LABELS

POLISH_FEATURES = 10       # pepper Options.py:2
POLISH_HIDDEN = 128        # TrainOptions.HIDDEN_SIZE (pepper Options.py:19)
POLISH_CLASSES = 5
POLISH_SEQ = 1000
POLISH_WINDOW = 100
POLISH_JUMP = 50

VSYN_SEED = 20260926
PSYN_SEED = 20260927


def _uniform(rng, shape, bound):
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def variant_param_shapes(image_features=VARIANT_FEATURES, gru_layers=1,
                         num_classes_type=VARIANT_CLASSES, window=VARIANT_WINDOW):
    """(name, shape, init_bound) in the reference's state_dict order."""
    H = VARIANT_LSTM_HIDDEN
    out = []
    for mod, in0 in (("encoder", image_features), ("decoder", 2 * H)):
        for layer in range(gru_layers):
            in_sz = in0 if layer == 0 else 2 * H
            for suffix in ("", "_reverse"):
                k = 1.0 / np.sqrt(H)
                out.append((f"{mod}.weight_ih_l{layer}{suffix}", (4 * H, in_sz), k))
                out.append((f"{mod}.weight_hh_l{layer}{suffix}", (4 * H, H), k))
                out.append((f"{mod}.bias_ih_l{layer}{suffix}", (4 * H,), k))
                out.append((f"{mod}.bias_hh_l{layer}{suffix}", (4 * H,), k))
    L = VARIANT_LINEAR
    dims = [("linear_1", 2 * H * window, L), ("linear_2", L, L), ("linear_3", L, L),
            ("linear_4", L, L), ("linear_5", L, L), ("output_layer_type", L, num_classes_type)]
    for name, fin, fout in dims:
        k = 1.0 / np.sqrt(fin)
        out.append((f"{name}.weight", (fout, fin), k))
        out.append((f"{name}.bias", (fout,), k))
    return out


def polish_param_shapes(image_features=POLISH_FEATURES, gru_layers=1, hidden=POLISH_HIDDEN,
                        num_classes=POLISH_CLASSES):
    H = hidden
    out = []
    for mod, in0 in (("gru_encoder", image_features), ("gru_decoder", 2 * H)):
        for layer in range(gru_layers):
            in_sz = in0 if layer == 0 else 2 * H
            for suffix in ("", "_reverse"):
                k = 1.0 / np.sqrt(H)
                out.append((f"{mod}.weight_ih_l{layer}{suffix}", (3 * H, in_sz), k))
                out.append((f"{mod}.weight_hh_l{layer}{suffix}", (3 * H, H), k))
                out.append((f"{mod}.bias_ih_l{layer}{suffix}", (3 * H,), k))
                out.append((f"{mod}.bias_hh_l{layer}{suffix}", (3 * H,), k))
    k = 1.0 / np.sqrt(2 * H)
    out.append(("dense1.weight", (num_classes, 2 * H), k))
    out.append(("dense1.bias", (num_classes,), k))
    return out


def _state_dict(shapes, seed, gain):
    sd = OrderedDict()
    for idx, (name, shape, bound) in enumerate(shapes):
        rng = np.random.default_rng([seed, idx])
        sd[name] = _uniform(rng, shape, bound * gain)
    return sd


def variant_state_dict(seed=0, gain=1.0, **kw):
    """numpy state_dict of the variant model (PyTorch-default init bounds x ``gain``).

    gain=1 resembles a freshly initialised network (outputs near 1/3 each); gain>1
    saturates gates and spreads the softmax, which is the harder parity case.
    """
    return _state_dict(variant_param_shapes(**kw), seed, gain)


def polish_state_dict(seed=0, gain=1.0, **kw):
    return _state_dict(polish_param_shapes(**kw), seed, gain)


def checkpoint_dict(state_dict, hidden_size, gru_layers=1, epochs=1, module_prefix=False):
    """Reference checkpoint schema (torch tensors are made by the caller)."""
    sd = OrderedDict((("module." + k if module_prefix else k), v) for k, v in state_dict.items())
    return {"model_state_dict": sd, "hidden_size": hidden_size, "gru_layers": gru_layers,
            "epochs": epochs}


# ---- synthetic summaries ----------------------------------------------------------------
def variant_windows(n, seed=VSYN_SEED, window=VARIANT_WINDOW, features=VARIANT_FEATURES):
    """V-syn: int8 [n, 33, 26] candidate windows with a summary-like value distribution.

    Column meaning: /root/reference/pepper_variant/modules/cpp/region_summary.h:23-48.
    Counts are stored negated by the encoder (region_summary.cpp:381-430); the centre
    row carries the candidate-specific overwrite (region_summary.cpp:848-905).
    """
    rng = np.random.default_rng(seed)
    x = np.zeros((n, window, features), dtype=np.int16)
    x[:, :, 0] = rng.integers(1, 6, size=(n, window))
    for col in (4, 15):
        x[:, :, col] = -np.clip(rng.poisson(30, size=(n, window)), 0, 125)
    base_cols = list(range(8, 15)) + list(range(19, 26))
    sparse = rng.random(size=(n, window, len(base_cols))) < 0.10
    vals = -np.clip(rng.poisson(3, size=(n, window, len(base_cols))), 0, 125)
    x[:, :, base_cols] = np.where(sparse, vals, 0)
    mid = window // 2
    kind = rng.integers(0, 3, size=n)  # 0 SNP, 1 INS, 2 DEL
    rows = np.arange(n)
    alt = rng.integers(1, 5, size=n)
    length = np.clip(rng.geometric(0.4, size=n), 1, 60)
    fwd = np.clip(rng.poisson(6, size=n), 0, 125)
    rev = np.clip(rng.poisson(6, size=n), 0, 125)
    snp, ins, dele = kind == 0, kind == 1, kind == 2
    x[rows[snp], mid, 1] = alt[snp]
    x[rows[snp], mid, 5] = fwd[snp]
    x[rows[snp], mid, 16] = rev[snp]
    x[rows[snp], mid, 7 + alt[snp]] = fwd[snp]       # negated base column -> positive
    x[rows[snp], mid, 18 + alt[snp]] = rev[snp]
    x[rows[ins], mid, 2] = length[ins]
    x[rows[ins], mid, 6] = fwd[ins]
    x[rows[ins], mid, 17] = rev[ins]
    x[rows[ins], mid, 12] = fwd[ins]
    x[rows[ins], mid, 23] = rev[ins]
    x[rows[dele], mid, 3] = length[dele]
    x[rows[dele], mid, 7] = fwd[dele]
    x[rows[dele], mid, 18] = rev[dele]
    x[rows[dele], mid, 13] = fwd[dele]
    x[rows[dele], mid, 24] = rev[dele]
    return x.astype(np.int8)


def variant_windows_device(n, seed=VSYN_SEED, device="cuda", window=VARIANT_WINDOW, features=VARIANT_FEATURES):
    """V-syn with the distribution of variant_windows(), drawn by torch's generator ON THE DEVICE (bench.py needs 2^20
    distinct windows per GPU; the numpy recipe above takes two minutes for that on a few host cores).  Same columns,
    same laws, different bits: fixtures and parity tests keep using the numpy recipe.  -> int8 tensor [n, 33, 26]."""
    import torch
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(int(seed))
    out = torch.empty((n, window, features), dtype=torch.int8, device=dev)
    base_cols = torch.tensor(list(range(8, 15)) + list(range(19, 26)), device=dev)
    mid = window // 2
    for a in range(0, n, 65536):
        m = min(65536, n - a)
        x = torch.zeros((m, window, features), dtype=torch.int16, device=dev)
        x[:, :, 0] = torch.randint(1, 6, (m, window), generator=g, device=dev, dtype=torch.int16)
        for col in (4, 15):
            x[:, :, col] = -torch.poisson(torch.full((m, window), 30.0, device=dev), generator=g).clamp_(0, 125).to(torch.int16)
        sparse = torch.rand((m, window, 14), generator=g, device=dev) < 0.10
        vals = -torch.poisson(torch.full((m, window, 14), 3.0, device=dev), generator=g).clamp_(0, 125).to(torch.int16)
        x[:, :, base_cols] = torch.where(sparse, vals, torch.zeros_like(vals))
        kind = torch.randint(0, 3, (m,), generator=g, device=dev)
        alt = torch.randint(1, 5, (m,), generator=g, device=dev)
        length = torch.empty(m, device=dev).geometric_(0.4, generator=g).clamp_(1, 60).to(torch.int16)
        fwd = torch.poisson(torch.full((m,), 6.0, device=dev), generator=g).clamp_(0, 125).to(torch.int16)
        rev = torch.poisson(torch.full((m,), 6.0, device=dev), generator=g).clamp_(0, 125).to(torch.int16)
        for k, (c_len, c_f, c_r, b_f, b_r) in enumerate(((1, 5, 16, None, None), (2, 6, 17, 12, 23), (3, 7, 18, 13, 24))):
            rows = torch.nonzero(kind == k).squeeze(1)
            x[rows, mid, c_len] = alt[rows].to(torch.int16) if k == 0 else length[rows]
            x[rows, mid, c_f] = fwd[rows]
            x[rows, mid, c_r] = rev[rows]
            if k == 0:
                x[rows, mid, 7 + alt[rows]] = fwd[rows]
                x[rows, mid, 18 + alt[rows]] = rev[rows]
            else:
                x[rows, mid, b_f] = fwd[rows]
                x[rows, mid, b_r] = rev[rows]
        out[a:a + m] = x.to(torch.int8)
    return out


def polish_chunks_device(n, seed=PSYN_SEED, device="cuda", seq=POLISH_SEQ, features=POLISH_FEATURES):
    """P-syn with the distribution of polish_chunks(), drawn on the device.  -> uint8 tensor [n, 1000, 10]."""
    import torch
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(int(seed))
    alpha = torch.tensor([4, 1, 1, 1, 4, 1, 1, 1, 0.3, 0.3], device=dev) * 0.25
    out = torch.empty((n, seq, features), dtype=torch.uint8, device=dev)
    for a in range(0, n, 8192):
        m = min(8192, n - a)
        # a Dirichlet draw = normalised Gamma(alpha, 1) draws
        gam = torch._standard_gamma(alpha.expand(m, seq, features).contiguous(), generator=g)
        frac = gam / gam.sum(-1, keepdim=True).clamp_min(1e-30)
        img = torch.floor(frac * 254.0).to(torch.uint8)
        padded = torch.rand(m, generator=g, device=dev) < 0.02
        tail = torch.randint(1, seq // 2, (m,), generator=g, device=dev)
        pos = torch.arange(seq, device=dev).unsqueeze(0)
        img[(padded.unsqueeze(1) & (pos >= (seq - tail).unsqueeze(1)))] = 0
        out[a:a + m] = img
    return out


def polish_chunks(n, seed=PSYN_SEED, seq=POLISH_SEQ, features=POLISH_FEATURES):
    """P-syn: uint8 [n, 1000, 10] chunks; each row a composition summing to <= 254.

    Row semantics: /root/reference/pepper/modules/src/pileup_summary/summary_generator.cpp:274-306
    (count / coverage * 254, truncated).  2 % of chunks get zero-padded tails as produced by
    chunk_images (/root/reference/pepper/modules/python/AlignmentSummarizer.py:18-56).
    """
    rng = np.random.default_rng(seed)
    alpha = np.array([4, 1, 1, 1, 4, 1, 1, 1, 0.3, 0.3]) * 0.25
    frac = rng.dirichlet(alpha, size=(n, seq))
    img = np.floor(frac * 254.0).astype(np.uint8)
    padded = rng.random(n) < 0.02
    tail = rng.integers(1, seq // 2, size=n)
    for i in np.nonzero(padded)[0]:
        img[i, seq - tail[i]:, :] = 0
    return img


def simulate_clipped_reads(rng, reference, region_start, n_reads, sub=0.04, ins=0.03, dele=0.04, min_len=30, full_span=0.0):
    """Noisy reads of a reference window (nanopore-like error mix), clipped like the polish BAM reader clips them;
    a fraction `full_span` of them covers the whole window (long reads over a 1 kb region)."""
    bases = "ACGT"
    n = len(reference)
    pos, seqs = [], []
    for _ in range(n_reads):
        a = int(rng.integers(0, max(1, n - min_len)))
        b = int(rng.integers(min(n, a + min_len), n + 1)) if rng.random() < 0.5 else n
        if full_span and rng.random() < full_span:
            a, b = 0, n
        out = []
        for ch in reference[a:b]:
            u = rng.random()
            if u < dele:
                continue
            if u < dele + sub:
                ch = bases[int(rng.integers(4))]
            out.append(ch)
            while rng.random() < ins:
                out.append(bases[int(rng.integers(4))])
        pos.append(region_start + a)
        seqs.append("".join(out) or "A")
    return pos, seqs


# ---- E-syn: pileups for the summary encoder ------------------------------------------------------------------------
ESYN_SEED = 20260928


def encoder_region(seed=ESYN_SEED, region=100_000, flank=100, depth=60, read_len=8000, region_start=10_000,
                   event_rate=0.02, sub_rate=0.04, snp_every=1000, indel_every=700):
    """One nanopore-like pileup as the flat arrays of pa_pileup (include/pepper_amd_encoder.h), vectorised.

    A `region` + 2 x `flank` + 1 reference window (AlignmentSummarizer.py:181-182 pads 100 each side) at ~`depth`x:
    reads of N(read_len, read_len / 4) reference span starting anywhere from half a read before the window to its end
    (clipped to it, as the BAM reader clips); random inserts / deletions of 1..5 bases at `event_rate` per position and
    sub_rate substitutions (sequencing noise); qualities uniform 3..39; a heterozygous or homozygous SNP about every
    `snp_every` positions (two haplotypes) and a systematic indel site about every `indel_every` positions where 35 % of
    the covering reads carry the same insert or deletion (what homopolymers do to nanopore reads) -- so that SNP, insert
    and delete candidates all occur.  -> (reference bytes, flat dict, region_start, region_end)."""
    rng = np.random.default_rng(seed)
    L = region + 2 * flank + 1
    hap = rng.integers(0, 4, size=(2, L), dtype=np.uint8)
    hap[1] = hap[0]
    sites = np.flatnonzero(rng.random(L) < 1.0 / snp_every)
    alt = (hap[0, sites] + rng.integers(1, 4, size=len(sites))) % 4
    hom = rng.random(len(sites)) < 0.4
    ref = hap[0].copy()
    hap[0, sites] = np.where(hom, alt, hap[0, sites])          # haplotype 0 carries the homozygous ones,
    hap[1, sites] = alt                                        # haplotype 1 all of them
    n_reads = int(depth * L / read_len) + 8
    start = rng.integers(-read_len // 2, L - 50, size=n_reads)
    span_end = np.minimum(L, start + rng.normal(read_len, read_len / 4, size=n_reads).astype(np.int64))
    s = np.maximum(0, start)
    keep = span_end - s >= 100
    s, e = s[keep], span_end[keep]
    n_reads = len(s)
    span = e - s
    which = rng.integers(0, 2, size=n_reads)
    # events as flat (read, reference position, insert?, length) lists: noise + the systematic sites, ordered per read
    n_noise = rng.poisson(span * event_rate)
    ev_read = np.repeat(np.arange(n_reads), n_noise)
    ev_pos = s[ev_read] + 20 + (rng.random(len(ev_read)) * (span[ev_read] - 40)).astype(np.int64)
    ev_ins = rng.random(len(ev_read)) < 0.5
    ev_len = rng.integers(1, 6, size=len(ev_read))
    isites = np.flatnonzero(rng.random(L) < 1.0 / indel_every)
    site_ins = rng.random(len(isites)) < 0.5
    site_len = rng.integers(1, 9, size=len(isites))
    carried = (isites[None, :] >= s[:, None] + 20) & (isites[None, :] < e[:, None] - 40) & (rng.random((n_reads, len(isites))) < 0.35)
    cr, cs = np.nonzero(carried)
    ev_read = np.concatenate([ev_read, cr])
    ev_pos = np.concatenate([ev_pos, isites[cs]])
    ev_ins = np.concatenate([ev_ins, site_ins[cs]])
    ev_len = np.concatenate([ev_len, site_len[cs]])
    order = np.lexsort((ev_pos, ev_read))
    ev_read, ev_pos, ev_ins, ev_len = ev_read[order], ev_pos[order], ev_ins[order], ev_len[order]
    # an event needs 8 matched bases after the end of the one before it and 20 before the end of the read: drop the ones
    # that do not, until the survivors all do (a dropped event can uncover a closer predecessor)
    while True:
        prev_end = np.concatenate([[0], ev_pos[:-1] + np.where(ev_ins[:-1], 0, ev_len[:-1])])
        first = np.concatenate([[True], ev_read[1:] != ev_read[:-1]])
        prev_end = np.where(first, s[ev_read], prev_end)
        ok = (ev_pos - prev_end >= 8) & (ev_pos + np.where(ev_ins, 0, ev_len) <= e[ev_read] - 20)
        if ok.all():
            break
        ev_read, ev_pos, ev_ins, ev_len = ev_read[ok], ev_pos[ok], ev_ins[ok], ev_len[ok]
    gap = ev_pos - prev_end                                     # matched bases before the event
    n_keep = np.bincount(ev_read, minlength=n_reads)
    kk = np.arange(len(ev_read)) - np.repeat(np.cumsum(n_keep) - n_keep, n_keep)
    ev_end = ev_pos + np.where(ev_ins, 0, ev_len)
    last_end = s.copy()                                         # reads without events: one match run from s
    np.maximum.at(last_end, ev_read, ev_end)
    tail = e - last_end
    # operations per read: (M gap, I/D len) x n_keep, then M tail
    ops_per = 2 * n_keep + 1
    coff = np.zeros(n_reads + 1, np.int64)
    np.cumsum(ops_per, out=coff[1:])
    total_ops = int(coff[-1])
    cigar_op = np.zeros(total_ops + 1, np.int32)
    cigar_len = np.zeros(total_ops + 1, np.int32)
    at = coff[ev_read] + 2 * kk
    cigar_len[at] = gap
    cigar_op[at + 1] = np.where(ev_ins, 1, 2)
    cigar_len[at + 1] = ev_len
    cigar_len[coff[1:] - 1] = tail
    # read segments = operations that consume read bases (M and I); each has a length and, for M, a reference start
    seg_len = np.where(cigar_op[:total_ops] == 2, 0, cigar_len[:total_ops]).astype(np.int64)
    seg_ref = np.zeros(total_ops, np.int64)
    seg_ref[at] = prev_end
    seg_ref[coff[1:] - 1] = last_end
    op_read = np.repeat(np.arange(n_reads), ops_per)
    seg_start = np.cumsum(seg_len) - seg_len
    total = int(seg_len.sum())
    is_ins_op = cigar_op[:total_ops] == 1
    # every read base of a match run = haplotype[which][seg_ref + offset in the run]: one gather through a per-run offset
    hap_flat = hap.reshape(-1)
    seg_off = np.where(is_ins_op, 0, seg_ref + which[op_read] * L) - seg_start
    src = np.repeat(seg_off, seg_len)
    src += np.arange(total)
    np.clip(src, 0, 2 * L - 1, out=src)                         # inserted bases: any in-range index, overwritten below
    seq = hap_flat[src]
    ins_ops = np.flatnonzero(is_ins_op)
    ins_at = np.repeat(seg_start[ins_ops], seg_len[ins_ops]) + (np.arange(int(seg_len[ins_ops].sum())) -
                                                                  np.repeat(np.cumsum(seg_len[ins_ops]) - seg_len[ins_ops], seg_len[ins_ops]))
    seq[ins_at] = rng.integers(0, 4, size=len(ins_at), dtype=np.uint8)
    # carriers of one systematic insert site share its bases: seeded by the site position
    site_of_op = np.full(total_ops, -1, np.int64)
    sys_ev = np.isin(ev_pos, isites) & ev_ins
    site_of_op[at[sys_ev] + 1] = ev_pos[sys_ev]
    sys_ops = np.flatnonzero(site_of_op >= 0)
    if len(sys_ops):
        sys_at = np.repeat(seg_start[sys_ops], seg_len[sys_ops]) + (np.arange(int(seg_len[sys_ops].sum())) -
                                                                      np.repeat(np.cumsum(seg_len[sys_ops]) - seg_len[sys_ops], seg_len[sys_ops]))
        k_in = np.arange(len(sys_at)) - np.repeat(np.cumsum(seg_len[sys_ops]) - seg_len[sys_ops], seg_len[sys_ops])
        seq[sys_at] = ((np.repeat(site_of_op[sys_ops], seg_len[sys_ops]) * 7 + k_in * 3) % 4).astype(np.uint8)
    sub_at = np.cumsum(rng.geometric(sub_rate, size=int(total * sub_rate * 1.2) + 64)) - 1
    sub_at = sub_at[sub_at < total]
    seq[sub_at] = (seq[sub_at] + rng.integers(1, 4, size=len(sub_at), dtype=np.uint8)) % 4
    alphabet = np.frombuffer(b"ACGT", np.uint8)
    read_bases = np.zeros(n_reads + 1, np.int64)
    np.cumsum(np.bincount(op_read, weights=seg_len, minlength=n_reads).astype(np.int64), out=read_bases[1:])
    flat = dict(read_pos=(region_start + s).astype(np.int64), read_reverse=(rng.random(n_reads) < 0.5).astype(np.uint8),
                read_mapq=np.full(n_reads, 60, np.int32), seq_offset=read_bases,
                seq=np.concatenate([alphabet[seq], np.zeros(1, np.uint8)]),
                qual=np.concatenate([rng.integers(3, 40, size=total, dtype=np.uint8), np.zeros(1, np.uint8)]),
                cigar_offset=coff, cigar_op=cigar_op, cigar_len=cigar_len, n_reads=int(n_reads))
    return alphabet[ref].tobytes(), flat, region_start, region_start + L - 1


def encoder_regions(n, seed=ESYN_SEED, workers=0, **kw):
    """n regions (seeds seed, seed + 1, ...), generated on `workers` processes (0: one per 4 regions, at most 16)."""
    if workers <= 0:
        workers = max(1, min(16, n // 4, os.cpu_count() or 1))
    if workers == 1 or n == 1:
        return [encoder_region(seed + k, **kw) for k in range(n)]
    # spawned workers (numpy only); the parent's __main__ is hidden while they start so that none of them re-imports a
    # main module that pulls in torch (the trick of pepper_amd.hostpipe._start_all)
    import sys
    from multiprocessing import get_context
    main = sys.modules.get("__main__")
    saved_spec, saved_file = getattr(main, "__spec__", None), getattr(main, "__file__", None)
    had_file = main is not None and hasattr(main, "__file__")
    try:
        if main is not None:
            main.__spec__ = None
            if had_file:
                del main.__file__
        pool = get_context("spawn").Pool(workers)
    finally:
        if main is not None:
            main.__spec__ = saved_spec
            if had_file:
                main.__file__ = saved_file
    with pool:
        futs = [pool.apply_async(encoder_region, (seed + k,), kw) for k in range(n)]
        return [f.get() for f in futs]


# ---- WG-syn: a whole genome's worth of variant windows, shard sizes like the chromosomes ------------------
# GRCh38 primary assembly lengths in bases, chr1..22, X, Y (one image file per chromosome, as a per-chromosome run writes them)
GRCH38_LENGTHS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
                  133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616,
                  64444167, 46709983, 50818468, 156040895, 57227415]


def wg_syn_shards(total_windows, multiple=512):
    """Windows per chromosome file, proportional to the chromosome lengths, each a multiple of `multiple` (the reference's
    HDF5 batch), summing to about total_windows.  -> list of 24 counts, chr1..22, X, Y."""
    total_len = float(sum(GRCH38_LENGTHS))
    return [max(multiple, int(round(total_windows * n / total_len / multiple)) * multiple) for n in GRCH38_LENGTHS]

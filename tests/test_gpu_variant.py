"""GPU parity of the variant model: HIP path (through the C ABI) vs golden vectors from the
reference classes and vs the oracle on seeded inputs.  Tolerance: 1e-4 on probabilities and
logits (BASELINE.json north_star)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import models_np
from pepper_amd import _lib, synthetic

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(autouse=True, params=["big-call", "small-call", "no-unit-split"])
def schedule(request):
    """Every test of this file under every schedule.  "small-call", the defaults: calls of at most 512 windows run their
    step loops with a tile's hidden units split over eight workgroups that exchange h_t through memory every step (api.hip
    `unit_split`; four workgroups of 64 units for 513-1024 windows), calls of at most 3072 windows take the GEMM + Xp decoder and 32-row workgroups in both step loops
    (`small_batch`, `small_rows`).  "no-unit-split" (PA_UNIT_SPLIT=0): the latter for the calls of at most 512 windows too.
    "big-call" (PA_SMALL_BATCH=0 PA_SMALL_ROWS=0 PA_UNIT_SPLIT=0): the schedule of large calls -- the decoder with the
    projection inside its step loop, 64-row workgroups -- for the small calls the parity tests make.  (All read at model
    creation.)"""
    saved = {k: os.environ.get(k) for k in ("PA_SMALL_BATCH", "PA_SMALL_ROWS", "PA_UNIT_SPLIT")}
    for k in saved:
        if request.param == "big-call" or (request.param == "no-unit-split" and k == "PA_UNIT_SPLIT"):
            os.environ[k] = "0"
        else:
            os.environ.pop(k, None)
    yield request.param
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


class NativeVariant:
    """Thin test harness over the raw C ABI (host-pointer entry points)."""

    def __init__(self, sd, gru_layers=1, max_chunk=0):
        self.lib = _lib.load()
        cfg = _lib.VariantConfig(26, 33, gru_layers, 3, 0, max_chunk)
        names, data, numel, n, keep = _lib.marshal_state_dict(sd)
        self.h = ctypes.c_void_p()
        _lib.check(self.lib.pa_variant_create(ctypes.byref(cfg), names, data, numel, n, None,
                                              ctypes.byref(self.h)))

    def forward(self, images):
        images = np.ascontiguousarray(images, dtype=np.int8)
        n = images.shape[0]
        probs = np.empty((n, 3), np.float32)
        logits = np.empty((n, 3), np.float32)
        _lib.check(self.lib.pa_variant_forward_host(self.h, images.ctypes.data, n, probs.ctypes.data,
                                                    logits.ctypes.data))
        return probs, logits

    def close(self):
        self.lib.pa_variant_destroy(self.h)


def _golden(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("tag", ["g1", "g3"])
def test_variant_matches_reference_golden(golden_dir, tag):
    g = _golden(golden_dir, f"variant_{tag}.npz")
    sd = synthetic.variant_state_dict(seed=int(g["seed"]), gain=float(g["gain"]))
    m = NativeVariant(sd)
    probs, logits = m.forward(g["images"])
    m.close()
    assert np.abs(probs - g["probs"]).max() < TOL
    assert np.abs(logits - g["logits"]).max() < TOL * max(1.0, np.abs(g["logits"]).max())
    assert (probs.argmax(1) == g["probs"].argmax(1)).all()


def test_variant_two_layer_golden(golden_dir):
    g = _golden(golden_dir, "variant_l2.npz")
    sd = synthetic.variant_state_dict(seed=int(g["seed"]), gain=float(g["gain"]), gru_layers=2)
    m = NativeVariant(sd, gru_layers=2)
    probs, _ = m.forward(g["images"])
    m.close()
    assert np.abs(probs - g["probs"]).max() < TOL


@pytest.mark.parametrize("n", [1, 63, 64, 65, 130, 512, 515, 1024])
def test_variant_ragged_batches_vs_oracle(n):
    sd = synthetic.variant_state_dict(seed=5, gain=2.0)
    x = synthetic.variant_windows(n, seed=77 + n)
    m = NativeVariant(sd)
    probs, logits = m.forward(x)
    m.close()
    ref, inter = models_np.variant_forward(sd, x, return_intermediates=True)
    assert np.abs(probs - ref).max() < TOL
    assert np.abs(logits - inter["logits"]).max() < TOL * max(1.0, np.abs(inter["logits"]).max())
    assert np.abs(probs.sum(1) - 1).max() < 1e-5


def test_variant_empty_batch():
    m = NativeVariant(synthetic.variant_state_dict(seed=5))
    probs = np.empty((0, 3), np.float32)
    _lib.check(m.lib.pa_variant_forward_host(m.h, None, 0, probs.ctypes.data, None))
    m.close()


def test_variant_chunking_and_order_invariance():
    """Size-independent properties: device chunking must not change results; windows are
    independent, so permuting the batch permutes the output bit-exactly."""
    sd = synthetic.variant_state_dict(seed=6, gain=2.0)
    x = synthetic.variant_windows(1000, seed=3)
    whole = NativeVariant(sd)
    p0, _ = whole.forward(x)
    perm = np.random.default_rng(0).permutation(len(x))
    p1, _ = whole.forward(x[perm])
    whole.close()
    chunked = NativeVariant(sd, max_chunk=192)
    p2, _ = chunked.forward(x)
    chunked.close()
    assert np.abs(p0 - p2).max() < 1e-6
    assert np.abs(p0[perm] - p1).max() < 1e-6


def test_variant_missing_key_fails_loudly():
    sd = synthetic.variant_state_dict(seed=5)
    del sd["linear_3.bias"]
    with pytest.raises(_lib.PepperAmdError, match="linear_3.bias"):
        NativeVariant(sd)


def test_module_wrapper_and_checkpoint_loader(tmp_path):
    """TransducerGRU / ModelHandler mirror: module.-prefixed checkpoint, int8 and float inputs,
    CUDA and CPU tensors, train_mode logits."""
    from pepper_amd.variant.models.ModelHander import ModelHandler
    sd = synthetic.variant_state_dict(seed=8, gain=2.0)
    ckpt = synthetic.checkpoint_dict({k: torch.from_numpy(v) for k, v in sd.items()}, hidden_size=128,
                                     module_prefix=True)
    path = str(tmp_path / "model.pkl")
    torch.save(ckpt, path)
    model, hidden_size, gru_layers, epochs = ModelHandler.load_simple_model_for_training(path, 26, 28, 3)
    assert (hidden_size, gru_layers, epochs) == (128, 1, 1)
    x = synthetic.variant_windows(70, seed=9)
    ref, inter = models_np.variant_forward(sd, x, return_intermediates=True)
    xi = torch.from_numpy(x)
    out_cpu = model.eval()(xi.type(torch.FloatTensor), False)
    assert not out_cpu.is_cuda and np.abs(out_cpu.numpy() - ref).max() < TOL
    out_gpu = model(xi.cuda(), False)
    assert out_gpu.is_cuda and np.abs(out_gpu.cpu().numpy() - ref).max() < TOL
    out_f = model(xi.float().cuda(), False)
    # int8 input runs the fused kernel, float input the GEMM + Xp path: same math, different
    # accumulation order
    assert np.abs(out_f.cpu().numpy() - out_gpu.cpu().numpy()).max() < 2e-5
    lg = model(xi.cuda(), True)
    assert np.abs(lg.cpu().numpy() - inter["logits"]).max() < TOL * max(1.0, np.abs(inter["logits"]).max())
    # non-integral float input exercises the f32 loader for real
    xf = xi.float() * 0.37
    ref_f = models_np.variant_forward(sd, xf.numpy())
    assert np.abs(model(xf.cuda()).cpu().numpy() - ref_f).max() < TOL


def test_variant_full_bench_size_properties():
    """BASELINE-size batch (16384 + ragged tail, two device chunks): (i) a sample of windows against
    the oracle, (ii) chunking invariance, (iii) row-permutation equivariance, (iv) probabilities sum
    to one -- the size-independent checks at the size bench.py times."""
    n = 16384 + 37
    sd = synthetic.variant_state_dict(seed=15, gain=2.0)
    x = synthetic.variant_windows(n, seed=2718)
    m = NativeVariant(sd)                      # default max_chunk 16384 -> chunks of 16384 and 37
    p0, l0 = m.forward(x)
    perm = np.random.default_rng(1).permutation(n)
    p1, _ = m.forward(x[perm])
    m.close()
    small = NativeVariant(sd, max_chunk=4096)
    p2, _ = small.forward(x)
    small.close()
    assert np.isfinite(p0).all() and np.abs(p0.sum(1) - 1).max() < 1e-5
    assert np.abs(p0 - p2).max() < 1e-6
    assert np.abs(p0[perm] - p1).max() < 1e-6
    pick = np.random.default_rng(2).choice(n, 192, replace=False)
    pick[:3] = (0, 16383, n - 1)
    ref, inter = models_np.variant_forward(sd, x[pick], return_intermediates=True)
    assert np.abs(p0[pick] - ref).max() < TOL
    assert np.abs(l0[pick] - inter["logits"]).max() < TOL * max(1.0, np.abs(inter["logits"]).max())


def test_variant_run_to_run_determinism():
    """No atomics and a fixed split-K order anywhere on the model path: the same handle (and a fresh one) must return
    bit-identical results for the same input."""
    sd = synthetic.variant_state_dict(seed=17, gain=2.0)
    x = synthetic.variant_windows(3000, seed=31)
    a = NativeVariant(sd)
    p0, l0 = a.forward(x)
    p1, l1 = a.forward(x)
    a.close()
    b = NativeVariant(sd)
    p2, l2 = b.forward(x)
    b.close()
    assert np.array_equal(p0, p1) and np.array_equal(l0, l1)
    assert np.array_equal(p0, p2) and np.array_equal(l0, l2)


def _overflow_rows(m):
    rows = ctypes.c_int64()
    _lib.check(m.lib.pa_variant_overflow_rows(m.h, ctypes.byref(rows)))
    return rows.value


@pytest.mark.parametrize("case", ["linear_1_bias", "mlp_x38", "mlp_x40"])
def test_activations_beyond_the_f16_range_get_the_f32_results(case):
    """The split-f16 operand format holds |x| < 65504.  Checkpoints whose weights are all small (so the split kernels are
    chosen) but whose dense-layer ACTIVATIONS leave that range -- a huge linear_1 bias; linear_2..4 scaled so that the
    fourth layer's output passes 65504 in some rows (x38) or most rows (x40) -- must give the f32 reference's probabilities
    (simple_model.py:60-78 is f32 throughout), never inf / NaN: the MLP kernel re-runs the 64-row tiles concerned in f32 and
    counts them."""
    sd = synthetic.variant_state_dict(seed=5, gain=2.0)
    if case == "linear_1_bias":
        sd["linear_1.bias"][:32] = 2e5
    else:
        for name in ("linear_2", "linear_3", "linear_4"):
            sd[name + ".weight"] *= float(case[5:])
        sd["linear_5.weight"] *= 1e-4
    assert max(np.abs(v).max() for k, v in sd.items() if "weight" in k) < 64          # stays on the split kernels
    x = synthetic.variant_windows(300, seed=3)
    with np.errstate(over="ignore"):
        ref, inter = models_np.variant_forward(sd, x, return_intermediates=True)
    m = NativeVariant(sd)
    assert _overflow_rows(m) == 0
    probs, logits = m.forward(x)
    rows = _overflow_rows(m)
    m.close()
    assert np.isfinite(probs).all() and np.isfinite(logits).all()
    assert 0 < rows <= 300
    assert np.abs(probs - ref).max() < TOL
    assert np.abs(logits - inter["logits"]).max() < TOL * max(1.0, np.abs(inter["logits"]).max())
    # and a checkpoint inside the range never takes that path
    sd0 = synthetic.variant_state_dict(seed=5, gain=2.0)
    m0 = NativeVariant(sd0)
    m0.forward(x)
    assert _overflow_rows(m0) == 0
    m0.close()


def test_a_call_whose_workgroups_do_not_meet_is_run_again(monkeypatch, schedule):
    """The unit-split step loop needs the eight workgroups of a tile resident together; a group that does not meet (here:
    one member is told never to arrive, PA_UNIT_SPLIT_SABOTAGE) gives up after ~25 ms and the call is run again with the
    ordinary schedule -- same results, counted in pa_variant_split_fallbacks, and the out-of-range row counter does not see
    the garbage of the abandoned pass.  The handle then leaves the split alone for 256 small calls and comes back to it."""
    if schedule != "small-call":
        pytest.skip("the split is off in this schedule")
    sd = synthetic.variant_state_dict(seed=5)
    x = synthetic.variant_windows(700, seed=77)
    monkeypatch.setenv("PA_UNIT_SPLIT", "0")
    plain = NativeVariant(sd)                              # never splits
    monkeypatch.delenv("PA_UNIT_SPLIT")
    clean = NativeVariant(sd)                              # splits, undisturbed
    monkeypatch.setenv("PA_UNIT_SPLIT_SABOTAGE", "1")
    m = NativeVariant(sd)                                  # its first split launch has a member that never arrives
    monkeypatch.delenv("PA_UNIT_SPLIT_SABOTAGE")
    n, rows = ctypes.c_int64(-1), ctypes.c_int64(-1)
    got = m.forward(x[:300])[0]
    _lib.check(m.lib.pa_variant_split_fallbacks(m.h, ctypes.byref(n)))
    _lib.check(m.lib.pa_variant_overflow_rows(m.h, ctypes.byref(rows)))
    assert n.value == 1 and rows.value == 0
    assert np.array_equal(got, plain.forward(x[:300])[0])            # the second run IS the ordinary schedule
    assert np.abs(got - clean.forward(x[:300])[0]).max() <= 1e-4
    for k in range(260):                                   # through the hold-off
        m.forward(x[k:k + 16])
    assert np.array_equal(m.forward(x[300:650])[0], clean.forward(x[300:650])[0])    # ... and back on the split path
    _lib.check(m.lib.pa_variant_split_fallbacks(m.h, ctypes.byref(n)))
    _lib.check(clean.lib.pa_variant_split_fallbacks(clean.h, ctypes.byref(rows)))
    assert n.value == 1 and rows.value == 0
    # the four-member form of calls of 513-1024 windows
    monkeypatch.setenv("PA_UNIT_SPLIT_SABOTAGE", "1")
    m4 = NativeVariant(sd)
    monkeypatch.delenv("PA_UNIT_SPLIT_SABOTAGE")
    got = m4.forward(x)[0]
    _lib.check(m4.lib.pa_variant_split_fallbacks(m4.h, ctypes.byref(n)))
    assert n.value == 1 and np.array_equal(got, plain.forward(x)[0]) and np.abs(got - clean.forward(x)[0]).max() <= 1e-4
    for h in (plain, clean, m, m4):
        h.close()

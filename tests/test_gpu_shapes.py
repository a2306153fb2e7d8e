"""GPU parity at feature widths / window lengths other than the released models' (kernels are parametric in
F and T; the first layer's input projection then runs as a separate GEMM instead of inside the step loop).

* polish stack at F=100, 100-step windows: the literal synthetic shape of BASELINE.json's north_star
  ("region x 100-bp window x 100-feature"); not a reference shape, so the checker is the
  torch.nn restatement with seeded weights.
* variant stack at 21 x 48: the reference's HP image shape (pepper_variant/modules/python/Options.py:17-22).
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import models_np, torch_port
from pepper_amd import _lib, synthetic

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _rows(rng, n, seq, features):
    """uint8 rows that sum to <= 254 like a normalised pileup row, spread over `features` columns."""
    frac = rng.dirichlet(np.full(features, 4.0 / features), size=(n, seq))
    return np.floor(frac * 254.0).astype(np.uint8)


@pytest.mark.parametrize("features", [100, 16, 17, 40])
def test_polish_feature_widths(features):
    from pepper_amd.polish.models.simple_model import TransducerGRU
    sd = synthetic.polish_state_dict(seed=40 + features, gain=1.0, image_features=features)
    img = _rows(np.random.default_rng(features), 70, 1000, features)
    ref = torch_port.load_numpy_state_dict(torch_port.PolishPort(image_features=features), sd)
    m = TransducerGRU(1, features, 1, 128, 5, bidirectional=True).load_state_dict(sd)
    x0 = torch.from_numpy(img[:, :100]).float()
    h0 = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (70, 2, 128)).astype(np.float32))
    logits, hidden = m(x0, h0)
    with torch.no_grad():
        rl, rh = ref(x0, h0)
    assert np.abs(logits.numpy() - rl.numpy()).max() < TOL * max(1.0, float(rl.abs().max()))
    assert np.abs(hidden.numpy() - rh.numpy()).max() < TOL
    labels, phred, acc = m.predict_chunks(torch.from_numpy(img), return_acc=True)
    m.close()
    rlab, rphred, inter = models_np.polish_predict_chunks(sd, img, 128, return_intermediates=True)
    assert np.abs(acc.numpy() - inter["acc"]).max() < TOL
    top2 = np.sort(inter["acc"], axis=2)[:, :, -2:]
    tie = (top2[:, :, 1] - top2[:, :, 0]) < 2 * TOL
    assert ((labels.numpy() == rlab) | tie).all()
    assert (phred.numpy() == rphred).mean() > 0.99


@pytest.mark.parametrize("features,window", [(48, 21), (26, 21), (33, 33), (100, 16)])
def test_variant_feature_widths_and_windows(features, window):
    lib = _lib.load()
    sd = synthetic.variant_state_dict(seed=50 + features, gain=1.0, image_features=features, window=window)
    rng = np.random.default_rng(features * 100 + window)
    x = -rng.poisson(3.0, size=(300, window, features)).clip(0, 125).astype(np.int8)
    x[:, :, 0] = rng.integers(1, 6, size=(300, window))
    cfg = _lib.VariantConfig(features, window, 1, 3, 0, 0)
    names, data, numel, n, keep = _lib.marshal_state_dict(sd)
    h = ctypes.c_void_p()
    _lib.check(lib.pa_variant_create(ctypes.byref(cfg), names, data, numel, n, None, ctypes.byref(h)))
    probs = np.empty((300, 3), np.float32)
    logits = np.empty((300, 3), np.float32)
    _lib.check(lib.pa_variant_forward_host(h, x.ctypes.data, 300, probs.ctypes.data, logits.ctypes.data))
    lib.pa_variant_destroy(h)
    ref = torch_port.load_numpy_state_dict(torch_port.VariantPort(image_features=features, window=window), sd)
    with torch.no_grad():
        xt = torch.from_numpy(x).float()
        rp = ref(xt).numpy()
        rl = ref(xt, True).numpy()
    assert np.abs(probs - rp).max() < TOL
    assert np.abs(logits - rl).max() < TOL * max(1.0, np.abs(rl).max())

"""CPU leg of the weight-family stress test (tests/weight_families.py): the numpy emulation of the product's
arithmetic -- every contraction as hi*hi + hi*lo + lo*hi on f16 halves of f32 operands -- against the float64
restatement.  The GPU leg is tests/test_gpu_arith_modes.py::test_split_arithmetic_on_weight_families."""
import warnings

import numpy as np
import pytest

import weight_families as wf
from oracle import models_np


@pytest.mark.parametrize("family", ["mixed_row_scales", "heavy_tailed", "large_bias"])
def test_emulated_split_arithmetic_meets_the_bar(family):
    sd = wf.make(family, 70)
    x = wf.stress_windows(6, 7)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        p64, l64 = models_np.variant_forward_f64(sd, x)
        p32, inter = models_np.variant_forward(sd, x, return_intermediates=True)
        pe, le = wf.variant_forward_emulated(sd, x)
    assert np.isfinite(pe).all()
    # the float32 restatement and the emulated split arithmetic are both far inside 1e-4 of exact arithmetic
    assert max(wf.errors(p32, inter["logits"], p64, l64)) < 2e-5
    assert max(wf.errors(pe, le, p64, l64)) < 2e-5


def test_why_large_weights_leave_the_split_path():
    """The absolute floor of the split format (2^-25 on an activation) times a weight in the thousands is visible: the
    emulated split arithmetic misses the bar at |w| = 2e4 while the float32 restatement does not -- which is why
    pa_*_create sends such checkpoints to the exact-f32 kernels (api.hip kSplitMaxWeight = 64)."""
    sd = wf.make("near_f16_limit", 70)
    x = wf.stress_windows(6, 7)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        p64, l64 = models_np.variant_forward_f64(sd, x)
        p32, inter = models_np.variant_forward(sd, x, return_intermediates=True)
        pe, le = wf.variant_forward_emulated(sd, x)
    assert max(wf.errors(p32, inter["logits"], p64, l64)) < 2e-5
    assert max(wf.errors(pe, le, p64, l64)) > 1e-4

"""Design study (CPU, numpy; no GPU): what would cheaper machine products cost in accuracy?

Every contraction of the variant model is evaluated as the split-f16 kernels evaluate it -- hi*hi + hi*lo + lo*hi on
f16 halves of the f32 operands, three MFMAs per product -- or with the two correction terms on the block-scaled FP8
path (v_mfma_scale_f32_32x32x64_f8f6f4, twice the f16 rate: both operands of a correction MFMA are then e4m3 with one
power-of-two scale per 32 values along k), which would make a product cost two machine units instead of three
(DESIGN.md section 8.1).  Schemes:
    f16x3        what ships
    fp8corr_x    FP8 corrections in the input projections only (not recurrent: their error does not feed back)
    fp8corr_all  FP8 corrections everywhere
    f16x2        no hi(a)*lo(w) term at all (for scale)
against the float64 restatement, over the weight families of tests/weight_families.py and the uniform recipe at several
gains.  Prints one JSON line per case:  python tools/emulate_product_schemes.py [--windows 6] > profiles/..."""
import argparse
import json
import os
import sys
import warnings

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import weight_families as wf  # noqa: E402
from oracle import models_np  # noqa: E402
from pepper_amd import synthetic  # noqa: E402


def split(v):
    hi = v.astype(np.float16)
    lo = (v.astype(np.float32) - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def mx_e4m3(v, block=32):
    """Round to e4m3 (4 significant bits, normal range 2^-6 .. 448, subnormals down to 2^-9) after scaling each run of
    `block` values along the last axis by the power of two that puts its largest magnitude just under 448 (the MX
    block-scale the instruction takes).  float64 in / out."""
    shape = v.shape
    k = shape[-1]
    pad = (-k) % block
    a = np.pad(v, [(0, 0)] * (v.ndim - 1) + [(0, pad)]).reshape(-1, block)
    amax = np.abs(a).max(axis=1, keepdims=True)
    scale = np.where(amax > 0, 2.0 ** (np.floor(np.log2(np.where(amax > 0, amax, 1.0))) - 8), 1.0)    # amax / scale in [256, 512) -> clip below
    x = a / scale
    x = np.clip(x, -448.0, 448.0)
    mag = np.abs(x)
    e = np.floor(np.log2(np.where(mag > 0, mag, 1.0)))
    e = np.maximum(e, -6.0)                      # subnormal step below 2^-6
    step = 2.0 ** (e - 3)
    q = np.sign(x) * np.round(mag / step) * step
    out = (q * scale).reshape(*shape[:-1], k + pad)
    return out[..., :k]


def mm(a, w, scheme):
    ah, al = split(a)
    wh, wl = split(w)
    main = ah @ wh.T
    if scheme == "f16x3":
        return (main + ah @ wl.T + al @ wh.T).astype(np.float32)
    if scheme == "f16x2":
        return (main + al @ wh.T).astype(np.float32)
    if scheme == "fp8corr":
        return (main + mx_e4m3(ah) @ mx_e4m3(wl).T + mx_e4m3(al) @ mx_e4m3(wh).T).astype(np.float32)
    raise KeyError(scheme)


def forward(sd, images, scheme):
    f4 = np.float32
    x_scheme = {"f16x3": "f16x3", "f16x2": "f16x2", "fp8corr_x": "fp8corr", "fp8corr_all": "fp8corr"}[scheme]
    h_scheme = {"f16x3": "f16x3", "f16x2": "f16x2", "fp8corr_x": "f16x3", "fp8corr_all": "fp8corr"}[scheme]
    sig = lambda v: (1.0 / (1.0 + np.exp(-np.clip(v.astype(np.float64), -700, 700)))).astype(f4)   # noqa: E731

    def direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
        B, T, F = x.shape
        H = w_hh.shape[1]
        h, c = np.zeros((B, H), f4), np.zeros((B, H), f4)
        y = np.zeros((B, T, H), f4)
        xp = mm(x.reshape(B * T, F), w_ih, x_scheme).reshape(B, T, 4 * H) + (b_ih + b_hh)
        for t in (range(T - 1, -1, -1) if reverse else range(T)):
            g = xp[:, t] + mm(h, w_hh, h_scheme)
            c = sig(g[:, H:2 * H]) * c + sig(g[:, :H]) * np.tanh(g[:, 2 * H:3 * H])
            h = (sig(g[:, 3 * H:]) * np.tanh(c)).astype(f4)
            y[:, t] = h
        return y

    x = np.asarray(images).astype(f4)
    for prefix in ("encoder", "decoder"):
        x = np.concatenate([direction(x, sd[f"{prefix}.weight_ih_l0{s}"], sd[f"{prefix}.weight_hh_l0{s}"],
                                      sd[f"{prefix}.bias_ih_l0{s}"], sd[f"{prefix}.bias_hh_l0{s}"], rev)
                            for s, rev in (("", False), ("_reverse", True))], axis=2)
    a = x.reshape(x.shape[0], -1)
    for name in ("linear_1", "linear_2", "linear_3", "linear_4", "linear_5"):
        a = models_np.selu(mm(a, sd[f"{name}.weight"], h_scheme if name != "linear_1" else x_scheme) + sd[f"{name}.bias"])
    logits = (a.astype(np.float64) @ sd["output_layer_type.weight"].astype(np.float64).T + sd["output_layer_type.bias"]).astype(f4)
    return models_np.softmax(logits, axis=1), logits


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=6)
    args = ap.parse_args()
    cases = [("uniform_gain_%g" % g, synthetic.variant_state_dict(seed=3, gain=g)) for g in (0.25, 1.0, 2.0, 4.0, 8.0)]
    cases += [(name, wf.make(name, 70)) for name in ("mixed_row_scales", "heavy_tailed", "large_bias", "near_tie_head")]
    x = wf.stress_windows(args.windows, 7)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, sd in cases:
            p64, l64 = models_np.variant_forward_f64(sd, x)
            row = {"weights": name, "windows": args.windows, "max_abs_logit": float(np.abs(l64).max())}
            for scheme in ("f16x3", "fp8corr_x", "fp8corr_all", "f16x2"):
                p, lg = forward(sd, x, scheme)
                row[scheme] = {"max_abs_dlogit": float(np.abs(lg - l64).max()), "max_abs_dprob": float(np.abs(p - p64).max())}
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()

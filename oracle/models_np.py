"""ORACLE (test infrastructure, never the product path): CPU restatement of PEPPER's two
RNN predictors in plain numpy float32.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; pepper_amd/ never does (tests/test_no_oracle_in_product.py enforces it).

What is restated, and from where:
  * variant model  (bi-LSTM x2 + 5x(Linear+SELU) + Linear + Softmax)
      /root/reference/pepper_variant/modules/python/models/simple_model.py:23-46 (layers)
      /root/reference/pepper_variant/modules/python/models/simple_model.py:48-82 (forward)
  * polish model   (bi-GRU x2 + Linear, hidden hand-off encoder->decoder->next window)
      /root/reference/pepper/modules/python/models/simple_model.py:12-24 (layers), :27-42 (forward)
  * polish sliding-window loop (19 windows, softmax overlap-add, argmax, phred)
      /root/reference/pepper/modules/python/models/predict_distributed_cpu.py:43-90

The cell arithmetic itself lives in a third-party dependency that is not under
/root/reference: torch.nn.LSTM / GRU / Linear / SELU / Softmax (requirements.txt pins
torch 1.10.0; this container has torch 2.10.0).  The published equations restated here:
  LSTM  i,f,g,o = split(W_ih x + b_ih + W_hh h + b_hh);  c' = s(f) c + s(i) tanh(g);
        h' = s(o) tanh(c')                                  (gate order i,f,g,o)
  GRU   r = s(W_ir x + b_ir + W_hr h + b_hr); z likewise;
        n = tanh(W_in x + b_in + r * (W_hn h + b_hn)); h' = (1-z) n + z h   (order r,z,n)
  SELU  scale * (max(0,x) + min(0, alpha (exp(x)-1)))
Pinning: the reference repo has no tests or golden vectors for this path (SURVEY.md
section 4), so this restatement is pinned against outputs of the reference's own model
classes imported in the build container (tests/golden/make_golden.py ->
tests/golden/variant_*.npz, polish_*.npz; checked by tests/test_oracle_golden.py).
"""
import numpy as np

SELU_ALPHA = np.float32(1.6732632423543772848170429916717)
SELU_SCALE = np.float32(1.0507009873554804934193349852946)


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x))).astype(np.float32)


def selu(x):
    return (SELU_SCALE * np.where(x > 0, x, SELU_ALPHA * np.expm1(x))).astype(np.float32)


def softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(axis=axis, keepdims=True)).astype(np.float32)


def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """x [B,T,F] -> y [B,T,H]; zero initial (h, c)."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = np.zeros((B, H), np.float32)
    c = np.zeros((B, H), np.float32)
    y = np.zeros((B, T, H), np.float32)
    xp = x.astype(np.float32) @ w_ih.T + (b_ih + b_hh)
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        g = xp[:, t] + h @ w_hh.T
        i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
        c = _sigmoid(f) * c + _sigmoid(i) * np.tanh(gg)
        h = (_sigmoid(o) * np.tanh(c)).astype(np.float32)
        y[:, t] = h
    return y, h, c


def gru_direction(x, h0, w_ih, w_hh, b_ih, b_hh, reverse):
    """x [B,T,F], h0 [B,H] -> y [B,T,H], h_n [B,H]."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = h0.astype(np.float32)
    y = np.zeros((B, T, H), np.float32)
    xp = x.astype(np.float32) @ w_ih.T + b_ih
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        hp = h @ w_hh.T + b_hh
        r = _sigmoid(xp[:, t, :H] + hp[:, :H])
        z = _sigmoid(xp[:, t, H:2 * H] + hp[:, H:2 * H])
        n = np.tanh(xp[:, t, 2 * H:] + r * hp[:, 2 * H:])
        h = ((1.0 - z) * n + z * h).astype(np.float32)
        y[:, t] = h
    return y, h


def _bi_lstm(sd, prefix, x, layers):
    for layer in range(layers):
        outs = []
        for suffix, rev in (("", False), ("_reverse", True)):
            y, _, _ = lstm_direction(x, sd[f"{prefix}.weight_ih_l{layer}{suffix}"],
                                     sd[f"{prefix}.weight_hh_l{layer}{suffix}"],
                                     sd[f"{prefix}.bias_ih_l{layer}{suffix}"],
                                     sd[f"{prefix}.bias_hh_l{layer}{suffix}"], rev)
            outs.append(y)
        x = np.concatenate(outs, axis=2)
    return x


def variant_forward(sd, images, gru_layers=1, return_intermediates=False):
    """images [B,33,26] (int8 or float) -> softmax probs [B,3] float32.

    Follows simple_model.py:48-82 (dropouts are identity in eval mode).
    """
    x = np.asarray(images).astype(np.float32)
    enc = _bi_lstm(sd, "encoder", x, gru_layers)
    dec = _bi_lstm(sd, "decoder", enc, gru_layers)
    a = dec.reshape(dec.shape[0], -1)
    for name in ("linear_1", "linear_2", "linear_3", "linear_4", "linear_5"):
        a = selu(a @ sd[f"{name}.weight"].T + sd[f"{name}.bias"])
    logits = (a @ sd["output_layer_type.weight"].T + sd["output_layer_type.bias"]).astype(np.float32)
    probs = softmax(logits, axis=1)
    if return_intermediates:
        return probs, {"enc": enc, "dec": dec, "logits": logits}
    return probs


def polish_forward(sd, x, hidden, gru_layers=1):
    """One window: x [B,T,10], hidden [B,2,H] -> (logits [B,T,5], hidden [B,2,H]).

    Follows pepper simple_model.py:27-42: decoder h0 = encoder h_n; returned hidden =
    decoder h_n; index 0 = forward final state, index 1 = reverse final state.
    (Released models use gru_layers=1; for L>1 hidden is [B,2L,H] in PyTorch's
    layer-major, direction-minor order.)
    """
    x = np.asarray(x).astype(np.float32)
    h_in = np.asarray(hidden, np.float32)

    def bi_gru(prefix, inp, h0):
        hn = np.zeros_like(h0)
        for layer in range(gru_layers):
            outs = []
            for d, (suffix, rev) in enumerate((("", False), ("_reverse", True))):
                y, h = gru_direction(inp, h0[:, 2 * layer + d],
                                     sd[f"{prefix}.weight_ih_l{layer}{suffix}"],
                                     sd[f"{prefix}.weight_hh_l{layer}{suffix}"],
                                     sd[f"{prefix}.bias_ih_l{layer}{suffix}"],
                                     sd[f"{prefix}.bias_hh_l{layer}{suffix}"], rev)
                outs.append(y)
                hn[:, 2 * layer + d] = h
            inp = np.concatenate(outs, axis=2)
        return inp, hn

    y1, h_enc = bi_gru("gru_encoder", x, h_in)
    y2, h_dec = bi_gru("gru_decoder", y1, h_enc)
    logits = (y2 @ sd["dense1.weight"].T + sd["dense1.bias"]).astype(np.float32)
    return logits, h_dec


def polish_predict_chunks(sd, images, hidden_size, window=100, jump=50, overlap=50,
                          gru_layers=1, return_intermediates=False):
    """images uint8/float [B,1000,10] -> (labels uint8 [B,1000], phred uint8 [B,1000]).

    Follows predict_distributed_cpu.py:43-90: zero hidden per batch, windows at
    i = 0, 50, ..., 900, softmax over classes accumulated into [B,1000,5], max -> (value,
    label); counts = 1 on the first/last 50 positions and 2 elsewhere;
    phred = -10 log10(1 - value/counts) with inf -> 100; both cast to uint8
    (pepper DataStorePredict.py:70-76).
    """
    x = np.asarray(images).astype(np.float32)
    B, S, _ = x.shape
    C = sd["dense1.bias"].shape[0]
    hidden = np.zeros((B, 2 * gru_layers, hidden_size), np.float32)
    acc = np.zeros((B, S, C), np.float32)
    hiddens = []
    for i in range(0, S, jump):
        if i + window > S:
            break
        logits, hidden = polish_forward(sd, x[:, i:i + window], hidden, gru_layers)
        acc[:, i:i + window] += softmax(logits, axis=2)
        hiddens.append(hidden.copy())
    values = acc.max(axis=2)
    labels = acc.argmax(axis=2)
    counts = np.full((B, S), 2.0, np.float32)
    counts[:, :overlap] = 1.0
    counts[:, S - overlap:] = 1.0
    with np.errstate(divide="ignore"):
        phred = -10.0 * np.log10((1.0 - values / counts).astype(np.float32))
    phred[np.isinf(phred)] = 100
    out = (labels.astype(np.uint8), phred.astype(np.float32).astype(np.uint8))
    if return_intermediates:
        return out + ({"acc": acc, "hiddens": np.stack(hiddens, 0), "phred_f32": phred},)
    return out


# ---- float64 restatement of the variant forward ------------------------------------------------------------------
# For stress families of weights (mixed row scales, heavy tails, values near the f16 limit) the float32 restatement
# above is itself a few 1e-6 away from exact arithmetic; tests that judge an arithmetic scheme on such weights compare
# against this float64 form of the same equations (same citations as variant_forward).
def variant_forward_f64(sd, images, gru_layers=1):
    f8 = np.float64
    sig = lambda v: 1.0 / (1.0 + np.exp(-np.clip(v, -700.0, 700.0)))   # noqa: E731

    def direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
        B, T, _ = x.shape
        H = w_hh.shape[1]
        h, c = np.zeros((B, H), f8), np.zeros((B, H), f8)
        y = np.zeros((B, T, H), f8)
        xp = x @ w_ih.astype(f8).T + (b_ih.astype(f8) + b_hh.astype(f8))
        whh = w_hh.astype(f8).T
        for t in (range(T - 1, -1, -1) if reverse else range(T)):
            g = xp[:, t] + h @ whh
            c = sig(g[:, H:2 * H]) * c + sig(g[:, :H]) * np.tanh(g[:, 2 * H:3 * H])
            h = sig(g[:, 3 * H:]) * np.tanh(c)
            y[:, t] = h
        return y

    x = np.asarray(images).astype(f8)
    for prefix in ("encoder", "decoder"):
        for layer in range(gru_layers):
            x = np.concatenate([direction(x, sd[f"{prefix}.weight_ih_l{layer}{s}"], sd[f"{prefix}.weight_hh_l{layer}{s}"],
                                          sd[f"{prefix}.bias_ih_l{layer}{s}"], sd[f"{prefix}.bias_hh_l{layer}{s}"], rev)
                                for s, rev in (("", False), ("_reverse", True))], axis=2)
    a = x.reshape(x.shape[0], -1)
    for name in ("linear_1", "linear_2", "linear_3", "linear_4", "linear_5"):
        v = a @ sd[f"{name}.weight"].astype(f8).T + sd[f"{name}.bias"].astype(f8)
        a = float(SELU_SCALE) * np.where(v > 0, v, float(SELU_ALPHA) * np.expm1(np.minimum(v, 0.0)))
    logits = a @ sd["output_layer_type.weight"].astype(f8).T + sd["output_layer_type.bias"].astype(f8)
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True), logits

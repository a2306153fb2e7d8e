"""Generate golden vectors from the REFERENCE model classes (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports /root/reference's TransducerGRU classes (pure torch + Options constants), loads the
deterministic synthetic weights of pepper_amd.synthetic, runs them on deterministic inputs
and stores inputs + expected outputs as small .npz fixtures.  /root/reference does not exist
on the GPU box; only the fixtures (data) travel.  The reference inference loops need h5py /
onnxruntime / the compiled pybind module, none of which exist here, so the ~30 lines of
window-loop logic of pepper/.../predict_distributed_cpu.py:43-90 are driven below around the
imported reference model (ORT-vs-PyTorch CPU differences are not separately pinned).
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from pepper_amd import synthetic  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def to_torch(sd):
    return {k: torch.from_numpy(v.copy()) for k, v in sd.items()}


def variant_inputs(n):
    x = synthetic.variant_windows(n, seed=1234)
    # edge rows: all-zero window, int8 extremes (the encoder leaves cols 4/8-10/25 unclamped,
    # so wrapped values down to -128 / up to 127 do reach the model: SURVEY.md section 7)
    x[0] = 0
    x[1] = 127
    x[2] = -128
    x[3, :, :] = np.arange(26, dtype=np.int8)[None, :] - 13
    return x


@torch.no_grad()
def make_variant(tag, seed, gain, n=64):
    from pepper_variant.modules.python.models.simple_model import TransducerGRU
    sd = synthetic.variant_state_dict(seed=seed, gain=gain)
    model = TransducerGRU(image_features=26, gru_layers=1, hidden_size=128, num_classes=28,
                          num_classes_type=3, bidirectional=True)
    model.load_state_dict(to_torch(sd))
    model.eval()
    x = variant_inputs(n)
    xf = torch.from_numpy(x).type(torch.FloatTensor)
    probs = model(xf, False).numpy()
    logits = model(xf, True).numpy()
    enc, _ = model.encoder(xf[:2])
    dec, _ = model.decoder(enc)
    np.savez_compressed(os.path.join(OUT, f"variant_{tag}.npz"), seed=seed, gain=gain, images=x,
                        probs=probs, logits=logits, enc2=enc.numpy(), dec2=dec.numpy())
    print(tag, "probs[4:7]", probs[4:7])


@torch.no_grad()
def make_variant_two_layer(tag, seed, gain, n=8):
    """gru_layers comes from the checkpoint (ModelHander.py:21,27); exercise L=2 once."""
    from pepper_variant.modules.python.models.simple_model import TransducerGRU
    sd = synthetic.variant_state_dict(seed=seed, gain=gain, gru_layers=2)
    model = TransducerGRU(26, 2, 128, 28, 3, bidirectional=True)
    model.load_state_dict(to_torch(sd))
    model.eval()
    x = variant_inputs(n)
    probs = model(torch.from_numpy(x).float(), False).numpy()
    np.savez_compressed(os.path.join(OUT, f"variant_{tag}.npz"), seed=seed, gain=gain, images=x,
                        probs=probs, gru_layers=2)


@torch.no_grad()
def make_polish(tag, seed, gain, n=4):
    from pepper.modules.python.models.simple_model import TransducerGRU
    import torch.nn as nn
    sd = synthetic.polish_state_dict(seed=seed, gain=gain)
    model = TransducerGRU(1, 10, 1, 128, 5, bidirectional=True)
    model.load_state_dict(to_torch(sd))
    model.eval()
    imgs = synthetic.polish_chunks(n, seed=4321)
    imgs[0, 700:] = 0          # padded tail as chunk_images produces
    images = torch.from_numpy(imgs).type(torch.FloatTensor)
    # --- loop of predict_distributed_cpu.py:43-90, reference model in place of ORT ---
    hidden = torch.zeros(images.size(0), 2, 128)
    acc = torch.zeros((images.size(0), images.size(1), 5))
    hiddens, logits0 = [], None
    for i in range(0, 1000, 50):
        if i + 100 > 1000:
            break
        out, hidden = model(images[:, i:i + 100], hidden)
        if logits0 is None:
            logits0 = out.numpy().copy()
        hiddens.append(hidden.numpy().copy())
        layers = nn.Sequential(nn.Softmax(dim=2), nn.ZeroPad2d((0, 0, i, 1000 - (i + 100))))
        acc = torch.add(acc, layers(out))
    values, labels = torch.max(acc, 2)
    counts = torch.ones((values.size(0), values.size(1) - 100))
    counts = nn.ZeroPad2d((50, 50))(counts) + 1
    phred = -10 * torch.log10(1.0 - (values / counts))
    phred[phred == float("inf")] = 100
    np.savez_compressed(os.path.join(OUT, f"polish_{tag}.npz"), seed=seed, gain=gain, images=imgs,
                        logits_w0=logits0, hiddens=np.stack(hiddens, 0), acc=acc.numpy(),
                        labels=labels.numpy().astype(np.uint8),
                        phred=phred.numpy().astype(np.uint8), phred_f32=phred.numpy())
    print(tag, "labels", labels[1, :12].tolist(), "phred", phred[1, :6].tolist())


@torch.no_grad()
def make_polish_two_layer(tag, seed, gain, n=3):
    """Module-level forward with gru_layers=2 (hidden [B,4,H]); the reference window loop itself
    hard-codes TrainOptions.GRU_LAYERS = 1, so only forward(x, hidden) is meaningful for L > 1."""
    from pepper.modules.python.models.simple_model import TransducerGRU
    sd = synthetic.polish_state_dict(seed=seed, gain=gain, gru_layers=2)
    model = TransducerGRU(1, 10, 2, 128, 5, bidirectional=True)
    model.load_state_dict(to_torch(sd))
    model.eval()
    imgs = synthetic.polish_chunks(n, seed=777)[:, :100]
    rng = np.random.default_rng(5)
    hidden = rng.uniform(-0.5, 0.5, size=(n, 4, 128)).astype(np.float32)
    logits, hout = model(torch.from_numpy(imgs).float(), torch.from_numpy(hidden))
    np.savez_compressed(os.path.join(OUT, f"polish_{tag}.npz"), seed=seed, gain=gain, x=imgs, hidden=hidden,
                        logits=logits.numpy(), hidden_out=hout.numpy(), gru_layers=2)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "polish_l2":
        make_polish_two_layer("l2", seed=23, gain=2.0)
        sys.exit(0)
    torch.set_num_threads(8)
    make_variant("g1", seed=11, gain=1.0)
    make_variant("g3", seed=12, gain=3.0)
    make_variant_two_layer("l2", seed=13, gain=2.0)
    make_polish("g1", seed=21, gain=1.0)
    make_polish("g3", seed=22, gain=3.0)
    make_polish_two_layer("l2", seed=23, gain=2.0)

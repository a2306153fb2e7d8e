"""ORACLE / CPU BASELINE (test infrastructure, never the product path): the reference's
two predictors assembled from stock torch.nn modules on the CPU.

The reference's own CPU path is torch.nn.LSTM/GRU/Linear/SELU/Softmax executed by ATen
(/root/reference/pepper_variant/modules/python/models/predict_distributed_cpu.py:102-147
``predict_pytorch``; polish: /root/reference/pepper/modules/python/models/predict.py and
predict_distributed_cpu.py:43-90 with ONNX Runtime in place of the module).  The reference
Python cannot travel to the GPU box, so bench.py's ``cpu_baseline`` leg times THIS port
(kind="port"): the same torch.nn layer stack, same shapes, same weights, same ATen kernels.

Layer stacks restated from
  /root/reference/pepper_variant/modules/python/models/simple_model.py:23-46, 48-82
  /root/reference/pepper/modules/python/models/simple_model.py:12-24, 27-42
Pinned against the reference classes by tests/golden/make_golden.py +
tests/test_oracle_golden.py (same golden vectors as oracle/models_np.py).
"""
import numpy as np
import torch
import torch.nn as nn


class VariantPort(nn.Module):
    def __init__(self, image_features=26, gru_layers=1, num_classes_type=3, window=33):
        super().__init__()
        self.encoder = nn.LSTM(image_features, 256, num_layers=gru_layers, bidirectional=True,
                               batch_first=True)
        self.decoder = nn.LSTM(512, 256, num_layers=gru_layers, bidirectional=True,
                               batch_first=True)
        self.linear_1 = nn.Linear(512 * window, 512)
        self.linear_2 = nn.Linear(512, 512)
        self.linear_3 = nn.Linear(512, 512)
        self.linear_4 = nn.Linear(512, 512)
        self.linear_5 = nn.Linear(512, 512)
        self.output_layer_type = nn.Linear(512, num_classes_type)
        self.activation = nn.SELU()

    def forward(self, x, train_mode=False):
        x, _ = self.encoder(x)
        x, _ = self.decoder(x)
        x = torch.flatten(x, start_dim=1, end_dim=2)
        for lin in (self.linear_1, self.linear_2, self.linear_3, self.linear_4, self.linear_5):
            x = self.activation(lin(x))
        x = self.output_layer_type(x)
        return x if train_mode else torch.softmax(x, dim=1)


class PolishPort(nn.Module):
    def __init__(self, image_features=10, gru_layers=1, hidden_size=128, num_classes=5):
        super().__init__()
        self.gru_encoder = nn.GRU(image_features, hidden_size, num_layers=gru_layers,
                                  bidirectional=True, batch_first=True)
        self.gru_decoder = nn.GRU(2 * hidden_size, hidden_size, num_layers=gru_layers,
                                  bidirectional=True, batch_first=True)
        self.dense1 = nn.Linear(2 * hidden_size, num_classes)

    def forward(self, x, hidden):
        hidden = hidden.transpose(0, 1).contiguous()
        x, h = self.gru_encoder(x, hidden)
        x, h = self.gru_decoder(x, h)
        return self.dense1(x), h.transpose(0, 1).contiguous()


def load_numpy_state_dict(module, sd):
    module.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
    return module.eval()


@torch.no_grad()
def polish_predict_chunks(model, images, hidden_size, gru_layers=1, window=100, jump=50, overlap=50):
    """torch restatement of predict_distributed_cpu.py:43-90 (see oracle/models_np.py)."""
    images = torch.as_tensor(images).float()
    B, S, _ = images.shape
    hidden = torch.zeros(B, 2 * gru_layers, hidden_size)
    acc = torch.zeros(B, S, model.dense1.out_features)
    for i in range(0, S, jump):
        if i + window > S:
            break
        logits, hidden = model(images[:, i:i + window], hidden)
        acc[:, i:i + window] += torch.softmax(logits, dim=2)
    values, labels = torch.max(acc, 2)
    counts = torch.full((B, S), 2.0)
    counts[:, :overlap] = 1.0
    counts[:, S - overlap:] = 1.0
    phred = -10 * torch.log10(1.0 - values / counts)
    phred[phred == float("inf")] = 100
    return labels.numpy().astype(np.uint8), phred.numpy().astype(np.uint8)

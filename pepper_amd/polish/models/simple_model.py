"""HIP-backed counterpart of the polish ``TransducerGRU``.

Mirrors /root/reference/pepper/modules/python/models/simple_model.py:5-48:
``forward(x, hidden) -> (logits [B,T,5], hidden [B,2L,H])`` with the encoder->decoder->next-window
hidden hand-off, plus ``predict_chunks`` = the whole sliding-window loop of
/root/reference/pepper/modules/python/models/predict_distributed_cpu.py:43-90 on the device.
"""
import ctypes
import os

import torch

from pepper_amd import _lib
from pepper_amd.polish.Options import ImageSizeOptions, TrainOptions


class TransducerGRU(object):
    def __init__(self, image_channels, image_features, gru_layers, hidden_size, num_classes,
                 bidirectional=True, device=None, max_chunk=0):
        if not bidirectional:
            raise ValueError("the reference inference path only instantiates bidirectional=True")
        self.image_features = image_features
        self.hidden_size = hidden_size
        self.num_layers = gru_layers
        self.num_classes = num_classes
        self.max_chunk = max_chunk
        self.device = torch.cuda.current_device() if device is None and torch.cuda.is_available() else (device or 0)
        self._handle = None
        self._stream = None
        self._state = None

    def load_state_dict(self, state_dict, strict=True):
        lib = _lib.load()
        self.close()
        cfg = _lib.PolishConfig(self.image_features, self.hidden_size, self.num_layers, self.num_classes,
                                ImageSizeOptions.SEQ_LENGTH, TrainOptions.TRAIN_WINDOW,
                                TrainOptions.WINDOW_JUMP, ImageSizeOptions.SEQ_OVERLAP, self.device,
                                self.max_chunk)
        names, data, numel, n, keep = _lib.marshal_state_dict(state_dict)
        # (priority -1: a stream of the device's high-priority queues -- a caller whose passes must not queue behind other
        # streams' long kernels in a shared hardware queue, polish/fused.py)
        self._stream = torch.cuda.Stream(device=self.device, priority=int(getattr(self, "stream_priority", 0)))
        handle = ctypes.c_void_p()
        _lib.check(lib.pa_polish_create(ctypes.byref(cfg), names, data, numel, n,
                                        ctypes.c_void_p(self._stream.cuda_stream), ctypes.byref(handle)))
        self._handle = handle
        self._state = state_dict
        return self

    def clone(self, stream_priority=None):
        """A second, independent handle on the same weights (own stream, own staging buffers): what runs a second block on
        the device while this one's pass is under way (pepper_amd/hostpipe.py polish_lanes).  stream_priority: 0 / -1 (high) for
        the clone's stream; None: this object's."""
        other = TransducerGRU(1, self.image_features, self.num_layers, self.hidden_size, self.num_classes, device=self.device,
                              max_chunk=self.max_chunk)
        other.stream_priority = int(getattr(self, "stream_priority", 0) if stream_priority is None else stream_priority)
        return other.load_state_dict(self._state)

    def eval(self):
        return self

    def cuda(self, device=None):
        return self

    def cpu(self):
        return self

    def close(self):
        if self._handle is not None:
            _lib.load().pa_polish_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        if self._handle is None:
            raise _lib.PepperAmdError("TransducerGRU has no weights: call load_state_dict first")
        return self._handle

    def init_hidden(self, batch_size, num_layers, bidirectional=True):
        return torch.zeros(batch_size, (2 if bidirectional else 1) * num_layers, self.hidden_size)

    def __call__(self, x, hidden):
        return self.forward(x, hidden)

    def _enter(self, dev):
        cur = torch.cuda.current_stream(dev)
        self._stream.wait_stream(cur)
        return cur

    def _leave(self, cur, tensors):
        for t in tensors:
            if t is not None:
                t.record_stream(self._stream)
        cur.wait_stream(self._stream)

    def forward(self, x, hidden):
        lib = _lib.load()
        x = torch.as_tensor(x)
        on_cpu = not x.is_cuda
        dev = torch.device("cuda", self.device)
        x = x.to(dev, torch.float32).contiguous()
        hidden = torch.as_tensor(hidden).to(dev, torch.float32).contiguous()
        n, T = x.shape[0], x.shape[1]
        if x.shape[2] != self.image_features or tuple(hidden.shape) != (n, 2 * self.num_layers, self.hidden_size):
            raise ValueError("bad x / hidden shape")
        logits = torch.empty((n, T, self.num_classes), dtype=torch.float32, device=dev)
        hidden_out = torch.empty_like(hidden)
        cur = self._enter(dev)
        _lib.check(lib.pa_polish_forward_device(self.handle, x.data_ptr(), hidden.data_ptr(), n, T,
                                                logits.data_ptr(), hidden_out.data_ptr()))
        self._leave(cur, (x, hidden, logits, hidden_out))
        return (logits.cpu(), hidden_out.cpu()) if on_cpu else (logits, hidden_out)

    def predict_chunks_into(self, images, labels, phred):
        """Host arrays in, host arrays out (numpy uint8, C-contiguous; page-locked memory makes the copies asynchronous):
        images [B,1000,10] -> labels [B,1000], phred [B,1000] filled in place.  What the lane pipeline calls
        (pepper_amd/hostpipe.py)."""
        import numpy as np
        for a in (images, labels, phred):
            if a.dtype != np.uint8 or not a.flags.c_contiguous:
                raise ValueError("predict_chunks_into wants C-contiguous uint8 arrays")
        n = images.shape[0]
        if labels.shape != (n, images.shape[1]) or phred.shape != labels.shape:
            raise ValueError("labels / phred must be [B, seq_length]")
        _lib.check(_lib.load().pa_polish_predict_host(self.handle, images.ctypes.data, n, labels.ctypes.data,
                                                      phred.ctypes.data, None))

    def predict_chunk_parts_into(self, parts):
        """predict_chunks_into over several host blocks as one sequence of chunks: parts = [(images [n_p,1000,10], labels
        [n_p,1000], phred [n_p,1000]), ...].  One series of full-sized device passes whatever the blocks' sizes
        (pa_polish_predict_host_parts)."""
        import numpy as np
        k = len(parts)
        for images, labels, phred in parts:
            for a in (images, labels, phred):
                if a.dtype != np.uint8 or not a.flags.c_contiguous:
                    raise ValueError("predict_chunk_parts_into wants C-contiguous uint8 arrays")
            if labels.shape != (images.shape[0], images.shape[1]) or phred.shape != labels.shape:
                raise ValueError("labels / phred must be [B, seq_length]")
        ptr = lambda col: (ctypes.c_void_p * k)(*[p[col].ctypes.data for p in parts])   # noqa: E731
        counts = (ctypes.c_int64 * k)(*[p[0].shape[0] for p in parts])
        _lib.check(_lib.load().pa_polish_predict_host_parts(self.handle, k, ptr(0), counts, ptr(1), ptr(2)))

    def predict_chunks(self, images, return_acc=False):
        """images uint8 [B,1000,10] -> (labels uint8 [B,1000], phred uint8 [B,1000][, acc])."""
        lib = _lib.load()
        images = torch.as_tensor(images)
        on_cpu = not images.is_cuda
        dev = torch.device("cuda", self.device)
        if images.dtype != torch.uint8:
            raise ValueError("polish images are uint8 (pepper DataStore.py:60)")
        if on_cpu:
            # host buffers: device passes with the copies of the neighbouring passes beside the kernels
            from pepper_amd.variant.models.simple_model import _pinned_empty
            images = images.contiguous()
            n, S = images.shape[0], images.shape[1]
            labels = _pinned_empty((n, S), torch.uint8)
            phred = _pinned_empty((n, S), torch.uint8)
            acc = _pinned_empty((n, S, self.num_classes), torch.float32) if return_acc else None
            _lib.check(lib.pa_polish_predict_host(self.handle, images.data_ptr(), n, labels.data_ptr(), phred.data_ptr(),
                                                  acc.data_ptr() if acc is not None else None))
            return (labels, phred) + ((acc,) if return_acc else ())
        images = images.to(dev).contiguous()
        n, S = images.shape[0], images.shape[1]
        labels = torch.empty((n, S), dtype=torch.uint8, device=dev)
        phred = torch.empty((n, S), dtype=torch.uint8, device=dev)
        acc = torch.empty((n, S, self.num_classes), dtype=torch.float32, device=dev) if return_acc else None
        cur = self._enter(dev)
        _lib.check(lib.pa_polish_predict_device(self.handle, images.data_ptr(), n, labels.data_ptr(),
                                                phred.data_ptr(), acc.data_ptr() if acc is not None else None))
        self._leave(cur, (images, labels, phred, acc))
        out = (labels, phred) + ((acc,) if return_acc else ())
        return tuple(t.cpu() for t in out) if on_cpu else out

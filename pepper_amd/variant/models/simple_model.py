"""HIP-backed counterpart of the variant ``TransducerGRU``.

Mirrors /root/reference/pepper_variant/modules/python/models/simple_model.py:6-87: same
constructor arguments, same state_dict keys/shapes, ``forward(x, train_mode=False)`` returning
softmax probabilities [B,3] (logits when ``train_mode``).  All arithmetic runs in
libpepper_amd.so on the GPU; torch is used only for device memory and stream ordering.
"""
import ctypes

import torch

from pepper_amd import _lib
from pepper_amd.variant.Options import ImageSizeOptions


def _pinned_empty(shape, dtype):
    """Page-locked result buffer (the D2H copies of the host entry points are asynchronous only into pinned memory);
    falls back to pageable memory where pinning is refused."""
    try:
        return torch.empty(shape, dtype=dtype, pin_memory=True)
    except RuntimeError:
        return torch.empty(shape, dtype=dtype)


# Per thread: the priority of the stream a handle made on this thread gets (0; -1 = one of the device's high-priority queues).  The
# loaders keep the reference's signatures, so a caller that wants its forwards not to queue behind other streams' long kernels in a
# shared hardware queue (variant/fused.py) sets NEW_HANDLES.stream_priority around its load.
import threading as _threading
NEW_HANDLES = _threading.local()

class TransducerGRU(object):
    def __init__(self, image_features, gru_layers, hidden_size, num_classes, num_classes_type,
                 bidirectional=True, device=None, max_chunk=0):
        if not bidirectional:
            raise ValueError("the reference inference path only instantiates bidirectional=True")
        self.image_features = image_features
        self.hidden_size = hidden_size          # kept for parity; layer widths are fixed at 256/512
        self.bidirectional = bidirectional
        self.num_layers = gru_layers
        self.num_classes = num_classes
        self.num_classes_type = num_classes_type
        self.window = ImageSizeOptions.CANDIDATE_WINDOW_SIZE + 1
        self.max_chunk = max_chunk
        self.device = torch.cuda.current_device() if device is None and torch.cuda.is_available() else (device or 0)
        self._handle = None
        self._stream = None
        self.training = False

    # ---- nn.Module-like surface used by predict() ------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        lib = _lib.load()
        self.close()
        cfg = _lib.VariantConfig(self.image_features, self.window, self.num_layers,
                                 self.num_classes_type, self.device, self.max_chunk)
        names, data, numel, n, keep = _lib.marshal_state_dict(state_dict)
        self._stream = torch.cuda.Stream(device=self.device, priority=int(getattr(NEW_HANDLES, "stream_priority", 0)))
        handle = ctypes.c_void_p()
        _lib.check(lib.pa_variant_create(ctypes.byref(cfg), names, data, numel, n,
                                         ctypes.c_void_p(self._stream.cuda_stream), ctypes.byref(handle)))
        self._handle = handle
        return self

    def eval(self):
        self.training = False
        return self

    def cuda(self, device=None):
        return self

    def cpu(self):
        return self

    def close(self):
        if self._handle is not None:
            _lib.load().pa_variant_destroy(self._handle)
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

    def __call__(self, x, train_mode=False):
        return self.forward(x, train_mode)

    def forward(self, x, train_mode=False):
        """x: [B, 33, 26] int8 (as stored in the images HDF5) or float tensor, CPU or GPU."""
        lib = _lib.load()
        x = torch.as_tensor(x)
        if x.dim() != 3 or x.shape[1] != self.window or x.shape[2] != self.image_features:
            raise ValueError(f"expected [B,{self.window},{self.image_features}], got {tuple(x.shape)}")
        on_cpu = not x.is_cuda
        dev = torch.device("cuda", self.device)
        if on_cpu and x.dtype == torch.int8:
            # host buffers (what the predict loop hands over: a file's packed int8 windows, page-locked): device passes
            # with the H2D copy of the next pass and the D2H copy of the previous one beside the kernels
            x = x.contiguous()
            n = x.shape[0]
            probs = _pinned_empty((n, self.num_classes_type), torch.float32)
            logits = _pinned_empty((n, self.num_classes_type), torch.float32) if train_mode else None
            _lib.check(lib.pa_variant_forward_host(self.handle, x.data_ptr(), n, probs.data_ptr(),
                                                   logits.data_ptr() if logits is not None else None))
            return logits if train_mode else probs
        if x.dtype not in (torch.int8, torch.float32):
            x = x.to(torch.float32)
        x = x.to(dev).contiguous()
        n = x.shape[0]
        probs = torch.empty((n, self.num_classes_type), dtype=torch.float32, device=dev)
        logits = torch.empty_like(probs) if train_mode else None
        cur = torch.cuda.current_stream(dev)
        self._stream.wait_stream(cur)
        fn = lib.pa_variant_forward_device if x.dtype == torch.int8 else lib.pa_variant_forward_device_f32
        _lib.check(fn(self.handle, x.data_ptr(), n, probs.data_ptr(),
                      logits.data_ptr() if logits is not None else None))
        for t in (x, probs, logits):
            if t is not None:
                t.record_stream(self._stream)
        cur.wait_stream(self._stream)
        out = logits if train_mode else probs
        return out.cpu() if on_cpu else out

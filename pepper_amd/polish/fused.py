"""polish() with image generation and consensus inference fused (opt-in: polish(..., fused_inference=True) /
PEPPER_AMD_FUSED_POLISH=1).

The reference's steps talk through files (/root/reference/pepper/modules/python/polish.py:94-117: make_images writes the image HDF5
files, call_consensus reads them back).  Here an image-generation worker keeps the chunks the chain left on the device
(pa_polish_chain_device_chunks), gathers a few thousand of them, and hands them to the polish model there
(pa_polish_predict_device: the 19-window loop with hidden carry, labels and phred per position); both HDF5 stores are still
written -- the image files by the workers as before, one prediction file per worker with the reference's layout
(predictions/<contig>/<contig>-<start>-<end>/<chunk id>/...), which perform_stitch globs as it does the callers' files.
"""
import os
import queue
import threading

import numpy as np
import torch

from pepper_amd.polish.DataStorePredict import DataStore
from pepper_amd.polish.Options import ImageSizeOptions
from pepper_amd.polish.models.ModelHander import ModelHandler


class _DeviceChunks(object):
    """A device address as torch sees it (no copy): uint8 [n, seq, features]."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "|u1", "data": (int(ptr), False), "version": 3}


class FusedConsensus(object):
    """One per polish() run: up to HANDLES model handles per device, a small pool of pass threads that run the model over full
    gather buffers and write the predictions, and, per worker, two device buffers the chain's chunks are gathered in
    (PASS_CHUNKS each: a pass worth launching; the small-call
    schedule of the step loops is several times slower per chunk)."""
    PASS_CHUNKS = int(os.environ.get("PEPPER_AMD_FUSED_PASS_CHUNKS", 4096))
    HANDLES = 2          # passes in flight per device (own stream and workspace each)

    def __init__(self, model_path, output_directory):
        self.model_path = model_path
        self.output_directory = output_directory
        self.models, self.models_lock = {}, threading.Lock()
        self.chunks = 0
        self.handles = max(1, int(os.environ.get("PEPPER_AMD_FUSED_HANDLES", self.HANDLES)))
        self.stream_priority = int(os.environ.get("PEPPER_AMD_FUSED_STREAM_PRIORITY", -1))
        # the threads that run the model passes and write their predictions (the image workers only gather chunks): a pass waits
        # for a free handle of its device in _model, so a few threads more than handles keep every handle busy
        from concurrent.futures import ThreadPoolExecutor
        self.passes = ThreadPoolExecutor(max_workers=2 * self.handles + 2, thread_name_prefix="fused-consensus-pass")

    def _model(self, device):
        """(the device's entry, a free model handle of it): made on first use, at most `handles` of them."""
        with self.models_lock:
            entry = self.models.get(device)
            if entry is None:
                entry = self.models[device] = {"free": queue.Queue(), "made": 0, "all": []}
            make = entry["free"].empty() and entry["made"] < self.handles
            if make:
                entry["made"] += 1
                first = entry["all"][0] if entry["all"] else None
        if not make:
            while True:
                try:
                    model = entry["free"].get(timeout=1.0)
                except queue.Empty:
                    if entry.get("failed") is None:
                        continue
                    model = None
                if model is None:       # (a load that failed hands its error to everyone waiting for a handle)
                    raise RuntimeError("fused consensus: the model could not be loaded") from entry.get("failed")
                return entry, model
        try:
            torch.cuda.set_device(device)
            if first is not None:
                model = first.clone()
            else:
                loaded = ModelHandler.load_simple_model_for_training(self.model_path, input_channels=ImageSizeOptions.IMAGE_CHANNELS,
                                                                     image_features=ImageSizeOptions.IMAGE_HEIGHT,
                                                                     seq_len=ImageSizeOptions.SEQ_LENGTH,
                                                                     num_classes=ImageSizeOptions.TOTAL_LABELS)[0]
                # The passes' stream is one of the device's HIGH-PRIORITY queues: a pass is 19 windows x a few short kernels, and on
                # an ordinary stream every one of them waits in a hardware queue it shares with an image worker's stream behind
                # that worker's 10 ms alignment kernel (16 queues, ~20 streams) -- 3 500 launches x ~12 ms was the whole of what
                # the fused form lost.  (The loader's signature is the reference's: the handle is re-made with the priority.)
                model = loaded.clone(stream_priority=self.stream_priority)
                loaded.close()
        except BaseException as err:
            with self.models_lock:
                entry["made"] -= 1
                entry["failed"] = err
            for _ in range(64):
                entry["free"].put(None)
            raise
        with self.models_lock:
            entry["all"].append(model)
        return entry, model

    def worker(self, thread_id, device):
        return _Worker(self, thread_id, device)

    def close(self):
        self.passes.shutdown(wait=True)
        for entry in self.models.values():
            for model in entry["all"]:
                model.close()
        self.models.clear()


class _Worker(object):
    """One image worker's side: TWO gather buffers.  A full one is handed to the owner's pass threads (model pass + prediction
    write in the background) while the worker goes on filling the other with its next chain calls -- the worker waits only when
    both are in flight.  (Until round 6 the worker ran the pass itself: sixteen workers queued for two model handles with their
    chains idle, and fused polish was slower than the three steps it replaces.)"""

    def __init__(self, owner, thread_id, device):
        self.owner, self.device = owner, device
        self.seq, self.features = ImageSizeOptions.SEQ_LENGTH, ImageSizeOptions.IMAGE_HEIGHT
        torch.cuda.set_device(device)
        n = owner.PASS_CHUNKS
        self.sets = [{"buffer": torch.empty((n, self.seq, self.features), dtype=torch.uint8, device=torch.device("cuda", device)),
                      "position": np.empty((n, self.seq), np.int64), "index": np.empty((n, self.seq), np.int64), "meta": [],
                      "pending": None} for _ in range(2)]
        self.cur = 0
        self.n = 0
        # the gather copies run on this worker's own stream: the device's default stream is shared by every thread of the process,
        # and a copy queued there waits behind whatever another thread made it wait for (a whole model pass)
        self.copy_stream = torch.cuda.Stream(device=device)
        self.store = DataStore(owner.output_directory + "pepper_prediction_fused_" + str(thread_id) + ".hdf", mode='w')
        self.store_lock = threading.Lock()          # (two passes of this worker may finish at the same time)
        self.failed = False

    @property
    def buffer(self):
        return self.sets[self.cur]["buffer"]

    @property
    def position(self):
        return self.sets[self.cur]["position"]

    @property
    def index(self):
        return self.sets[self.cur]["index"]

    @property
    def meta(self):
        return self.sets[self.cur]["meta"]

    def add(self, contig, starts, stops, chunk_counts, device_images, position, index):
        """The chunks of one chain call: device_images = their address on the device, position / index = numpy views of the
        chain's page-locked copies ([total, seq] int64: copied here, the chain overwrites them in its next run)."""
        meta = [(contig, int(a), int(b), cid) for a, b, c in zip(starts, stops, chunk_counts) for cid in range(int(c))]
        total, at = len(meta), 0
        torch.cuda.set_device(self.device)
        while at < total:
            take = min(total - at, self.owner.PASS_CHUNKS - self.n)
            src = torch.as_tensor(_DeviceChunks(device_images + at * self.seq * self.features, (take, self.seq, self.features)),
                                  device=torch.device("cuda", self.device))
            with torch.cuda.stream(self.copy_stream):
                self.buffer[self.n:self.n + take].copy_(src)
            self.position[self.n:self.n + take] = position[at:at + take]
            self.index[self.n:self.n + take] = index[at:at + take]
            self.meta.extend(meta[at:at + take])
            self.n += take
            at += take
            if self.n == self.owner.PASS_CHUNKS:
                self.copy_stream.synchronize()                 # (the pass thread reads the buffer on the model's stream)
                self.flush()
        self.copy_stream.synchronize()                         # (the chain overwrites its chunks in its next run)

    def add_host(self, region, images, positions, chunk_ids):
        """An interval that went through the host form (a pile beyond the reservoir cap, a span the packed reader refused): its
        chunks as lists of [seq, features] uint8 arrays and [seq, 2] (position, index) arrays."""
        contig, start, end = region
        for image, pos, cid in zip(images, positions, chunk_ids):
            torch.cuda.set_device(self.device)
            with torch.cuda.stream(self.copy_stream):
                self.buffer[self.n].copy_(torch.from_numpy(np.ascontiguousarray(image, np.uint8)))
            pos = np.asarray(pos, np.int64).reshape(self.seq, 2)
            self.position[self.n], self.index[self.n] = pos[:, 0], pos[:, 1]
            self.meta.append((str(contig), int(start), int(end), int(cid)))
            self.n += 1
            if self.n == self.owner.PASS_CHUNKS:
                self.copy_stream.synchronize()
                self.flush()

    def _wait(self, k):
        pending, self.sets[k]["pending"] = self.sets[k]["pending"], None
        if pending is not None:
            pending.result()                 # (re-raises what the pass raised)

    def _run_pass(self, k, n):
        """On a pass thread: the model over the first n chunks of set k, then their predictions into this worker's file."""
        st = self.sets[k]
        entry, model = self.owner._model(self.device)
        try:
            torch.cuda.set_device(self.device)
            with torch.cuda.stream(model._stream):          # (this thread's "current stream": not the process-wide default one)
                labels, phred = model.predict_chunks(st["buffer"][:n])
                labels, phred = labels.cpu().numpy(), phred.cpu().numpy()
        finally:
            entry["free"].put(model)
        meta = st["meta"][:n]
        contigs = np.array([m[0] for m in meta], dtype='S')
        with self.store_lock:
            self.store.write_predictions_block(contigs, np.array([m[1] for m in meta], np.int64), np.array([m[2] for m in meta], np.int64),
                                               np.array([m[3] for m in meta], np.int64), st["position"][:n], st["index"][:n], labels, phred)
        with self.owner.models_lock:
            self.owner.chunks += n
        del st["meta"][:n]

    def flush(self):
        """Hand the current set to the pass threads and move to the other one (waiting for ITS pass, if that is still running)."""
        if self.n == 0:
            return
        k, n = self.cur, self.n
        self.sets[k]["pending"] = self.owner.passes.submit(self._run_pass, k, n)
        self.cur, self.n = k ^ 1, 0
        self._wait(self.cur)

    def close(self, failed=False):
        if failed:
            for k in (0, 1):
                try:
                    self._wait(k)
                except BaseException:      # noqa: BLE001 -- the run is being abandoned already
                    pass
            self.store.abort()
            return
        if self.n:
            torch.cuda.set_device(self.device)
            self.copy_stream.synchronize()
        self.flush()
        for k in (0, 1):
            self._wait(k)
        self.store.close()

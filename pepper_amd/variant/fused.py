"""call_variant with image generation and inference fused (opt-in: options.fused_inference / PEPPER_AMD_FUSED_CALL_VARIANT=1).

The reference's three steps talk through files (/root/reference/pepper_variant/modules/python/CallVariant.py:74-104: make_images
writes the image HDF5 files, run_inference reads them back).  Here an image-generation worker hands a group's candidate windows
to the model WHERE THE ENCODER LEFT THEM on the device (pa_encoder_device_images -> pa_variant_forward_device: no D2H -> HDF5 ->
H2D round trip for the model's input) and queues the predictions for one writer thread; both HDF5 files are still written, with
the reference's layouts -- the image files by the workers as before, `pepper_prediction.hdf` (predictions/batch_<n> groups of
options.batch_size candidates) by the writer in arrival order.  Which batch a candidate lands in differs from the unfused run (the
candidate finder sorts sites itself, CandidateFinder.py:356-581); every candidate's record is the same.
"""
import ctypes
import os
import queue
import threading
import time

import numpy as np
import torch

from pepper_amd import _lib
from pepper_amd.variant.DataStorePredict import DataStore
from pepper_amd.variant.Options import ImageSizeOptions
from pepper_amd.variant.models.ModelHander import ModelHandler


class FusedPredictor(object):
    """One per call_variant run: up to HANDLES model handles per device (made on first use; a worker takes a free one for its
    forward) and one writer thread over a queue of (contig, arrays, probabilities)."""

    def __init__(self, options, output_filepath):
        self.options = options
        self.batch_size = int(options.batch_size)
        self.models = {}
        self.models_lock = threading.Lock()
        # (the append-only writer: ~9 000 batch groups per 256 Mb cost the writer thread 0.3 ms each through libhdf5)
        self.store = DataStore(output_filepath + "pepper_prediction.hdf", mode='w', bulk=True)
        self.queue = queue.Queue(maxsize=64)
        self.error = None
        self.batch_no = 0
        self.windows = 0
        self.forward_seconds = 0.0
        self.timer_lock = threading.Lock()                   # the stage timers are added to from several threads
        self.write_seconds = self.select_seconds = 0.0       # what the two threads spent working (not waiting)
        self.select_error = None
        self.pending = {}            # per image worker: arrays of its candidates that have not filled a batch yet
        self.filename = output_filepath + "pepper_prediction.hdf"
        # the candidate finder's selection + record text of a batch (FastCandidates.native_batch_arrays, inside the I/O library)
        # runs on a third thread as soon as the batch exists, so that step 3 starts with most of its per-batch work done
        self.segments = {}
        self.select_queue = queue.Queue()
        self.selector = None
        if not getattr(options, "fused_candidates_off", False):
            # (two threads: a batch is ~0.4 ms here against ~0.3 ms in the writer, and the library calls release the interpreter)
            self.selector = [threading.Thread(target=self._select_loop, name="fused-candidate-selection-%d" % k, daemon=True)
                             for k in range(max(1, int(os.environ.get("PEPPER_AMD_FUSED_SELECTORS", 2))))]
            for t in self.selector:
                t.start()
        self.writer = threading.Thread(target=self._write_loop, name="fused-prediction-writer", daemon=True)
        self.writer.start()

    # ---- model ----
    HANDLES = 2          # forwards in flight per device: while one handle's results come back, the other's kernels run

    def _model(self, device):
        """A free model handle of the device (made on first use, at most HANDLES of them: own stream and workspace each)."""
        with self.models_lock:
            entry = self.models.get(device)
            if entry is None:
                entry = self.models[device] = {"free": queue.Queue(), "made": 0, "all": []}
            make = entry["free"].empty() and entry["made"] < max(1, int(os.environ.get("PEPPER_AMD_FUSED_HANDLES", self.HANDLES)))
            if make:
                entry["made"] += 1
        if make:
            try:
                torch.cuda.set_device(device)
                # PEPPER_AMD_FUSED_STREAM_PRIORITY=-1: the forwards' stream on the device's high-priority queues (default 0: measured no
                # gain, and the extra hardware queues can push a sixteen-worker process past what the device keeps resident)
                from pepper_amd.variant.models import simple_model
                simple_model.NEW_HANDLES.stream_priority = int(os.environ.get("PEPPER_AMD_FUSED_STREAM_PRIORITY", 0))
                try:
                    model = ModelHandler.load_simple_model_for_training(
                        self.options.model_path, image_features=ImageSizeOptions.IMAGE_HEIGHT, num_classes=ImageSizeOptions.TOTAL_LABELS,
                        num_type_classes=ImageSizeOptions.TOTAL_TYPE_LABELS)[0]
                finally:
                    simple_model.NEW_HANDLES.stream_priority = 0
                model.eval()
            except BaseException as err:
                # a load that failed (no memory, a bad model_path) must not leave the other workers waiting for a handle that will
                # never be put back: the slot is given up and every waiter is handed the error
                with self.models_lock:
                    entry["made"] -= 1
                    entry["failed"] = err
                for _ in range(64):
                    entry["free"].put(None)
                raise
            with self.models_lock:
                entry["all"].append(model)
            return entry, model
        while True:
            try:
                model = entry["free"].get(timeout=1.0)
            except queue.Empty:
                if entry.get("failed") is not None:
                    raise RuntimeError("fused inference: the model could not be loaded") from entry["failed"]
                continue
            if model is None:
                raise RuntimeError("fused inference: the model could not be loaded") from entry.get("failed")
            return entry, model

    def forward_device(self, device, images_ptr, n):
        """n int8 windows [n, 33, 26] at device address images_ptr (the encoder's results of its last run: complete, and valid
        until that encoder's next call) -> float32 probabilities [n, 3] on the host."""
        if n == 0:
            return np.zeros((0, ImageSizeOptions.TOTAL_TYPE_LABELS), np.float32)
        entry, model = self._model(device)
        lib = _lib.load()
        t0 = time.perf_counter()
        try:
            torch.cuda.set_device(device)
            with torch.cuda.stream(model._stream):       # (allocation and the copy back on the forward's stream, not the process-wide default)
                probs = torch.empty((n, model.num_classes_type), dtype=torch.float32, device=torch.device("cuda", device))
                _lib.check(lib.pa_variant_forward_device(model.handle, ctypes.c_void_p(images_ptr), n, probs.data_ptr(), None))
                model._stream.synchronize()
                return probs.cpu().numpy()
        finally:
            with self.timer_lock:
                self.forward_seconds += time.perf_counter() - t0    # (summed over the handles: not a wall time)
            entry["free"].put(model)

    def forward_host(self, device, images):
        """int8 windows on the host (the host-clipped form of image generation) -> probabilities."""
        if len(images) == 0:
            return np.zeros((0, ImageSizeOptions.TOTAL_TYPE_LABELS), np.float32)
        entry, model = self._model(device)
        try:
            torch.cuda.set_device(device)
            return model(torch.from_numpy(np.ascontiguousarray(images)), False).numpy()
        finally:
            entry["free"].put(model)

    # ---- predictions ----
    def submit(self, contig, out, probs):
        """out: one interval's arrays as the encoder returns them (positions, depths, candidates_blob / offsets or candidates,
        candidate_frequency); probs float32 [n, 3]."""
        if self.error is not None:
            raise self.error
        if len(out["positions"]):
            self.queue.put((threading.get_ident(), contig, out, probs))

    NEAR = 1 << 20       # a batch holds candidates of ONE worker's consecutive intervals: the next one starts within this many bases

    def _flush(self, pending, final):
        """batch_size candidates per predictions/batch_<n> group, as the reference's DataLoader batches them, out of ONE worker's
        stream of intervals (`pending`: its pieces that have not filled a batch yet); the groups are written from bulk arrays (one
        library call each), the candidate strings as the encoder left them (NUL-terminated, back to back)."""
        while pending and (final or sum(len(p[1]) for p in pending) >= self.batch_size):
            take, have = [], 0
            while pending and have < self.batch_size:
                piece = pending[0]
                room = self.batch_size - have
                if len(piece[1]) <= room:
                    take.append(pending.pop(0))
                    have += len(piece[1])
                else:
                    contigs, pos, dep, blob, off, freq, probs = piece
                    cut = int(off[room])                     # first byte of the first candidate that does not fit
                    take.append((contigs[:room], pos[:room], dep[:room], blob[:cut], off[:room], freq[:room], probs[:room]))
                    pending[0] = (contigs[room:], pos[room:], dep[room:], blob[cut:], off[room:] - cut, freq[room:], probs[room:])
                    have += room
            blob = b"".join(t[3] for t in take)
            bases = np.cumsum([0] + [len(t[3]) for t in take[:-1]])
            contigs = np.concatenate([t[0] for t in take])
            positions, depths = np.concatenate([t[1] for t in take]), np.concatenate([t[2] for t in take])
            freqs, probs = np.concatenate([t[5] for t in take]).reshape(-1, 1), np.concatenate([t[6] for t in take])
            self.store.write_prediction_arrays(self.batch_no, contigs, positions, depths, np.frombuffer(blob + b"\0", np.uint8),
                                               np.concatenate([t[4] + base for t, base in zip(take, bases)]), freqs, probs)
            if self.selector is not None:
                self.select_queue.put(("batch_" + str(self.batch_no), contigs, positions, depths, freqs, probs, blob))
            self.batch_no += 1
            self.windows += have

    def _select_loop(self):
        """Per batch what FastCandidates._part does from the file: None results (a batch of several contigs, a candidate list the
        library does not take) are left for step 3 to do from the file."""
        from pepper_amd.variant import FastCandidates
        options = self.options
        try:
            fasta_handler = FastCandidates._fasta(options)
            rules = FastCandidates._rules(options)
            while True:
                item = self.select_queue.get()
                if item is None:
                    break
                if rules is None:             # (thresholds the library does not take: step 3 does the job the Python way)
                    continue
                key, contigs, positions, depths, freqs, probs, blob = item
                t0 = time.perf_counter()
                if len(contigs) == 0 or (contigs != contigs[0]).any():
                    continue
                first = bytes(contigs[0])
                seg = FastCandidates.native_batch_arrays(options, rules, fasta_handler, first, len(positions), positions, depths, freqs,
                                                         probs, blob, self.filename + "/" + key)
                if seg is not None:
                    self.segments[(self.filename, key)] = seg
                with self.timer_lock:
                    self.select_seconds += time.perf_counter() - t0    # (summed over the threads)
        except BaseException as err:      # noqa: BLE001 -- whatever was not selected here is selected in step 3 from the file
            with self.timer_lock:
                first_error, self.select_error = self.select_error is None, err
            if first_error:               # said once: the run still completes (step 3 does these batches from the file), only slower
                import sys
                sys.stderr.write("[pepper_amd] fused candidate selection stopped (%r): find_candidates will select from the "
                                 "predictions file instead\n" % (err,))
            while self.select_queue.get() is not None:
                pass

    def _write_loop(self):
        try:
            while True:
                item = self.queue.get()
                if item is None:
                    break
                source, contig, out, probs = item
                t0 = time.perf_counter()
                n = len(out["positions"])
                # (contigs 'S', positions, depths, the interval's candidate strings, each candidate's offset in them, support, p)
                offs = np.asarray(out["candidates_offsets"], np.int64)
                blob = bytes(out["candidates_blob"])[int(offs[0]):int(offs[n]) if len(offs) > n else None]
                piece = (np.array([contig] * n, dtype='S'), np.asarray(out["positions"], np.int32),
                         np.asarray(out["depths"]).astype(np.uint8), blob, offs[:n] - offs[0],
                         np.asarray(out["candidate_frequency"]).astype(np.uint8), np.asarray(probs, np.float32))
                # the workers' intervals arrive interleaved: a batch is cut from one worker's stream, and closed short where that
                # stream jumps (its next run of intervals, another contig) -- the candidate finder fetches the reference once per
                # batch for the span the batch covers (FastCandidates.native_batch_arrays)
                pending = self.pending.setdefault(source, [])
                if pending:
                    last = pending[-1]
                    gap = int(piece[1][0]) - int(last[1][-1])
                    if last[0][0] != piece[0][0] or gap < -self.NEAR or gap > self.NEAR:
                        self._flush(pending, True)
                pending.append(piece)
                self._flush(pending, False)
                self.write_seconds += time.perf_counter() - t0
            for pending in self.pending.values():
                self._flush(pending, True)
        except BaseException as err:      # noqa: BLE001 -- surfaces in submit() / close()
            self.error = err
            while True:                   # keep draining so that no producer blocks on a full queue
                if self.queue.get() is None:
                    break

    def close(self, failed=False):
        """failed: the run is being abandoned (an image worker raised): whatever was written is withdrawn -- a partial
        pepper_prediction.hdf under its final name would be read by a re-run of step 3 as if it were complete."""
        t0 = time.perf_counter()
        self.queue.put(None)
        self.writer.join()
        if self.selector is not None:
            for _ in self.selector:
                self.select_queue.put(None)
            for t in self.selector:
                t.join()
        self.drain_seconds = time.perf_counter() - t0      # what the two threads still had to do when image generation was over
        if self.error is not None or failed:
            self.store.abort()                             # (no partial predictions file for the next step's listing)
        else:
            self.store.close()
        for entry in self.models.values():
            for model in entry["all"]:
                model.close()
        self.models.clear()
        if self.error is not None:
            raise self.error
        return self.batch_no, self.windows

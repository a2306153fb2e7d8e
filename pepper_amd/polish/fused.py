"""polish() with image generation and consensus inference fused (opt-in: polish(..., fused_inference=True) /
PEPPER_AMD_FUSED_POLISH=1).

The reference's steps talk through files (/root/reference/pepper/modules/python/polish.py:94-117: make_images writes the image HDF5
files, call_consensus reads them back).  Here the image-generation workers copy the chunks the chain left on the device
(pa_polish_chain_device_chunks) into ONE gather buffer per device, shared by all of them; every time it holds a full-sized pass
(16 384 chunks) a background thread hands it to the polish model there (pa_polish_predict_device: the 19-window loop with hidden
carry, labels and phred per position) and writes the predictions, while the workers go on filling the next buffer.  Both HDF5
stores are still written -- the image files by the workers as before, the prediction files (one per pass thread) with the
reference's layout (predictions/<contig>/<contig>-<start>-<end>/<chunk id>/...), which perform_stitch globs as it does the
callers' files; an interval's chunks always travel in one pass, so a region group is never split over two files.

Why one shared buffer: a pass is 19 windows x a few kernels whatever its size, and the step loops fill the chip only at 16 384
chunks (128 rows per workgroup x 2 directions = 256 workgroups).  Round 5 gathered per worker (4 096 chunks, 64 workgroups) and
ran the pass on the worker's own thread: sixteen workers queued for two model handles with their chains idle, 3 500 small
launches each waited in a hardware queue behind some worker's 10 ms alignment kernel, and the fused form was slower than the
three steps it replaces.
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


class _GatherSet(object):
    """One pass worth of chunks on a device: the images there, their positions / indices and names on the host."""

    def __init__(self, device, chunks, seq, features):
        self.buffer = torch.empty((chunks, seq, features), dtype=torch.uint8, device=torch.device("cuda", device))
        self.position = np.empty((chunks, seq), np.int64)
        self.index = np.empty((chunks, seq), np.int64)
        self.meta = [None] * chunks          # (contig, start, end, chunk id) per chunk
        self.reserved = 0                    # chunks handed out to workers
        self.done = 0                        # ... and filled in
        self.sealed = False                  # no more reservations: the pass starts when done == reserved
        self.busy = False                    # its pass is running


class _Gather(object):
    """The gather buffers of one device and the hand-over between the image workers (reserve / commit) and the pass threads."""
    # Gather sets per device, made on demand: a worker that finds every set being filled or run gets a new one rather than
    # waiting, up to this many (164 MB of device memory and 262 MB of host arrays each at 16 384 chunks).  The passes share the
    # SIMDs with the alignment kernels and fall behind while the chains run; with room to gather ahead the workers never wait
    # for them, and what is left of the passes runs at full speed once the chains are done.
    SETS = max(2, int(os.environ.get("PEPPER_AMD_FUSED_GATHER_SETS", 6)))

    def __init__(self, owner, device):
        self.owner, self.device = owner, device
        self.seq, self.features = ImageSizeOptions.SEQ_LENGTH, ImageSizeOptions.IMAGE_HEIGHT
        self.chunks = owner.PASS_CHUNKS
        self.lock = threading.Condition()
        self.sets = []                       # made on first use: the third only if the first two are both busy
        self.cur = None
        self.futures = []

    def _free_set(self):
        """(lock held, never waits) a set nobody is filling or running: an idle one, a new one while fewer than SETS exist, else None."""
        for st in self.sets:
            if not st.busy and not st.sealed and st.reserved == 0:
                return st
        if len(self.sets) < self.SETS:
            torch.cuda.set_device(self.device)
            st = _GatherSet(self.device, self.chunks, self.seq, self.features)
            self.sets.append(st)
            return st
        return None

    def reserve(self, n):
        """Room for n chunks that must travel together (n <= PASS_CHUNKS) -> (set, offset).  A set that cannot take them is closed
        short and the next one started; when every set is being filled or run, the caller waits for a pass to finish.  (One loop
        under one lock, re-reading `cur` after every wait: two workers that waited side by side must not each install a set of
        their own -- the first one's would be left half filled for ever.)"""
        with self.lock:
            while True:
                self.owner.check()
                if self.cur is None:
                    self.cur = self._free_set()
                    if self.cur is None:
                        self.lock.wait(0.5)
                        continue
                if self.cur.reserved + n > self.chunks:
                    self._seal(self.cur)
                    self.cur = None
                    continue
                st, at = self.cur, self.cur.reserved
                st.reserved += n
                return st, at

    def commit(self, st, n):
        with self.lock:
            st.done += n
            if st.sealed and st.done == st.reserved:
                self._start(st)

    def _seal(self, st):
        st.sealed = True
        if st.done == st.reserved:
            self._start(st)

    def _start(self, st):
        if st.reserved == 0:
            st.sealed = False
            return
        st.busy = True
        self.futures.append(self.owner.passes.submit(self._run_pass, st))

    def _predict_and_write(self, st, n):
        entry, model = self.owner._model(self.device)
        try:
            torch.cuda.set_device(self.device)
            with torch.cuda.stream(model._stream):          # (this thread's "current stream": not the process-wide default one)
                labels, phred = model.predict_chunks(st.buffer[:n])
                labels, phred = labels.cpu().numpy(), phred.cpu().numpy()
        finally:
            entry["free"].put(model)
        meta = st.meta[:n]
        contigs = np.array([m[0] for m in meta], dtype='S')
        self.owner.store().write_predictions_block(contigs, np.array([m[1] for m in meta], np.int64), np.array([m[2] for m in meta], np.int64),
                                                   np.array([m[3] for m in meta], np.int64), st.position[:n], st.index[:n], labels, phred)

    def _run_pass(self, st):
        """On a pass thread: the model over the set's chunks, their predictions into this thread's file, the set back to the workers."""
        try:
            n = st.reserved
            self._predict_and_write(st, n)
            with self.owner.models_lock:
                self.owner.chunks += n
                self.owner.passes_run += 1
        except BaseException as err:      # noqa: BLE001 -- handed to the workers (check) and to close()
            self.owner.fail(err)
            raise
        finally:
            with self.lock:
                st.reserved = st.done = 0
                st.sealed = st.busy = False
                self.lock.notify_all()

    def finish(self):
        """Every worker has finished: the last, short pass, then wait for all of them."""
        with self.lock:
            if self.cur is not None:
                self._seal(self.cur)
                self.cur = None
            futures, self.futures = self.futures, []
        for f in futures:
            f.result()


class FusedConsensus(object):
    """One per polish() run: up to HANDLES model handles per device, a small pool of pass threads that run the model over full
    gather buffers and write the predictions, and per device the shared gather buffers (_Gather)."""
    PASS_CHUNKS = int(os.environ.get("PEPPER_AMD_FUSED_PASS_CHUNKS", 16384))      # a pass that fills the chip
    HANDLES = 2          # passes in flight per device (own stream and workspace each)

    def __init__(self, model_path, output_directory):
        self.model_path = model_path
        self.output_directory = output_directory
        self.models, self.models_lock = {}, threading.Lock()
        self.chunks = 0
        self.passes_run = 0
        self.handles = max(1, int(os.environ.get("PEPPER_AMD_FUSED_HANDLES", self.HANDLES)))
        self.stream_priority = int(os.environ.get("PEPPER_AMD_FUSED_STREAM_PRIORITY", 0))
        # the threads that run the model passes and write their predictions (the image workers only gather chunks)
        from concurrent.futures import ThreadPoolExecutor
        self.passes = ThreadPoolExecutor(max_workers=self.handles + 1, thread_name_prefix="fused-consensus-pass")
        self.gathers = {}
        self.stores, self._local = [], threading.local()
        self.error = None
        self.failed = False

    # ---- errors ----
    def fail(self, err):
        with self.models_lock:
            if self.error is None:
                self.error = err

    def check(self):
        if self.error is not None:
            raise RuntimeError("fused consensus: a model pass failed") from self.error

    # ---- the prediction files: one per pass thread ----
    def store(self):
        st = getattr(self._local, "store", None)
        if st is None:
            with self.models_lock:
                k = len(self.stores)
                st = DataStore(self.output_directory + "pepper_prediction_fused_" + str(k) + ".hdf", mode='w')
                self.stores.append(st)
            self._local.store = st
        return st

    def gather(self, device):
        with self.models_lock:
            g = self.gathers.get(device)
            if g is None:
                g = self.gathers[device] = _Gather(self, device)
            return g

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
                # PEPPER_AMD_FUSED_STREAM_PRIORITY=-1 puts the passes' stream on the device's high-priority queues.  Measured with
                # full-sized passes: no gain (7.8 s either way on the 64 Mb job), and the extra hardware queues push a process that
                # already drives sixteen past what the device keeps resident (docs/LEDGER_r05.md, "Late finding") -- inside
                # bench.py's child process the fused run was 1.2 s slower with it.  Default 0.  (The loader's signature is the
                # reference's, so the handle is re-made when a priority is asked for.)
                if self.stream_priority:
                    model = loaded.clone(stream_priority=self.stream_priority)
                    loaded.close()
                else:
                    model = loaded
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

    def worker(self, thread_id, device, stream=None):
        return _Worker(self, thread_id, device, stream)

    def close(self, failed=False):
        """After make_images: the last pass, the files closed (or, after a failure anywhere, withdrawn), the handles released."""
        failed = failed or self.failed or self.error is not None
        err = self.error
        try:
            if not failed:
                for g in list(self.gathers.values()):
                    g.finish()
        except BaseException as e:      # noqa: BLE001
            failed, err = True, (err or e)
        self.passes.shutdown(wait=True)
        for st in self.stores:
            if failed:
                st.abort()
            else:
                st.close()
        for entry in self.models.values():
            for model in entry["all"]:
                model.close()
        self.models.clear()
        self.gathers.clear()
        if err is not None:
            raise RuntimeError("fused consensus failed") from err


class _Worker(object):
    """One image worker's side: copies its chain calls' chunks into the device's shared gather buffer."""

    def __init__(self, owner, thread_id, device, stream=None):
        self.owner, self.device = owner, device
        self.seq, self.features = ImageSizeOptions.SEQ_LENGTH, ImageSizeOptions.IMAGE_HEIGHT
        self.gather = owner.gather(device)
        torch.cuda.set_device(device)
        # The gather copies run on the worker's CHAIN stream (the encoder was made on a torch stream for this): queued behind the
        # chain's own kernels, which have finished when add() is called, they start at once.  On the device's default stream --
        # shared by every thread of the process -- a copy waits behind whatever another thread made that stream wait for (a whole
        # model pass); on a stream of its own it waits behind the kernels of whichever other worker shares its hardware queue.
        self.copy_stream = stream if stream is not None else torch.cuda.Stream(device=device)
        self.unsettled = []                  # (set, chunks) of copies under way

    def add(self, contig, starts, stops, chunk_counts, device_images, position, index):
        """The chunks of one chain call: device_images = their address on the device, position / index = numpy views of the
        chain's page-locked copies ([total, seq] int64: copied here, the chain overwrites them in its next run).  The chunks of an
        interval stay together (one pass, one prediction file); a call's intervals are cut into pieces of at most a pass."""
        counts = [int(c) for c in chunk_counts]
        torch.cuda.set_device(self.device)
        self.settle()
        r0, at = 0, 0
        while r0 < len(counts):
            r1, take = r0, 0
            while r1 < len(counts) and (take + counts[r1] <= self.gather.chunks or r1 == r0):
                take += counts[r1]
                r1 += 1
            if take > self.gather.chunks:
                raise RuntimeError("an interval of %d chunks does not fit a pass of %d" % (take, self.gather.chunks))
            if take:
                st, off = self.gather.reserve(take)
                try:
                    src = torch.as_tensor(_DeviceChunks(device_images + at * self.seq * self.features, (take, self.seq, self.features)),
                                          device=torch.device("cuda", self.device))
                    with torch.cuda.stream(self.copy_stream):
                        st.buffer[off:off + take].copy_(src)
                    st.position[off:off + take] = position[at:at + take]
                    st.index[off:off + take] = index[at:at + take]
                    k = off
                    for r in range(r0, r1):
                        a, b = int(starts[r]), int(stops[r])
                        for cid in range(counts[r]):
                            st.meta[k] = (contig, a, b, cid)
                            k += 1
                except BaseException:
                    self.gather.commit(st, take)               # (the set must not wait for ever for this piece; the run fails anyway)
                    raise
                # The copy is NOT waited for here: it sits in a hardware queue behind whatever kernel of another worker shares
                # that queue (~10-20 ms), time this worker spends better writing its image file and reading its next span.
                # settle() -- called before the chain's next run, which overwrites the source -- waits and commits.
                self.unsettled.append((st, take))
            at += take
            r0 = r1

    def settle(self):
        """Wait for the copies add() started and hand their pieces to the gather (before the chain's buffers are written again)."""
        if self.unsettled:
            self.copy_stream.synchronize()
            pieces, self.unsettled = self.unsettled, []
            for st, take in pieces:
                self.gather.commit(st, take)

    def add_host(self, region, images, positions, chunk_ids):
        """An interval that went through the host form (a pile beyond the reservoir cap, a span the packed reader refused): its
        chunks as lists of [seq, features] uint8 arrays and [seq, 2] (position, index) arrays."""
        contig, start, end = region
        n = len(images)
        if n == 0:
            return
        torch.cuda.set_device(self.device)
        self.settle()
        st, off = self.gather.reserve(n)
        try:
            block = np.ascontiguousarray(np.stack([np.asarray(im, np.uint8) for im in images]))
            with torch.cuda.stream(self.copy_stream):
                st.buffer[off:off + n].copy_(torch.from_numpy(block))
            for k, (pos, cid) in enumerate(zip(positions, chunk_ids)):
                pos = np.asarray(pos, np.int64).reshape(self.seq, 2)
                st.position[off + k], st.index[off + k] = pos[:, 0], pos[:, 1]
                st.meta[off + k] = (str(contig), int(start), int(end), int(cid))
            self.copy_stream.synchronize()
        finally:
            self.gather.commit(st, n)

    def flush(self):
        """(kept for callers of the round-5 form: the passes start by themselves)"""

    def close(self, failed=False):
        try:
            self.settle()
        except BaseException:      # noqa: BLE001
            if not failed:
                raise
        if failed:
            self.owner.failed = True
        else:
            self.owner.check()

"""Host side of the predict loops at device speed: reader and writer PROCESSES per lane of image files.

libhdf5 is not thread-safe -- every call of one process goes through one lock -- and the step hand-off formats are made
of many small groups (variant: six datasets per 512-window batch on the way out; polish: seven datasets per 1000-row
chunk on the way in and four on the way out), so one process tops out far below one MI355X (r01: variant HDF5 -> HDF5
1.45 M windows/s against 2.5 M on the device, polish 4.7 k chunks/s against 185 k).  Files shard naturally
(RunInference.py:104-110, call_consensus.py:93-97), so the image files of a caller are dealt over `lanes`; a lane is

    reader process  --slots-->  the caller's GPU loop  --slots-->  writer process  -> pepper_prediction_<rank>[_<lane>].hdf

Slots are shared-memory segments (multiprocessing.shared_memory) that the GPU process page-locks once
(hipHostRegister), so a reader's libhdf5 read lands where the H2D copy starts and the D2H copy lands where the writer's
libhdf5 write starts: no copies in between, no pickling of bulk arrays.  A region's chunks (polish) and a file's batch
numbering (variant) never span lanes because an image file never does; the next stage globs every *.hdf of the
directory (FindCandidates.py:151-166, perform_stitch.py:44-72), exactly as it does for the per-GPU / per-thread files the
reference writes.

Worker processes are spawned (the parent holds a HIP context) and import only numpy + the libhdf5 binding.
"""
import collections
import os
import sys
import time
import traceback
from multiprocessing import get_context, shared_memory

import numpy as np

ALIGN = 256


def _align(n):
    return (n + ALIGN - 1) // ALIGN * ALIGN


def shm_room(nbytes):
    """Is there room for nbytes of shared memory?  (tmpfs allocates lazily: an over-full /dev/shm ends in SIGBUS)"""
    try:
        st = os.statvfs("/dev/shm")
        return st.f_bavail * st.f_frsize > nbytes + (1 << 30)
    except OSError:
        return False


_pagelock_warned = False


def _attach(name):
    """Attach to a segment the parent owns.  Python 3.10 registers every attach with the resource tracker, which would then
    unlink the parent's segment when this process ends (and, a reader and a writer attaching the same segment, trips over
    its own bookkeeping: "KeyError" tracebacks from resource_tracker.py): attach without registering."""
    from multiprocessing import resource_tracker
    register = resource_tracker.register
    resource_tracker.register = lambda *a, **k: None
    try:
        return shared_memory.SharedMemory(name=name)
    finally:
        resource_tracker.register = register


class Slots(object):
    """`count` shared segments of `nbytes`, created by the process that runs the device.  register_async() page-locks
    them for the GPU (pa_host_register = hipHostRegister, ~4 GB/s on the MI355X host) on a background thread, in the
    order given, while the workers start and read; ready(i) waits for segment i."""

    def __init__(self, count, nbytes, pin_bytes=None):
        import threading
        self.nbytes = int(nbytes)
        self.pin_bytes = self.nbytes if pin_bytes is None else min(int(pin_bytes), self.nbytes)   # leading bytes the GPU touches
        self.segments = []
        self.registered = []
        try:
            for _ in range(count):
                seg = shared_memory.SharedMemory(create=True, size=max(1, self.nbytes))
                self.segments.append(seg)
                # tmpfs allocates lazily and statvfs is checked by every rank for itself: reserve the pages now, so that N
                # ranks (or a small --shm-size) overcommitting /dev/shm fail HERE, where the caller can still fall back to
                # its in-process loop, instead of a reader dying of SIGBUS in the middle of the run
                os.posix_fallocate(seg._fd, 0, max(1, self.nbytes))
        except OSError as e:
            self.close()
            raise NoSharedMemory("no room for %d shared-memory slots of %d MB in /dev/shm: %s" % (count, self.nbytes >> 20, e))
        self.names = [s.name for s in self.segments]
        self._ready = [threading.Event() for _ in range(count)]

    def _register_one(self, i):
        try:
            from pepper_amd import _lib
            lib = _lib.load()
            ptr = np.frombuffer(self.segments[i].buf, np.uint8).ctypes.data
            if lib.pa_host_register(ptr, self.pin_bytes) == 0:
                self.registered.append(ptr)
            else:
                raise RuntimeError((lib.pa_last_error() or b"pa_host_register failed").decode())
        except Exception as e:
            # pageable slots still work (the copies stage through the runtime, no longer beside the kernels): say so once
            global _pagelock_warned
            if not _pagelock_warned:
                _pagelock_warned = True
                sys.stderr.write("[pepper_amd] page-locking a %d MB lane slot failed (%s): host<->device copies of the lanes "
                                 "will be staged, expect a lower rate\n" % (self.nbytes >> 20, e))
        finally:
            self._ready[i].set()

    def ready(self, i):
        self._ready[i].wait()

    def view(self, i, offset, shape, dtype):
        return np.ndarray(shape, dtype, buffer=self.segments[i].buf, offset=offset)

    def close(self):
        if self.registered:
            try:
                from pepper_amd import _lib
                lib = _lib.load()
                for ptr in self.registered:
                    lib.pa_host_unregister(ptr)
            except Exception:
                pass
            self.registered = []
        for s in self.segments:
            try:
                s.unlink()                  # the name goes first: close() refuses while numpy views are alive
            except Exception:
                pass
            try:
                s.close()
            except Exception:
                pass
        self.segments = []


def register_async(slot_sets, use_gpu=True):
    """Page-lock every segment of every Slots object on one background thread: slot 0 of each lane first, then slot 1 ...
    (the order the lanes will need them).  Without a GPU (CPU tests) the segments are just marked ready."""
    import threading
    order = [(sl, i) for i in range(max(len(sl.segments) for sl in slot_sets)) for sl in slot_sets if i < len(sl.segments)]

    def work():
        for sl, i in order:
            if use_gpu:
                sl._register_one(i)
            else:
                sl._ready[i].set()
    t = threading.Thread(target=work, name="pepper-amd-pagelock", daemon=True)
    t.start()
    return t


def _have_gpu():
    try:
        from pepper_amd import _lib
        return _lib.load().pa_device_count() > 0
    except Exception:
        return False


def deal_files(files, lanes):
    """Largest file first onto the least loaded lane (files keep their order inside a lane)."""
    sizes = [os.path.getsize(f) for f in files]
    load = [0] * lanes
    out = [[] for _ in range(lanes)]
    for i in sorted(range(len(files)), key=lambda k: (-sizes[k], k)):
        lane = min(range(lanes), key=lambda c: (load[c], c))
        out[lane].append(i)
        load[lane] += sizes[i]
    return [[files[i] for i in sorted(idx)] for idx in out if idx]


class LaneError(RuntimeError):
    pass


class NoSharedMemory(LaneError):
    """/dev/shm cannot hold the staging slots: the caller falls back to its in-process loop."""


def make_slots(lanes, count, nbytes, pin_bytes=None):
    """One Slots object per lane; the ones already created are released when a later one finds no room."""
    made = []
    try:
        for _ in range(lanes):
            made.append(Slots(count, nbytes, pin_bytes))
    except NoSharedMemory:
        for sl in made:
            sl.close()
        raise
    return made


class SlotTooSmall(LaneError):
    """One summaries group does not fit a staging slot: the caller falls back to its in-process loop."""


def _trace(t0, what):
    if os.environ.get("PEPPER_AMD_LANE_TRACE"):
        sys.stderr.write("[lanes] %-28s %8.3f s\n" % (what, time.perf_counter() - t0))


def _start_all(procs):
    """Start the worker processes WITHOUT letting them re-import the caller's main module.  multiprocessing's spawn makes
    every child run the parent's __main__ top level (that is how it finds functions defined there); ours live in this
    module, and a main module that imports torch costs each worker 1.5 s (minutes on a cold page cache) and needs an
    `if __name__ == "__main__"` guard to be safe.  Hiding __main__'s file / spec while the children are created is what an
    interactive session looks like to spawn: nothing to re-import."""
    main = sys.modules.get("__main__")
    saved_spec, saved_file = getattr(main, "__spec__", None), getattr(main, "__file__", None)
    had_file = main is not None and hasattr(main, "__file__")
    try:
        if main is not None:
            main.__spec__ = None
            if had_file:
                del main.__file__
        # a spawned interpreter takes ~20 ms of this process's time to start; four threads start them side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=4) as starters:
            list(starters.map(lambda p: p.start(), procs))
    finally:
        if main is not None:
            main.__spec__ = saved_spec
            if had_file:
                main.__file__ = saved_file


class _Preparing(object):
    """The caller's prepare() (checkpoint load, model creation: ~0.2 s for the variant model) on a thread from the first moment
    of a lanes call, beside the creation of the slots (tmpfs pages are reserved up front: ~0.15 s per GB) and the start of the
    worker processes; join() before the first device pass re-raises what it raised."""

    def __init__(self, prepare):
        self.error = None
        self.thread = None
        if prepare is not None:
            import threading
            self.thread = threading.Thread(target=self._run, args=(prepare,), daemon=True)
            self.thread.start()

    def _run(self, prepare):
        try:
            prepare()
        except BaseException as e:       # noqa: B036 -- handed to the joining thread
            self.error = e

    def wait(self):
        if self.thread is not None:
            self.thread.join()
            self.thread = None

    def join(self):
        self.wait()
        if self.error is not None:
            error, self.error = self.error, None
            raise error


def _next_message(result_q, procs, poll=None):
    """result_q.get() that notices dead workers: a child that dies before it can report (a crash in native code, or a
    caller's script without the `if __name__ == "__main__":` guard that spawned children need) must not leave the GPU
    loop waiting forever.  poll: seconds after which to return None instead of waiting on (the caller has device passes
    in flight to look after)."""
    import queue
    if poll is not None:
        try:
            return result_q.get(timeout=poll)
        except queue.Empty:
            return None
    while True:
        try:
            return result_q.get(timeout=2.0)
        except queue.Empty:
            dead = [p for p in procs if p.exitcode not in (None, 0)]
            if dead:
                raise LaneError("worker process %s exited with code %s before reporting (is the calling script's entry point "
                                "guarded by `if __name__ == '__main__':`?)" % (dead[0].name, dead[0].exitcode))
            if procs and all(p.exitcode is not None for p in procs):
                raise LaneError("all worker processes exited without finishing their lanes")


def _close_all(segments):
    for s in segments:
        try:
            s.close()
        except BufferError:      # a numpy view is still alive; the mapping goes with the process
            pass


def _guarded(fn, lane, result_q, args):
    """Worker body: an exception travels to the parent as ('error', lane, text) on the result queue."""
    try:
        fn(lane, result_q, *args)
    except BaseException:
        result_q.put(("error", lane, traceback.format_exc()))


# ============================================================================================================
# polish: summaries/<name> chunk groups -> predictions/<contig>/<contig>-<start>-<end>/<chunk_id>
# ============================================================================================================
class PolishLayout(object):
    """Byte layout of one slot holding a block of up to `block` chunks."""

    def __init__(self, block, seq_len, features):
        self.block, self.seq_len, self.features = block, seq_len, features
        # what the GPU reads and writes first, in one range (the only part that is page-locked: 12 KB of a chunk's 28 KB); the
        # position / index rows only travel from the reader to the writer
        self.o_image = 0
        self.o_labels = _align(block * seq_len * features)
        self.o_phred = self.o_labels + _align(block * seq_len)
        self.pin_bytes = self.o_phred + _align(block * seq_len)
        self.o_position = self.pin_bytes
        self.o_index = self.o_position + _align(block * seq_len * 8)
        self.nbytes = self.o_index + _align(block * seq_len * 8)

    def views(self, buf, n):
        s, f = self.seq_len, self.features
        mk = lambda off, shape, dt: np.ndarray(shape, dt, buffer=buf, offset=off)   # noqa: E731
        return (mk(self.o_image, (n, s, f), np.uint8), mk(self.o_position, (n, s), np.int64), mk(self.o_index, (n, s), np.int64),
                mk(self.o_labels, (n, s), np.uint8), mk(self.o_phred, (n, s), np.uint8))


def _worker_trace(what, lane, chunks, busy, waiting, t_start):
    """PEPPER_AMD_LANE_TRACE=1: what a worker process spent (CPU seconds of the process, seconds inside the HDF5 calls, seconds
    waiting for a slot / a block)."""
    if os.environ.get("PEPPER_AMD_LANE_TRACE"):
        sys.stderr.write("[lanes] %s %2d: %6d chunks, %.2f s in HDF5 calls (%.1f us per chunk), %.2f s waiting, %.2f s of CPU, "
                         "%.2f s after start\n" % (what, lane, chunks, busy, 1e6 * busy / max(1, chunks), waiting,
                                                   time.process_time(), time.perf_counter() - t_start))


def _polish_reader(lane, result_q, files, slot_names, layout_args, free_q):
    t_start = time.perf_counter()
    from pepper_amd import h5
    layout = PolishLayout(*layout_args)
    segs = [_attach(n) for n in slot_names]
    busy = waiting = 0.0
    chunks = 0
    try:
        first = True
        for path in files:
            with h5.File(path, 'r') as f:
                if 'summaries' not in f:
                    continue
                names = f.keys('summaries')
                a = 0
                while a < len(names):
                    # a lane's first block is a quarter block: the device starts on it while the readers fill whole ones
                    part = names[a:a + (max(1, layout.block // 4) if first else layout.block)]
                    a += len(part)
                    first = False
                    t0 = time.perf_counter()
                    slot = free_q.get()
                    t1 = time.perf_counter()
                    image, position, index, _, _ = layout.views(segs[slot].buf, len(part))
                    contigs, start, end, chunk = f.read_polish_chunks(part, layout.seq_len, layout.features,
                                                                      out=(image, position, index))[:4]
                    result_q.put(("block", lane, slot, len(part), (contigs, start, end, chunk)))
                    waiting += t1 - t0
                    busy += time.perf_counter() - t1
                    chunks += len(part)
        result_q.put(("read_done", lane))
        _worker_trace("reader", lane, chunks, busy, waiting, t_start)
    finally:
        _close_all(segs)


def _polish_writer(lane, result_q, output_filename, slot_names, layout_args, write_q, free_q):
    t_start = time.perf_counter()
    from pepper_amd.polish.DataStorePredict import DataStore
    layout = PolishLayout(*layout_args)
    segs = [_attach(n) for n in slot_names]
    store = DataStore(output_filename, mode='w')
    busy = waiting = 0.0
    chunks = 0
    try:
        while True:
            t0 = time.perf_counter()
            item = write_q.get()
            t1 = time.perf_counter()
            if item is None:
                break
            slot, n, (contigs, start, end, chunk) = item
            _, position, index, labels, phred = layout.views(segs[slot].buf, n)
            store.write_predictions_block(contigs, start, end, chunk, position, index, labels, phred)
            free_q.put(slot)
            waiting += t1 - t0
            busy += time.perf_counter() - t1
            chunks += n
        t0 = time.perf_counter()
        store.close()
        busy += time.perf_counter() - t0
        result_q.put(("write_done", lane))
        _worker_trace("writer", lane, chunks, busy, waiting, t_start)
    except BaseException:
        store.abort()               # (a lane that raised publishes no partial store under the final name)
        raise
    finally:
        _close_all(segs)


def polish_reader(lane, result_q, *args):
    _guarded(_polish_reader, lane, result_q, args)


def polish_writer(lane, result_q, *args):
    _guarded(_polish_writer, lane, result_q, args)


def polish_lanes(files, output_stem, predict_block, lanes, block=8192, seq_len=1000, features=10, slots_per_lane=3, log=None,
                 prepare=None, predict_parts=None, more_predict=None, in_flight=1, pass_blocks=1):
    """Run the polish predict loop over `files` with `lanes` reader/writer process pairs.

    predict_block(image u8 [n, seq, features], labels u8 [n, seq], phred u8 [n, seq]) runs the device pass on host
    arrays that live in page-locked shared memory and fills labels / phred; prepare() (optional) runs in this process, on a
    thread, while the slots are made and the workers start.  Output files: `<output_stem>.hdf` for one
    lane, `<output_stem>_<lane>.hdf` otherwise.  Returns the number of chunks processed.

    The polish kernel gives a workgroup 128 chunks of one direction and walks their 1 900 time steps in sequence, so a
    device pass takes 60-75 ms whether it holds 512 chunks or 16 384: one lane's block of 4 096 chunks alone leaves three
    quarters of the chip idle (and five such passes side by side on their own streams took 205 ms each).  Hence
    pass_blocks > 1 with predict_parts([(image, labels, phred), ...]) -- the blocks that arrived while the device was busy,
    up to pass_blocks of them, go to the device as ONE pass (pa_polish_predict_host_parts) -- and in_flight > 1 with
    more_predict() (returns one more independent predictor of the same kind: its own model handle, streams and staging
    buffers) -- that many passes are under way at once, each on its own thread (the library call releases the GIL), so
    that the copies of one lie beside the kernels of the other.  The first block does not wait for company.  Blocks reach
    a lane's writer in the order its reader produced them."""
    t_begin = time.perf_counter()
    groups = deal_files(files, max(1, lanes))
    lanes = len(groups)
    if lanes == 0:
        return 0
    layout = PolishLayout(block, seq_len, features)
    largs = (block, seq_len, features)
    ctx = get_context("spawn")
    preparing = _Preparing(prepare)
    try:
        slots = make_slots(lanes, slots_per_lane, layout.nbytes, layout.pin_bytes)
    except BaseException:
        preparing.wait()
        raise
    result_q = ctx.Queue()
    free_qs = [ctx.Queue() for _ in range(lanes)]
    write_qs = [ctx.Queue() for _ in range(lanes)]
    procs = []
    done = 0
    locker = pool = None          # bound before the try: the finally reads them when prepare() or a worker start raises
    try:
        for k in range(lanes):
            for s in range(slots_per_lane):
                free_qs[k].put(s)
            out = output_stem + ".hdf" if lanes == 1 else "%s_%d.hdf" % (output_stem, k)
            procs.append(ctx.Process(target=polish_reader, args=(k, result_q, groups[k], slots[k].names, largs, free_qs[k]),
                                     daemon=True))
            procs.append(ctx.Process(target=polish_writer, args=(k, result_q, out, slots[k].names, largs, write_qs[k], free_qs[k]),
                                     daemon=True))
        _start_all(procs[0::2] + procs[1::2])    # the readers first: a process takes ~25 ms to start, the writers have time
        _trace(t_begin, "workers started")
        preparing.join()           # e.g. the checkpoint loaded and the model built while the slots were made and the readers started
        _trace(t_begin, "caller prepared")
        # page-locking after prepare(): hipHostRegister holds the process's mm lock, and a model creation (hipMalloc,
        # uploads) running beside it took 0.4 s instead of 0.07 s
        # PEPPER_AMD_POLISH_PIN=0: leave the slots pageable (the copies are then staged by the runtime; the path needs ~1 GB/s)
        locker = register_async(slots, _have_gpu() and os.environ.get("PEPPER_AMD_POLISH_PIN", "1") != "0")

        def one_by_one(parts):
            for part in parts:
                predict_block(*part)
        depth = max(1, int(in_flight)) if more_predict is not None else 1
        per_pass = max(1, int(pass_blocks))
        predictors = [predict_parts if predict_parts is not None else one_by_one]
        extra = []
        if depth > 1:
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=depth)
            extra = [pool.submit(more_predict) for _ in range(depth - 1)]    # built beside the first pass
        pending = collections.deque()           # (lane, slot, n, meta) received and not yet on the device, oldest first
        inflight = collections.deque()          # (future or None, [blocks]) passes under way, oldest first
        read_done = set()                       # lanes whose end marker waits for their last blocks to retire
        spans = []                              # (start, end, chunks) of every device pass, for the trace
        launched = 0
        writing = lanes

        def run(predict, take):
            views = [layout.views(slots[lane].segments[slot].buf, n) for lane, slot, n, _ in take]
            t0 = time.perf_counter()
            try:
                predict([(v[0], v[3], v[4]) for v in views])
            finally:
                del views
                spans.append((t0, time.perf_counter(), sum(item[2] for item in take)))

        def launch():
            nonlocal launched
            take = [pending.popleft() for _ in range(min(per_pass, len(pending)))]
            for lane, slot, _, _ in take:
                slots[lane].ready(slot)
            if launched == 0:
                _trace(t_begin, "first block on the GPU")
            if pool is None:
                run(predictors[0], take)
                inflight.append((None, take))
            else:
                while len(predictors) < min(depth, launched + 1):
                    more = extra[len(predictors) - 1].result()
                    predictors.append(more if predict_parts is not None else
                                      (lambda parts, one=more: [one(*part) for part in parts]))
                inflight.append((pool.submit(run, predictors[launched % len(predictors)], take), take))
            launched += 1

        def retire():
            nonlocal done
            fut, take = inflight.popleft()
            if fut is not None:
                fut.result()
            for lane, slot, n, meta in take:
                write_qs[lane].put((slot, n, meta))
                done += n
            if log is not None:
                log(done)
            for lane in {item[0] for item in take} & read_done:
                if all(item[0] != lane for item in pending) and all(item[0] != lane for _, blocks in inflight for item in blocks):
                    read_done.discard(lane)
                    write_qs[lane].put(None)             # the end marker follows the lane's last block

        reading = lanes

        def handle(msg):
            nonlocal reading, writing
            kind, lane = msg[0], msg[1]
            if kind == "error":
                raise (SlotTooSmall if "SlotTooSmall" in msg[2] else LaneError)("lane %d failed:\n%s" % (lane, msg[2]))
            if kind == "block":
                _, _, slot, n, meta = msg
                pending.append((lane, slot, n, meta))
            elif kind == "read_done":
                reading -= 1
                if any(item[0] == lane for item in pending) or any(item[0] == lane for _, blocks in inflight for item in blocks):
                    read_done.add(lane)
                else:
                    write_qs[lane].put(None)
            elif kind == "write_done":
                writing -= 1

        import queue
        while writing:
            # a pass costs about the same whatever it holds: the first of the passes under way takes what there is (an idle
            # device is worse than a small pass), a further one starts only full -- or when nothing more is coming
            while pending and len(inflight) < depth and (not inflight or len(pending) >= per_pass or reading == 0):
                launch()
                if pool is None:
                    retire()
            if not writing:
                break
            # with passes under way, do not sleep on the queue past the moment the oldest one is done: its slots (freed by
            # the writers) may be what the readers are waiting for
            msg = _next_message(result_q, procs, poll=0.002 if inflight else None)
            if msg is not None:
                handle(msg)
                while True:                               # whatever else has arrived by now travels with it
                    try:
                        handle(result_q.get_nowait())
                    except queue.Empty:
                        break
            # (after a message as well as after a quiet poll: a steady trickle of blocks must not keep a finished pass waiting)
            while inflight and inflight[0][0].done():
                retire()
        _trace(t_begin, "all lanes written")
        if spans and os.environ.get("PEPPER_AMD_LANE_TRACE"):
            busy = sum(b - a for a, b, _ in spans)
            sys.stderr.write("[lanes] %d device passes of %.0f chunks and %.1f ms on average, %.2f of them at once over %.2f s\n"
                             % (len(spans), sum(c for _, _, c in spans) / len(spans), 1e3 * busy / len(spans),
                                busy / max(1e-9, max(b for _, b, _ in spans) - spans[0][0]),
                                max(b for _, b, _ in spans) - spans[0][0]))
        locker.join()
        for p in procs:
            p.join(timeout=60)
        _trace(t_begin, "workers joined")
    finally:
        preparing.wait()
        if pool is not None:
            pool.shutdown(wait=True)
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(timeout=5)            # (terminated workers are reaped here: none outlives the call)
        if locker is not None:
            locker.join()
        for s in slots:
            s.close()
    return done


# ============================================================================================================
# variant: summaries/<region> image groups -> predictions/batch_<n>
# ============================================================================================================
VARIANT_BLOCK_WINDOWS = 65536       # a reader hands over about this many windows at a time (56 MB of int8 summaries)
VARIANT_SLOT_BYTES = 3 * VARIANT_BLOCK_WINDOWS * 33 * 26 // 2      # a block plus the group that crosses the mark: 84 MB


def _variant_reader(lane, result_q, image_directory, files, slot_names, slot_bytes, free_q, block_windows=VARIANT_BLOCK_WINDOWS):
    """Files of the lane, in order; each file's summaries groups in name order (as SequenceDataset reads them), handed over
    in blocks of whole groups of about VARIANT_BLOCK_WINDOWS windows: the GPU starts after the first block, not after the
    first file, and a slot is a block, not a file."""
    from pepper_amd import h5
    segs = [_attach(n) for n in slot_names]

    def flush(slot, rows, shape, parts, file_end):
        contigs, positions, depths, cands, freqs = parts
        if rows:
            width = max(c.dtype.itemsize for c in contigs)
            blob = np.frombuffer(b"".join(cands) + b"\0", np.uint8)
            ends = np.flatnonzero(blob[:-1] == 0)
            offsets = np.concatenate(([0], ends[:-1] + 1)).astype(np.int64) if len(ends) else np.zeros(0, np.int64)
            meta = (np.concatenate([c.astype("S%d" % width) for c in contigs]), np.concatenate(positions), np.concatenate(depths),
                    blob, offsets, np.concatenate(freqs), (rows,) + tuple(shape))
        else:
            meta = None
        result_q.put(("block", lane, slot, meta, file_end))

    try:
        for path in files:
            slot, rows, shape = None, 0, None
            parts = ([], [], [], [], [])
            with h5.File(path, 'r') as f:
                names = f.keys('summaries') if 'summaries' in f else []
                for name in names:
                    base = 'summaries/' + name + '/'
                    dims = f.info(base + 'images')[0]
                    n = int(dims[0])
                    if n == 0:
                        continue
                    if shape is None:
                        shape = tuple(dims[1:])
                    elif tuple(dims[1:]) != shape:
                        raise ValueError("image shapes differ between groups: %r vs %r" % (tuple(dims[1:]), shape))
                    per = int(np.prod(shape))
                    if n * per > slot_bytes:
                        raise SlotTooSmall("group %s of %s holds %d windows, more than a staging slot of %d bytes" % (name, path, n, slot_bytes))
                    if slot is not None and ((rows + n) * per > slot_bytes or rows >= block_windows):
                        flush(slot, rows, shape, parts, False)
                        slot, rows, parts = None, 0, ([], [], [], [], [])
                    if slot is None:
                        slot = free_q.get()
                    images = np.ndarray((n,) + shape, np.int8, buffer=segs[slot].buf, offset=rows * per)
                    f.read_into(base + 'images', images)
                    del images
                    rows += n
                    parts[0].append(f[base + 'contigs'])
                    parts[1].append(f[base + 'positions'])
                    parts[2].append(f[base + 'depths'])
                    parts[3].append(f.read_strings_raw(base + 'candidates'))
                    parts[4].append(f[base + 'candidate_frequency'])
            if slot is None:
                slot = free_q.get()
            flush(slot, rows, shape if shape is not None else (33, 26), parts, True)      # (possibly empty) end-of-file block
        result_q.put(("read_done", lane))
    finally:
        _close_all(segs)


def _variant_writer(lane, result_q, output_filename, batch_size, write_q):
    """predictions/batch_<n> groups of batch_size windows, cut per image file exactly as the in-process loop cuts them (a
    file's windows in order, the last batch of a file partial, numbering continuing over the lane's files): rows of a
    block that do not fill a batch wait for the next block of the same file."""
    from pepper_amd.variant.DataStorePredict import DataStore
    store = DataStore(output_filename, mode='w')
    batch_no = 0
    held = None            # (contigs, positions, depths, candidate strings list, freqs, probs) not yet written

    def rows_of(item):
        contigs, positions, depths, blob, offsets, freqs, probs = item
        raw = blob.tobytes().split(b"\0")[:len(offsets)]
        return [contigs, positions, depths, raw, freqs, probs]

    def write(rows, count):
        nonlocal batch_no
        contigs, positions, depths, raw, freqs, probs = rows
        for s in range(0, count, batch_size):
            e = min(count, s + batch_size)
            blob = np.frombuffer(b"\0".join(raw[s:e]) + b"\0\0", np.uint8)
            lens = np.fromiter((len(r) + 1 for r in raw[s:e]), np.int64, e - s)
            offsets = np.concatenate(([0], np.cumsum(lens)[:-1])) if e > s else np.zeros(0, np.int64)
            store.write_prediction_arrays(batch_no, contigs[s:e], positions[s:e], depths[s:e], blob, offsets, freqs[s:e], probs[s:e])
            batch_no += 1

    try:
        while True:
            item = write_q.get()
            if item is None:
                break
            payload, file_end = item
            rows = rows_of(payload) if payload is not None else None
            if held is not None and rows is not None:
                width = max(held[0].dtype.itemsize, rows[0].dtype.itemsize)
                rows = [np.concatenate([held[0].astype("S%d" % width), rows[0].astype("S%d" % width)]),
                        np.concatenate([held[1], rows[1]]), np.concatenate([held[2], rows[2]]), held[3] + rows[3],
                        np.concatenate([held[4], rows[4]]), np.concatenate([held[5], rows[5]])]
            elif rows is None:
                rows = held
            held = None
            if rows is None:
                continue
            n = len(rows[1])
            full = n if file_end else (n // batch_size) * batch_size
            if full:
                write(rows, full)
            if full < n:
                held = [rows[0][full:], rows[1][full:], rows[2][full:], rows[3][full:], rows[4][full:], rows[5][full:]]
        result_q.put(("write_done", lane, batch_no))
    finally:
        store.close()


def variant_reader(lane, result_q, *args):
    _guarded(_variant_reader, lane, result_q, args)


def variant_writer(lane, result_q, *args):
    _guarded(_variant_writer, lane, result_q, args)


def variant_lanes(image_directory, files, output_stem, forward_block, batch_size, lanes, slots_per_lane=0, log=None,
                  block_windows=VARIANT_BLOCK_WINDOWS, prepare=None, second_forward=None):
    """Run the variant predict loop over `files` with `lanes` reader/writer process pairs.

    forward_block(images int8 [n, window, features]) -> float32 probabilities [n, classes] runs the device pass on a
    host array in page-locked shared memory.  second_forward: None, or a callable returning a second, independent
    forward_block (its own model handle and streams): two blocks are then in flight, each on its own thread (the library
    call releases the GIL), so that the H2D of one block's first pass, the D2H of the other's last one and this loop's
    own bookkeeping no longer sit between device passes; results still reach a lane's writer in block order.  Output: `<output_stem>.hdf` for one lane, `<output_stem>_<lane>.hdf`
    otherwise; batch_<n> numbering runs over the files of a lane, as it runs over the files of a caller in the reference
    (predict_distributed_gpu.py:40-67).  Returns (batches written, windows processed)."""
    t_begin = time.perf_counter()
    groups = deal_files(files, max(1, lanes))
    lanes = len(groups)
    if lanes == 0:
        return 0, 0
    # a slot holds one block of whole groups: block_windows windows plus room for the group that crosses the mark (a file
    # smaller than that needs only its own size; a single group larger than a slot raises SlotTooSmall).  Small slots
    # matter: page-locking holds the process's mm lock, and 4 GB of it slowed the concurrent checkpoint load by 0.5 s
    slot_bytes = min(max(os.path.getsize(f) for f in files), max(VARIANT_SLOT_BYTES, 3 * block_windows * 33 * 26 // 2))
    if slots_per_lane <= 0:
        slots_per_lane = 2
    ctx = get_context("spawn")
    preparing = _Preparing(prepare)
    try:
        slots = make_slots(lanes, slots_per_lane, slot_bytes)
    except BaseException:
        preparing.wait()
        raise
    result_q = ctx.Queue()
    free_qs = [ctx.Queue() for _ in range(lanes)]
    write_qs = [ctx.Queue() for _ in range(lanes)]
    procs = []
    windows = batches = 0
    locker = None
    pool = second = None          # bound before the try: the finally reads them when prepare() or a worker start raises
    try:
        for k in range(lanes):
            for s in range(slots_per_lane):
                free_qs[k].put(s)
            out = output_stem + ".hdf" if lanes == 1 else "%s_%d.hdf" % (output_stem, k)
            procs.append(ctx.Process(target=variant_reader, args=(k, result_q, image_directory, groups[k], slots[k].names,
                                                                   slot_bytes, free_qs[k], block_windows), daemon=True))
            procs.append(ctx.Process(target=variant_writer, args=(k, result_q, out, batch_size, write_qs[k]), daemon=True))
        _start_all(procs)
        _trace(t_begin, "workers started")
        preparing.join()           # e.g. the checkpoint loaded and the model built while the slots were made and the readers started
        _trace(t_begin, "caller prepared")
        # page-locking after prepare(): hipHostRegister holds the process's mm lock, and a model creation (hipMalloc,
        # uploads) running beside it took 0.4 s instead of 0.07 s
        locker = register_async(slots, _have_gpu())
        writing = lanes
        files_done = 0
        forwards = [forward_block]
        if second_forward is not None:
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=2)
            second = pool.submit(second_forward)        # built beside the first block's device pass
        inflight = collections.deque()                  # (future or None, lane, slot, meta, file_end), oldest first
        submitted = 0

        spans = []                                       # (start, end, windows) of every device pass, for the trace

        def run(forward, lane, slot, shape):
            images = slots[lane].view(slot, 0, shape, np.int8)
            t0 = time.perf_counter()
            try:
                return np.asarray(forward(images))
            finally:
                del images
                spans.append((t0, time.perf_counter(), shape[0]))

        def retire():
            nonlocal windows, files_done
            fut, lane, slot, meta, file_end = inflight.popleft()
            if meta is not None:
                contigs, positions, depths, blob, offsets, freqs, shape = meta
                probs = fut.result() if pool is not None else fut
                write_qs[lane].put(((contigs, positions, depths, blob, offsets, freqs, probs), file_end))
                windows += shape[0]
            else:
                write_qs[lane].put((None, file_end))
            free_qs[lane].put(slot)                       # the forward has consumed the images
            if file_end:
                files_done += 1
                if log is not None:
                    log(files_done)

        depth = 2 if pool is not None else 1
        while writing:
            # with passes in flight, do not sleep on the queue past the moment the oldest one is done: its slot may be what
            # the only reader still running is waiting for
            msg = _next_message(result_q, procs, poll=0.002 if inflight else None)
            if msg is None:
                while inflight and (inflight[0][0] is None or pool is None or inflight[0][0].done()):
                    retire()
                continue
            kind, lane = msg[0], msg[1]
            if kind == "error":
                raise (SlotTooSmall if "SlotTooSmall" in msg[2] else LaneError)("lane %d failed:\n%s" % (lane, msg[2]))
            if kind == "block":
                _, _, slot, meta, file_end = msg
                while len(inflight) >= depth:
                    retire()
                fut = None
                if meta is not None:
                    slots[lane].ready(slot)
                    if submitted == 0:
                        _trace(t_begin, "first block on the GPU")
                    if pool is None:
                        fut = run(forward_block, lane, slot, meta[6])
                    else:
                        if submitted == 1:
                            forwards.append(second.result())
                        fut = pool.submit(run, forwards[submitted % len(forwards)], lane, slot, meta[6])
                    submitted += 1
                inflight.append((fut, lane, slot, meta, file_end))
                if pool is None:
                    retire()
            elif kind == "read_done":
                while inflight:                          # the end marker follows the lane's last block
                    retire()
                write_qs[lane].put(None)
            elif kind == "write_done":
                writing -= 1
                batches += msg[2]
        _trace(t_begin, "all lanes written")
        if spans and os.environ.get("PEPPER_AMD_LANE_TRACE"):
            busy = sum(b - a for a, b, _ in spans)
            window = max(b for _, b, _ in spans) - spans[0][0]
            sys.stderr.write("[lanes] %d device passes of %.0f windows and %.1f ms on average, %.2f of them at once over %.2f s "
                             "(%.2f M windows/s inside)\n" % (len(spans), sum(c for _, _, c in spans) / len(spans), 1e3 * busy / len(spans),
                                                              busy / max(1e-9, window), window, 1e-6 * sum(c for _, _, c in spans) / max(1e-9, window)))
        locker.join()
        for p in procs:
            p.join(timeout=60)
        _trace(t_begin, "workers joined")
    finally:
        preparing.wait()
        if pool is not None:
            pool.shutdown(wait=True)
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(timeout=5)            # (terminated workers are reaped here: none outlives the call)
        if locker is not None:
            locker.join()
        for s in slots:
            s.close()
    return batches, windows


def default_lanes(files, requested, small=64 << 20, most=8):
    """options.num_workers > 0: that many lanes (at most one per file).  0: process lanes only when the job is big
    enough to pay for them (start-up of the workers + page-locking of the slots is ~0.3 s), and then `most` of them
    (measured on the MI355X host: 4 readers keep the variant loop at the device rate).  most=None, the polish loop: 3/8 of
    hostinfo.usable_cpus(), between 3 and 12 -- a lane is two processes that are mostly idle since the chunk reads bypass
    libhdf5 (~15-30 us per chunk) and the prediction files are laid out by h5build.cpp (~3 us per chunk); the device pass
    (~200 k chunks/s) is what the lanes have to feed.  On the 16 CPUs the project's GPU boxes grant: 4 lanes 116 k chunks/s,
    6 lanes 120 k, 8 lanes 102 k (262 144 chunks, start-up included)."""
    if not files or os.environ.get("PEPPER_AMD_NO_LANES") == "1":
        return 0
    if requested and requested > 0:
        return min(int(requested), len(files))
    total = sum(os.path.getsize(f) for f in files)
    if total < small or len(files) < 2:
        return 0
    if most is None:
        from pepper_amd.hostinfo import usable_cpus
        most = max(3, min(12, usable_cpus() * 3 // 8))
    # callers of one host share its CPUs: with R ranks of a torch.distributed job here (one per GPU), each takes its R-th
    most = max(2, most // _ranks_on_this_host())
    return min(len(files), most)


def _ranks_on_this_host():
    """Ranks of the running torch.distributed job on this host (the launchers of this package are single-node: the world
    size), 1 outside a job.  LOCAL_WORLD_SIZE (torchrun) wins when set."""
    try:
        local = int(os.environ.get("LOCAL_WORLD_SIZE", "0"))
        if local > 0:
            return local
        torch_dist = sys.modules.get("torch.distributed")
        if torch_dist is not None and torch_dist.is_available() and torch_dist.is_initialized():
            return max(1, int(torch_dist.get_world_size()))
    except Exception:
        pass
    return 1


if __name__ == "__main__":
    sys.exit(0)

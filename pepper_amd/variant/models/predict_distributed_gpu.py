"""GPU prediction driver with the reference's entry points.

Mirrors /root/reference/pepper_variant/modules/python/models/predict_distributed_gpu.py:19-87:
  predict(options, input_filepath, input_files, output_filepath, threads)
  predict_distributed_gpu(options, filepath, input_files, output_filepath, threads_per_caller)
Same inputs (a checkpoint path in options.model_path, images HDF5 files), same output file
(<output_filepath>pepper_prediction.hdf with predictions/batch_<n> groups of options.batch_size
candidates, batch numbering continuing across files).  Differences, all deliberate:
  * the forward runs in libpepper_amd.so on one MI355X per process (no DataParallel, which
    re-replicates 47 MB of weights every forward: SURVEY.md 8(a) A8); a whole file's windows go
    to the device as packed int8 in large chunks while the HDF5 groups keep the reference's
    batch_size granularity;
  * multi-GPU = one process per GPU over file shards (RunInference.distributed_gpu); rank r > 0
    or world > 1 writes pepper_prediction_<r>.hdf, which the downstream reader already globs
    (FindCandidates.py:151-166);
  * failures raise instead of being logged and swallowed.
"""
import sys
from datetime import datetime

import os

import torch

from pepper_amd.variant.DataStorePredict import DataStore
from pepper_amd.variant.Options import ImageSizeOptions
from pepper_amd.variant.models.ModelHander import ModelHandler
from pepper_amd.variant.models.dataloader_predict import SequenceDataset


def _log(msg):
    sys.stderr.write("[" + str(datetime.now().strftime('%m-%d-%Y %H:%M:%S')) + "] " + msg + "\n")
    sys.stderr.flush()


class _StagingBuffers(object):
    """Two reusable page-locked int8 buffers for the image blocks: the reader thread fills one from the HDF5 file while
    the other is on its way to the device, no fresh (page-faulting, pageable) allocation per file."""
    LIMIT = 4 << 30

    def __init__(self):
        self.buffers = [None, None]
        self.turn = 0

    def alloc(self, n, window, features):
        import numpy as np
        need = n * window * features
        if need > self.LIMIT:
            return np.empty((n, window, features), np.int8)
        k, self.turn = self.turn, (self.turn + 1) % len(self.buffers)
        t = self.buffers[k]
        if t is None or t.numel() < need:
            t = torch.empty(max(need + need // 8, 1), dtype=torch.int8)
            try:
                t = t.pin_memory()
            except RuntimeError:
                pass
            self.buffers[k] = t
        return t[:need].view(n, window, features).numpy()


def predict(options, input_filepath, input_files, output_filepath, threads, rank=None, device=None,
            model=None):
    if getattr(options, "use_hp_info", False):
        raise NotImplementedError("--use_hp_info inference is non-functional in the reference at this "
                                  "commit (SURVEY.md 2.1 V13) and is not provided")
    if device is None:
        device = torch.cuda.current_device()
    torch.cuda.set_device(device)
    holder = {"model": model}

    def get_model():
        if holder["model"] is None:
            torch.cuda.set_device(device)                   # also reached from a lanes call's preparing thread
            holder["model"] = ModelHandler.load_simple_model_for_training(
                options.model_path, image_features=ImageSizeOptions.IMAGE_HEIGHT,
                num_classes=ImageSizeOptions.TOTAL_LABELS, num_type_classes=ImageSizeOptions.TOTAL_TYPE_LABELS)[0]
            holder["model"].eval()
        return holder["model"]
    suffix = "" if rank is None else "_" + str(rank)
    # big jobs (or options.num_workers > 0, the reference's DataLoader(num_workers=...), RunInferenceArguments.py:75-82): reader
    # and writer processes per lane of image files around this process's GPU loop (libhdf5 has one lock per process); the
    # checkpoint is loaded while they start and read
    from pepper_amd import hostpipe
    lanes = hostpipe.default_lanes(input_files, int(getattr(options, "num_workers", 0) or 0), small=512 << 20, most=4)
    if lanes > 0 and hostpipe.shm_room(2 * lanes * hostpipe.VARIANT_SLOT_BYTES):
        torch.set_num_threads(max(1, int(threads)))

        def log(done):
            _log("INFO: FILES COMPLETED: " + str(done) + "/" + str(len(input_files)) + ".")
        def forward_with(get):
            def forward(images):
                torch.cuda.set_device(device)           # the blocks run on pool threads
                return get()(torch.from_numpy(images), False).numpy()
            return forward

        def second_forward():
            # an independent handle (own streams, own staging buffers) for the second block in flight
            torch.cuda.set_device(device)
            other = ModelHandler.load_simple_model_for_training(
                options.model_path, image_features=ImageSizeOptions.IMAGE_HEIGHT,
                num_classes=ImageSizeOptions.TOTAL_LABELS, num_type_classes=ImageSizeOptions.TOTAL_TYPE_LABELS)[0]
            other.eval()
            return forward_with(lambda: other)
        try:
            return hostpipe.variant_lanes(input_filepath, input_files, output_filepath + "pepper_prediction" + suffix,
                                          forward_with(get_model), options.batch_size, lanes, log=log, prepare=get_model,
                                          second_forward=None if os.environ.get("PEPPER_AMD_ONE_BLOCK_IN_FLIGHT") == "1" else second_forward)
        except (hostpipe.SlotTooSmall, hostpipe.NoSharedMemory) as e:
            # a single summaries group above 84 MB (98 k windows; a 100 kb region has a few hundred): the in-process loop
            # below takes whole files and has no such limit
            _log("INFO: " + str(e).strip().splitlines()[-1] + " -- continuing in one process.")
            from pepper_amd.variant.RunInference import remove_stale_predictions
            remove_stale_predictions(output_filepath, pattern="pepper_prediction" + suffix, exact=True)
    output_filename = output_filepath + "pepper_prediction" + suffix + ".hdf"
    prediction_data_file = DataStore(output_filename, mode='w')
    torch.set_num_threads(max(1, int(threads)))
    _log("INFO: TOTAL FILES: " + str(len(input_files)) + ".")

    batch_completed = 0
    total_windows = 0
    # three stages in flight: the next image file is being read and the previous file's predictions are being written
    # (one thread each; libhdf5 calls are serialised by the h5 lock, ctypes releases the GIL) while the GPU works on
    # the current file.  The writer is a single FIFO worker, so batch_<n> groups are created in order.
    from concurrent.futures import ThreadPoolExecutor
    reader = ThreadPoolExecutor(max_workers=1)
    writer = ThreadPoolExecutor(max_workers=1)
    writes = []
    staging = _StagingBuffers()
    pending = reader.submit(SequenceDataset, input_filepath, input_files[0], None, staging.alloc) if input_files else None
    model = get_model()          # the checkpoint is loaded while the first image file is being read

    def write_file(first_batch, input_data, probs):
        # bulk arrays straight into one library call per batch_<n> group (no per-candidate Python objects)
        batch_no = first_batch
        for s in range(0, len(input_data), options.batch_size):
            e = min(len(input_data), s + options.batch_size)
            prediction_data_file.write_prediction_arrays(batch_no, input_data.all_contigs[s:e], input_data.all_positions[s:e],
                                                         input_data.all_depths[s:e], input_data.candidate_blob,
                                                         input_data.candidate_offsets[s:e],
                                                         input_data.all_candidate_frequency[s:e], probs[s:e])
            batch_no += 1

    try:
        for file_id, input_file in enumerate(input_files):
            input_data = pending.result()
            pending = (reader.submit(SequenceDataset, input_filepath, input_files[file_id + 1], None, staging.alloc)
                       if file_id + 1 < len(input_files) else None)
            n = len(input_data)
            if n:
                # one packed int8 H2D copy per file, one device pass; float32 probs come back
                images = torch.from_numpy(input_data.all_images)
                probs = model(images, False).numpy()
                writes.append(writer.submit(write_file, batch_completed, input_data, probs))
                batch_completed += (n + options.batch_size - 1) // options.batch_size
            total_windows += n
            if len(writes) > 2:
                writes.pop(0).result()          # surfaces writer errors early, bounds the queue
            _log("INFO: FILES COMPLETED: " + str(file_id + 1) + "/" + str(len(input_files)) + ".")
        for w in writes:
            w.result()
    finally:
        reader.shutdown(wait=True)
        writer.shutdown(wait=True)
        prediction_data_file.close()
    return batch_completed, total_windows


def predict_distributed_gpu(options, filepath, input_files, output_filepath, threads_per_caller, rank=None,
                            device=None):
    """Create the prediction table of an image set using a trained model (reference signature)."""
    return predict(options, filepath, input_files, output_filepath, threads_per_caller, rank=rank, device=device)

"""Polish GPU prediction driver with the reference's entry points.

Mirrors /root/reference/pepper/modules/python/models/predict_distributed_gpu.py:24-166:
  predict(input_filepath, file_chunks, output_filepath, model_path, batch_size, num_workers, rank, device_id)
  predict_distributed_gpu(filepath, file_chunks, output_filepath, model_path, batch_size, device_ids, num_workers)
One process per GPU over file shards, output <output_filepath>pepper_prediction_<rank>.hdf.  The
whole window loop (19 windows, hidden carry, softmax overlap-add, max, phred) runs on the device;
labels/phred follow the reference's CPU path -- the reference's GPU path overwrites base_values
with base_labels before the phred formula (predict_distributed_gpu.py:103), which is not reproduced.
"""
import os
import sys
from datetime import datetime

import torch

from pepper_amd.polish.DataStorePredict import DataStore
from pepper_amd.polish.Options import ImageSizeOptions
from pepper_amd.polish.models.ModelHander import ModelHandler
from pepper_amd.polish.models.dataloader_predict import SequenceDataset


def _log(msg):
    sys.stderr.write("[" + str(datetime.now().strftime('%m-%d-%Y %H:%M:%S')) + "] " + msg + "\n")
    sys.stderr.flush()


DEVICE_CHUNKS = 16384    # chunks per device pass (19 windows of 100 rows each; 128 chunks per workgroup and direction: 256 workgroups)
LANE_CHUNKS = int(os.environ.get("PEPPER_AMD_POLISH_BLOCK", 2048))   # chunks a reader lane hands over at a time (a slot: 57 MB, 25 MB of it page-locked)
LANE_SLOTS = int(os.environ.get("PEPPER_AMD_POLISH_SLOTS", 4))       # slots per lane: being read, waiting / on the device, being written
# The device loop of the lanes: a pass walks 1 900 time steps in sequence and takes 60-75 ms whether it holds 512 chunks or
# 16 384 (one workgroup per 128 chunks and direction), so the blocks that arrive while the device is busy go to it together --
# up to LANE_PASS_BLOCKS of them as one pass -- and LANE_PASSES_IN_FLIGHT passes are under way at once, each through its own
# model handle (the copies of one beside the kernels of the other).
LANE_PASS_BLOCKS = int(os.environ.get("PEPPER_AMD_POLISH_PASS_BLOCKS", max(1, DEVICE_CHUNKS // LANE_CHUNKS)))
LANE_PASSES_IN_FLIGHT = int(os.environ.get("PEPPER_AMD_POLISH_IN_FLIGHT", 2))

def predict(input_filepath, file_chunks, output_filepath, model_path, batch_size, num_workers, rank, device_id,
            model=None):
    torch.cuda.set_device(device_id)
    holder = {"model": model}

    def get_model():
        if holder["model"] is None:
            torch.cuda.set_device(device_id)                # also reached from a lanes call's preparing thread
            holder["model"] = ModelHandler.load_simple_model_for_training(
                model_path, input_channels=ImageSizeOptions.IMAGE_CHANNELS, image_features=ImageSizeOptions.IMAGE_HEIGHT,
                seq_len=ImageSizeOptions.SEQ_LENGTH, num_classes=ImageSizeOptions.TOTAL_LABELS)[0]
            holder["model"].eval()
        return holder["model"]
    # big jobs (or num_workers > 0, the reference's DataLoader(num_workers=...)): reader and writer processes per lane of
    # image files around this process's GPU loop; libhdf5's one-lock-per-process is what bounds the loop below.  The
    # checkpoint is loaded while the workers start and read.
    from pepper_amd import hostpipe
    lanes = hostpipe.default_lanes(file_chunks, num_workers, most=None)
    layout = hostpipe.PolishLayout(LANE_CHUNKS, ImageSizeOptions.SEQ_LENGTH, ImageSizeOptions.IMAGE_HEIGHT)
    if lanes > 0 and hostpipe.shm_room(LANE_SLOTS * lanes * layout.nbytes):
        def log(done):
            if rank == 0:
                _log("INFO: CHUNKS PROCESSED " + str(done) + ".")
        def predict_with(get):
            def predict_parts(parts):
                torch.cuda.set_device(device_id)            # the passes run on pool threads
                get().predict_chunk_parts_into(parts)
            return predict_parts

        def more_predict():
            torch.cuda.set_device(device_id)
            other = get_model().clone()
            return predict_with(lambda: other)
        try:
            hostpipe.polish_lanes(file_chunks, output_filepath + "pepper_prediction_" + str(rank), None, lanes,
                                  block=LANE_CHUNKS, seq_len=ImageSizeOptions.SEQ_LENGTH, features=ImageSizeOptions.IMAGE_HEIGHT,
                                  slots_per_lane=LANE_SLOTS, log=log, prepare=get_model, predict_parts=predict_with(get_model),
                                  more_predict=more_predict, in_flight=LANE_PASSES_IN_FLIGHT, pass_blocks=LANE_PASS_BLOCKS)
            return rank
        except hostpipe.NoSharedMemory as e:
            # raised before any worker started or any file was written (the ranks of one host share /dev/shm)
            _log("INFO: " + str(e) + " -- continuing in one process.")
    model = get_model()
    output_filename = output_filepath + "pepper_prediction_" + str(rank) + ".hdf"
    prediction_data_file = DataStore(output_filename, mode='w')
    input_data = SequenceDataset(input_filepath, file_chunks)
    done = 0
    # batch_size is the reference's DataLoader batch (128 chunks = one workgroup tile per direction here); the device
    # pass takes up to DEVICE_CHUNKS chunks at once -- outputs are per chunk, so the files do not change -- and the
    # next block is read (libhdf5 under the h5 lock, GIL released) while the GPU works on this one
    from concurrent.futures import ThreadPoolExecutor
    device_batch = max(int(batch_size), DEVICE_CHUNKS)
    blocks = input_data.blocks(device_batch, ImageSizeOptions.SEQ_LENGTH, ImageSizeOptions.IMAGE_HEIGHT)
    reader = ThreadPoolExecutor(max_workers=1)
    writer = ThreadPoolExecutor(max_workers=1)       # single FIFO worker: groups are created in reading order
    pending = reader.submit(next, blocks, None)
    writes = []
    try:
        while True:
            block = pending.result()
            if block is None:
                break
            pending = reader.submit(next, blocks, None)
            contig, contig_start, contig_end, chunk_id, images, position, index = block
            labels, phred = model.predict_chunks(torch.from_numpy(images))
            writes.append(writer.submit(prediction_data_file.write_predictions_block, contig, contig_start, contig_end,
                                        chunk_id, position, index, labels.numpy(), phred.numpy()))
            if len(writes) > 2:
                writes.pop(0).result()               # surfaces writer errors early, bounds the queue
            done += 1
            if rank == 0:
                _log("INFO: BATCHES PROCESSED " + str(done) + ".")
        for w in writes:
            w.result()
    except BaseException:
        # a partial store must not appear under the final name (perform_stitch reads whatever *.hdf the directory holds)
        reader.shutdown(wait=True)
        writer.shutdown(wait=True)
        input_data.close()
        prediction_data_file.abort()
        raise
    reader.shutdown(wait=True)
    writer.shutdown(wait=True)
    input_data.close()
    prediction_data_file.close()
    return rank


def _setup(rank, device_ids, args, all_input_files, port, backend):
    import torch.distributed as dist
    from pepper_amd.parallel import broadcast_checkpoint
    filepath, output_filepath, model_path, batch_size, num_workers = args
    device = device_ids[rank]
    torch.cuda.set_device(device)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ['MASTER_PORT'] = str(port)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=len(device_ids), device_id=torch.device("cuda", device))
    else:
        dist.init_process_group(backend, rank=rank, world_size=len(device_ids))
    try:
        state, meta = broadcast_checkpoint(model_path if rank == 0 else None, src=0,
                                           device=torch.device("cuda", device) if backend == "nccl" else None)
        model = ModelHandler.get_new_gru_model(ImageSizeOptions.IMAGE_CHANNELS, ImageSizeOptions.IMAGE_HEIGHT,
                                               meta["gru_layers"], meta["hidden_size"], ImageSizeOptions.TOTAL_LABELS)
        model.load_state_dict(state)
        predict(filepath, all_input_files[rank] if rank < len(all_input_files) else [], output_filepath, model_path,
                batch_size, num_workers, rank, device, model=model)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def predict_distributed_gpu(filepath, file_chunks, output_filepath, model_path, batch_size, device_ids, num_workers):
    """One model per entry of device_ids over the given file chunks (reference signature).  An ordinal listed twice
    gets two callers on that GPU (the weight broadcast then runs over gloo: RCCL refuses two ranks per device)."""
    from pepper_amd.variant.RunInference import dist_backend, free_port, remove_stale_predictions
    if len(file_chunks) > len(device_ids) and len(device_ids) > 1:
        raise ValueError("predict_distributed_gpu: %d file chunks for %d devices" % (len(file_chunks), len(device_ids)))
    remove_stale_predictions(output_filepath)
    if len(device_ids) == 1:
        # every file goes to the one device, whatever the caller's chunking was
        return predict(filepath, [f for chunk in file_chunks for f in chunk], output_filepath, model_path, batch_size,
                       num_workers, 0, device_ids[0])
    import torch.multiprocessing as mp
    args = (filepath, output_filepath, model_path, batch_size, num_workers)
    mp.spawn(_setup, args=(device_ids, args, file_chunks, free_port(), dist_backend(device_ids)), nprocs=len(device_ids),
             join=True)

"""call_consensus with the reference's signature.

Mirrors /root/reference/pepper/modules/python/call_consensus.py:13-160: argument validation, 'hdf'
image files sharded round-robin over the GPUs (file i -> caller i % callers), one
pepper_prediction_<rank>.hdf per GPU.  Validation failures raise instead of exit(1); gpu=False
raises (no CPU fallback in the MI355X drop-in).
"""
import os
import sys
from datetime import datetime
from os import listdir
from os.path import isfile, join

import torch

from pepper_amd.polish.models.predict_distributed_gpu import predict_distributed_gpu


def _log(msg):
    sys.stderr.write("[" + str(datetime.now().strftime('%m-%d-%Y %H:%M:%S')) + "] " + msg + "\n")


def get_file_paths_from_directory(directory_path):
    return sorted(join(directory_path, file) for file in listdir(directory_path)
                  if isfile(join(directory_path, file)) and file[-3:] == 'hdf')


def handle_output_directory(output_dir):
    if not os.path.exists(output_dir):
        os.makedirs(output_dir, exist_ok=True)
    if output_dir[-1] != '/':
        output_dir += '/'
    return output_dir


def polish_genome_distributed_gpu(image_dir, model_path, batch_size, num_workers, output_dir, device_ids):
    _log("INFO: DISTRIBUTED GPU SETUP")
    if device_ids is None:
        device_ids = list(range(torch.cuda.device_count()))
    elif isinstance(device_ids, str):
        device_ids = [int(i) for i in device_ids.split(',')]
    total_callers = len(device_ids)
    if total_callers == 0:
        raise RuntimeError("ERROR: NO GPU AVAILABLE BUT GPU MODE IS SET")
    input_files = get_file_paths_from_directory(image_dir)
    # the reference deals the files round robin (call_consensus.py:93-97); with more than one GPU they go largest
    # first onto the least loaded one instead (same rule as pepper_amd.variant.RunInference.shard_files)
    from pepper_amd.variant.RunInference import shard_files
    file_chunks = shard_files(input_files, total_callers,
                              sizes=[os.path.getsize(f) for f in input_files] if total_callers > 1 else None)
    device_ids = device_ids[:max(1, len(file_chunks))]
    _log("INFO: TOTAL THREADS: " + str(len(device_ids)))
    predict_distributed_gpu(image_dir, file_chunks, output_dir, model_path, batch_size, device_ids, num_workers)
    _log("INFO: PREDICTION GENERATED SUCCESSFULLY.")


def call_consensus(image_dir, model_path, batch_size, num_workers, output_dir, device_ids, gpu, threads):
    if not os.path.isfile(model_path):
        raise FileNotFoundError("ERROR: CAN NOT LOCATE MODEL FILE.")
    if not os.path.isdir(image_dir):
        raise FileNotFoundError("ERROR: CAN NOT LOCATE IMAGE DIRECTORY.")
    if batch_size <= 0:
        raise ValueError("ERROR: batch_size NEEDS TO BE >0.")
    if num_workers < 0:
        raise ValueError("ERROR: num_workers NEEDS TO BE >=0.")
    if threads <= 0:
        raise ValueError("ERROR: THREAD NEEDS TO BE >=0.")
    output_dir = handle_output_directory(output_dir)
    if not gpu:
        raise RuntimeError("pepper_amd is the MI355X drop-in for the GPU inference path and has no CPU fallback")
    if not torch.cuda.is_available():
        raise RuntimeError("ERROR: TORCH IS NOT BUILT WITH CUDA/HIP OR NO GPU IS VISIBLE.")
    polish_genome_distributed_gpu(image_dir, model_path, batch_size, num_workers, output_dir, device_ids)

"""run_inference with the reference's signature, one process per MI355X.

Mirrors /root/reference/pepper_variant/modules/python/RunInference.py:12-138:
  get_file_paths_from_directory, distributed_gpu(options, image_dir, output_dir),
  run_inference(options, image_dir, output_dir)
Sharding is the reference's: image files round-robin, file i -> caller i % callers
(RunInference.py:104-110).  The reference drives every GPU from one process through
nn.DataParallel; here each GPU gets its own process (torch.multiprocessing spawn, or the ranks of
an existing torchrun launch), loads the checkpoint on rank 0 and receives the weights by one
RCCL broadcast over xGMI, and writes its own pepper_prediction_<rank>.hdf.
"""
import os
import sys
import time
from datetime import datetime
from os import listdir
from os.path import isfile, join

import torch


def _log(msg):
    sys.stderr.write("[" + str(datetime.now().strftime('%m-%d-%Y %H:%M:%S')) + "] " + msg + "\n")
    sys.stderr.flush()


def get_file_paths_from_directory(directory_path):
    """Returns all paths of files in a directory whose name ends in 'hdf5' (sorted for a
    deterministic shard assignment; the reference uses listdir order)."""
    return sorted(join(directory_path, file) for file in listdir(directory_path)
                  if isfile(join(directory_path, file)) and file[-4:] == 'hdf5')


def handle_output_directory(output_dir):
    """ImageGenerationUI.py:79-91: make the directory, return it with a trailing slash."""
    if not os.path.exists(output_dir):
        os.makedirs(output_dir, exist_ok=True)
    if output_dir[-1] != '/':
        output_dir += '/'
    return output_dir


def shard_files(input_files, callers, sizes=None):
    """Files -> one list per caller, empty lists dropped.  Without sizes: the reference's round robin,
    file_chunks[i % callers].append(input_files[i]) (RunInference.py:104-110).  With sizes (bytes per file): largest
    file first onto the least loaded caller -- image files differ by a lot between chromosomes, and a GPU that finishes
    early just idles; the prediction files are merged by directory listing downstream, so the assignment is free."""
    chunks = [[] for _ in range(callers)]
    if sizes is None:
        for i, f in enumerate(input_files):
            chunks[i % callers].append(f)
    else:
        load = [0] * callers
        for i in sorted(range(len(input_files)), key=lambda k: (-sizes[k], k)):
            r = min(range(callers), key=lambda c: (load[c], c))
            chunks[r].append(input_files[i])
            load[r] += sizes[i]
        order = {f: i for i, f in enumerate(input_files)}
        for c in chunks:
            c.sort(key=order.get)
    return [c for c in chunks if c]


def resolve_device_ids(options):
    """One entry per caller (= process).  --device_ids is taken as written, duplicates included: "0,0" puts two callers on
    GPU 0, as the reference's list does (RunInference.py:41-60).  options.callers_per_gpu is accepted and NOT multiplied
    in: in the reference it only lengthens the device list handed to one nn.DataParallel process (default 4, sized for
    11 GB cards); here a caller is a process that fills an MI355X on its own, and four of them time-slicing one GPU
    would only add context switches."""
    if getattr(options, "device_ids", None) is None:
        return list(range(torch.cuda.device_count()))
    if isinstance(options.device_ids, str):
        return [int(i) for i in options.device_ids.split(',') if i.strip() != ""]
    return [int(i) for i in options.device_ids]


def free_port():
    """MASTER_PORT for a spawn: PEPPER_AMD_MASTER_PORT if set, else a port the kernel hands out (two runs on one host
    must not collide on a fixed number)."""
    if os.environ.get("PEPPER_AMD_MASTER_PORT"):
        return int(os.environ["PEPPER_AMD_MASTER_PORT"])
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def dist_backend(device_ids):
    """"nccl" (= RCCL over xGMI) when every caller has its own GPU.  RCCL refuses two ranks on one device, so when an
    ordinal is listed twice the one weight broadcast goes over gloo on host memory instead (47 MB, once)."""
    if os.environ.get("PEPPER_AMD_DIST_BACKEND"):
        return os.environ["PEPPER_AMD_DIST_BACKEND"]
    return "nccl" if len(set(device_ids)) == len(device_ids) else "gloo"


def remove_stale_predictions(output_dir, pattern="pepper_prediction", exact=False):
    """The file names depend on the number of callers (pepper_prediction.hdf vs pepper_prediction_<rank>.hdf) and the
    next stage globs every *.hdf of the directory (FindCandidates.py:151-166): leftovers of an earlier run with a
    different GPU count would be mixed into the VCF.  exact=True removes only <pattern>.hdf and <pattern>_<lane>.hdf --
    what ONE caller wrote -- so that rank 1's clean-up ("pepper_prediction_1") leaves pepper_prediction_10.hdf,
    pepper_prediction_11_0.hdf ... of the other running ranks alone."""
    import re
    own = re.compile(re.escape(pattern) + r"(_\d+)?\.hdf") if exact else None
    for name in listdir(output_dir):
        if not isfile(join(output_dir, name)):
            continue
        if (own.fullmatch(name) if exact else (name.startswith(pattern) and name.endswith(".hdf"))):
            os.remove(join(output_dir, name))


def _worker(rank, world, device_ids, options, image_dir, file_chunks, output_dir, port, backend):
    import torch.distributed as dist
    from pepper_amd.parallel import broadcast_checkpoint
    from pepper_amd.variant.Options import ImageSizeOptions
    from pepper_amd.variant.models.ModelHander import ModelHandler
    from pepper_amd.variant.models.predict_distributed_gpu import predict
    device = device_ids[rank]
    torch.cuda.set_device(device)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        state, meta = broadcast_checkpoint(options.model_path if rank == 0 else None, src=0,
                                           device=torch.device("cuda", device) if backend == "nccl" else None)
        model = ModelHandler.get_new_gru_model(ImageSizeOptions.IMAGE_HEIGHT, meta["gru_layers"],
                                               meta["hidden_size"], ImageSizeOptions.TOTAL_LABELS,
                                               ImageSizeOptions.TOTAL_TYPE_LABELS)
        model.load_state_dict(state)
        threads = max(1, int(options.threads / world))
        predict(options, image_dir, file_chunks[rank] if rank < len(file_chunks) else [], output_dir, threads,
                rank=rank, device=device, model=model)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def distributed_gpu(options, image_dir, output_dir):
    start_time = time.time()
    device_ids = resolve_device_ids(options)
    _log("INFO: AVAILABLE GPU DEVICES: " + str(device_ids))
    if len(device_ids) == 0:
        raise RuntimeError("ERROR: NO GPU AVAILABLE BUT GPU MODE IS SET")
    input_files = get_file_paths_from_directory(image_dir)
    file_chunks = shard_files(input_files, len(device_ids),
                              sizes=[os.path.getsize(f) for f in input_files] if len(device_ids) > 1 else None)
    world = max(1, min(len(device_ids), len(file_chunks)))
    remove_stale_predictions(output_dir)
    threads_per_caller = max(1, int(options.threads / world))
    _log("INFO: TOTAL CALLERS: " + str(world))
    _log("INFO: TOTAL THREADS PER CALLER: " + str(threads_per_caller))

    if world == 1:
        from pepper_amd.variant.models.predict_distributed_gpu import predict_distributed_gpu
        predict_distributed_gpu(options, image_dir, input_files, output_dir, threads_per_caller,
                                device=device_ids[0])
    else:
        import torch.multiprocessing as mp
        mp.spawn(_worker, args=(world, device_ids, options, image_dir, file_chunks, output_dir, free_port(),
                                dist_backend(device_ids[:world])),
                 nprocs=world, join=True)

    _log("INFO: PREDICTION GENERATED SUCCESSFULLY.")
    end_time = time.time()
    mins = int((end_time - start_time) / 60)
    secs = int((end_time - start_time)) % 60
    _log("ELAPSED TIME: " + str(mins) + " Min " + str(secs) + " Sec")


def run_inference(options, image_dir, output_dir):
    output_dir = handle_output_directory(output_dir)
    if getattr(options, "dry", False):
        raise NotImplementedError("--dry (fake one-hot predictor over train-mode labels, "
                                  "predict_distributed_cpu_fake.py) is a debug aid outside the accelerated path")
    if not getattr(options, "gpu", False):
        raise RuntimeError("pepper_amd is the MI355X drop-in for the GPU inference path and has no CPU "
                           "fallback: run with --gpu, or use the reference's own distributed_cpu")
    distributed_gpu(options, image_dir, output_dir)

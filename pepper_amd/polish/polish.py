"""`polish(bam, fasta, output_path, threads, region, model_path, batch_size, gpu_mode, device_ids, num_workers)`:
images -> consensus -> stitched FASTA (/root/reference/pepper/modules/python/polish.py:14-125).  The three steps are
the package's make_images / call_consensus / perform_stitch; argument checks raise instead of exiting."""
import os
import sys
import time
from datetime import datetime

from pepper_amd.polish.ImageGenerationUI import UserInterfaceSupport
from pepper_amd.polish.call_consensus import call_consensus
from pepper_amd.polish.make_images import make_images
from pepper_amd.polish.perform_stitch import perform_stitch


def _log(message):
    sys.stderr.write("[" + datetime.now().strftime('%m-%d-%Y %H:%M:%S') + "] " + message + "\n")
    sys.stderr.flush()


def polish(bam_filepath, fasta_filepath, output_path, threads, region, model_path, batch_size, gpu_mode, device_ids,
           num_workers, stage_walls=None, fused_inference=None):
    """The reference's ten arguments; stage_walls: a dict that receives the three steps' wall times; fused_inference (default:
    PEPPER_AMD_FUSED_POLISH=1): the image workers hand their chunks to the model on the device instead of call_consensus reading
    the image files back (pepper_amd/polish/fused.py); both stores are still written."""
    for path, what in ((bam_filepath, "BAM"), (fasta_filepath, "FASTA"), (model_path, "MODEL")):
        if not os.path.isfile(path):
            raise FileNotFoundError("CAN NOT LOCATE " + what + " FILE: " + str(path))
    if threads <= 0:
        raise ValueError("THREAD NEEDS TO BE >=0.")
    if batch_size <= 0:
        raise ValueError("batch_size NEEDS TO BE >0.")
    if num_workers < 0:
        raise ValueError("num_workers NEEDS TO BE >=0.")
    if not gpu_mode:
        raise RuntimeError("pepper_amd has no CPU inference path: gpu_mode must be set")
    timestr = time.strftime("%m%d%Y_%H%M%S")
    output_dir = UserInterfaceSupport.handle_output_directory(output_path)
    image_output_directory = output_dir + "images_" + str(timestr) + "/"
    prediction_output_directory = output_dir + "predictions_" + str(timestr) + "/"
    _log("INFO: RUN-ID: " + str(timestr))
    _log("STEP 1: GENERATING IMAGES -> " + image_output_directory)
    image_stats = {} if stage_walls is not None else None       # the image workers' stage times, summed over the workers
    t0 = time.perf_counter()
    if fused_inference is None:
        fused_inference = os.environ.get("PEPPER_AMD_FUSED_POLISH") == "1"
    if fused_inference:
        from pepper_amd.polish.fused import FusedConsensus
        UserInterfaceSupport.handle_output_directory(prediction_output_directory)
        _log("STEP 1+2: GENERATING IMAGES AND RUNNING INFERENCE (FUSED) -> " + prediction_output_directory)
        sink = FusedConsensus(model_path, prediction_output_directory)
        try:
            make_images(bam_filepath, fasta_filepath, region, image_output_directory, threads, device_ids=device_ids, fused=sink,
                        stats=image_stats)
        finally:
            sink.close()
        t1 = t2 = time.perf_counter()
    else:
        make_images(bam_filepath, fasta_filepath, region, image_output_directory, threads, device_ids=device_ids, stats=image_stats)
        t1 = time.perf_counter()
        _log("STEP 2: RUNNING INFERENCE -> " + prediction_output_directory)
        call_consensus(image_output_directory, model_path, batch_size, num_workers, prediction_output_directory, device_ids,
                       gpu_mode, threads)
        t2 = time.perf_counter()
    _log("STEP 3: RUNNING STITCH -> " + output_dir)
    perform_stitch(prediction_output_directory, output_dir, threads)
    if stage_walls is not None:
        stage_walls.update(make_images=t1 - t0, call_consensus=t2 - t1, perform_stitch=time.perf_counter() - t2)
        stage_walls["image_stage_seconds_summed_over_workers"] = image_stats

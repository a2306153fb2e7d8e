"""make_images -> run_inference -> find_candidates in one call.

replaces: /root/reference/pepper_variant/modules/python/CallVariant.py:12-109 (`call_variant`).
Same option names and the same three output locations (images_<run>/, predictions_<run>/, VCFs in
output_dir).  Input checks raise instead of calling exit().  `options.bam` / `options.fasta` are read by the package's
own readers (pepper_amd.variant.bam: BGZF members inflated on the device, records walked there;
pepper_amd.variant.fasta); `options.bam_handler_factory` / `options.fasta_handler_factory`, when given,
replace them (objects with the reference handlers' methods: tests inject synthetic read sets that way).
"""
import os
import sys
import time
from datetime import datetime

from pepper_amd.variant.FindCandidates import process_candidates
from pepper_amd.variant.ImageGenerationUI import ImageGenerationUtils
from pepper_amd.variant.RunInference import run_inference


def _log(message):
    sys.stderr.write("[" + str(datetime.now().strftime('%m-%d-%Y %H:%M:%S')) + "] INFO: " + message + "\n")


def call_variant(options):
    start_time = time.time()
    if getattr(options, "bam_handler_factory", None) is None and not os.path.isfile(options.bam):
        raise FileNotFoundError("ERROR: CAN NOT LOCATE BAM FILE.")
    if getattr(options, "fasta_handler_factory", None) is None and not os.path.isfile(options.fasta):
        raise FileNotFoundError("ERROR: CAN NOT LOCATE FASTA FILE.")
    if not os.path.isfile(options.model_path):
        raise FileNotFoundError("ERROR: CAN NOT LOCATE MODEL FILE.")
    if options.threads <= 0:
        raise ValueError("ERROR: THREAD NEEDS TO BE >=0.")
    if options.batch_size <= 0:
        raise ValueError("ERROR: batch_size NEEDS TO BE >0.")
    if options.num_workers < 0:
        raise ValueError("ERROR: num_workers NEEDS TO BE >=0.")

    timestr = time.strftime("%m%d%Y_%H%M%S")
    output_dir = ImageGenerationUtils.handle_output_directory(options.output_dir)
    image_output_directory = output_dir + "images_" + str(timestr) + "/"
    prediction_output_directory = output_dir + "predictions_" + str(timestr) + "/"
    candidate_output_directory = output_dir

    _log("RUN-ID: " + str(timestr))
    _log("IMAGE OUTPUT: " + str(image_output_directory))
    _log("STEP 1/3 GENERATING IMAGES:")
    options.image_output_directory = image_output_directory
    walls = getattr(options, "stage_walls", None)           # a dict the caller wants the three steps' wall times in
    precomputed = None
    t0 = time.perf_counter()
    fused = bool(getattr(options, "fused_inference", False)) or os.environ.get("PEPPER_AMD_FUSED_CALL_VARIANT") == "1"
    if fused:
        # opt-in: the encoder's windows go to the model where they lie on the device; both HDF5 files are still written
        # (pepper_amd/variant/fused.py).  Steps 1 and 2 are then one step.
        from pepper_amd.variant.fused import FusedPredictor
        os.makedirs(prediction_output_directory, exist_ok=True)
        _log("STEP 1+2/3 GENERATING IMAGES AND RUNNING INFERENCE (FUSED)")
        _log("OUTPUT: " + str(prediction_output_directory))
        options.fused_sink = FusedPredictor(options, prediction_output_directory)
        precomputed = None
        try:
            ImageGenerationUtils.generate_images(options)
        finally:
            sink, options.fused_sink = options.fused_sink, None
            # (an image worker that raised: the predictions written so far are withdrawn, not published under the final name)
            sink.close(failed=sys.exc_info()[0] is not None)
            precomputed = sink.segments or None
            if walls is not None:
                walls["fused_writer_drain"] = getattr(sink, "drain_seconds", 0.0)
                walls["fused_forward_summed_over_handles"] = sink.forward_seconds
                walls["fused_writer_busy_summed"], walls["fused_selection_busy_summed"] = sink.write_seconds, sink.select_seconds
            if sink.select_error is not None:
                _log("INFO: THE FUSED RUN'S CANDIDATE SELECTION STOPPED (" + repr(sink.select_error) + "): STEP 3 DOES THE REST FROM THE FILE")
            _log("INFO: FUSED: " + str(len(sink.segments)) + " OF " + str(sink.batch_no) + " BATCHES SELECTED AHEAD OF STEP 3")
        t1 = t2 = time.perf_counter()
    else:
        ImageGenerationUtils.generate_images(options)
        t1 = time.perf_counter()

        _log("STEP 2/3 RUNNING INFERENCE")
        _log("OUTPUT: " + str(prediction_output_directory))
        run_inference(options, image_output_directory, prediction_output_directory)
        t2 = time.perf_counter()

    _log("STEP 3/3 FINDING CANDIDATES")
    _log("OUTPUT: " + str(candidate_output_directory))
    totals = process_candidates(options, prediction_output_directory, candidate_output_directory,
                                precomputed=precomputed if fused else None)
    if walls is not None:
        walls.update(make_images=t1 - t0, run_inference=t2 - t1, find_candidates=time.perf_counter() - t2)

    elapsed = time.time() - start_time
    _log("TOTAL ELAPSED TIME FOR FINDING CANDIDATES: " + str(int(elapsed / 60)) + " Min " + str(int(elapsed) % 60) + " Sec")
    return image_output_directory, prediction_output_directory, totals

"""Which step of polish() makes the NEXT image generation of the same process slower?  make_images / call_consensus / stitch in any
order inside one process, each step's wall time on stdout.  With GPU_MAX_HW_QUEUES=32 every M after the second C is 1.5x slower
(profiles/r05_hw_queues_sequence_*.txt, docs/LEDGER_r05.md "Late finding"); 16, the package's setting, shows no step.
usage: GPU_MAX_HW_QUEUES=<n> python tools/hw_queue_sequence.py <work dir> <sequence of M C S Z, e.g. MMMCMMSMMCMM> [draft bases]"""
import os, shutil, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_e2e as be
from pepper_amd.hostinfo import usable_cpus
from pepper_amd.polish.make_images import make_images
from pepper_amd.polish.call_consensus import call_consensus
from pepper_amd.polish.perform_stitch import perform_stitch
work, seq = sys.argv[1], sys.argv[2]
info = be.synth(work, float(sys.argv[3]) if len(sys.argv) > 3 else 32e6, 60)
model = os.path.join(work, "polish.pkl"); be.checkpoint(model, "polish")
threads = max(1, usable_cpus())
img, pred, out = work + "/seq_img/", work + "/seq_pred/", work + "/seq_out/"
for step in seq:
    t0 = time.perf_counter()
    if step == "M":
        shutil.rmtree(img, ignore_errors=True)
        st = {}
        make_images(work + "/reads.bam", work + "/draft.fa", None, img, threads, device_ids="0", stats=st)
        extra = " score %.1f band %.1f enc %.1f hdf5 %.2f" % (st.get("chain_score_kernel", 0), st.get("chain_band_kernel", 0), st.get("chain_encode", 0), st.get("hdf5", 0))
    elif step == "C":
        shutil.rmtree(pred, ignore_errors=True)
        call_consensus(img, model, 512, 0, pred, "0", True, threads); extra = ""
    elif step == "S":
        shutil.rmtree(out, ignore_errors=True)
        perform_stitch(pred, out, threads); extra = ""
    elif step == "Z":
        time.sleep(2.0); extra = ""
    print("%s %.2f%s" % (step, time.perf_counter() - t0, extra), flush=True)

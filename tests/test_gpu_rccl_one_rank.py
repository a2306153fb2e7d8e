"""The RCCL branch of the multi-GPU plumbing on hardware, with the one rank a 1-GPU box has: a gloo default group, the bounded RCCL
attempt on top of it (pepper_amd.parallel.agree_on_rccl: communicator creation + probe all-reduce, exactly what bench.py's
dist_setup runs per rank), then the two weight broadcasts over the RCCL group from device memory (broadcast_checkpoint: the header
as an object broadcast + ONE flat fp32 blob; broadcast_numpy_state_dict) -- compared with the checkpoint as loaded.  One rank moves
no bytes over xGMI; what this pins is that the library path (ncclCommInitRank, the collectives' launch and completion, the polling
wait) runs on the box the N-rank run will use.  In a child process with a deadline: a communication library that blocks must not
take the suite with it.  Reference: pepper_variant/modules/python/models/predict_distributed_gpu.py:24-30 (DDP's init + broadcast),
RunInference.py:101-116."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r'''
import json, os, sys
from datetime import timedelta
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["PEPPER_AMD_REPO"])
from pepper_amd import parallel, synthetic
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=0, world_size=1)
dev = torch.device("cuda", 0)
def try_rccl():
    torch.cuda.set_device(0)
    group = dist.new_group(backend="nccl", timeout=timedelta(seconds=60), device_id=dev)
    probe = torch.ones(1, device=dev)
    parallel.wait_bounded(dist.all_reduce(probe, group=group, async_op=True), 60, "RCCL probe all-reduce")
    assert int(probe.item()) == 1
    return group
group, failed, why = parallel.agree_on_rccl(1, try_rccl, timeout_s=90)
out = {"rccl": group is not None, "failed": failed, "why": why}
if group is not None:
    path = os.environ["PEPPER_AMD_CKPT"]
    want, want_meta = parallel.load_checkpoint_state(path)
    got, meta = parallel.broadcast_checkpoint(path, src=0, device=dev, group=group)
    out["checkpoint_equal"] = meta == want_meta and list(got) == list(want) and all(torch.equal(got[k], want[k]) for k in want)
    shapes = [(k, tuple(v.shape)) for k, v in want.items()]
    sd = parallel.broadcast_numpy_state_dict(lambda: {k: v.numpy() for k, v in want.items()}, shapes, device=dev, group=group)
    out["numpy_equal"] = all(np.array_equal(sd[k], want[k].numpy()) for k in want)
    out["backend"] = dist.get_backend(group)
    out["blob_floats"] = int(sum(v.numel() for v in want.values()))
print("RESULT " + json.dumps(out))
sys.stdout.flush()
os._exit(0)          # (no communicator teardown: a 1-rank destroy has nothing to prove and has been seen to wait)
'''


def test_rccl_group_of_one_rank_carries_the_weight_broadcasts(tmp_path):
    import json
    import torch
    from pepper_amd import synthetic
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ckpt = str(tmp_path / "variant.pkl")
    sd = synthetic.variant_state_dict(seed=5)
    torch.save({"model_state_dict": {"module." + k: torch.from_numpy(v) for k, v in sd.items()}, "hidden_size": 256, "gru_layers": 1,
                "epochs": 1}, ckpt)
    env = dict(os.environ, PEPPER_AMD_REPO=repo, PEPPER_AMD_CKPT=ckpt, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300, cwd=repo)
    lines = [l for l in run.stdout.splitlines() if l.startswith("RESULT ")]
    assert run.returncode == 0 and lines, run.stdout[-2000:] + run.stderr[-3000:]
    out = json.loads(lines[-1][7:])
    if not out["rccl"]:
        pytest.skip("RCCL did not come up on this box: " + out["why"])
    assert out["backend"] == "nccl" and out["checkpoint_equal"] and out["numpy_equal"] and out["blob_floats"] > 10_000_000, out

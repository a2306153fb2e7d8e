"""Multi-GPU plumbing of the hot path: region/file sharding needs no data-path collective; the
only exchange is one broadcast of the packed weight blob from the rank that read the checkpoint
(SURVEY.md 8(e)).  Backend-agnostic on purpose: "nccl" (= RCCL over xGMI) on the GPU box, "gloo"
in the CPU tests.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist


def load_checkpoint_state(model_path):
    """Reference checkpoint schema (train_distributed.py:36-42), 'module.' prefixes stripped
    (ModelHander.py:35-39).  -> (OrderedDict name -> float32 tensor, meta dict)."""
    ckpt = torch.load(model_path, map_location='cpu')
    state = OrderedDict()
    for k, v in ckpt['model_state_dict'].items():
        state[k[7:] if k[0:7] == 'module.' else k] = v.detach().to(torch.float32).contiguous()
    meta = {"hidden_size": int(ckpt['hidden_size']), "gru_layers": int(ckpt['gru_layers']),
            "epochs": int(ckpt.get('epochs', 0))}
    return state, meta


def broadcast_checkpoint(model_path, src=0, device=None, group=None):
    """Rank `src` reads the checkpoint; every rank returns (state_dict, meta).

    Two messages: a small object broadcast with the tensor names/shapes + meta, then ONE flat
    float32 blob (variant 47.4 MB, polish 1.62 MB) -- on RCCL a single xGMI broadcast.
    """
    rank = dist.get_rank(group)
    if rank == src:
        state, meta = load_checkpoint_state(model_path)
        header = [([(k, tuple(v.shape)) for k, v in state.items()], meta)]
    else:
        state, header = None, [None]
    dist.broadcast_object_list(header, src=src, group=group)
    layout, meta = header[0]
    total = int(sum(int(np.prod(s)) if len(s) else 1 for _, s in layout))
    dev = device if device is not None else torch.device("cpu")
    blob = torch.empty(total, dtype=torch.float32, device=dev)
    if rank == src:
        blob.copy_(torch.cat([v.reshape(-1) for v in state.values()]))
    dist.broadcast(blob, src=src, group=group)
    host = blob.cpu()
    out, off = OrderedDict(), 0
    for name, shape in layout:
        n = int(np.prod(shape)) if len(shape) else 1
        out[name] = host[off:off + n].reshape(shape).clone()
        off += n
    return out, meta


def shard_round_robin(items, world, rank):
    """items[i] goes to rank i % world (RunInference.py:104-110, call_consensus.py:93-97)."""
    return [x for i, x in enumerate(items) if i % world == rank]


def broadcast_numpy_state_dict(make_state_dict, shapes, src=0, device=None, group=None):
    """Synthetic-weights variant of broadcast_checkpoint: only rank `src` calls make_state_dict();
    `shapes` = [(name, shape, ...)] is known to every rank, so ONE flat fp32 broadcast suffices."""
    rank = dist.get_rank(group)
    total = int(sum(int(np.prod(s[1])) for s in shapes))
    dev = device if device is not None else torch.device("cpu")
    blob = torch.empty(total, dtype=torch.float32, device=dev)
    if rank == src:
        sd = make_state_dict()
        blob.copy_(torch.from_numpy(np.concatenate([np.asarray(sd[s[0]], np.float32).ravel() for s in shapes])))
    dist.broadcast(blob, src=src, group=group)
    host = blob.cpu().numpy()
    out, off = OrderedDict(), 0
    for s in shapes:
        n = int(np.prod(s[1]))
        out[s[0]] = host[off:off + n].reshape(s[1]).copy()
        off += n
    return out


class AttemptTimedOut(RuntimeError):
    """bounded(): the call did not come back in time; its thread is still inside it."""


def bounded(fn, timeout_s):
    """fn() on a helper thread -> its result, or what it raised; AttemptTimedOut after timeout_s seconds, with the (daemon) thread
    left where it is.  For calls that block inside a communication library with no deadline of their own -- a communicator
    waiting for a peer that has already given up."""
    import threading
    box = {}

    def run():
        try:
            box["value"] = fn()
        except BaseException as err:      # noqa: BLE001
            box["error"] = err
    t = threading.Thread(target=run, name="bounded-attempt", daemon=True)
    t.start()
    t.join(timeout_s)
    if t.is_alive():
        raise AttemptTimedOut("no answer within %.0f s" % timeout_s)
    if "error" in box:
        raise box["error"]
    return box.get("value")


def wait_bounded(work, timeout_s, what="collective"):
    """Poll an async collective's completion (no blocking wait: a stuck device kernel cannot be waited out) -> None, or
    AttemptTimedOut after timeout_s."""
    import time
    deadline = time.monotonic() + timeout_s
    while not work.is_completed():
        if time.monotonic() > deadline:
            raise AttemptTimedOut("%s not complete after %.0f s" % (what, timeout_s))
        time.sleep(0.005)
    work.wait()             # (complete: this only hands over the result / raises what the collective raised)


def agree_on_rccl(world, try_rccl, timeout_s=None):
    """All ranks decide TOGETHER whether the RCCL group is used: the default group is gloo (up wherever the rendezvous is),
    `try_rccl()` -- this rank's attempt to create the RCCL group and run a first collective on it, returning the group or raising
    -- runs on every rank, and the verdicts are summed over gloo.  -> (group or None, ranks that failed, this rank's reason).
    A rank-local fallback (RCCL failed here, so re-initialise with gloo here) deadlocks the ranks where it did not fail.

    timeout_s bounds the attempt: RCCL coming up on seven ranks and throwing on the eighth leaves the seven INSIDE their first
    collective (or inside communicator creation), waiting for a peer that is already on its way to the vote; without a bound
    they sit there until the backend's watchdog fires, minutes later, and takes the process down with it.  With it the attempt
    runs on a helper thread and a rank whose attempt has not answered in time votes "failed" like one whose attempt raised."""
    ok, why, group = 1.0, "", None
    try:
        group = bounded(try_rccl, timeout_s) if timeout_s else try_rccl()
    except Exception as err:        # noqa: BLE001 -- whatever the backend raises
        ok, why = 0.0, repr(err)[:160]
    verdict = torch.tensor([ok])
    dist.all_reduce(verdict)
    failed = world - int(verdict.item())
    return (group if failed == 0 else None), failed, why

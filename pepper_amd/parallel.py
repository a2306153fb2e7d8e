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


def agree_on_rccl(world, try_rccl):
    """All ranks decide TOGETHER whether the RCCL group is used: the default group is gloo (up wherever the rendezvous is),
    `try_rccl()` -- this rank's attempt to create the RCCL group and run a first collective on it, returning the group or raising
    -- runs on every rank, and the verdicts are summed over gloo.  -> (group or None, ranks that failed, this rank's reason).
    A rank-local fallback (RCCL failed here, so re-initialise with gloo here) deadlocks the ranks where it did not fail."""
    ok, why, group = 1.0, "", None
    try:
        group = try_rccl()
    except Exception as err:        # noqa: BLE001 -- whatever the backend raises
        ok, why = 0.0, repr(err)[:160]
    verdict = torch.tensor([ok])
    dist.all_reduce(verdict)
    failed = world - int(verdict.item())
    return (group if failed == 0 else None), failed, why

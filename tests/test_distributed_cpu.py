"""N>1 path on CPU: world_size-2 gloo process group exercising the only collective of the hot path
(weight broadcast) and the reference's file sharding rule."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pepper_amd import synthetic
from pepper_amd.parallel import broadcast_checkpoint, shard_round_robin


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker_np(rank, world, port, out_dir):
    from pepper_amd.parallel import broadcast_numpy_state_dict
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shapes = synthetic.variant_param_shapes()
        sd = broadcast_numpy_state_dict((lambda: synthetic.variant_state_dict(seed=9)) if rank == 0 else None, shapes)
        digest = {k: float(np.abs(v).sum()) for k, v in sd.items()}
        torch.save(digest, os.path.join(out_dir, f"np{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_bench_weight_broadcast_world2(tmp_path):
    """bench.py's N>1 leg: rank 0 builds the synthetic checkpoint, others get it by one broadcast."""
    mp.spawn(_worker_np, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    want = {k: float(np.abs(v).sum()) for k, v in synthetic.variant_state_dict(seed=9).items()}
    for r in range(2):
        got = torch.load(str(tmp_path / f"np{r}.pt"), weights_only=False)
        assert got == want


def _worker(rank, world, port, ckpt_path, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        state, meta = broadcast_checkpoint(ckpt_path if rank == 0 else None, src=0)
        files = [f"img_{i}.hdf5" for i in range(7)]
        mine = shard_round_robin(files, world, rank)
        torch.save({"state": state, "meta": meta, "files": mine}, os.path.join(out_dir, f"r{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_weight_broadcast_and_sharding_world2(tmp_path):
    sd = synthetic.polish_state_dict(seed=3)
    ckpt = synthetic.checkpoint_dict({k: torch.from_numpy(v) for k, v in sd.items()}, hidden_size=128,
                                     gru_layers=1, epochs=7, module_prefix=True)
    path = str(tmp_path / "model.pkl")
    torch.save(ckpt, path)
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), path, str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(str(tmp_path / f"r{r}.pt"), weights_only=False) for r in range(world)]
    for g in got:
        assert g["meta"] == {"hidden_size": 128, "gru_layers": 1, "epochs": 7}
        assert list(g["state"].keys()) == list(sd.keys())          # 'module.' stripped, order kept
        for k, v in sd.items():
            assert np.array_equal(g["state"][k].numpy(), v), k
    assert got[0]["files"] == ["img_0.hdf5", "img_2.hdf5", "img_4.hdf5", "img_6.hdf5"]
    assert got[1]["files"] == ["img_1.hdf5", "img_3.hdf5", "img_5.hdf5"]


def test_shard_files_matches_reference_rule():
    from pepper_amd.variant.RunInference import shard_files
    files = [f"f{i}" for i in range(5)]
    assert shard_files(files, 2) == [["f0", "f2", "f4"], ["f1", "f3"]]
    assert shard_files(files[:1], 4) == [["f0"]]                    # empty chunks dropped
    assert shard_files([], 4) == []
    # with file sizes: largest first onto the least loaded caller (whole-genome shards differ by chromosome length)
    sizes = [248, 242, 198, 190, 181, 171, 159, 145, 138, 133, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47, 51]
    names = ["chr%d" % (i + 1) for i in range(22)]
    got = shard_files(names, 8, sizes=sizes)
    loads = [sum(sizes[names.index(f)] for f in chunk) for chunk in got]
    assert sorted(f for chunk in got for f in chunk) == sorted(names) and len(got) == 8
    assert max(loads) <= 1.08 * (sum(sizes) / 8)
    assert max(sum(sizes[i] for i in range(r, 22, 8)) for r in range(8)) > 1.25 * (sum(sizes) / 8)      # round robin would not be
    assert all(chunk == sorted(chunk, key=names.index) for chunk in got)


def test_one_callers_clean_up_leaves_the_other_callers_files(tmp_path):
    """The SlotTooSmall fall-back of rank 1 removes what rank 1's lanes wrote (pepper_prediction_1.hdf,
    pepper_prediction_1_<lane>.hdf) and nothing of ranks 10, 11 ... that share the prefix."""
    from pepper_amd.variant.RunInference import remove_stale_predictions
    names = ["pepper_prediction_1.hdf", "pepper_prediction_1_0.hdf", "pepper_prediction_1_3.hdf", "pepper_prediction_10.hdf",
             "pepper_prediction_11_0.hdf", "pepper_prediction_12.hdf", "pepper_prediction.hdf", "other.hdf"]
    for n in names:
        (tmp_path / n).write_bytes(b"x")
    remove_stale_predictions(str(tmp_path), pattern="pepper_prediction_1", exact=True)
    assert sorted(os.listdir(tmp_path)) == sorted(["pepper_prediction_10.hdf", "pepper_prediction_11_0.hdf",
                                                   "pepper_prediction_12.hdf", "pepper_prediction.hdf", "other.hdf"])
    remove_stale_predictions(str(tmp_path))                      # start of a run: every prediction file of the directory
    assert sorted(os.listdir(tmp_path)) == ["other.hdf"]


def test_wg_syn_shards_and_the_two_deals():
    """WG-syn: 24 shards with the chromosomes' proportions; over 8 callers the reference's round robin leaves
    one caller well above the mean, the size-ordered deal of RunInference.shard_files within a few per cent."""
    from pepper_amd.variant.RunInference import shard_files
    shards = synthetic.wg_syn_shards(1 << 22)
    assert len(shards) == 24 and all(s % 512 == 0 and s > 0 for s in shards)
    assert abs(sum(shards) - (1 << 22)) < 24 * 512
    assert shards[0] == max(shards) and shards[20] == min(shards)            # chr1 the longest, chr21 the shortest
    names = ["chr%d" % (k + 1) for k in range(22)] + ["chrX", "chrY"]
    size = dict(zip(names, shards))

    def imbalance(chunks):
        loads = [sum(size[n] for n in c) for c in chunks]
        return max(loads) / (sum(loads) / len(loads))
    rr, so = shard_files(names, 8), shard_files(names, 8, shards)
    assert sorted(n for c in rr for n in c) == sorted(names) == sorted(n for c in so for n in c)
    assert imbalance(rr) > 1.15 and imbalance(so) < 1.05
    for world in (1, 2, 4):
        assert imbalance(shard_files(names, world, shards)) < 1.02


def _agree_worker(rank, world, port, fail_on, out_dir):
    import os
    import torch.distributed as dist
    from pepper_amd.parallel import agree_on_rccl
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def attempt():
        if rank in fail_on:
            raise RuntimeError("no RCCL on rank %d" % rank)
        return "the-group"
    group, failed, why = agree_on_rccl(world, attempt)
    dist.barrier()
    with open(os.path.join(out_dir, "r%d" % rank), "w") as fh:
        fh.write("%s %d %s" % (group, failed, "why" if why else "-"))
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_on", [(), (1,), (0, 1)])
def test_ranks_agree_on_the_collective_backend(tmp_path, fail_on):
    """bench.py's N>1 set-up: RCCL failing on SOME ranks must leave every rank on the gloo group (a rank-local fallback hangs the
    others in an RCCL collective); with it everywhere, every rank gets the group."""
    port = _free_port()
    mp.spawn(_agree_worker, args=(2, port, fail_on, str(tmp_path)), nprocs=2, join=True)
    got = [open(str(tmp_path / ("r%d" % r))).read().split() for r in range(2)]
    assert got[0][:2] == got[1][:2] == (["the-group", "0"] if not fail_on else ["None", str(len(fail_on))])
    assert [g[2] for g in got] == ["why" if r in fail_on else "-" for r in range(2)]


def _blocked_worker(rank, world, port, out_dir):
    """Rank 1's attempt raises before it enters the probe; rank 0 is INSIDE the probe all-reduce of a second group, waiting for
    it -- the case the attempt's bound is for (the earlier test's failing rank raised and nobody blocked)."""
    import os
    import time
    from datetime import timedelta
    import torch
    import torch.distributed as dist
    from pepper_amd.parallel import agree_on_rccl, wait_bounded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    state = {}

    def attempt():
        group = dist.new_group(backend="gloo", timeout=timedelta(seconds=60))     # (both ranks get this far: creation is collective)
        if rank == 1:
            raise RuntimeError("the first collective failed on rank 1")
        state["entered"] = time.monotonic()
        wait_bounded(dist.all_reduce(torch.ones(1), group=group, async_op=True), 2.0, "probe all-reduce")
        return group
    t0 = time.monotonic()
    group, failed, why = agree_on_rccl(world, attempt, timeout_s=30.0)
    waited = time.monotonic() - t0
    # the default group still works for both after the vote: the weight broadcast's stand-in
    x = torch.tensor([float(rank + 1)])
    dist.all_reduce(x)
    with open(os.path.join(out_dir, "b%d" % rank), "w") as fh:
        fh.write("%s %d %.2f %d %s" % (group, failed, waited, int(x.item()), why.replace(" ", "_") or "-"))
    os._exit(0)              # (bench.py's leave_group: the abandoned group is not torn down)


def test_a_rank_blocked_in_the_probe_is_released_by_the_bound(tmp_path):
    port = _free_port()
    mp.spawn(_blocked_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [open(str(tmp_path / ("b%d" % r))).read().split() for r in range(2)]
    assert got[0][:2] == got[1][:2] == ["None", "2"]           # both attempts count as failed: one raised, one ran out of time
    assert 1.5 < float(got[0][2]) < 15.0                       # rank 0 sat in the probe for its 2 s, not for the backend's minutes
    assert got[0][3] == got[1][3] == "3"
    assert "AttemptTimedOut" in got[0][4] and "rank_1" in got[1][4]


def _stuck_worker(rank, world, port, out_dir):
    """The attempt blocks somewhere that has no deadline at all (communicator creation): the outer bound of agree_on_rccl."""
    import os
    import threading
    import time
    import torch.distributed as dist
    from pepper_amd.parallel import agree_on_rccl
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def attempt():
        if rank == 0:
            threading.Event().wait()          # for ever
        return "the-group"
    t0 = time.monotonic()
    group, failed, why = agree_on_rccl(world, attempt, timeout_s=1.0)
    with open(os.path.join(out_dir, "s%d" % rank), "w") as fh:
        fh.write("%s %d %.2f" % (group, failed, time.monotonic() - t0))
    dist.barrier()
    os._exit(0)


def test_an_attempt_that_never_returns_is_voted_failed(tmp_path):
    port = _free_port()
    mp.spawn(_stuck_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [open(str(tmp_path / ("s%d" % r))).read().split() for r in range(2)]
    assert got[0][:2] == got[1][:2] == ["None", "1"] and float(got[0][2]) < 10.0

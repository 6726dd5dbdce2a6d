"""Multi-process (world_size 2, gloo, CPU) coverage of the N>1 path: bucketed gradient all-reduce
in backward order, DDP-mean semantics, utterance sharding, LR schedule and accumulation formula."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from coral_amd.trainer import GradSync, shard_indices

    buckets = {"front": [0, 40], "layer0": [40, 104], "layer1": [104, 168], "head": [168, 200]}
    g = torch.arange(200, dtype=torch.float32) * (rank + 1)
    sync = GradSync(g, buckets)
    order = []
    for name in ("head", "layer1", "layer0", "front"):  # the order backward() reports buckets
        sync.start(name)
        order.append(name)
    launched = list(sync.launched)
    sync.finish()
    want_sum = torch.arange(200, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok_sum = torch.equal(g, want_sum)
    sync.scale_()
    ok_mean = torch.allclose(g, want_sum / world)
    # a partial launch (frozen base: only the head bucket) leaves the rest untouched
    g2 = torch.ones(200) * (rank + 1)
    s2 = GradSync(g2, buckets)
    s2.start("head")
    s2.finish()
    ok_partial = bool((g2[168:] == 3).all() and (g2[:168] == rank + 1).all())
    # bf16 wire format: sums of bf16-rounded gradients, written back to the fp32 buffer
    g3 = (torch.arange(200, dtype=torch.float32) * 0.37 + rank).clone()
    want3 = sum(((torch.arange(200, dtype=torch.float32) * 0.37 + r).to(torch.bfloat16)).float() for r in range(world))
    s3 = GradSync(g3, buckets, compress=True)
    s3.start_all()
    s3.finish()
    ok_bf16 = bool((g3 - want3.to(torch.bfloat16).float()).abs().max() <= 1e-6 + 0.008 * want3.abs().max())
    # post-reduction callback (the trainer's per-bucket squared norm): runs once per bucket, after the bucket holds
    # the reduced values (also through the bf16 wire), in launch order
    g4 = (torch.arange(200, dtype=torch.float32) * 0.5 + rank).clone()
    want4 = sum((torch.arange(200, dtype=torch.float32) * 0.5 + r) for r in range(world))
    seen = []
    s4 = GradSync(g4, buckets)

    def post(name):
        lo, hi = buckets[name]
        seen.append((name, float((g4[lo:hi] ** 2).sum())))

    for name in ("head", "layer1", "layer0", "front"):
        s4.start(name, post=post)
    s4.finish()
    ok_post = [n for n, _ in seen] == ["head", "layer1", "layer0", "front"] and torch.equal(g4, want4) and all(
        abs(v - float((want4[buckets[n][0]:buckets[n][1]] ** 2).sum())) <= 1e-3 * v for n, v in seen)
    # sharded buckets (trainer zero_stage): the matrix part of a layer bucket is reduce-scattered - this rank's slice
    # holds the sum over ranks, the bucket's small part and the unsharded buckets are all-reduced as before
    g5 = (torch.arange(200, dtype=torch.float32) * 0.25 + rank).clone()
    want5 = sum((torch.arange(200, dtype=torch.float32) * 0.25 + r) for r in range(world))
    shard = {"layer0": (56, 104), "layer1": (120, 168)}
    s5 = GradSync(g5, buckets, shard=shard)
    seen5 = []
    for name in ("head", "layer1", "layer0", "front"):
        s5.start(name, post=lambda n: seen5.append(n))
    s5.finish()
    ok_shard = seen5 == ["head", "layer1", "layer0", "front"]
    for name, (mlo, hi) in shard.items():
        a, b = s5.slice_of(name)
        per = (hi - mlo) // world
        ok_shard &= (a, b) == (mlo + rank * per, mlo + (rank + 1) * per)
        ok_shard &= torch.equal(g5[a:b], want5[a:b]) and torch.equal(g5[buckets[name][0]:mlo], want5[buckets[name][0]:mlo])
    ok_shard &= torch.equal(g5[:40], want5[:40]) and torch.equal(g5[168:], want5[168:])
    try:
        GradSync(g5, buckets, shard={"layer0": (57, 104)})
        ok_shard = False
    except ValueError:
        pass
    q.put((rank, ok_sum, ok_mean, ok_partial, launched == order and ok_bf16 and ok_post and bool(ok_shard),
           shard_indices(8, rank, world)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1:5] for r in res] == [(True, True, True, True)] * 2
    assert res[0][5] == [0, 1, 2, 3] and res[1][5] == [4, 5, 6, 7]  # disjoint, covering shards


def _rng_worker(rank, world, port, tmp, q):
    """Two ranks write `rng_state_<rank>.pth` into one checkpoint directory the way CoralTrainer.train does (the main rank
    with the checkpoint, the others behind the barrier) and read their OWN file back: the resumed SpecAugment / LayerDrop
    streams continue per rank (HF's per-process `_save_rng_state`), not rank 0's on every rank."""
    import types
    from pathlib import Path

    import numpy as np

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from coral_amd.coral_trainer import CoralTrainer

    tr = CoralTrainer.__new__(CoralTrainer)  # (the host-side checkpoint logic only: no engine, no GPU)
    tr.model = types.SimpleNamespace(_rng=np.random.RandomState(4242 + 17 * rank))
    tr.is_main = rank == 0
    np.random.seed(100 + rank)
    torch.manual_seed(200 + rank)
    tr.model._rng.rand(5 + rank)  # the ranks' streams have advanced differently
    d = Path(tmp) / "checkpoint-3"
    if tr.is_main:
        tr._save_rng_state(d)
    torch.distributed.barrier()
    if not tr.is_main:
        tr._save_rng_state(d)
    torch.distributed.barrier()
    mine = d / f"rng_state_{rank}.pth"
    ok = mine.exists() and len(list(d.glob("rng_state_*.pth"))) == world
    want = tr.model._rng.rand(3).tolist()  # what the uninterrupted run would draw next
    st = torch.load(str(mine), weights_only=False)
    tr.model._rng.rand(7)  # (the "interrupted" process went on; a resume rewinds to the saved state)
    tr._set_rng_state(st)
    got = tr.model._rng.rand(3).tolist()
    q.put((rank, bool(ok), got == want, want))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_every_rank_saves_and_restores_its_own_rng_state(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rng_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] for r in res), res
    assert res[0][3] != res[1][3]  # two different streams were saved, not rank 0's twice


def test_schedule_and_accumulation():
    from coral_amd.trainer import cosine_lr, grad_accumulation_steps, shard_indices

    # transformers.get_cosine_schedule_with_warmup values (warm-up 1000, 100k steps, lr 1e-4)
    assert cosine_lr(0, 1e-4, 1000, 100_000) == 0.0
    assert abs(cosine_lr(500, 1e-4, 1000, 100_000) - 5e-5) < 1e-12
    assert abs(cosine_lr(1000, 1e-4, 1000, 100_000) - 1e-4) < 1e-12
    assert abs(cosine_lr(50_500, 1e-4, 1000, 100_000) - 5e-5) < 1e-9
    assert cosine_lr(100_000, 1e-4, 1000, 100_000) < 1e-12
    # R/src/coral/wav2vec2.py:159-181 with the CoRal defaults 256 / n / 8
    assert grad_accumulation_steps(256, 1, 8) == 32
    assert grad_accumulation_steps(256, 8, 8) == 4
    assert grad_accumulation_steps(64, 8, 8) == 1
    assert grad_accumulation_steps(8, 8, 8) == 1  # never below one
    with pytest.raises(ValueError):
        shard_indices(10, 0, 4)

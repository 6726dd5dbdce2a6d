"""Sharded evaluation under N > 1 ranks (`Trainer.evaluate` shards the evaluation dataloader and
gathers with pad_across_processes(-100) + gather_for_metrics, $TF/trainer.py:2653-2777): world-2 gloo on CPU with a stand-in
model - every rank decodes half of the batches, the id rows are all-gathered, and both ranks report the metrics of the
one-rank run, on a set whose size is not a multiple of world x batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Shape:
    pad_token_id = 0


class _FakeSeq2Seq:
    """generate() = a deterministic function of the clip (its first feature value), rows of different lengths."""

    shape = _Shape()

    def __init__(self):
        self.calls = 0
        self.seen = []

    def eval(self):
        return self

    def generate(self, feats, language=None, task=None, max_length=None):
        self.calls += 1
        rows = []
        for f in feats:
            k = int(f[0])
            self.seen.append(k)
            rows.append([50258, 7 + k % 5] + [10 + (k * j) % 23 for j in range(1, 2 + k % 6)])
        return rows


class _FakeCTC(_FakeSeq2Seq):
    def __call__(self, x, m):
        self._last = [int(r[0]) for r in x]
        self.calls += 1
        self.seen.extend(self._last)

    @property
    def engine(self):
        return self

    def greedy_decode(self):
        return [[3 + (k * j) % 11 for j in range(1, 2 + k % 4)] for k in self._last], None


def _collate(kind):
    def f(exs):
        key = "input_features" if kind == "s2s" else "input_values"
        lab = [e["labels"] for e in exs]
        W = max(len(x) for x in lab)
        L = torch.full((len(lab), W), -100, dtype=torch.int64)
        for i, x in enumerate(lab):
            L[i, :len(x)] = torch.tensor(x)
        d = {key: torch.tensor([[float(e["k"])] for e in exs]), "labels": L}
        if kind != "s2s":
            d["attention_mask"] = torch.ones(len(exs), 1)
        return d
    return f


def _metrics(P, Lb):
    # order-sensitive digest of what compute_metrics is given + a CER-like number
    P, Lb = np.asarray(P), np.asarray(Lb)
    w = np.arange(1, P.shape[0] + 1)[:, None]
    return dict(digest=float((P * w).sum() * 1e-3 + (np.where(Lb < 0, 0, Lb) * w).sum() * 1e-5), rows=int(P.shape[0]),
                cer=float((P[:, 1] % 7).mean()))


def _examples(n):
    return [dict(k=i, labels=[1 + (i * j) % 9 for j in range(1, 2 + i % 5)]) for i in range(n)]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from coral_amd.finetune import evaluate_split, evaluate_split_seq2seq

    n, B = 23, 4  # 6 batches (the last of 3 examples): rank 0 takes batches 0, 2, 4, rank 1 takes 1, 3, 5
    ex = _examples(n)
    m1 = _FakeSeq2Seq()
    r1 = evaluate_split_seq2seq(m1, ex, _collate("s2s"), _metrics, B, 225)
    m2 = _FakeCTC()
    r2 = evaluate_split(m2, ex, _collate("ctc"), _metrics, B)
    q.put((rank, r1, sorted(m1.seen), r2, sorted(m2.seen)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
def test_sharded_evaluation_matches_the_one_rank_run_and_every_rank_decodes_its_share():
    from coral_amd.finetune import eval_batches_of_rank, evaluate_split, evaluate_split_seq2seq

    n, B, world = 23, 4, 2
    ex = _examples(n)
    ref1 = evaluate_split_seq2seq(_FakeSeq2Seq(), ex, _collate("s2s"), _metrics, B, 225, rank=0, world=1)
    ref2 = evaluate_split(_FakeCTC(), ex, _collate("ctc"), _metrics, B, rank=0, world=1)
    assert ref1["rows"] == n and ref2["rows"] == n
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    seen_all = []
    for rank, r1, seen1, r2, seen2 in got:
        assert r1 == ref1, f"rank {rank}: seq2seq metrics differ from the one-rank run"
        assert r2 == ref2, f"rank {rank}: CTC metrics differ from the one-rank run"
        mine = [i for lo, hi in eval_batches_of_rank(n, B, rank, world) for i in range(lo, hi)]
        assert seen1 == mine and seen2 == mine, f"rank {rank} decoded {seen1}, its share is {mine}"
        seen_all += seen1
    assert sorted(seen_all) == list(range(n))  # every example exactly once over the ranks
    by_rank = {g_[0]: g_ for g_ in got}  # (the queue delivers in completion order)
    assert len(by_rank[0][2]) == 12 and len(by_rank[1][2]) == 11  # half each


def test_eval_batches_round_robin_and_gather_rejects_a_missing_example():
    from coral_amd.finetune import eval_batches_of_rank, gather_rows_in_order

    assert eval_batches_of_rank(10, 4, 0, 2) == [(0, 4), (8, 10)]
    assert eval_batches_of_rank(10, 4, 1, 2) == [(4, 8)]
    assert eval_batches_of_rank(3, 4, 1, 2) == []
    out = gather_rows_in_order([[1, 2], [3]], [1, 0], 2, -100, 0, 1)
    assert out.tolist() == [[3, -100], [1, 2]]

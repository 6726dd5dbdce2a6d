"""The multi-GPU engine path with two real ranks: two processes, one process group (gloo; the GPU box
has a single device, both ranks run on cuda:0), two optimiser steps of `DataParallelTrainer` on the wav2vec2 engine with
the backward-hooked per-bucket all-reduce, per-bucket norms on the communication stream and the per-bucket AdamW.

DDP semantics being matched (accelerate -> torch DDP under R/src/scripts/finetune_asr_model.py:9-12, $TF/trainer.py:
1750-1759,1961): gradients averaged over ranks, clip + AdamW replicated  =>  replicas stay bit-identical, and the run
equals ONE rank stepping on the two shards as two accumulation micro-batches (loss and gradients scaled by 1/2).
"""
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _run_two_ranks(tmp_path, wire, steps=2, zero=0, kind="wav2vec2"):
    tmp_path.mkdir(parents=True, exist_ok=True)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--nnodes=1", "--nproc-per-node=2",
           "--local-addr", "127.0.0.1", str(ROOT / "tests" / "dp_worker.py"), str(tmp_path), wire, str(steps), str(zero), kind]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return [torch.load(tmp_path / f"rank{k}.pt") for k in range(2)]


def _single_rank_reference(steps=2):
    sys.path.insert(0, str(ROOT / "tests"))
    import dp_worker

    from coral_amd.trainer import DataParallelTrainer

    eng, shard = dp_worker.build_case()
    tr = DataParallelTrainer(eng, learning_rate=1e-3, warmup_steps=0, max_steps=100, max_grad_norm=1.0, grad_accum=2)
    losses, norms = [], []
    for _ in range(steps):
        losses.append(float(tr.train_step([shard([0, 1]), shard([2, 3])])))
        norms.append(tr.grad_norm())
    tr.finish()
    torch.cuda.synchronize()
    return losses, norms, eng.store.p32.cpu()


@pytest.mark.parametrize("wire", ["fp32", "bf16"])
def test_two_ranks_match_one_rank_accumulating(tmp_path, wire):
    r0, r1 = _run_two_ranks(tmp_path, wire)
    # replicas: identical parameters (fp32 master and bf16 compute copy), bit for bit
    assert torch.equal(r0["p32"], r1["p32"]) and torch.equal(r0["p16"], r1["p16"])
    assert r0["norms"] == r1["norms"]
    losses, norms, p32 = _single_rank_reference()
    lr = 1e-3
    tol = 1e-5 if wire == "fp32" else 2e-3   # the bf16 wire rounds each rank's gradient to 8 bits once
    for s in range(2):
        mean = 0.5 * (r0["losses"][s] + r1["losses"][s])  # the 1-rank run returns sum(loss_i / 2)
        # (second step: AdamW's first update is lr * sign(g) wherever v is fresh, so gradient elements near zero whose
        # last bits depend on the order of summation - a + b on the wire against accumulation in place - move opposite
        # ways; on this tiny model that is worth up to ~2e-4 of the next loss, measured 1.1e-4)
        assert abs(mean - losses[s]) <= (1e-6 if s == 0 else max(10 * tol, 1e-3)) * abs(losses[s]), (s, mean, losses[s])
        assert abs(r0["norms"][s] - norms[s]) <= 10 * tol * norms[s], (s, r0["norms"][s], norms[s])
    # after two AdamW steps of size <= lr each: nearly every parameter moved exactly as in the 1-rank run
    d = (r0["p32"] - p32).abs()
    assert float(d.max()) <= 4 * lr   # two AdamW steps, each bounded by ~lr x |m̂| / sqrt(v̂), opposite signs at worst
    close = float((d <= 0.05 * lr).float().mean())
    assert close >= (0.995 if wire == "fp32" else 0.9), close
    assert float((r0["p32"] - p32).norm() / (2 ** 0.5 * lr * p32.numel() ** 0.5)) <= (0.02 if wire == "fp32" else 0.2)


@pytest.mark.parametrize("wire", ["fp32", "bf16"])
def test_sharded_optimizer_equals_the_replicated_trainer(tmp_path, wire):
    """zero_stage (the reference's `--zero-stage 2` launch, R/makefile:79-84): reduce-scatter of the layers' weight-matrix
    gradients, AdamW + moments on this rank's 1/N slice, all-gather of the bf16 compute copy.  Two real ranks, three
    steps: the gathered fp32 master parameters, the bf16 compute copy, the moments, the losses and the gradient norms
    equal the replicated trainer's bit for bit on the fp32 wire (two ranks: a + b either way) - and both ranks agree."""
    rep = _run_two_ranks(tmp_path / "rep", wire, steps=3)
    zer = _run_two_ranks(tmp_path / "zero", wire, steps=3, zero=2)
    for k in ("p32", "p16", "m", "v"):
        assert torch.equal(zer[0][k], zer[1][k]), k
    for r in (0, 1):
        # the first step's loss is bit-identical (same parameters); later ones to the bf16 weight copies that flip by one
        # ulp when the clip factor differs in its last bits (the squared norm is summed over other partials)
        assert zer[r]["losses"][0] == rep[r]["losses"][0]
        # (measured up to 2.2e-4 at the third step on this tiny model)
        assert all(abs(a - b) <= 1e-3 * abs(b) for a, b in zip(zer[r]["losses"], rep[r]["losses"])), (zer[r]["losses"], rep[r]["losses"])
    for i, (a, b) in enumerate(zip(zer[0]["norms"], rep[0]["norms"])):
        # step 1: the same parameters, sums of per-slice partials in another order; later steps inherit the one-ulp flips
        # of the bf16 copies described above (which steps flip depends on the values: 3.9e-5 seen on the bf16 wire in round 5)
        assert abs(a - b) <= (1e-6 if i == 0 else 1e-3) * b, (i, a, b)
    if wire == "fp32":
        lr = 1e-3
        d = (zer[0]["p32"] - rep[0]["p32"]).abs()
        # the clip factor comes from the norm, whose last bits depend on the order of the partial sums: parameters agree
        # to that (a relative 1e-6 of a learning-rate-sized step), the bf16 copies bit for bit almost everywhere
        # (parameters: a relative 1e-6 of a learning-rate-sized step from the clip factor, plus what a handful of flipped
        # bf16 weights did to the third step's gradients)
        assert float(d.max()) <= 0.05 * lr, float(d.max())
        assert float((d <= 1e-3 * lr).float().mean()) >= 0.999
        assert float((zer[0]["p16"] != rep[0]["p16"]).float().mean()) <= 1e-3
        assert float((zer[0]["m"] - rep[0]["m"]).abs().max()) <= 1e-2 * float(rep[0]["m"].abs().max())
    else:
        # bf16 wire: gradient elements at the noise floor (k_proj.bias, whose true gradient is zero - softmax does not see
        # a key bias -, and the q / k weights beside it: |g| ~ 3e-7) come off the two exchange paths with other last bits;
        # AdamW's normalisation turns their sign into a full +-lr step, three steps in a row at worst (round 5, after the
        # GELU evaluation changed the values: 497 of 4.8 M elements beyond lr, all of them such elements - tools/scratch
        # diagnosis in NOTEBOOK R5).  Everything else agrees.
        lr = 1e-3
        d = (zer[0]["p32"] - rep[0]["p32"]).abs()
        assert float(d.max()) <= 3.2 * lr, float(d.max())
        assert float((d <= 0.1 * lr).float().mean()) >= 0.99


def test_sharded_optimizer_on_the_whisper_engine(tmp_path):
    """zero_stage on the Whisper finetune step (configs[4] is whisper-large-turbo under the same `--zero-stage 2`
    launch): the encoder layers' weight matrices are reduce-scattered and updated on 1/N, decoder / embeddings stay
    replicated.  Two real ranks, three steps against the replicated trainer: both ranks agree bit for bit, the first
    loss is identical, parameters and moments agree to the clip factor's last bits."""
    rep = _run_two_ranks(tmp_path / "rep", "fp32", steps=3, kind="whisper")
    zer = _run_two_ranks(tmp_path / "zero", "fp32", steps=3, zero=2, kind="whisper")
    for k in ("p32", "p16", "m", "v"):
        assert torch.equal(zer[0][k], zer[1][k]), k
    for r in (0, 1):
        # (same parameters at the first step; the cross-entropy sums its rows with float atomics, so two runs agree to
        # the last bits only)
        assert abs(zer[r]["losses"][0] - rep[r]["losses"][0]) <= 1e-6 * abs(rep[r]["losses"][0])
        assert all(abs(a - b) <= 1e-3 * abs(b) for a, b in zip(zer[r]["losses"], rep[r]["losses"])), (zer[r]["losses"], rep[r]["losses"])
    for a, b in zip(zer[0]["norms"], rep[0]["norms"]):
        assert abs(a - b) <= 1e-5 * b
    lr = 1e-3
    d = (zer[0]["p32"] - rep[0]["p32"]).abs()
    assert float(d.max()) <= 0.1 * lr, float(d.max())
    assert float((d <= 1e-3 * lr).float().mean()) >= 0.995
    assert float((zer[0]["m"] - rep[0]["m"]).abs().max()) <= 1e-2 * float(rep[0]["m"].abs().max())


def test_rccl_one_rank_group_reproduces_the_plain_trainer(tmp_path):
    """backend "nccl" (RCCL) with one rank and the exchange path forced on: see tests/rccl_one_rank_worker.py."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = tmp_path / "one_rank.pt"
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "rccl_one_rank_worker.py"), str(out), "2", str(port)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = torch.load(out)
    plain, f32, b16 = res["plain"], res["fp32"], res["bf16"]
    # the evaluation's id gather through RCCL (device tensors) = the direct one-rank result
    assert torch.equal(res["gather_direct"], res["gather_rccl"]) and res["gather_rccl"].shape == (4, 6)
    assert res["gather_rccl"][1].tolist() == [1, 2, 3, 4, 5, 6] and res["gather_rccl"][0].tolist() == [-100] * 6
    # Without the clip (see the worker: the comparison must not hang on the last bit of the gradient norm) the identity
    # exchange changes nothing, bit for bit: fp32 wire through the C ABI's ca_* collectives (the default on RCCL) and
    # through torch.distributed's, and the sharded optimiser over the same one-rank group (RCCL's in-place
    # reduce-scatter and all-gather are identities)
    for k in ("fp32", "fp32_torch", "zero"):
        assert torch.equal(plain["p32"], res[k]["p32"]) and torch.equal(plain["p16"], res[k]["p16"]), k
        assert plain["losses"] == res[k]["losses"], k
        assert all(abs(a - b) <= 1e-6 * a for a, b in zip(plain["norms"], res[k]["norms"])), k  # (per-bucket sums vs one pass)
    # with the clip (max_grad_norm 1.0): one step, parameters equal to the rounding of the clip coefficient
    pc = res["plain_clip"]
    for k in ("fp32_clip", "zero_clip"):
        assert pc["losses"] == res[k]["losses"] and abs(pc["norms"][0] - res[k]["norms"][0]) <= 1e-6 * pc["norms"][0]
        assert float((pc["p32"] - res[k]["p32"]).abs().max()) <= 5e-7, k  # (one ulp of a parameter near 1 is 1.2e-7)
        assert float((pc["p16"] != res[k]["p16"]).float().mean()) <= 1e-4, k
    assert float((pc["p32"] - plain["p32"]).abs().max()) > 0  # (the clip did act: norm ~ 950)
    # bf16 wire: one rounding of the gradients to 8 bits
    lr = 1e-3
    assert plain["losses"][0] == b16["losses"][0]
    assert abs(plain["norms"][0] - b16["norms"][0]) <= 2e-3 * plain["norms"][0]
    d = (plain["p32"] - b16["p32"]).abs()
    assert float(d.max()) <= 4 * lr and float((d <= 0.05 * lr).float().mean()) >= 0.9


def test_comm_context_collectives_through_the_c_abi():
    """ca_comm_* (include/coral_amd.h) on a context of ONE rank: unique id -> init -> the three in-place collectives on
    the context's own stream, ordered against torch's stream by ca_comm_after / ca_comm_before.  Over one rank a SUM
    all-reduce, a reduce-scatter and an all-gather are identities: the buffers must come back unchanged - and the work
    enqueued after ca_comm_before must see them."""
    import ctypes as C

    from coral_amd import ops

    lib = ops.lib()
    ident = C.create_string_buffer(128)
    assert lib.ca_comm_unique_id(ident) == 0, lib.ca_last_error()
    assert any(ident.raw)  # RCCL filled it
    ctx = C.c_void_p()
    torch.cuda.set_device(0)
    assert lib.ca_comm_init(C.byref(ctx), ident, 0, 1) == 0, lib.ca_last_error()
    assert lib.ca_comm_rank(ctx) == 0 and lib.ca_comm_world(ctx) == 1 and lib.ca_comm_stream(ctx)
    x = torch.randn(1 << 20, device="cuda:0")
    h = x.to(torch.bfloat16)
    want, want_h = x.clone(), h.clone()
    cur = torch.cuda.current_stream().cuda_stream
    assert lib.ca_comm_after(ctx, cur) == 0
    assert lib.ca_allreduce_bucket(ctx, x.data_ptr(), x.numel(), 0) == 0, lib.ca_last_error()
    assert lib.ca_reduce_scatter_bucket(ctx, x.data_ptr(), x.numel(), 0) == 0, lib.ca_last_error()
    assert lib.ca_allgather_bucket(ctx, h.data_ptr(), h.numel(), 1) == 0, lib.ca_last_error()
    assert lib.ca_comm_before(ctx, cur) == 0
    y = x * 2  # enqueued on torch's stream behind the collectives
    torch.cuda.synchronize()
    assert torch.equal(x, want) and torch.equal(h, want_h) and torch.equal(y, want * 2)
    assert lib.ca_allreduce_bucket(ctx, x.data_ptr(), 0, 0) != 0 and lib.ca_allreduce_bucket(ctx, x.data_ptr(), 8, 7) != 0  # argument checks
    assert lib.ca_comm_destroy(ctx) == 0

"""RCCL plumbing on a 1-GPU box (tests/test_dp_gpu.py): one process, backend "nccl" (= RCCL) with a process group of
ONE rank, the trainer's exchange path forced on (CA_DP_FORCE=1).  A sum over one rank is the identity, so the run must
reproduce the plain single-process trainer: bit for bit on the fp32 wire, to bf16 rounding on the bf16 wire.  What this
covers that the two-rank gloo test cannot: RCCL initialisation and its asynchronous handles on the communication stream,
with the per-bucket callbacks and the per-bucket AdamW behind them."""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def run(forced, wire, steps, zero=0, clip=1e9):
    """clip = the trainer's max_grad_norm.  The default (1e9: the clip coefficient is exactly 1) makes the comparison
    independent of the last bits of the gradient norm, which the plain trainer sums from the weight-gradient kernels'
    per-tile partials and the exchange path per bucket: one ulp of the norm moves every parameter by one ulp, a few of
    the 4.8 M bf16 copies then round the other way, and two steps later nothing is bit-identical any more."""
    import dp_worker

    from coral_amd.trainer import DataParallelTrainer

    os.environ["CA_DP_FORCE"] = "1" if forced else "0"
    # (the plain trainer would keep a single micro-batch's weight-matrix gradients in bf16; the exchange path reduces
    # fp32 gradients - the comparison is about the exchange, so both run on the fp32 buffer)
    os.environ["CA_WGRAD_BF16"] = "0"
    eng, shard = dp_worker.build_case()
    tr = DataParallelTrainer(eng, learning_rate=1e-3, warmup_steps=0, max_steps=100, max_grad_norm=clip,
                             compress_grads=(wire == "bf16"), zero_stage=zero)
    assert tr.dist == forced and tr.overlap == forced and tr.world == 1 and tr.zero == bool(zero)
    # RCCL ranks exchange through the C ABI's ca_* collectives (include/coral_amd.h), CA_COMM_CAPI=0 = torch.distributed's
    assert (tr.sync.capi is not None) == (forced and os.environ.get("CA_COMM_CAPI", "1") != "0")
    if zero and forced:
        assert tr._collectives_selfcheck(eng.store.device, None, tr.sync.capi)
    mb = shard([0, 1, 2, 3])
    losses, norms = [], []
    for _ in range(steps):
        losses.append(float(tr.train_step([mb])))
        norms.append(tr.grad_norm())
    tr.finish()
    tr.consolidate()
    torch.cuda.synchronize()
    out = dict(losses=losses, norms=norms, p32=eng.store.p32.cpu(), p16=eng.store.p16.float().cpu(),
               launched=list(tr.sync.buckets))
    tr.close()  # the communicator and its stream go back (eight trainers are built in this process)
    assert tr.sync.capi is None
    return out


def main():
    out, steps = Path(sys.argv[1]), int(sys.argv[2])
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{sys.argv[3]}", rank=0, world_size=1,
                                         device_id=torch.device("cuda:0"))
    res = {"plain": run(False, "fp32", steps), "fp32": run(True, "fp32", steps), "bf16": run(True, "bf16", steps),
           # the sharded optimiser's RCCL calls (in-place reduce_scatter_tensor / all_gather_into_tensor) over one rank
           "zero": run(True, "fp32", steps, zero=2)}
    # with the clip active (max_grad_norm 1.0, the reference's value): ONE step, compared to the norm's rounding
    res["plain_clip"] = run(False, "fp32", 1, clip=1.0)
    res["fp32_clip"] = run(True, "fp32", 1, clip=1.0)
    res["zero_clip"] = run(True, "fp32", 1, zero=2, clip=1.0)
    # the same exchange through torch.distributed's own RCCL calls (the A/B switch)
    os.environ["CA_COMM_CAPI"] = "0"
    res["fp32_torch"] = run(True, "fp32", steps)
    os.environ.pop("CA_COMM_CAPI")
    # sharded evaluation's id gather (coral_amd/finetune.py) over the same group: device tensors through RCCL's
    # all_reduce(MAX) + all_gather; one rank holds every row, so the collective route must equal the direct one
    from coral_amd.finetune import gather_rows_in_order

    rows, starts = [[5, 6, 7], [], [9], [1, 2, 3, 4, 5, 6]], [2, 0, 3, 1]
    res["gather_direct"] = torch.from_numpy(gather_rows_in_order(rows, starts, 4, -100, 0, 1))
    res["gather_rccl"] = torch.from_numpy(gather_rows_in_order(rows, starts, 4, -100, 0, 1, collective=True))
    torch.save(res, out)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

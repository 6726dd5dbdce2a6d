"""`finetune(config)` — mirror of R/src/coral/finetune.py:21-95 on the MI355X engine: build the
processor, model, data stream, collator and trainer from the ModelSetup, `trainer.train(...)`, save the
model in HF layout.  The loop itself lives in `coral_amd.coral_trainer.CoralTrainer`."""

from __future__ import annotations

import logging
import os

import torch

from .data import load_data_for_finetuning
from .model_setup import load_model_setup

logger = logging.getLogger(__package__)


def finetune(config, n_examples: int | None = None) -> dict:
    """Finetune a model on a dataset - the reference's function, statement for statement where the hot path is
    concerned (R/src/coral/finetune.py:21-95); experiment tracking, n-gram training and hub upload are out of scope
    (DESIGN.md §7).  Returns a dict for callers and tests (the reference returns None)."""
    from .coral_trainer import EarlyStoppingCallback

    is_main_process = os.getenv("RANK", "0") == "0"
    model_setup = load_model_setup(config)
    if config.model.type == "whisper":  # the GPU log-mel front end lives on the model's engine
        model = model_setup.load_model()
        processor = model_setup.load_processor()
    else:
        processor = model_setup.load_processor()
        model = model_setup.load_model()
    if is_main_process:
        processor.save_pretrained(config.model_dir)
    dataset = load_data_for_finetuning(config, processor, n_examples, model=model)

    vals = {name: split for name, split in dataset.items() if name.startswith("val")}
    eval_dataset = None if not vals else (list(vals.values())[0] if len(vals) == 1 else vals)
    if eval_dataset is None and is_main_process:
        logger.info("No validation set found. Disabling early stopping.")

    trainer = model_setup.load_trainer_class()(
        model=model,
        data_collator=model_setup.load_data_collator(),
        args=model_setup.load_training_arguments(),
        compute_metrics=model_setup.load_compute_metrics(),
        train_dataset=dataset["train"],
        eval_dataset=eval_dataset,
        processing_class=getattr(processor, "tokenizer"),
        callbacks=[EarlyStoppingCallback(early_stopping_patience=config.early_stopping_patience)]
        if eval_dataset is not None and config.early_stopping else None,
    )
    out = trainer.train(resume_from_checkpoint=config.get("resume_from_checkpoint", False),
                        ignore_data_skip=config.get("ignore_data_skip", False))
    if is_main_process:
        model.save_pretrained(config.model_dir)
    return dict(history=trainer.state["log_history"], model=model, processor=processor, trainer=trainer,
                steps_done=out.global_step, state=trainer.state, train_output=out)


def _dist_rank_world():
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_rank(), torch.distributed.get_world_size()
    return 0, 1


def eval_batches_of_rank(n_examples: int, batch_size: int, rank: int, world: int) -> list[tuple[int, int]]:
    """The evaluation batches `rank` decodes: batch k = examples [k B, (k + 1) B) goes to rank k % world - the
    round-robin of accelerate's `BatchSamplerShard` under `Trainer.evaluate` ($TF/trainer.py:2653-2777: the evaluation
    dataloader is sharded over the ranks, every rank decodes 1 / world of the set)."""
    batches = [(i, min(i + batch_size, n_examples)) for i in range(0, n_examples, batch_size)]
    return batches[rank::world]


def gather_rows_in_order(rows: list[list[int]], starts: list[int], n_examples: int, fill: int, rank: int, world: int,
                         collective: bool | None = None):
    """Every rank's id rows -> ONE [n_examples, W] int64 array in dataset order, on every rank.

    rows[j] belongs to example starts[j].  The rows are padded with `fill` to the widest row of any rank and the row
    counts to the largest of any rank with all-`fill` rows (what `pad_across_processes(pad_index=-100)` +
    `gather_for_metrics` do, $TF/trainer.py:2754-2777; the filler rows are what R/src/coral/compute_metrics.py:63-66
    special-cases) - here they carry the example index -1 and are dropped after the gather instead.  Ids, not logits:
    46 x fewer bytes than the reference moves.  One all_gather of a [rows, 1 + W] matrix per evaluation set.
    `collective=True` takes the collective route over a one-rank group too (tests/rccl_one_rank_worker.py: RCCL with
    device tensors on a 1-GPU box)."""
    import numpy as np

    if world == 1 and not collective:
        W = max((len(r) for r in rows), default=1)
        out = np.full((n_examples, max(W, 1)), fill, dtype=np.int64)
        for r, i in zip(rows, starts):
            out[i, :len(r)] = r
        return out
    dist = torch.distributed
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    dims = torch.tensor([len(rows), max((len(r) for r in rows), default=1)], dtype=torch.int64, device=dev)
    dist.all_reduce(dims, op=dist.ReduceOp.MAX)
    R, W = int(dims[0]), max(1, int(dims[1]))
    local = np.full((R, 1 + W), fill, dtype=np.int64)
    local[:, 0] = -1
    for j, (r, i) in enumerate(zip(rows, starts)):
        local[j, 0] = i
        local[j, 1:1 + len(r)] = r
    mine = torch.from_numpy(local).to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    out = np.full((n_examples, W), fill, dtype=np.int64)
    seen = np.zeros(n_examples, dtype=bool)
    for part in parts:
        a = part.cpu().numpy()
        keep = a[:, 0] >= 0
        out[a[keep, 0]] = a[keep, 1:]
        seen[a[keep, 0]] = True
    if not seen.all():
        raise RuntimeError(f"sharded evaluation: {int((~seen).sum())} of {n_examples} examples were decoded by no rank")
    return out


def evaluate_split_seq2seq(model, examples, collator, compute_metrics, batch_size, max_length, rank=None, world=None) -> dict:
    """`predict_with_generate` evaluation (R/src/coral/whisper.py:221-222): greedy generation on the GPU,
    CER/WER of the decoded strings on the host.  Under N > 1 ranks every rank generates for its share of the batches
    (`eval_batches_of_rank`), the id rows are all-gathered (`gather_rows_in_order`) and every rank computes the same
    metrics from the whole set - early stopping stays in lock-step."""
    if rank is None or world is None:
        rank, world = _dist_rank_world()
    model.eval()
    preds, labels, starts = [], [], []
    for lo, hi in eval_batches_of_rank(len(examples), batch_size, rank, world):
        batch = collator(examples[lo:hi])
        ids = model.generate(batch["input_features"], language="danish", task="transcribe", max_length=max_length)
        preds.extend(ids.tolist() if hasattr(ids, "tolist") else ids)
        labels.extend(batch["labels"].tolist())
        starts.extend(range(lo, hi))
    P = gather_rows_in_order(preds, starts, len(examples), model.shape.pad_token_id, rank, world)
    Lb = gather_rows_in_order(labels, starts, len(examples), -100, rank, world)
    return compute_metrics(P, Lb)


def evaluate_split(model, examples, collator, compute_metrics, batch_size, rank=None, world=None) -> dict:
    """Greedy CTC evaluation: argmax + collapse on the GPU, CER/WER on the host; sharded over the ranks like
    `evaluate_split_seq2seq`."""
    if rank is None or world is None:
        rank, world = _dist_rank_world()
    model.eval()
    preds, labels, starts = [], [], []
    for lo, hi in eval_batches_of_rank(len(examples), batch_size, rank, world):
        batch = collator(examples[lo:hi])
        with torch.no_grad():
            model(batch["input_values"], batch["attention_mask"])
        ids, _ = model.engine.greedy_decode()
        preds.extend(list(r) for r in ids)
        labels.extend(batch["labels"].tolist())
        starts.extend(range(lo, hi))
    P = gather_rows_in_order(preds, starts, len(examples), model.shape.pad_token_id, rank, world)
    Lb = gather_rows_in_order(labels, starts, len(examples), -100, rank, world)
    # ids are already collapsed: decode without grouping so genuine double letters survive
    tok = compute_metrics.keywords["processor"].tokenizer if hasattr(compute_metrics, "keywords") else None
    if tok is not None:
        from .metrics import cer, wer

        ps = [tok.decode(r, group_tokens=False).lower().strip() for r in P]
        Lb2 = Lb.copy()
        Lb2[Lb2 == -100] = tok.pad_token_id
        ls = [tok.decode(r, group_tokens=False).lower().strip() for r in Lb2]
        return dict(cer=cer(ps, ls), wer=wer(ps, ls))
    return compute_metrics(P, Lb)

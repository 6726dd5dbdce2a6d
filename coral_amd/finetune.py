"""`finetune(config)` — mirror of R/src/coral/finetune.py:21-95 on the MI355X engine: build the
processor, model, data stream, collator and trainer from the ModelSetup, run `max_steps` optimiser
steps (evaluating every `eval_steps`), save the model in HF layout."""

from __future__ import annotations

import json
import logging
import os
import shutil
import time
from pathlib import Path

import torch

from .data import load_data_for_finetuning
from .model_setup import load_model_setup

logger = logging.getLogger(__package__)


def _all_ranks_agree(flag: bool, device) -> bool:
    """True iff `flag` is True on every rank (one scalar MIN all-reduce; a rank that ran out of data must not leave
    the others blocked in the gradient all-reduce)."""
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return flag
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
    return bool(int(t.item()))


def _checkpoint_dirs(model_dir: Path) -> list[Path]:
    out = [d for d in Path(model_dir).glob("checkpoint-*") if d.is_dir() and d.name.split("-")[-1].isdigit()]
    return sorted(out, key=lambda d: int(d.name.split("-")[-1]))


def save_checkpoint(model, trainer, model_dir: Path, step: int, state: dict, save_total_limit: int, keep: Path | None):
    """`checkpoint-<step>/`: the model in HF layout, the optimiser moments and the trainer state
    (Trainer._save_checkpoint + rotation, $TF/trainer.py:3079,3326; `save_total_limit` never deletes the best one)."""
    from safetensors.torch import save_file

    d = Path(model_dir) / f"checkpoint-{step}"
    model.save_pretrained(d)
    trainer.finish()
    torch.cuda.synchronize()
    save_file(dict(m=trainer.m.cpu(), v=trainer.v.cpu()), str(d / "optimizer.safetensors"))
    (d / "trainer_state.json").write_text(json.dumps(dict(global_step=step, **state), indent=1))
    if save_total_limit and save_total_limit > 0:
        protected = {d, keep}
        deletable = [c for c in _checkpoint_dirs(model_dir) if c not in protected]  # oldest first
        total = len(_checkpoint_dirs(model_dir))
        while total > save_total_limit and deletable:
            shutil.rmtree(deletable.pop(0), ignore_errors=True)
            total -= 1
    return d


def load_checkpoint(model, trainer, ckpt: Path) -> dict:
    """Resume: parameters, optimiser moments and step count (the cosine schedule continues where it stopped)."""
    from safetensors.torch import load_file

    from .modeling import load_checkpoint_tensors

    sd = load_checkpoint_tensors(ckpt)
    eng = trainer.engine
    if any(k.startswith("model.") for k in sd) or hasattr(eng, "exported_names"):
        sd = {(k if k.startswith("model.") else "model." + k): v for k, v in sd.items() if k != "proj_out.weight"}
    if hasattr(eng, "exported_names"):
        eng.load_state_dict(sd)
    else:  # wav2vec2: `masked_spec_embed` is only in the file when SpecAugment is configured (as HF saves it)
        rep = eng.load_state_dict(sd, strict=False, init_missing=False)
        if [n for n in rep["missing"] if n != "wav2vec2.masked_spec_embed"]:
            raise KeyError(f"{ckpt}: checkpoint lacks {rep['missing']}")
    if hasattr(eng, "refresh_derived"):
        eng.refresh_derived()
    opt = load_file(str(ckpt / "optimizer.safetensors"))
    trainer.m.copy_(opt["m"])
    trainer.v.copy_(opt["v"])
    state = json.loads((ckpt / "trainer_state.json").read_text())
    trainer.opt_step = int(state["global_step"])
    return state


def finetune(config, n_examples: int | None = None) -> dict:
    is_main = os.getenv("RANK", "0") == "0"
    setup = load_model_setup(config)
    if config.model.type == "whisper":  # the GPU log-mel front end lives on the model's engine
        model = setup.load_model()
        processor = setup.load_processor()
    else:
        processor = setup.load_processor()
        model = setup.load_model()
    if is_main:
        processor.save_pretrained(config.model_dir)
    dataset = load_data_for_finetuning(config, processor, n_examples, model=model)
    collator = setup.load_data_collator()
    args = setup.load_training_arguments()
    compute_metrics = setup.load_compute_metrics()
    trainer = setup.load_trainer_class()(
        model, learning_rate=args.learning_rate, betas=(args.adam_beta1, args.adam_beta2),
        max_grad_norm=args.max_grad_norm, warmup_steps=args.warmup_steps, max_steps=args.max_steps,
        grad_accum=args.gradient_accumulation_steps)
    device = trainer.engine.device
    model_dir = Path(config.model_dir)
    B = args.per_device_train_batch_size
    accum = args.gradient_accumulation_steps
    history = []
    state = dict(best_metric=None, best_step=None, bad_evals=0)
    start_step = 0
    resume = config.get("resume_from_checkpoint", False)
    if resume:  # True = the newest checkpoint under model_dir (Trainer.train(resume_from_checkpoint=True)); or a path
        ckpt = Path(resume) if isinstance(resume, str) else (_checkpoint_dirs(model_dir) or [None])[-1]
        if ckpt is None or not Path(ckpt).exists():
            raise FileNotFoundError(f"resume_from_checkpoint={resume!r}: no checkpoint-* directory under {model_dir}")
        st = load_checkpoint(model, trainer, Path(ckpt))
        start_step = int(st["global_step"])
        state.update({k: st[k] for k in ("best_metric", "best_step", "bad_evals") if k in st})
        if is_main:
            logger.info("resumed from %s at step %d", ckpt, start_step)

    it = iter(dataset["train"])
    epoch = 0
    if start_step and not config.get("ignore_data_skip", False):
        # Trainer skips the batches the first run consumed so that the data order continues (`ignore_data_skip`)
        for _ in range(start_step * accum * B):
            try:
                next(it)
            except StopIteration:
                it = iter(dataset["train"])

    def next_micro_batch():
        """One per-device batch; a dry stream starts the next epoch (Trainer re-iterates an IterableDataset until
        `max_steps`, $TF/trainer.py:1678-1726), on every rank together."""
        nonlocal it, epoch
        for attempt in range(2):
            feats = []
            try:
                while len(feats) < B:
                    feats.append(next(it))
            except StopIteration:
                pass
            if _all_ranks_agree(len(feats) == B, device):  # dataloader_drop_last=True: a short batch is dropped
                return collator(feats)
            epoch += 1
            it = iter(dataset["train"])
        raise RuntimeError(f"the training stream yields fewer than per_device_batch_size={B} examples per epoch on "
                           "some rank: nothing to train on")

    best_dir = None
    metric_key = "cer" if config.model.type == "wav2vec2" else "wer"
    t0 = time.time()
    step = start_step - 1
    for step in range(start_step, args.max_steps):
        micro = [next_micro_batch() for _ in range(accum)]
        loss = trainer.train_step(micro)
        if (step + 1) % args.logging_steps == 0 or step == start_step:
            history.append(dict(step=step + 1, loss=float(loss), lr=trainer.lr, epoch=epoch, elapsed=time.time() - t0))
            if is_main:
                logger.info("step %d loss %.4f", step + 1, float(loss))
        stop = False
        if (step + 1) % args.eval_steps == 0 or step + 1 == args.max_steps:
            if config.model.type == "whisper":
                metrics = evaluate_split_seq2seq(model, dataset["val"], collator, compute_metrics, B,
                                                 args.generation_max_length)
            else:
                metrics = evaluate_split(model, dataset["val"], collator, compute_metrics, B)
            history.append(dict(step=step + 1, **{f"val_{k}": v for k, v in metrics.items()}))
            cur = metrics.get(metric_key)
            if cur is not None and (state["best_metric"] is None or cur < state["best_metric"]):
                state.update(best_metric=cur, best_step=step + 1, bad_evals=0)
            else:
                state["bad_evals"] += 1
            # EarlyStoppingCallback (R/src/coral/finetune.py:66-75): stop after `patience` evaluations without a new best
            stop = bool(config.early_stopping) and state["bad_evals"] >= int(config.early_stopping_patience)
        save_now = args.save_strategy != "no" and ((step + 1) % args.save_steps == 0 or
                                                   (state["best_step"] == step + 1 and args.load_best_model_at_end))
        if save_now and is_main:
            keep = model_dir / f"checkpoint-{state['best_step']}" if state["best_step"] else None
            d = save_checkpoint(model, trainer, model_dir, step + 1, state, args.save_total_limit, keep)
            if state["best_step"] == step + 1:
                best_dir = d
        if save_now and torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.barrier()
        if stop:
            if is_main:
                logger.info("early stopping at step %d (best %s=%.4f at step %s)", step + 1, metric_key,
                            state["best_metric"], state["best_step"])
            break
    trainer.finish()  # the last optimiser step may still be running on the trainer's side stream
    torch.cuda.synchronize()
    if args.load_best_model_at_end and state["best_step"] and state["best_step"] != step + 1:
        best = model_dir / f"checkpoint-{state['best_step']}"
        if best.exists():  # `load_best_model_at_end` (R/src/coral/wav2vec2.py:233): the saved model is the best one
            load_checkpoint(model, trainer, best)
            if is_main:
                logger.info("loaded the best model (step %d)", state["best_step"])
    if is_main:
        model.save_pretrained(config.model_dir)
    return dict(history=history, model=model, processor=processor, trainer=trainer, steps_done=step + 1, state=state)


def evaluate_split_seq2seq(model, examples, collator, compute_metrics, batch_size, max_length) -> dict:
    """`predict_with_generate` evaluation (R/src/coral/whisper.py:221-222): greedy generation on the GPU,
    CER/WER of the decoded strings on the host."""
    model.eval()
    preds, labels = [], []
    for i in range(0, len(examples), batch_size):
        batch = collator(examples[i:i + batch_size])
        ids = model.generate(batch["input_features"], language="danish", task="transcribe", max_length=max_length)
        preds.extend(ids.tolist() if hasattr(ids, "tolist") else ids)
        labels.extend(batch["labels"].tolist())
    import numpy as np

    pad = model.shape.pad_token_id
    P = np.full((len(preds), max(len(p) for p in preds)), pad, dtype=np.int64)
    for i, p in enumerate(preds):
        P[i, :len(p)] = p
    Lb = np.full((len(labels), max(len(x) for x in labels)), -100, dtype=np.int64)
    for i, x in enumerate(labels):
        Lb[i, :len(x)] = x
    return compute_metrics(P, Lb)


def evaluate_split(model, examples, collator, compute_metrics, batch_size) -> dict:
    """Greedy CTC evaluation: argmax + collapse on the GPU, CER/WER on the host."""
    model.eval()
    preds, labels = [], []
    for i in range(0, len(examples), batch_size):
        batch = collator(examples[i:i + batch_size])
        with torch.no_grad():
            model(batch["input_values"], batch["attention_mask"])
        ids, _ = model.engine.greedy_decode()
        width = max(1, max(len(x) for x in ids))
        for row in ids:
            preds.append(row + [model.shape.pad_token_id] * (width - len(row)))
        labels.extend(batch["labels"].tolist())
    import numpy as np

    W = max(len(p) for p in preds)
    P = np.full((len(preds), W), model.shape.pad_token_id, dtype=np.int64)
    for i, p in enumerate(preds):
        P[i, :len(p)] = p
    Lw = max(len(x) for x in labels)
    Lb = np.full((len(labels), Lw), -100, dtype=np.int64)
    for i, x in enumerate(labels):
        Lb[i, :len(x)] = x
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

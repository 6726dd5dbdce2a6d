"""`finetune(config)` — mirror of R/src/coral/finetune.py:21-95 on the MI355X engine: build the
processor, model, data stream, collator and trainer from the ModelSetup, run `max_steps` optimiser
steps (evaluating every `eval_steps`), save the model in HF layout."""

from __future__ import annotations

import logging
import os
import time

import torch

from .data import load_data_for_finetuning
from .model_setup import load_model_setup

logger = logging.getLogger(__package__)


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
    it = iter(dataset["train"])
    B = args.per_device_train_batch_size
    history = []
    t0 = time.time()
    for step in range(args.max_steps):
        micro = []
        for _ in range(args.gradient_accumulation_steps):
            feats = []
            for _ in range(B):
                try:
                    feats.append(next(it))
                except StopIteration:
                    break
            if len(feats) < B:  # dataloader_drop_last=True
                break
            micro.append(collator(feats))
        if len(micro) < args.gradient_accumulation_steps:
            break
        loss = trainer.train_step(micro)
        if (step + 1) % args.logging_steps == 0 or step == 0:
            history.append(dict(step=step + 1, loss=float(loss), lr=trainer.lr, elapsed=time.time() - t0))
            if is_main:
                logger.info("step %d loss %.4f", step + 1, float(loss))
        if (step + 1) % args.eval_steps == 0 or step + 1 == args.max_steps:
            if config.model.type == "whisper":
                metrics = evaluate_split_seq2seq(model, dataset["val"], collator, compute_metrics, B,
                                                 args.generation_max_length)
            else:
                metrics = evaluate_split(model, dataset["val"], collator, compute_metrics, B)
            history.append(dict(step=step + 1, **{f"val_{k}": v for k, v in metrics.items()}))
    trainer.finish()  # the last optimiser step may still be running on the trainer's side stream
    torch.cuda.synchronize()
    if is_main:
        model.save_pretrained(config.model_dir)
    return dict(history=history, model=model, processor=processor)


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

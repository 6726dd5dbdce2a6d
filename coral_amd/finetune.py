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

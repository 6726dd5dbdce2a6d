"""`evaluate(config)` — mirror of R/src/coral/evaluate.py:29-85 for the CTC path: load the saved
model + processor, transcribe the evaluation examples in batches (greedy CTC on the GPU), normalise
both sides like the reference (lower / strip) and report CER / WER.  The demographic slicing of the
reference (`get_score_df`, :161-216) is pandas reporting and out of scope."""

from __future__ import annotations

import csv
import logging
from pathlib import Path

import torch

from .data import synthetic_examples
from .metrics import cer, wer
from .model_setup import load_model_setup

logger = logging.getLogger(__package__)


def transcribe(model, processor, arrays: list, batch_size: int = 16) -> list[str]:
    """ASR-pipeline equivalent ($TF/pipelines/automatic_speech_recognition.py:345,569-575,679):
    feature-extract, forward, argmax over all frames, CTC collapse, decode."""
    model.eval()
    out = []
    for i in range(0, len(arrays), batch_size):
        feats = [processor(a, sampling_rate=processor.feature_extractor.sampling_rate) for a in arrays[i:i + batch_size]]
        batch = processor.feature_extractor.pad(feats, padding="longest")
        with torch.no_grad():
            model(torch.from_numpy(batch["input_values"]), torch.from_numpy(batch["attention_mask"]))
        ids, _ = model.engine.greedy_decode()
        out += [processor.tokenizer.decode(r, group_tokens=False) for r in ids]
    return out


def evaluate(config, examples: list | None = None) -> dict:
    """config: evaluation.yaml keys (+ `model_dir`).  examples: list of {"audio": array, "text": str};
    defaults to a seeded synthetic set (no hub access here)."""
    from .config import DictConfig

    mcfg = DictConfig(model=DictConfig(type="wav2vec2", sampling_rate=config.sampling_rate, decoder=None),
                      model_dir=config.get("model_dir", config.model_id), padding="longest",
                      max_seconds_per_example=config.max_seconds_per_example)
    saved = load_model_setup(mcfg).load_saved()
    model, processor = saved.model, saved.processor
    if examples is None:
        examples = [dict(audio=ex["input_values"], text=ex["text"])
                    for ex in synthetic_examples(processor, 2 * config.batch_size, 99, config.min_seconds_per_example,
                                                 min(3.0, config.max_seconds_per_example), config.sampling_rate)]
    preds = transcribe(model, processor, [e["audio"] for e in examples], config.batch_size)
    preds = [p.lower().strip() for p in preds]
    labels = [e["text"].lower().strip() if config.lower_case else e["text"].strip() for e in examples]
    scores = dict(cer=cer(preds, labels), wer=wer(preds, labels), n=len(examples))
    if config.store_results:
        name = str(config.model_id).replace("/", "--") + "." + str(config.dataset).split("::")[0].replace("/", "--")
        path = Path(f"{name}.csv")
        with path.open("w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["prediction", "label"])
            w.writerows(zip(preds, labels))
        scores["csv"] = str(path)
    return scores

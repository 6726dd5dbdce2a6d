"""`evaluate(config)` — mirror of R/src/coral/evaluate.py:29-85: load the saved model + processor (wav2vec2 or
Whisper, by the saved architecture), transcribe the evaluation examples in batches (greedy CTC, or log-mel +
greedy `generate(language="danish", task="transcribe")`, on the GPU), normalise both sides like the reference
(lower / strip) and report CER / WER.  The demographic slicing of the
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


def transcribe_whisper(model, processor, arrays: list, batch_size: int = 16, max_length: int | None = None):
    """The Whisper branch of the ASR pipeline ($TF/pipelines/automatic_speech_recognition.py:345,529,600): pad / trim
    every clip to 30 s, log-mel on the GPU, `model.generate(input_features, language="danish", task="transcribe")`
    (R/src/coral/evaluate.py:56-60), decode the generated ids without special tokens.
    -> (texts, id rows).  Without the byte-level BPE files (offline) the texts are the ids rendered as words."""
    model.eval()
    texts, rows = [], []
    max_length = int(max_length or model.shape.max_target_positions)
    for i in range(0, len(arrays), batch_size):
        feats = processor.feature_extractor(arrays[i:i + batch_size], sampling_rate=processor.feature_extractor.sampling_rate)
        ids = model.generate(feats, language="danish", task="transcribe", max_length=max_length)
        ids = ids.tolist() if hasattr(ids, "tolist") else [list(map(int, r)) for r in ids]
        rows += ids
        texts += processor.batch_decode(ids, skip_special_tokens=True)
    return texts, rows


def saved_model_type(model_dir) -> str:
    """"wav2vec2" or "whisper", from the `architectures` / `model_type` of the saved config.json (the ASR pipeline
    dispatches on the loaded model's class the same way, $TF/pipelines/automatic_speech_recognition.py:195-215)."""
    import json

    cfg = json.loads((Path(model_dir) / "config.json").read_text())
    arch = " ".join(cfg.get("architectures") or []) + " " + str(cfg.get("model_type", ""))
    if "whisper" in arch.lower():
        return "whisper"
    if "wav2vec2" in arch.lower():
        return "wav2vec2"
    raise ValueError(f"{model_dir}: unsupported architecture {arch.strip()!r}")


def evaluate(config, examples: list | None = None) -> dict:
    """config: evaluation.yaml keys (+ `model_dir`).  examples: list of {"audio": array, "text": str};
    defaults to a seeded synthetic set (no hub access here).  Serves both model types, like the reference's
    pipeline-based `evaluate` (R/src/coral/evaluate.py:29-85, :123-158)."""
    from .config import DictConfig

    model_dir = config.get("model_dir", config.model_id)
    mtype = saved_model_type(model_dir)
    mcfg = DictConfig(model=DictConfig(type=mtype, sampling_rate=config.sampling_rate, decoder=None),
                      model_dir=model_dir, padding="longest",
                      max_seconds_per_example=config.max_seconds_per_example)
    saved = load_model_setup(mcfg).load_saved()
    model, processor = saved.model, saved.processor
    id_rows = None
    if mtype == "whisper":
        if examples is None:
            import numpy as np

            rng = np.random.RandomState(99)
            examples = []
            for _ in range(2 * config.batch_size):
                n = int(rng.uniform(config.min_seconds_per_example, min(3.0, config.max_seconds_per_example)) * config.sampling_rate)
                w = np.clip(0.1 * rng.randn(n), -1, 1).astype(np.float32)
                examples.append(dict(audio=w / np.abs(w).max(), text=""))
        preds, id_rows = transcribe_whisper(model, processor, [e["audio"] for e in examples], config.batch_size,
                                            config.get("generation_max_length", None))
    else:
        if examples is None:
            examples = [dict(audio=ex["input_values"], text=ex["text"])
                        for ex in synthetic_examples(processor, 2 * config.batch_size, 99, config.min_seconds_per_example,
                                                     min(3.0, config.max_seconds_per_example), config.sampling_rate)]
        preds = transcribe(model, processor, [e["audio"] for e in examples], config.batch_size)
    preds = [p.lower().strip() for p in preds]
    labels = [e["text"].lower().strip() if config.lower_case else e["text"].strip() for e in examples]
    scores = dict(model_type=mtype, n=len(examples))
    if any(labels):  # CER / WER need reference texts (a synthetic Whisper set has none: ids-only output)
        scores.update(cer=cer(preds, labels), wer=wer(preds, labels))
    if config.store_results:
        name = str(config.model_id).replace("/", "--") + "." + str(config.dataset).split("::")[0].replace("/", "--")
        path = Path(f"{name}.csv")
        with path.open("w", newline="") as f:
            w = csv.writer(f)
            if id_rows is not None and getattr(processor, "tokenizer", None) is None:
                w.writerow(["prediction", "label", "token_ids"])
                w.writerows((p, l, " ".join(map(str, r))) for p, l, r in zip(preds, labels, id_rows))
            else:
                w.writerow(["prediction", "label"])
                w.writerows(zip(preds, labels))
        scores["csv"] = str(path)
    if id_rows is not None:
        scores["token_ids"] = id_rows
    return scores

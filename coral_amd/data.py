"""Data plane of the hot path.  The reference's `coral.data` (HF `datasets` streaming, text
normalisation, augmentation; R/src/coral/data.py) is host-side I/O that needs the network and is out
of scope; what the hot path needs from it is a stream of examples
`{"input_values": f32[n], "labels": [ids], "input_length": n}` — exactly what `process_example`
emits (R/src/coral/data.py:747-757).  This module provides that stream for the `synthetic` dataset
key and for local `.npz` shards, so the entry points run end-to-end offline."""

from __future__ import annotations

from pathlib import Path

import numpy as np


def synthetic_examples(processor, n: int, seed: int, min_seconds=1.0, max_seconds=10.0, sampling_rate=16_000,
                       fixed_length: bool = False, raw: bool = False):
    """Seeded 0.1*randn utterances, peak-normalised like `ta.PeakNormalization`
    (R/src/coral/data.py:710), with random transcriptions over the tokenizer's characters.
    raw=True leaves the audio as the dataset holds it before `process_example` (`example["audio"]["array"]`,
    R/src/coral/data.py:704-706): normalisation, augmentation and featurisation then happen on the GPU
    (coral_amd/input_pipeline.py, the training path); raw=False featurises on the host like `processor(audio)`
    at R/src/coral/data.py:747 (validation examples, whose text is scored on the host)."""
    rng = np.random.RandomState(seed)
    chars = [c for c in processor.tokenizer.get_vocab() if len(c) == 1 and c != "|"]
    for _ in range(n):
        secs = max_seconds if fixed_length else rng.uniform(min_seconds, max_seconds)
        wave = np.clip(0.1 * rng.randn(int(secs * sampling_rate)), -1, 1).astype(np.float32)
        wave /= np.abs(wave).max()
        n_chars = int(rng.randint(5, max(6, int(secs * 12))))
        text = "".join(rng.choice(chars + [" "] * 6, size=n_chars)).strip() or "a"
        if raw:
            ex = dict(audio=dict(array=wave, sampling_rate=sampling_rate), input_length=len(wave))
        else:
            ex = processor(wave, sampling_rate=sampling_rate)
            ex["input_length"] = len(ex["input_values"])
        ex["labels"] = processor(text=text, truncation=True)["input_ids"]
        ex["text"] = text
        yield ex


def synthetic_whisper_examples(processor, n: int, seed: int, prefix, eos: int, min_seconds=1.0, max_seconds=10.0,
                               sampling_rate=16_000, max_tokens: int = 64, raw: bool = False):
    """Whisper flavour of `process_example` (R/src/coral/data.py:747-757): `input_features` is the log-mel
    of the 30 s padded clip (computed on the GPU by the processor's feature extractor), `labels` the
    token ids `<|sot|><|da|><|transcribe|><|notimestamps|> … <|endoftext|>`.  The byte-level BPE files
    are not available offline, so the transcription ids are drawn directly (text tokens < 50257)."""
    rng = np.random.RandomState(seed)
    for _ in range(n):
        secs = rng.uniform(min_seconds, max_seconds)
        wave = np.clip(0.1 * rng.randn(int(secs * sampling_rate)), -1, 1).astype(np.float32)
        wave /= np.abs(wave).max()
        ids = rng.randint(0, 50257, size=int(rng.randint(4, max_tokens))).tolist()
        if raw:  # (the batch's log-mel is taken on the GPU by the trainer's input pipeline)
            yield dict(audio=dict(array=wave, sampling_rate=sampling_rate), labels=list(prefix) + ids + [eos],
                       input_length=len(wave))
            continue
        feats = processor.feature_extractor(wave, sampling_rate=sampling_rate)[0]
        yield dict(input_features=feats, labels=list(prefix) + ids + [eos], input_length=len(wave))


class ExampleStream:
    """A re-iterable example stream (what a streaming `IterableDataset` is to `Trainer`): every `iter()` starts a
    fresh pass, so the training loop can begin another epoch when a pass runs dry."""

    def __init__(self, *factories):
        self.factories = factories

    def __iter__(self):
        for f in self.factories:
            yield from f()


def load_data_for_finetuning(config, processor, n_examples: int | None = None, model=None):
    """-> {"train": iterable, "val": list}.  Only `datasets=synthetic` and local `.npz` directories
    (arrays `audio`, `text`) are supported offline."""
    import os

    from .trainer import grad_accumulation_steps

    rank, world = int(os.environ.get("RANK", "0") or 0), int(os.environ.get("WORLD_SIZE", "1") or 1)
    try:
        import torch

        ndev = max(torch.cuda.device_count(), 1)
    except Exception:
        ndev = 1
    # one pass of the synthetic stream covers the whole run: per-device batch x accumulation x steps
    accum = grad_accumulation_steps(config.total_batch_size, ndev, config.per_device_batch_size)
    n_run = n_examples or config.per_device_batch_size * accum * config.max_steps
    out_train = []
    raw = bool(config.get("device_input_pipeline", True))  # training examples keep their raw audio (module docstring)
    if config.model.type == "whisper":
        from .whisper_setup import prefix_ids

        shape = model.shape
        for key, ds in config.datasets.items():
            if ds["id"] != "synthetic":
                raise RuntimeError(f"dataset {key!r} ({ds['id']}) needs the HuggingFace hub and the Whisper tokenizer "
                                   "files; this environment is offline — use datasets=synthetic")
        mk = lambda k, sd, hi, raw=False: synthetic_whisper_examples(  # noqa: E731
            processor, k, sd, prefix_ids(shape), shape.eos_token_id, config.min_seconds_per_example, hi,
            config.model.sampling_rate, raw=raw)
        return {"train": ExampleStream(lambda: mk(n_run, config.seed + 1000 * rank, config.max_seconds_per_example, raw)),
                "val": list(mk(4, config.seed + 7, 3.0))}
    for key, ds in config.datasets.items():
        if ds["id"] == "synthetic":
            fixed = config.padding == "max_length"
            out_train.append(lambda fixed=fixed: synthetic_examples(
                processor, n_run, config.seed + 1000 * rank, config.min_seconds_per_example,
                config.max_seconds_per_example, config.model.sampling_rate, fixed_length=fixed, raw=raw))
        elif Path(ds["id"]).is_dir():
            out_train.append(lambda ds=ds: _npz_examples(Path(ds["id"]), processor, ds["text_column"],
                                                         config.model.sampling_rate, rank, world, raw=raw))
        else:
            raise RuntimeError(f"dataset {key!r} ({ds['id']}) needs the HuggingFace hub; this environment is "
                               "offline — use datasets=synthetic or a local directory of .npz shards")

    val = list(synthetic_examples(processor, 8, config.seed + 7, 1.0, 3.0, config.model.sampling_rate))
    return {"train": ExampleStream(*out_train), "val": val}


def _npz_examples(root: Path, processor, text_column: str, sampling_rate: int, rank: int = 0, world: int = 1,
                  raw: bool = False):
    """Local shards, dealt round-robin to the ranks (file i goes to rank i % world) so that no two ranks train on the
    same example — what `split_dataset_by_node` does for the reference's streaming datasets under accelerate."""
    for f in sorted(root.glob("*.npz"))[rank::world]:
        z = np.load(f, allow_pickle=True)
        if raw:
            ex = dict(audio=dict(array=z["audio"].astype(np.float32), sampling_rate=sampling_rate),
                      input_length=len(z["audio"]))
        else:
            ex = processor(z["audio"].astype(np.float32), sampling_rate=sampling_rate)
            ex["input_length"] = len(ex["input_values"])
        ex["labels"] = processor(text=str(z[text_column]), truncation=True)["input_ids"]
        yield ex

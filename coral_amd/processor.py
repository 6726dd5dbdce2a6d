"""Processor objects of the wav2vec2 path: character CTC tokenizer + waveform feature extractor.

Mirrors what `Wav2Vec2ModelSetup.load_processor` builds (R/src/coral/wav2vec2.py:49-102):
`Wav2Vec2CTCTokenizer` over the dumped `vocab.json` (<pad> = CTC blank, `|` = word delimiter,
model_max_length 512) and `Wav2Vec2FeatureExtractor(do_normalize=True, return_attention_mask=True)`.
Only the behaviour the path uses is implemented (HF itself is never imported).
"""

from __future__ import annotations

import json
from itertools import groupby
from pathlib import Path

import numpy as np

SPECIAL_TOKENS = ("<s>", "</s>", "<unk>", "<pad>")


def dump_vocabulary(characters_to_keep: str, model_dir: str | Path) -> dict:
    """R/src/coral/wav2vec2.py:308-329: sorted unique characters + '|' -> vocab.json."""
    chars = sorted(set(characters_to_keep + "|"))
    vocab = {c: i for i, c in enumerate(chars)}
    model_dir = Path(model_dir)
    model_dir.mkdir(parents=True, exist_ok=True)
    (model_dir / "vocab.json").write_text(json.dumps(vocab, ensure_ascii=False))
    return vocab


class CTCTokenizer:
    """Character tokenizer with the special tokens appended after the vocabulary file's entries."""

    def __init__(self, vocab: dict, model_max_length: int = 512):
        self.vocab = dict(vocab)
        for tok in SPECIAL_TOKENS:
            if tok not in self.vocab:
                self.vocab[tok] = len(self.vocab)
        self.inv = {i: t for t, i in self.vocab.items()}
        self.pad_token, self.unk_token, self.bos_token, self.eos_token = "<pad>", "<unk>", "<s>", "</s>"
        self.word_delimiter_token = "|"
        self.model_max_length = model_max_length

    @classmethod
    def from_pretrained(cls, model_dir: str | Path, **_):
        return cls(json.loads((Path(model_dir) / "vocab.json").read_text()))

    pad_token_id = property(lambda self: self.vocab["<pad>"])
    unk_token_id = property(lambda self: self.vocab["<unk>"])
    bos_token_id = property(lambda self: self.vocab["<s>"])
    eos_token_id = property(lambda self: self.vocab["</s>"])

    def get_vocab(self) -> dict:
        return dict(self.vocab)

    def __len__(self):
        return len(self.vocab)

    def encode(self, text: str, truncation: bool = True) -> list[int]:
        """Characters -> ids; ' ' -> '|' ($TF/models/wav2vec2/tokenization_wav2vec2.py:_tokenize)."""
        ids = [self.vocab.get(self.word_delimiter_token if c == " " else c, self.unk_token_id) for c in text]
        return ids[: self.model_max_length] if truncation else ids

    def decode(self, ids, group_tokens: bool = True, skip_special_tokens: bool = False) -> str:
        """CTC decode: collapse repeats, drop <pad>, '|' -> ' ', strip
        ($TF/models/wav2vec2/tokenization_wav2vec2.py:297-358)."""
        toks = [self.inv.get(int(i), self.unk_token) for i in ids]
        if skip_special_tokens:
            toks = [t for t in toks if t not in SPECIAL_TOKENS]
        if group_tokens:
            toks = [k for k, _ in groupby(toks)]
        toks = [t for t in toks if t != self.pad_token]
        return "".join(" " if t == self.word_delimiter_token else t for t in toks).strip()

    def batch_decode(self, sequences, **kw) -> list[str]:
        return [self.decode(s, **kw) for s in sequences]


class WaveformFeatureExtractor:
    """zero-mean / unit-variance normalisation + padding with an attention mask
    ($TF/models/wav2vec2/feature_extraction_wav2vec2.py:77-97,99-236)."""

    def __init__(self, sampling_rate: int = 16_000, padding_value: float = 0.0, do_normalize: bool = True):
        self.sampling_rate = sampling_rate
        self.padding_value = padding_value
        self.do_normalize = do_normalize

    def __call__(self, audio, sampling_rate: int | None = None) -> dict:
        """One example -> {"input_values": f32 array} (what `processor(audio)` yields per example,
        R/src/coral/data.py:747)."""
        if sampling_rate is not None and sampling_rate != self.sampling_rate:
            raise ValueError(f"expected {self.sampling_rate} Hz audio, got {sampling_rate}")
        a = np.asarray(audio, dtype=np.float32)
        if self.do_normalize:
            a = (a - a.mean()) / np.sqrt(a.var() + 1e-7)
        return {"input_values": a.astype(np.float32)}

    def pad(self, features: list[dict], padding="longest", max_length: int | None = None) -> dict:
        arrays = [np.asarray(f["input_values"], dtype=np.float32) for f in features]
        if padding in (False, "do_not_pad"):
            n = max(len(a) for a in arrays)
            if any(len(a) != n for a in arrays):
                raise ValueError("do_not_pad needs equal-length inputs to form a batch")
        elif padding == "max_length":
            n = int(max_length)
            arrays = [a[:n] for a in arrays]
        else:
            n = max(len(a) for a in arrays)
        vals = np.full((len(arrays), n), self.padding_value, dtype=np.float32)
        mask = np.zeros((len(arrays), n), dtype=np.int32)
        for i, a in enumerate(arrays):
            vals[i, : len(a)] = a
            mask[i, : len(a)] = 1
        return {"input_values": vals, "attention_mask": mask}


class Wav2Vec2Processor:
    """feature_extractor + tokenizer pair (`Wav2Vec2Processor` look-alike)."""

    def __init__(self, feature_extractor: WaveformFeatureExtractor, tokenizer: CTCTokenizer):
        self.feature_extractor = feature_extractor
        self.tokenizer = tokenizer

    def __call__(self, audio=None, sampling_rate=None, text=None, truncation=True):
        if audio is not None:
            return self.feature_extractor(audio, sampling_rate)
        return {"input_ids": self.tokenizer.encode(text, truncation)}

    def batch_decode(self, ids, **kw):
        return self.tokenizer.batch_decode(ids, **kw)

    def save_pretrained(self, model_dir):
        model_dir = Path(model_dir)
        model_dir.mkdir(parents=True, exist_ok=True)
        base = {k: v for k, v in self.tokenizer.vocab.items() if k not in SPECIAL_TOKENS}
        (model_dir / "vocab.json").write_text(json.dumps(base, ensure_ascii=False))
        (model_dir / "preprocessor_config.json").write_text(json.dumps(
            {"do_normalize": True, "feature_size": 1, "padding_value": 0.0, "return_attention_mask": True,
             "sampling_rate": self.feature_extractor.sampling_rate,
             "feature_extractor_type": "Wav2Vec2FeatureExtractor"}, indent=2))

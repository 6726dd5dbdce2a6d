"""Batch collation for the CTC path — mirror of `DataCollatorCTCWithPadding`
(R/src/coral/data_collators.py:17-95): pad `input_values` to longest / max_length =
sample_rate * max_seconds_per_example, pad labels and fill them with -100.  This defines the batch
layout the hot path consumes: {input_values f32[B,N], attention_mask i32[B,N], labels i64[B,L]}."""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class DataCollatorCTCWithPadding:
    processor: object
    sample_rate: int
    max_seconds_per_example: float
    padding: bool | str = "longest"

    def __call__(self, features: list[dict]) -> dict:
        if "input_values" in features[0]:
            audio = [dict(input_values=f["input_values"]) for f in features]
        elif "audio" in features[0]:
            audio = [dict(input_values=f["audio"]["array"]) for f in features]
        else:
            raise ValueError("Features must contain either 'input_values' or 'audio' key.")
        padding = "longest" if self.padding is True else self.padding
        batch = self.processor.feature_extractor.pad(
            audio, padding=padding, max_length=int(self.sample_rate * self.max_seconds_per_example))
        max_lab = min(self.processor.tokenizer.model_max_length, 512)
        labs = [list(f["labels"]) for f in features]
        L = max_lab if padding == "max_length" else max((len(x) for x in labs), default=0)
        labels = np.full((len(labs), L), -100, dtype=np.int64)
        for i, x in enumerate(labs):
            x = x[:L]
            labels[i, : len(x)] = x
        return {"input_values": torch.from_numpy(batch["input_values"]),
                "attention_mask": torch.from_numpy(batch["attention_mask"]),
                "labels": torch.from_numpy(labels)}

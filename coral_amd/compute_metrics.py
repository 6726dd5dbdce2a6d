"""`compute_error_rate_metrics` mirror (R/src/coral/compute_metrics.py:18-94): logits -> argmax ->
CTC collapse -> lower/strip -> CER / WER.  On the GPU path the argmax+collapse already happened in
ca_ctc_greedy_decode; this host function accepts either logits [B,T,V] or id rows."""

from __future__ import annotations

import numpy as np

from .metrics import cer, wer


def compute_error_rate_metrics(predictions, label_ids, processor) -> dict:
    tok = processor.tokenizer
    pad = tok.pad_token_id
    labels = np.array(label_ids, copy=True)
    labels[labels == -100] = pad
    predictions = np.asarray(predictions)
    if predictions.ndim == 3:
        predictions = np.array(predictions, copy=True)
        # rows that are all -100 (pad_across_processes filler) decode to the blank
        predictions[np.all(predictions == -100, axis=-1), pad] = 0
        pred_ids = predictions.argmax(-1)
        preds = tok.batch_decode(pred_ids)
    elif predictions.ndim == 2:
        preds = tok.batch_decode(predictions)
    else:
        raise ValueError(f"Expected predictions to have either 2 or 3 dimensions, but found {predictions.ndim} dimensions.")
    labs = tok.batch_decode(labels, group_tokens=False)
    preds = [p.lower().strip() for p in preds]
    labs = [x.lower().strip() for x in labs]
    return dict(cer=cer(preds, labs), wer=wer(preds, labs))

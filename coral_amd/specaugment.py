"""Host-side SpecAugment span sampling, bit-compatible with the reference's NumPy RNG stream.

The reference draws the masks on the host with `np.random` inside
`_compute_mask_indices` ($TF/models/wav2vec2/modeling_wav2vec2.py:101-217), driven by
CoRal's mask_time_prob / mask_feature_prob keys (R/src/coral/wav2vec2.py:114-118,
R/config/model/wav2vec2-large.yaml:18-21).  The engine takes the resulting boolean masks
(tiny H2D copies) and applies them on the GPU (ca_mask_frames).  Same draw order as the
reference: one `rand(1)` for the probabilistic rounding, then one `choice` per batch row.
"""

from __future__ import annotations

import numpy as np


def compute_mask_indices(shape, mask_prob: float, mask_length: int, lengths=None, min_masks: int = 0,
                         rng=np.random) -> np.ndarray:
    """bool [batch, length] span mask.  `lengths`: valid length per row (None = full)."""
    batch, seq_len = shape
    if mask_length < 1:
        raise ValueError("`mask_length` has to be bigger than 0.")
    if mask_length > seq_len:
        raise ValueError(f"`mask_length` {mask_length} has to be smaller than `sequence_length` {seq_len}")
    eps = float(rng.rand(1)[0])

    def n_spans(n):
        k = max(int(mask_prob * n / mask_length + eps), min_masks)
        if k * mask_length > seq_len:
            k = seq_len // mask_length
        if n - (mask_length - 1) < k:
            k = max(n - (mask_length - 1), 0)
        return k

    row_len = [seq_len] * batch if lengths is None else [int(x) for x in lengths]
    mask = np.zeros((batch, seq_len), dtype=bool)
    kmax = n_spans(seq_len)
    if kmax == 0:
        return mask
    for b, n in enumerate(row_len):
        k = n_spans(n)
        starts = rng.choice(np.arange(n - (mask_length - 1)), k, replace=False)
        # rows are padded to kmax spans with a repeat of the first start (a no-op), or with the
        # last (padding) position when nothing was drawn
        fill = seq_len - 1 if len(starts) == 0 else starts[0]
        starts = np.concatenate([starts, np.full(kmax - k, fill, dtype=np.int64)]).astype(np.int64)
        idx = (starts[:, None] + np.arange(mask_length)[None, :]).reshape(-1)
        mask[b, np.minimum(idx, seq_len - 1)] = True
    return mask


def sample_masks(batch: int, frames: int, hidden: int, frame_lengths, mask_time_prob: float,
                 mask_time_length: int, mask_feature_prob: float, mask_feature_length: int,
                 mask_time_min_masks: int = 2, mask_feature_min_masks: int = 0, rng=np.random):
    """Masks for one training step in the order `_mask_hidden_states` draws them
    ($TF/models/wav2vec2/modeling_wav2vec2.py:1272-1316): time first, then feature."""
    mt = mf = None
    if mask_time_prob > 0:
        mt = compute_mask_indices((batch, frames), mask_time_prob, mask_time_length, frame_lengths,
                                  mask_time_min_masks, rng)
    if mask_feature_prob > 0:
        mf = compute_mask_indices((batch, hidden), mask_feature_prob, mask_feature_length, None,
                                  mask_feature_min_masks, rng)
    return mt, mf

"""Host -> device input pipeline for raw 16 kHz PCM.

The reference featurises every example on the host inside `datasets.map` workers
(`processor(audio)` at R/src/coral/data.py:747: NumPy zero-mean/unit-variance for wav2vec2, a CPU
STFT + mel for Whisper) and pads again in the collator (R/src/coral/data_collators.py:72-77).  Here
the host only packs raw samples (int16 or fp32) into a pinned staging buffer; the copy to the GPU runs
on a side stream, double-buffered, and normalisation / padding / attention mask (`ca_pcm_prepare`) or
the log-mel front end (`ca_logmel`) run on the device.  Batch k+1 is staged and copied while the step
of batch k computes.
"""

from __future__ import annotations

import numpy as np
import torch

from . import ops


class DeviceInputPipeline:
    """`submit(list_of_pcm)` stages a batch, `get()` returns the device tensors of the oldest staged batch.

    kind="wav2vec2": -> {"input_values": f32 [B, N], "attention_mask": i32 [B, N]} with N = longest utterance
    (padding="longest") or `max_samples` (padding="max_length"), as `DataCollatorCTCWithPadding` yields.
    kind="whisper":  -> {"input_features": f32 [B, mels, 3000]} (30 s pad/trim + log-mel on the GPU)."""

    def __init__(self, device, batch: int, max_samples: int, kind: str = "wav2vec2", dtype=np.int16, depth: int = 2,
                 padding: str = "longest", peak_normalize: bool = False, mel_filters: torch.Tensor | None = None,
                 augment=None):
        if not torch.cuda.is_available():
            raise ops.CoralAmdError("DeviceInputPipeline needs a GPU")
        if kind not in ("wav2vec2", "whisper"):
            raise ValueError(kind)
        if kind == "whisper" and mel_filters is None:
            raise ValueError("the whisper pipeline needs the engine's mel filter bank")
        ops.lib()
        self.device = torch.device(device)
        self.kind, self.padding, self.peak = kind, padding, peak_normalize
        self.B, self.N, self.depth = batch, max_samples, depth
        tdt = torch.int16 if np.dtype(dtype) == np.int16 else torch.float32
        self.np_dtype = np.int16 if tdt == torch.int16 else np.float32
        self.host = [torch.zeros(batch, max_samples, dtype=tdt).pin_memory() for _ in range(depth)]
        self.host_len = [torch.zeros(batch, dtype=torch.int32).pin_memory() for _ in range(depth)]
        # Device slots are only ever written by the side-stream copies of submit() and read up to dev_len: no fill
        # (a zero-fill would be enqueued on the *current* stream and could land after the first side-stream copy).
        self.dev = [torch.empty(batch, max_samples, dtype=tdt, device=self.device) for _ in range(depth)]
        self.dev_len = [torch.empty(batch, dtype=torch.int32, device=self.device) for _ in range(depth)]
        self.copy_stream = torch.cuda.Stream(device=self.device)
        # whatever the allocating stream still has queued on this memory (a previous owner's kernels) comes first
        self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))
        self.ready = [torch.cuda.Event() for _ in range(depth)]
        self.free = [torch.cuda.Event() for _ in range(depth)]  # the compute stream is done with slot i
        self.mel_filters = mel_filters
        self.augment = augment  # coral_amd.augment.DeviceAugment or None (training only, like `augment_audio`)
        self._queue: list[tuple] = []  # (slot, rows, longest, host lengths)
        self._next = 0

    def submit(self, audios) -> None:
        """Pack one batch of 1-D PCM arrays (ragged) into the next staging slot and start its H2D copy."""
        if len(audios) > self.B:
            raise ValueError(f"batch of {len(audios)} exceeds the pipeline's {self.B}")
        if len(self._queue) >= self.depth:
            raise RuntimeError("all staging slots are in flight: call get() first")
        k = self._next
        self._next = (k + 1) % self.depth
        self.free[k].synchronize()  # the device side of this slot is no longer being read
        h, hl = self.host[k].numpy(), self.host_len[k].numpy()
        longest = 0
        for i, a in enumerate(audios):
            a = np.asarray(a)
            if a.dtype != self.np_dtype:
                a = a.astype(self.np_dtype)
            n = min(len(a), self.N)
            h[i, :n] = a[:n]
            hl[i] = n
            longest = max(longest, n)
        with torch.cuda.stream(self.copy_stream):
            self.dev[k][:len(audios), :longest].copy_(self.host[k][:len(audios), :longest], non_blocking=True)
            self.dev_len[k].copy_(self.host_len[k], non_blocking=True)
            self.ready[k].record(self.copy_stream)
        self._queue.append((k, len(audios), longest, [int(v) for v in hl[:len(audios)]]))

    def get(self) -> dict:
        """Device tensors of the oldest submitted batch (enqueued on the current stream)."""
        if not self._queue:
            raise RuntimeError("nothing submitted")
        k, rows, longest, host_lengths = self._queue.pop(0)
        cur = torch.cuda.current_stream()
        cur.wait_event(self.ready[k])
        src, src_ld = self.dev[k], self.N
        if self.augment is not None:
            # normalise -> augment -> featurise, the order of R/src/coral/data.py:708-747
            n_aug = longest if self.kind == "wav2vec2" else min(self.N, longest)
            raw = torch.empty(rows, n_aug, dtype=torch.float32, device=self.device)
            ops.pcm_prepare(self.dev[k], self.dev_len[k], raw, None, rows, n_aug, self.N, peak_normalize=True,
                            zero_mean_unit_var=False)
            src, src_ld = self.augment(raw, self.dev_len[k][:rows]), n_aug
        peak = self.peak and self.augment is None
        if self.kind == "wav2vec2":
            n_out = self.N if self.padding == "max_length" else longest
            y = torch.empty(rows, n_out, dtype=torch.float32, device=self.device)
            mask = torch.empty(rows, n_out, dtype=torch.int32, device=self.device)
            ops.pcm_prepare(src, self.dev_len[k], y, mask, rows, n_out, src_ld, peak_normalize=peak)
            # (sample_lengths: the valid samples per row as host integers, so that the model wrapper can place its
            # SpecAugment spans without reading the device mask back)
            out = {"input_values": y, "attention_mask": mask, "sample_lengths": host_lengths}
        else:
            from .whisper import HOP, N_SAMPLES

            wave = torch.empty(rows, N_SAMPLES, dtype=torch.float32, device=self.device)
            # pad / trim to 30 s; Whisper's extractor does no per-utterance normalisation
            ops.pcm_prepare(src, self.dev_len[k], wave, None, rows, N_SAMPLES, src_ld, peak_normalize=peak,
                            zero_mean_unit_var=False)
            mels = self.mel_filters.shape[0] if self.mel_filters.shape[0] in (80, 128) else self.mel_filters.shape[1]
            feats = torch.empty(rows, mels, N_SAMPLES // HOP, dtype=torch.float32, device=self.device)
            ws = torch.empty(ops.logmel_workspace_bytes(rows), dtype=torch.uint8, device=self.device)
            ops.logmel(wave, self.mel_filters, feats, ws, rows, N_SAMPLES, mels)
            out = {"input_features": feats}
        self.free[k].record(cur)
        return out

"""On-device waveform augmentation.

The reference composes, per example on the host (R/src/coral/data.py:708-738, torch_audiomentations):
    PeakNormalization(p=1) -> Gain(p=1) -> AddBackgroundNoise(p=0.7) -> AddColoredNoise(p=0.2)
    -> OneOf([BandPassFilter, BandStopFilter, HighPassFilter, LowPassFilter], p=0.2)
with that library's default parameter ranges.  torch_audiomentations / julius are not installed here, so
the ranges and filter designs below restate their documented defaults: gain U[-18, 6] dB; SNR U[3, 30] dB;
coloured noise with spectral decay f_decay U[-2, 2] (power ~ 1/f^decay); low-pass cutoff 150-7500 Hz,
high-pass 20-2400 Hz, band-pass/-stop centre 200-4000 Hz with bandwidth fraction 0.5-1.99, all sampled on
the mel scale; filters are julius-style windowed-sinc low-passes (8 zero crossings, Hann window, edges
replicated), a high-pass being x - lowpass(x).  The draws happen on the host with a seeded NumPy RNG (one
small H2D copy of the parameter arrays per batch); the arithmetic runs in coral_amd/csrc/augment.hip.
Being random, this stage has no parity target; tests check each operator against a NumPy restatement.
"""

from __future__ import annotations

import numpy as np
import torch

from . import ops

MAX_TAPS = 6401  # high-pass at 20 Hz / 16 kHz: half = int(8 / (20/16000) / 2) = 3200


def lowpass_taps(cutoff_hz: float, sample_rate: int, zeros: int = 8) -> np.ndarray:
    """julius.lowpass_filter design: Hann-windowed sinc, normalised to unit DC gain."""
    fc = cutoff_hz / sample_rate
    half = min(int(zeros / fc / 2), (MAX_TAPS - 1) // 2)
    t = np.arange(-half, half + 1, dtype=np.float64)
    window = 0.5 * (1.0 + np.cos(np.pi * t / (half + 1))) if half > 0 else np.ones(1)
    h = 2 * fc * window * np.sinc(2 * fc * t)
    return (h / h.sum()).astype(np.float32)


def coloured_taps(f_decay: float, sample_rate: int, n_taps: int = 1025) -> np.ndarray:
    """Linear-phase FIR whose magnitude follows 1 / f^(f_decay/2) (power ~ 1/f^f_decay) from 1 Hz-equivalent to
    Nyquist, the shaping AddColoredNoise applies to white noise in the frequency domain."""
    nfft = 4096
    mask = 1.0 / (np.linspace(1.0, np.sqrt(sample_rate / 2), nfft // 2 + 1) ** f_decay)
    h = np.fft.irfft(mask, nfft)
    h = np.roll(h, n_taps // 2)[:n_taps] * np.hanning(n_taps)
    return (h / np.sqrt((h ** 2).sum())).astype(np.float32)


def _mel(f):
    return 2595.0 * np.log10(1.0 + f / 700.0)


def _imel(m):
    return 700.0 * (10.0 ** (m / 2595.0) - 1.0)


class DeviceAugment:
    """`aug(x, lengths)` -> augmented fp32 [B, N] on the device (new tensor; x is not modified)."""

    def __init__(self, device, sample_rate: int = 16_000, seed: int = 4242, background_noises: list | None = None,
                 p_background: float = 0.7, p_coloured: float = 0.2, p_filter: float = 0.2):
        ops.lib()
        self.device = torch.device(device)
        self.sr = sample_rate
        self.rng = np.random.RandomState(seed)
        self.p_background, self.p_coloured, self.p_filter = p_background, p_coloured, p_filter
        self.noise_bank = None
        if background_noises:
            bank = np.concatenate([np.asarray(a, dtype=np.float32) for a in background_noises])
            self.noise_bank = torch.from_numpy(bank).to(self.device)
        self._counter = 0
        from .staging import PinnedStager

        self._stager = PinnedStager(self.device, depth=4)
        self._ncall = 0

    def _dev(self, a, dtype):
        """Host-drawn parameters -> device through pinned staging and a non-blocking copy (staging.py): a pageable
        `.to(device)` here is a synchronous copy on the compute stream - the host would wait for the GPU to drain the
        previous step before it could enqueue anything else (measured on the finetune path: 81.1 vs 76.4 ms/step)."""
        t = torch.as_tensor(np.asarray(a))
        self._ncall += 1
        return self._stager.to_device(t, dtype, f"aug{self._ncall}")

    def __call__(self, x: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
        B, N = x.shape
        rng, sr = self.rng, self.sr
        self._ncall = 0  # (staging keys = position of the transfer inside one call: the same ring every batch)
        cur = x
        # Gain(min_gain_in_db=-18, max_gain_in_db=6, p=1)
        gain = 10.0 ** (rng.uniform(-18.0, 6.0, size=B) / 20.0)
        nxt = torch.empty_like(x)
        ops.wave_scale(cur, self._dev(gain, torch.float32), nxt, B, N)
        cur = nxt
        # AddBackgroundNoise(p=0.7, SNR 3..30 dB): needs a noise bank (the reference downloads one)
        if self.noise_bank is not None:
            act = (rng.rand(B) < self.p_background).astype(np.int32)
            snr = rng.uniform(3.0, 30.0, size=B).astype(np.float32)
            off = rng.randint(0, self.noise_bank.numel(), size=B).astype(np.int64)
            nxt = torch.empty_like(x)
            ops.mix_noise(cur, lengths, self.noise_bank, 0, self.noise_bank.numel(), self._dev(off, torch.int64),
                          self._dev(snr, torch.float32), self._dev(act, torch.int32), nxt, B, N)
            cur = nxt
        # AddColoredNoise(p=0.2, SNR 3..30 dB, f_decay -2..2)
        act = (rng.rand(B) < self.p_coloured).astype(np.int32)
        if act.any():
            snr = rng.uniform(3.0, 30.0, size=B).astype(np.float32)
            decay = rng.uniform(-2.0, 2.0, size=B)
            white = torch.empty(B, N, dtype=torch.float32, device=self.device)
            self._counter += 1
            ops.white_noise(white, B * N, (int(rng.randint(0, 2 ** 31)) << 20) + self._counter)
            taps = np.zeros((B, 1025), dtype=np.float32)
            for b in range(B):
                taps[b] = coloured_taps(decay[b], sr)
            noise = torch.empty_like(white)
            ops.fir_filter(white, None, self._dev(taps, torch.float32), self._dev([1025] * B, torch.int32),
                           self._dev(act, torch.int32), noise, B, N, 1025)
            nxt = torch.empty_like(x)
            ops.mix_noise(cur, lengths, noise, N, N, None, self._dev(snr, torch.float32), self._dev(act, torch.int32),
                          nxt, B, N)
            cur = nxt
        # OneOf([BandPass, BandStop, HighPass, LowPass], p=0.2)
        act = rng.rand(B) < self.p_filter
        if act.any():
            kind = rng.randint(0, 4, size=B)
            lp = np.zeros((B, MAX_TAPS), dtype=np.float32)   # first stage: low-pass at the upper edge / the cutoff
            hp = np.zeros((B, MAX_TAPS), dtype=np.float32)   # second stage of the band filters: high-pass at the lower edge
            n1 = np.ones(B, dtype=np.int32)
            n2 = np.ones(B, dtype=np.int32)
            m1 = np.zeros(B, dtype=np.int32)
            m2 = np.zeros(B, dtype=np.int32)
            stop = np.zeros(B, dtype=bool)
            for b in range(B):
                if not act[b]:
                    continue
                if kind[b] in (0, 1):  # band-pass / band-stop: centre on the mel scale, bandwidth fraction 0.5..1.99
                    centre = _imel(rng.uniform(_mel(200.0), _mel(4000.0)))
                    bw = centre * rng.uniform(0.5, 1.99)
                    lo, hi = max(centre - bw / 2, 20.0), min(centre + bw / 2, sr / 2 - 100.0)
                    a, c = lowpass_taps(hi, sr), lowpass_taps(lo, sr)
                    lp[b, :len(a)], n1[b], m1[b] = a, len(a), 1
                    hp[b, :len(c)], n2[b], m2[b] = c, len(c), 2
                    stop[b] = kind[b] == 1
                elif kind[b] == 2:     # high-pass 20..2400 Hz
                    a = lowpass_taps(_imel(rng.uniform(_mel(20.0), _mel(2400.0))), sr)
                    lp[b, :len(a)], n1[b], m1[b] = a, len(a), 2
                else:                  # low-pass 150..7500 Hz
                    a = lowpass_taps(_imel(rng.uniform(_mel(150.0), _mel(7500.0))), sr)
                    lp[b, :len(a)], n1[b], m1[b] = a, len(a), 1
            mt1, mt2 = int(n1.max()) | 1, int(n2.max()) | 1
            s1 = torch.empty_like(x)
            ops.fir_filter(cur, lengths, self._dev(lp[:, :mt1], torch.float32), self._dev(n1, torch.int32),
                           self._dev(m1, torch.int32), s1, B, N, mt1)
            s2 = torch.empty_like(x)
            ops.fir_filter(s1, lengths, self._dev(hp[:, :mt2], torch.float32), self._dev(n2, torch.int32),
                           self._dev(m2, torch.int32), s2, B, N, mt2)
            if stop.any():  # band-stop = x - band-pass(x)
                sel = self._dev(stop, torch.bool)[:, None]
                s2 = torch.where(sel, cur - s2, s2)
            cur = s2
        return cur

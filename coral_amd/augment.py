"""On-device waveform augmentation.

The reference composes, per example on the host (R/src/coral/data.py:708-738, torch_audiomentations):
    PeakNormalization(p=1) -> Gain(p=1) -> AddBackgroundNoise(p=0.7) -> AddColoredNoise(p=0.2)
    -> OneOf([BandPassFilter, BandStopFilter, HighPassFilter, LowPassFilter], p=0.2)
with that library's default parameter ranges.  torch_audiomentations / julius are not installed here, so
the ranges and filter designs below restate their documented defaults: gain U[-18, 6] dB; SNR U[3, 30] dB;
coloured noise with spectral decay f_decay U[-2, 2] (power ~ 1/f^decay); low-pass cutoff 150-7500 Hz,
high-pass 20-2400 Hz, band-pass/-stop centre 200-4000 Hz with bandwidth fraction 0.5-1.99, all sampled on
the mel scale; filters are julius-style windowed-sinc low-passes (8 zero crossings, Hann window, edges
replicated), a high-pass being x - lowpass(x), a band-pass lowpass(high) - lowpass(low) at the lower edge's filter
length, a band-stop x - bandpass(x) (oracle/augment_ref.py restates the chain; parity unpinned: the libraries are absent).
The draws happen on the host with a seeded NumPy RNG (one
small H2D copy of the parameter arrays per batch); the arithmetic runs in coral_amd/csrc/augment.hip.
Being random, this stage has no golden output; `self.last` records what a call drew so that tests replay it through the
oracle.
"""

from __future__ import annotations

import numpy as np
import torch

from . import ops

MAX_TAPS = 6401  # high-pass at 20 Hz / 16 kHz: half = int(8 / (20/16000) / 2) = 3200


def lowpass_taps(cutoff_hz: float, sample_rate: int, zeros: int = 8, half: int | None = None) -> np.ndarray:
    """julius.LowPassFilters design: sinc under hann_window(2 half + 1, periodic=False), normalised to unit DC gain;
    half = int(zeros / (cutoff / sample_rate) / 2) unless given."""
    fc = cutoff_hz / sample_rate
    if half is None:
        half = int(zeros / fc / 2)
    half = min(half, (MAX_TAPS - 1) // 2)
    t = np.arange(-half, half + 1, dtype=np.float64)
    window = 0.5 * (1.0 + np.cos(np.pi * t / half)) if half > 0 else np.ones(1)
    h = 2 * fc * window * np.sinc(2 * fc * t)
    return (h / h.sum()).astype(np.float32)


def bandpass_taps(low_hz: float, high_hz: float, sample_rate: int, zeros: int = 8) -> np.ndarray:
    """julius.bandpass_filter as ONE filter: lowpass(high) - lowpass(low), both at the lower cutoff's size."""
    half = min(int(zeros / (low_hz / sample_rate) / 2), (MAX_TAPS - 1) // 2)
    return lowpass_taps(high_hz, sample_rate, zeros, half) - lowpass_taps(low_hz, sample_rate, zeros, half)


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
        # what was drawn for each example of this call (tests replay it through oracle/augment_ref.py)
        rec = [dict() for _ in range(B)]
        # Gain(min_gain_in_db=-18, max_gain_in_db=6, p=1)
        gain_db = rng.uniform(-18.0, 6.0, size=B)
        gain = 10.0 ** (gain_db / 20.0)
        for b in range(B):
            rec[b]["gain_db"] = float(gain_db[b])
        nxt = torch.empty_like(x)
        ops.wave_scale(cur, self._dev(gain, torch.float32), nxt, B, N)
        cur = nxt
        # AddBackgroundNoise(p=0.7, SNR 3..30 dB): needs a noise bank (the reference downloads one)
        if self.noise_bank is not None:
            act = (rng.rand(B) < self.p_background).astype(np.int32)
            snr = rng.uniform(3.0, 30.0, size=B).astype(np.float32)
            off = rng.randint(0, self.noise_bank.numel(), size=B).astype(np.int64)
            for b in range(B):
                if act[b]:
                    rec[b]["background"] = (int(off[b]), float(snr[b]))
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
            for b in range(B):
                if act[b]:
                    rec[b]["coloured"] = (float(snr[b]), float(decay[b]))
            self.last_white = white
            noise = torch.empty_like(white)
            ops.fir_filter(white, None, self._dev(taps, torch.float32), self._dev([1025] * B, torch.int32),
                           self._dev(act, torch.int32), noise, B, N, 1025)
            nxt = torch.empty_like(x)
            ops.mix_noise(cur, lengths, noise, N, N, None, self._dev(snr, torch.float32), self._dev(act, torch.int32),
                          nxt, B, N)
            cur = nxt
        # OneOf([BandPass, BandStop, HighPass, LowPass], p=0.2): one FIR pass, y (mode 1) or x - y (mode 2)
        act = rng.rand(B) < self.p_filter
        if act.any():
            kind = rng.randint(0, 4, size=B)
            taps = np.zeros((B, MAX_TAPS), dtype=np.float32)
            nt = np.ones(B, dtype=np.int32)
            mode = np.zeros(B, dtype=np.int32)
            for b in range(B):
                if not act[b]:
                    continue
                if kind[b] in (0, 1):  # band-pass / band-stop: centre on the mel scale, bandwidth fraction 0.5..1.99
                    centre = _imel(rng.uniform(_mel(200.0), _mel(4000.0)))
                    bw = centre * rng.uniform(0.5, 1.99)
                    # (julius would design a 64 001-tap filter for a 1-Hz lower edge: edges below 20 Hz are clamped)
                    lo, hi = max(centre - bw / 2, 20.0), centre + bw / 2
                    a = bandpass_taps(lo, hi, sr)
                    mode[b] = 2 if kind[b] == 1 else 1
                    rec[b]["filter"] = ("bandstop" if kind[b] == 1 else "bandpass", float(lo), float(hi))
                elif kind[b] == 2:     # high-pass 20..2400 Hz: x - lowpass(x)
                    fc = float(_imel(rng.uniform(_mel(20.0), _mel(2400.0))))
                    a, mode[b] = lowpass_taps(fc, sr), 2
                    rec[b]["filter"] = ("highpass", fc)
                else:                  # low-pass 150..7500 Hz
                    fc = float(_imel(rng.uniform(_mel(150.0), _mel(7500.0))))
                    a, mode[b] = lowpass_taps(fc, sr), 1
                    rec[b]["filter"] = ("lowpass", fc)
                taps[b, :len(a)], nt[b] = a, len(a)
            mt = int(nt.max()) | 1
            out = torch.empty_like(x)
            ops.fir_filter(cur, lengths, self._dev(taps[:, :mt], torch.float32), self._dev(nt, torch.int32),
                           self._dev(mode, torch.int32), out, B, N, mt)
            cur = out
        self.last = rec
        return cur

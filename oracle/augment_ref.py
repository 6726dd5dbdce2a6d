"""CPU restatement (NumPy, float64) of the waveform augmentation chain CoRal applies per training example
(R/src/coral/data.py:708-738).  TEST INFRASTRUCTURE: only tests/ may import this module; the product path is
coral_amd/augment.py + coral_amd/csrc/augment.hip.

PARITY UNPINNED.  The arithmetic lives in two third-party packages that are absent from this image and from
/root/reference: torch-audiomentations (pinned 0.12.0, R/uv.lock:3221-3232, `>=0.12.0` in R/pyproject.toml:32) and
julius (0.2.7, R/uv.lock:1253-1259), which torch-audiomentations calls for its filters.  The reference holds no golden
vectors for this stage (it is random per example), so nothing here could be checked against outputs of the real
libraries; each function restates the published algorithm of the class the reference instantiates, with that class's
default parameter ranges, and takes the DRAWN parameters as arguments so that a test can replay the draws
`DeviceAugment` made.

    ta.PeakNormalization(p=1.0)              R/src/coral/data.py:710,715   -> peak_normalize
    ta.Gain(p=1.0)                           :716                          -> gain            (U[-18, 6] dB)
    ta.AddBackgroundNoise(paths, p=0.7)      :717-719                      -> mix_at_snr      (SNR U[3, 30] dB)
    ta.AddColoredNoise(p=0.2)                :720                          -> colored_noise + mix_at_snr
                                                                              (SNR U[3, 30] dB, f_decay U[-2, 2])
    ta.OneOf([BandPass, BandStop, HighPass, LowPass], p=0.2)   :721-729    -> bandpass / bandstop / highpass / lowpass
"""
from __future__ import annotations

import numpy as np


def peak_normalize(x: np.ndarray) -> np.ndarray:
    """PeakNormalization(apply_to="all"): divide by max |x|; an all-zero example is left alone."""
    x = np.asarray(x, dtype=np.float64)
    peak = np.abs(x).max() if x.size else 0.0
    return x / peak if peak > 0 else x.copy()


def gain(x: np.ndarray, gain_db: float) -> np.ndarray:
    """Gain: x * 10^(dB / 20) with dB ~ U[min_gain_in_db = -18, max_gain_in_db = 6]."""
    return np.asarray(x, dtype=np.float64) * 10.0 ** (gain_db / 20.0)


def rms(x: np.ndarray) -> float:
    x = np.asarray(x, dtype=np.float64)
    return float(np.sqrt(np.mean(x * x))) if x.size else 0.0


def mix_at_snr(x: np.ndarray, noise: np.ndarray, snr_db: float) -> np.ndarray:
    """AddBackgroundNoise / AddColoredNoise: the noise is rescaled so that rms(x) / rms(scaled noise) = 10^(SNR/20) and
    added.  `noise` has the signal's length already (cropped or tiled by the caller)."""
    x = np.asarray(x, dtype=np.float64)
    noise = np.asarray(noise, dtype=np.float64)
    nr = rms(noise)
    if nr == 0.0:
        return x.copy()
    return x + noise * (rms(x) / (10.0 ** (snr_db / 20.0)) / nr)


def colored_noise(white: np.ndarray, f_decay: float, sample_rate: int) -> np.ndarray:
    """AddColoredNoise's generator applied to a given white sequence: the spectrum of the white noise is multiplied by
    1 / linspace(1, sqrt(sample_rate / 2), bins)^f_decay (power ~ 1/f^f_decay: -2 violet ... 0 white ... 2 brown),
    transformed back and normalised to unit RMS."""
    white = np.asarray(white, dtype=np.float64)
    spec = np.fft.rfft(white)
    mask = 1.0 / (np.linspace(1.0, np.sqrt(sample_rate / 2.0), spec.shape[0]) ** f_decay)
    out = np.fft.irfft(spec * mask, n=white.shape[0])
    r = rms(out)
    return out / r if r > 0 else out


def lowpass_design(cutoff: float, half_size: int | None = None, zeros: int = 8) -> np.ndarray:
    """julius.LowPassFilters: Hann-windowed sinc.  cutoff = f_c / sample_rate; half_size = int(zeros / cutoff / 2)
    unless given (a band filter designs both of its low-passes at the LOWER cutoff's size); window =
    hann(2 half + 1, periodic=False); h = 2 cutoff window sinc(2 cutoff t), normalised to unit sum."""
    if half_size is None:
        half_size = int(zeros / cutoff / 2)
    t = np.arange(-half_size, half_size + 1, dtype=np.float64)
    n = 2 * half_size + 1
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / (n - 1)) if n > 1 else np.ones(1)
    h = 2.0 * cutoff * window * np.sinc(2.0 * cutoff * t)
    return h / h.sum()


def fir_same(x: np.ndarray, h: np.ndarray) -> np.ndarray:
    """julius' `pad=True`: the input is padded by half_size on both sides with its edge samples (replicate), then
    correlated with the (symmetric) filter: the output has the input's length."""
    x = np.asarray(x, dtype=np.float64)
    half = (len(h) - 1) // 2
    if x.size == 0:
        return x.copy()
    xp = np.concatenate([np.full(half, x[0]), x, np.full(half, x[-1])])
    return np.convolve(xp, h[::-1], mode="valid")


def lowpass(x, cutoff_hz: float, sample_rate: int) -> np.ndarray:
    """LowPassFilter: julius.lowpass_filter(x, cutoff / sample_rate); cutoff ~ mel-uniform in [150, 7500] Hz."""
    return fir_same(x, lowpass_design(cutoff_hz / sample_rate))


def highpass(x, cutoff_hz: float, sample_rate: int) -> np.ndarray:
    """HighPassFilter: julius.highpass_filter = x - lowpass(x); cutoff ~ mel-uniform in [20, 2400] Hz."""
    x = np.asarray(x, dtype=np.float64)
    return x - lowpass(x, cutoff_hz, sample_rate)


def bandpass(x, low_hz: float, high_hz: float, sample_rate: int) -> np.ndarray:
    """BandPassFilter: julius.bandpass_filter(x, low / sr, high / sr) = lowpass_high(x) - lowpass_low(x), both designed
    with the half size of the LOWER cutoff.  low / high = centre -/+ bandwidth / 2, centre ~ mel-uniform in
    [200, 4000] Hz, bandwidth = centre x U[0.5, 1.99]."""
    half = int(8 / (low_hz / sample_rate) / 2)
    h = lowpass_design(high_hz / sample_rate, half) - lowpass_design(low_hz / sample_rate, half)
    return fir_same(x, h)


def bandstop(x, low_hz: float, high_hz: float, sample_rate: int) -> np.ndarray:
    """BandStopFilter: x - bandpass(x)."""
    x = np.asarray(x, dtype=np.float64)
    return x - bandpass(x, low_hz, high_hz, sample_rate)


def mel(f):
    """The scale the filter cutoffs / centres are drawn on (torch_audiomentations.utils.mel_scale: HTK form)."""
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def inverse_mel(m):
    return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)

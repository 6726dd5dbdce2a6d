"""oracle/augment_ref.py on the CPU: the identities its operators must satisfy by construction, and the product's
host-side filter designs (coral_amd/augment.py draws and designs on the host, the GPU only convolves) against it.
The oracle's parity is UNPINNED (torch-audiomentations / julius absent, no golden output in the reference): these
checks pin the restatement to the published definitions, not to the libraries' outputs."""
import numpy as np

from oracle import augment_ref as R

SR = 16000


def _x(n=4000, seed=0):
    return 0.3 * np.random.RandomState(seed).randn(n)


def test_filter_identities():
    x = _x()
    for fc in (150.0, 1000.0, 7500.0):
        h = R.lowpass_design(fc / SR)
        assert abs(h.sum() - 1.0) < 1e-12 and len(h) == 2 * int(8 / (fc / SR) / 2) + 1 and np.allclose(h, h[::-1])
        assert h[0] == 0.0 and h[-1] == 0.0  # hann_window(periodic=False) ends in zeros
        assert np.allclose(R.lowpass(x, fc, SR) + R.highpass(x, fc, SR), x, atol=1e-12)
    assert np.allclose(R.bandpass(x, 300, 2000, SR) + R.bandstop(x, 300, 2000, SR), x, atol=1e-12)
    # a constant passes a unit-DC-gain low-pass untouched (replicated edges), and is removed by the high-pass
    c = np.full(1000, 0.25)
    assert np.allclose(R.lowpass(c, 500, SR), c, atol=1e-12) and np.allclose(R.highpass(c, 500, SR), 0, atol=1e-12)
    # tones: 300 Hz survives the 1 kHz low-pass, 5 kHz does not; the band-pass keeps 1 kHz and drops both others
    t = np.arange(SR) / SR
    tone = lambda f: np.sin(2 * np.pi * f * t)  # noqa: E731
    amp = lambda y, f: np.abs(np.fft.rfft(y))[f] / (SR / 2)  # noqa: E731
    y = R.lowpass(tone(300) + tone(5000), 1000, SR)
    assert amp(y, 300) > 0.99 and amp(y, 5000) < 1e-3
    y = R.bandpass(tone(100) + tone(1000) + tone(6000), 500, 2000, SR)
    assert amp(y, 1000) > 0.98 and amp(y, 100) < 2e-2 and amp(y, 6000) < 1e-3


def test_gain_peak_and_snr_rules():
    x = _x()
    assert np.abs(R.peak_normalize(x)).max() == 1.0 and np.all(R.peak_normalize(np.zeros(5)) == 0)
    assert np.allclose(R.gain(x, -6.0), x * 10 ** (-0.3)) and np.allclose(R.gain(x, 0.0), x)
    noise = np.random.RandomState(1).randn(len(x))
    for snr in (3.0, 17.5, 30.0):
        y = R.mix_at_snr(x, noise, snr)
        assert abs(20 * np.log10(R.rms(x) / R.rms(y - x)) - snr) < 1e-9
    assert np.array_equal(R.mix_at_snr(x, np.zeros_like(x), 10.0), x)


def test_colored_noise_slopes():
    w = np.random.RandomState(2).randn(1 << 16)
    f = np.fft.rfftfreq(len(w), 1.0 / SR)
    lo, hi = (f >= 250) & (f < 500), (f >= 4000) & (f < 8000)
    prev = None
    for decay in (-2.0, 0.0, 2.0):
        y = R.colored_noise(w, decay, SR)
        assert abs(R.rms(y) - 1.0) < 1e-9
        p = np.abs(np.fft.rfft(y)) ** 2
        tilt = 10 * np.log10(p[hi].mean() / p[lo].mean())  # violet rises, white is flat, brown falls
        assert prev is None or tilt < prev - 3.0
        prev = tilt
    assert np.allclose(R.colored_noise(w, 0.0, SR), w / R.rms(w))
    assert abs(float(R.inverse_mel(R.mel(1234.5))) - 1234.5) < 1e-9


def test_product_filter_designs_equal_the_oracle():
    from coral_amd.augment import MAX_TAPS, bandpass_taps, lowpass_taps

    for fc in (20.0, 150.0, 999.0, 2400.0, 7500.0):
        a, b = lowpass_taps(fc, SR), R.lowpass_design(fc / SR)
        assert len(a) == len(b) <= MAX_TAPS and np.abs(a - b).max() < 1e-7
    for lo, hi in ((20.0, 400.0), (300.0, 2000.0), (1000.0, 7960.0)):
        half = int(8 / (lo / SR) / 2)
        want = R.lowpass_design(hi / SR, half) - R.lowpass_design(lo / SR, half)
        got = bandpass_taps(lo, hi, SR)
        assert len(got) == len(want) and np.abs(got - want).max() < 1e-7

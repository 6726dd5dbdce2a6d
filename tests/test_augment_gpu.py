"""On-device augmentation against oracle/augment_ref.py - the NumPy restatement of the
torch-audiomentations 0.12.0 / julius 0.2.7 classes CoRal composes (R/src/coral/data.py:708-738).  PARITY UNPINNED: those
libraries are absent here and the reference holds no golden output for this (random) stage, so the oracle restates their
published algorithms; the tests replay what `DeviceAugment` drew through it, operator by operator and as a whole chain."""
import numpy as np
import pytest
import torch

from oracle import augment_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SR = 16000


def _batch(seed=0, B=3, N=6000):
    rng = np.random.RandomState(seed)
    x = (0.3 * rng.randn(B, N)).astype(np.float32)
    lens = np.array([N, 4100, 257, 3000, 5999, 1200][:B], dtype=np.int32)
    for b in range(B):
        x[b, lens[b]:] = 0
    return x, lens


def _run_fir(x, lens, taps_list, modes):
    from coral_amd import ops

    B, N = x.shape
    mt = max(len(d) for d in taps_list) | 1
    taps = np.zeros((B, mt), dtype=np.float32)
    for b, d in enumerate(taps_list):
        taps[b, :len(d)] = d
    y = torch.empty(B, N, device=DEV)
    ops.fir_filter(torch.from_numpy(x).to(DEV), torch.from_numpy(lens).to(DEV), torch.from_numpy(taps).to(DEV),
                   torch.tensor([len(d) for d in taps_list], dtype=torch.int32, device=DEV),
                   torch.from_numpy(np.asarray(modes, dtype=np.int32)).to(DEV), y, B, N, mt)
    return y.cpu().numpy()


def test_filters_match_the_julius_restatement():
    """Low-pass, high-pass (x - lowpass), band-pass (lowpass_high - lowpass_low at the lower edge's length) and band-stop
    (x - bandpass) of ragged utterances, edges replicated, padding left at zero."""
    from coral_amd.augment import bandpass_taps, lowpass_taps

    x, lens = _batch(B=5)
    designs = [lowpass_taps(3000, SR), lowpass_taps(400, SR), bandpass_taps(300, 2500, SR), bandpass_taps(900, 1500, SR),
               lowpass_taps(1500, SR)]
    modes = [1, 2, 1, 2, 0]
    got = _run_fir(x, lens, designs, modes)
    want = [R.lowpass(x[0, :lens[0]], 3000, SR), R.highpass(x[1, :lens[1]], 400, SR),
            R.bandpass(x[2, :lens[2]], 300, 2500, SR), R.bandstop(x[3, :lens[3]], 900, 1500, SR), x[4, :lens[4]]]
    for b in range(5):
        assert np.abs(got[b, :lens[b]] - want[b]).max() <= 2e-5, b
        assert np.all(got[b, lens[b]:] == 0)


def test_mix_noise_is_the_snr_rule_of_the_restatement():
    from coral_amd import ops

    x, lens = _batch(1)
    B, N = x.shape
    noise = torch.empty(B, N, device=DEV)
    ops.white_noise(noise, B * N, 99)
    nz = noise.cpu().numpy()
    assert abs(nz.mean()) < 0.02 and abs(nz.std() - 1.0) < 0.02
    snr = np.array([3.0, 20.0, 10.0], dtype=np.float32)
    act = np.array([1, 1, 0], dtype=np.int32)
    y = torch.empty(B, N, device=DEV)
    ops.mix_noise(torch.from_numpy(x).to(DEV), torch.from_numpy(lens).to(DEV), noise, N, N, None,
                  torch.from_numpy(snr).to(DEV), torch.from_numpy(act).to(DEV), y, B, N)
    got = y.cpu().numpy()
    for b in range(B):
        n = lens[b]
        want = R.mix_at_snr(x[b, :n], nz[b, :n], float(snr[b])) if act[b] else x[b, :n]
        assert np.abs(got[b, :n] - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
        if act[b]:
            added = got[b, :n] - x[b, :n]
            assert abs(20 * np.log10(R.rms(x[b, :n]) / R.rms(added)) - snr[b]) <= 1e-2
        assert np.all(got[b, n:] == 0)


def test_coloured_noise_follows_the_spectral_decay_of_the_restatement():
    """AddColoredNoise shapes white noise by 1 / f^(decay / 2) in the frequency domain; the device applies a 1025-tap FIR
    of the same magnitude response to its own white noise: same noise sequence through both, the power spectra agree
    band by band (octave bands above the FIR's resolution) for violet ... brown."""
    from coral_amd import ops
    from coral_amd.augment import coloured_taps

    N = 1 << 16
    white = torch.empty(1, N, device=DEV)
    ops.white_noise(white, N, 1234)
    w = white.cpu().numpy()[0].astype(np.float64)
    for decay in (-2.0, -1.0, 0.0, 1.0, 2.0):
        taps = coloured_taps(decay, SR)
        y = torch.empty(1, N, device=DEV)
        ops.fir_filter(white, None, torch.from_numpy(taps[None]).to(DEV), torch.tensor([len(taps)], dtype=torch.int32, device=DEV),
                       None, y, 1, N, len(taps))
        got = y.cpu().numpy()[0].astype(np.float64)
        want = R.colored_noise(w, decay, SR)
        # (Hann analysis window: the FIR output is not periodic over N, and brown noise leaks from its lowest bins)
        win = np.hanning(N)
        pg, pw = np.abs(np.fft.rfft(win * got / R.rms(got))) ** 2, np.abs(np.fft.rfft(win * want)) ** 2
        f = np.fft.rfftfreq(N, 1.0 / SR)
        for lo in (62, 125, 250, 500, 1000, 2000, 4000):
            band = (f >= lo) & (f < 2 * lo)
            ratio_db = 10 * np.log10(pg[band].sum() / pw[band].sum())
            assert abs(ratio_db) <= 0.5, (decay, lo, ratio_db)  # measured <= 0.2 dB


def _replay(rec, x, n, bank, white, aug_mod):
    """One utterance through the oracle with the parameters DeviceAugment drew."""
    y = R.gain(x[:n], rec["gain_db"])
    if "background" in rec:
        off, snr = rec["background"]
        noise = bank[(off + np.arange(n)) % len(bank)]
        y = R.mix_at_snr(y, noise, snr)
    if "coloured" in rec:
        snr, decay = rec["coloured"]
        # (the device's generator: its white noise through the FIR form of the spectral mask, see the test above)
        noise = R.fir_same(white, aug_mod.coloured_taps(decay, SR).astype(np.float64))[:n]
        y = R.mix_at_snr(y, noise, snr)
    if "filter" in rec:
        kind, *p = rec["filter"]
        y = getattr(R, kind)(y, *p, SR)
    return y


@pytest.mark.parametrize("seed", [7, 8, 9, 10])
def test_whole_chain_replayed_through_the_oracle(seed):
    """Gain -> background noise at SNR -> coloured noise at SNR -> one of the four filters, every stage switched on: the
    device output equals the oracle's replay of the recorded draws; same seed = same bits; padding stays zero."""
    from coral_amd import augment as A

    x, lens = _batch(2, B=6, N=20000)
    lens = np.minimum(lens * 3, 20000).astype(np.int32)
    for b in range(6):
        x[b, lens[b]:] = 0
    xd, ld = torch.from_numpy(x).to(DEV), torch.from_numpy(lens).to(DEV)
    bank = np.random.RandomState(5).randn(30000).astype(np.float32)
    aug = A.DeviceAugment(DEV, seed=seed, background_noises=[bank], p_background=1.0, p_coloured=1.0, p_filter=1.0)
    a = aug(xd, ld)
    white = aug.last_white.cpu().numpy().astype(np.float64)
    b2 = A.DeviceAugment(DEV, seed=seed, background_noises=[bank], p_background=1.0, p_coloured=1.0, p_filter=1.0)(xd, ld)
    torch.cuda.synchronize()
    assert torch.equal(a, b2) and torch.isfinite(a).all()
    got = a.cpu().numpy()
    kinds = set()
    for b in range(6):
        n = int(lens[b])
        want = _replay(aug.last[b], x[b].astype(np.float64), n, bank.astype(np.float64), white[b], A)
        kinds.add(aug.last[b]["filter"][0])
        assert np.abs(got[b, :n] - want).max() <= 5e-5 * max(1.0, np.abs(want).max()), (b, aug.last[b])
        assert np.all(got[b, n:] == 0)
    assert kinds  # (over the four seeds all four filter kinds occur; each case above names the one it ran)


def test_gain_only_and_the_default_probabilities():
    from coral_amd.augment import DeviceAugment

    x, lens = _batch(2, B=3, N=20000)
    xd, ld = torch.from_numpy(x).to(DEV), torch.from_numpy(lens).to(DEV)
    aug = DeviceAugment(DEV, seed=3, p_background=0.0, p_coloured=0.0, p_filter=0.0)
    g = aug(xd, ld).cpu().numpy()
    for b in range(3):
        assert -18.0 <= aug.last[b]["gain_db"] <= 6.0 and set(aug.last[b]) == {"gain_db"}
        assert np.abs(g[b, :lens[b]] - R.gain(x[b, :lens[b]], aug.last[b]["gain_db"])).max() <= 1e-6
    # the reference's probabilities (R/src/coral/data.py:716-729): p = 0.7 / 0.2 / 0.2 over many examples
    aug = DeviceAugment(DEV, seed=11, background_noises=[np.ones(1000, dtype=np.float32)])
    n_bg = n_col = n_f = 0
    for _ in range(40):
        aug(xd, ld)
        n_bg += sum("background" in r for r in aug.last)
        n_col += sum("coloured" in r for r in aug.last)
        n_f += sum("filter" in r for r in aug.last)
    assert 0.55 <= n_bg / 120 <= 0.85 and 0.08 <= n_col / 120 <= 0.35 and 0.08 <= n_f / 120 <= 0.35


def test_low_pass_removes_an_out_of_band_tone():
    from coral_amd.augment import lowpass_taps

    N = 16000
    t = np.arange(N) / 16000.0
    x = (np.sin(2 * np.pi * 300 * t) + np.sin(2 * np.pi * 5000 * t)).astype(np.float32)[None]
    y = _run_fir(x, np.array([N], dtype=np.int32), [lowpass_taps(1000, SR)], [1])
    spec = np.abs(np.fft.rfft(y[0]))
    assert spec[5000] < 1e-3 * spec[300]

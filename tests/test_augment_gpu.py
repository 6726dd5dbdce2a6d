"""On-device augmentation operators against NumPy restatements.  The stage is random in
the reference (torch_audiomentations draws per example), so there is no golden output: each operator is
checked for its defining property and the composed chain for determinism and validity."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batch(seed=0, B=3, N=6000):
    rng = np.random.RandomState(seed)
    x = (0.3 * rng.randn(B, N)).astype(np.float32)
    lens = np.array([N, 4100, 257], dtype=np.int32)[:B]
    for b in range(B):
        x[b, lens[b]:] = 0
    return x, lens


def _fir_ref(x, n, taps, mode):
    half = len(taps) // 2
    idx = np.clip(np.arange(-half, n + half), 0, n - 1)
    y = np.convolve(x[:n][idx].astype(np.float64), taps[::-1].astype(np.float64), mode="valid")
    out = np.zeros_like(x)
    out[:n] = (x[:n] - y) if mode == 2 else y
    return out


def test_fir_filter_matches_numpy_with_replicated_edges():
    from coral_amd import ops
    from coral_amd.augment import lowpass_taps

    x, lens = _batch()
    B, N = x.shape
    designs = [lowpass_taps(3000, 16000), lowpass_taps(400, 16000), lowpass_taps(1500, 16000)]
    modes = np.array([1, 2, 0], dtype=np.int32)
    mt = max(len(d) for d in designs)
    taps = np.zeros((B, mt), dtype=np.float32)
    for b, d in enumerate(designs):
        taps[b, :len(d)] = d
    y = torch.empty(B, N, device=DEV)
    ops.fir_filter(torch.from_numpy(x).to(DEV), torch.from_numpy(lens).to(DEV), torch.from_numpy(taps).to(DEV),
                   torch.tensor([len(d) for d in designs], dtype=torch.int32, device=DEV),
                   torch.from_numpy(modes).to(DEV), y, B, N, mt)
    got = y.cpu().numpy()
    for b in range(B):
        want = x[b] if modes[b] == 0 else _fir_ref(x[b], lens[b], designs[b], modes[b])
        assert np.abs(got[b] - want).max() <= 2e-5, b
        assert np.all(got[b, lens[b]:] == 0)


def test_mix_noise_hits_the_requested_snr_and_respects_active():
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
        added = got[b, :n] - x[b, :n]
        if not act[b]:
            assert np.all(added == 0)
            continue
        measured = 20 * np.log10(np.sqrt((x[b, :n] ** 2).mean()) / np.sqrt((added ** 2).mean()))
        assert abs(measured - snr[b]) <= 1e-2
    assert np.all(got[0, lens[0]:] == 0) and np.all(got[1, lens[1]:] == 0)


def test_wave_scale_and_chain_is_deterministic_and_bounded():
    from coral_amd.augment import DeviceAugment

    x, lens = _batch(2, B=3, N=20000)
    xd, ld = torch.from_numpy(x).to(DEV), torch.from_numpy(lens).to(DEV)
    bank = [np.random.RandomState(5).randn(30000).astype(np.float32)]
    a = DeviceAugment(DEV, seed=7, background_noises=bank, p_background=1.0, p_coloured=1.0, p_filter=1.0)(xd, ld)
    b = DeviceAugment(DEV, seed=7, background_noises=bank, p_background=1.0, p_coloured=1.0, p_filter=1.0)(xd, ld)
    c = DeviceAugment(DEV, seed=8, background_noises=bank, p_background=1.0, p_coloured=1.0, p_filter=1.0)(xd, ld)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert torch.isfinite(a).all()
    for i in range(3):
        assert float(a[i, lens[i]:].abs().max()) == 0.0 if lens[i] < a.shape[1] else True
    # nothing switched on except the gain: output = input * 10^(g/20), g in [-18, 6] dB
    g = DeviceAugment(DEV, seed=3, p_background=0.0, p_coloured=0.0, p_filter=0.0)(xd, ld)
    ratio = (g[0, :100] / xd[0, :100]).cpu().numpy()
    assert np.allclose(ratio, ratio[0], rtol=1e-5) and 10 ** (-18 / 20) <= ratio[0] <= 10 ** (6 / 20)


def test_low_pass_removes_an_out_of_band_tone():
    from coral_amd import ops
    from coral_amd.augment import lowpass_taps

    N = 16000
    t = np.arange(N) / 16000.0
    x = (np.sin(2 * np.pi * 300 * t) + np.sin(2 * np.pi * 5000 * t)).astype(np.float32)[None]
    d = lowpass_taps(1000, 16000)
    y = torch.empty(1, N, device=DEV)
    ops.fir_filter(torch.from_numpy(x).to(DEV), None, torch.from_numpy(d[None]).to(DEV),
                   torch.tensor([len(d)], dtype=torch.int32, device=DEV), None, y, 1, N, len(d))
    spec = np.abs(np.fft.rfft(y.cpu().numpy()[0]))
    assert spec[5000] < 1e-3 * spec[300]

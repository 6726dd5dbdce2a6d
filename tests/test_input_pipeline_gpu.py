"""Raw-PCM input pipeline against the host featurisation it replaces:
`WaveformFeatureExtractor.__call__` + `.pad` (= $TF Wav2Vec2FeatureExtractor, pinned by
tests/golden/featext.npz on the CPU side) and pad/trim + log-mel for Whisper."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ragged(seed, dtype):
    rng = np.random.RandomState(seed)
    out = []
    for n in (16000, 9000, 23456, 400):
        x = np.clip(0.2 * rng.randn(n), -1, 1)
        out.append((x * 32767).astype(np.int16) if dtype == np.int16 else x.astype(np.float32))
    return out


@pytest.mark.parametrize("dtype", [np.int16, np.float32])
@pytest.mark.parametrize("padding", ["longest", "max_length"])
def test_wav2vec2_pipeline_matches_host_featurisation(dtype, padding):
    from coral_amd.input_pipeline import DeviceInputPipeline
    from coral_amd.processor import WaveformFeatureExtractor

    fe = WaveformFeatureExtractor()
    pipe = DeviceInputPipeline(DEV, batch=4, max_samples=32000, dtype=dtype, padding=padding)
    audios = _ragged(1, dtype)
    as_f32 = [a.astype(np.float32) / 32768.0 if dtype == np.int16 else a for a in audios]
    want = fe.pad([fe(a) for a in as_f32], padding=padding, max_length=32000)
    pipe.submit(audios)
    got = pipe.get()
    torch.cuda.synchronize()
    assert got["input_values"].shape == want["input_values"].shape
    assert np.array_equal(got["attention_mask"].cpu().numpy(), want["attention_mask"])
    assert np.abs(got["input_values"].cpu().numpy() - want["input_values"]).max() <= 3e-5


def test_peak_normalisation_and_double_buffering():
    from coral_amd.input_pipeline import DeviceInputPipeline
    from coral_amd.processor import WaveformFeatureExtractor

    fe = WaveformFeatureExtractor()
    pipe = DeviceInputPipeline(DEV, batch=4, max_samples=32000, dtype=np.float32, peak_normalize=True)
    b1, b2 = _ragged(2, np.float32), _ragged(3, np.float32)[::-1]
    pipe.submit(b1)
    pipe.submit(b2)
    with pytest.raises(RuntimeError):
        pipe.submit(b1)  # both staging slots are in flight
    for batch in (b1, b2):
        got = pipe.get()["input_values"].cpu().numpy()
        want = fe.pad([fe(a / np.abs(a).max()) for a in batch])["input_values"]
        assert np.abs(got - want).max() <= 3e-5
    pipe.submit(b1)  # slots are reusable
    assert pipe.get()["input_values"].shape[0] == 4


def test_whisper_pipeline_matches_processor():
    from coral_amd.input_pipeline import DeviceInputPipeline
    from coral_amd.whisper import N_SAMPLES, WhisperEngine, WhisperShape

    eng = WhisperEngine(WhisperShape(d_model=64, encoder_layers=1, decoder_layers=1, encoder_attention_heads=4,
                                     decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128), DEV)
    audios = _ragged(4, np.int16)
    pipe = DeviceInputPipeline(DEV, batch=4, max_samples=N_SAMPLES, kind="whisper", dtype=np.int16,
                               mel_filters=eng.mel_filters)
    pipe.submit(audios)
    got = pipe.get()["input_features"]
    host = np.zeros((4, N_SAMPLES), dtype=np.float32)
    for i, a in enumerate(audios):
        host[i, :len(a)] = a.astype(np.float32) / 32768.0
    want = eng.log_mel(torch.from_numpy(host))
    torch.cuda.synchronize()
    assert got.shape == want.shape == (4, 80, 3000)
    assert float((got - want).abs().max()) <= 1e-6


def test_pipeline_with_augmentation_keeps_shapes_masks_and_normalisation():
    from coral_amd.augment import DeviceAugment
    from coral_amd.input_pipeline import DeviceInputPipeline

    audios = _ragged(6, np.int16)
    pipe = DeviceInputPipeline(DEV, batch=4, max_samples=32000, dtype=np.int16,
                               augment=DeviceAugment(DEV, seed=1, p_coloured=1.0, p_filter=1.0))
    pipe.submit(audios)
    got = pipe.get()
    torch.cuda.synchronize()
    x, m = got["input_values"].cpu().numpy(), got["attention_mask"].cpu().numpy()
    assert x.shape == (4, 23456) and np.isfinite(x).all()
    for i, a in enumerate(audios):
        n = len(a)
        assert m[i].sum() == n and np.all(x[i, n:] == 0)
        assert abs(x[i, :n].mean()) < 1e-3 and abs(x[i, :n].var() - 1.0) < 1e-2  # still zero-mean / unit-variance

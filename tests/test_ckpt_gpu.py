"""Checkpoint interoperability with HuggingFace Transformers, engine side: model directories
written by `transformers` (tools/gen_goldens.py `hf_ckpt`: `save_pretrained` of a tiny Wav2Vec2ForCTC and
WhisperForConditionalGeneration, plus a head-less pretraining checkpoint in the legacy `pytorch_model.bin` form)
load through the engine's `from_pretrained` and reproduce the outputs HF computed from the same weights.  The other
direction — transformers loading what the engine saved — is tools/check_ckpt_with_hf.py + tests/test_ckpt_hf_side.py.
Reference behaviour: `load_saved` / `load_model` read such directories (R/src/coral/wav2vec2.py:104-133,253-305;
R/src/coral/whisper.py:67-109,234-267)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _w2v2_inputs(z):
    from oracle import wav2vec2_ref as ref

    g = torch.Generator().manual_seed(5)
    waves = [(0.1 * torch.randn(int(n), generator=g)).numpy() for n in z["lens"]]
    iv, am = ref.zero_mean_unit_var_norm(waves)
    return torch.from_numpy(iv), torch.from_numpy(am).long()


def test_hf_written_wav2vec2_ctc_directory_loads_and_matches(golden_dir):
    from coral_amd.modeling import Wav2Vec2ForCTC

    z = np.load(golden_dir / "hf_ckpt_w2v2.npz")
    model = Wav2Vec2ForCTC.from_pretrained(str(golden_dir / "hf_ckpt_w2v2")).eval()
    assert model.shape.conv_dim == (512, 32, 32, 32, 32, 32, 32) and model.shape.vocab_size == 46
    iv, am = _w2v2_inputs(z)
    out = model(iv, am, torch.from_numpy(z["labels"]))
    logits = out.logits.float().cpu().numpy()
    assert np.abs(logits - z["logits"]).max() <= 3e-2, np.abs(logits - z["logits"]).max()
    assert abs(float(out.loss) - float(z["loss"])) <= 5e-3 * float(z["loss"])


def test_headless_pretraining_checkpoint_gets_a_fresh_ctc_head(golden_dir, caplog):
    """What CoRal actually finetunes from: a pretrained base without `lm_head`, extra quantizer / project_* tensors,
    `weight_g` / `weight_v` names, `pytorch_model.bin` only; vocab_size comes from the tokenizer, not the checkpoint."""
    from coral_amd.modeling import Wav2Vec2ForCTC, load_checkpoint_tensors

    z = np.load(golden_dir / "hf_ckpt_w2v2_pretrain.npz")
    d = golden_dir / "hf_ckpt_w2v2_pretrain"
    sd = load_checkpoint_tensors(d)
    assert not any(k.startswith("lm_head") for k in sd) and any(k.endswith("weight_g") for k in sd)
    with caplog.at_level("WARNING"):
        model = Wav2Vec2ForCTC.from_pretrained(str(d), vocab_size=46, pad_token_id=45, seed=4242).eval()
    assert "lm_head.weight" in caplog.text and "newly initialised" in caplog.text
    eng = model.engine
    rep = eng.load_state_dict(sd, strict=False)
    assert sorted(rep["missing"]) == ["lm_head.bias", "lm_head.weight"]
    assert sorted(rep["unexpected"]) == sorted(z["extra_keys"].tolist())
    P = eng.state_dict()
    g = sd["wav2vec2.encoder.pos_conv_embed.conv.weight_g"].float()
    assert torch.equal(P["wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original0"].cpu(), g)
    assert P["lm_head.weight"].shape == (46, 64) and float(P["lm_head.weight"].std()) > 0
    # the encoder reproduces HF's hidden states: logits == last_hidden @ W^T + b with the engine's fresh head
    g5 = torch.Generator().manual_seed(5)
    from oracle import wav2vec2_ref as ref

    waves = [(0.1 * torch.randn(n, generator=g5)).numpy() for n in (4000, 3300)]
    iv, am = ref.zero_mean_unit_var_norm(waves)
    out = model(torch.from_numpy(iv), torch.from_numpy(am).long())
    want = torch.from_numpy(z["last_hidden"]) @ P["lm_head.weight"].cpu().T + P["lm_head.bias"].cpu()
    T = [eng.conv_lengths(n)[-1] for n in (4000, 3300)]
    got = out.logits.float().cpu()
    for b in range(2):
        assert (got[b, :T[b]] - want[b, :T[b]]).abs().max() <= 3e-2
    # strict loading still refuses it
    with pytest.raises(KeyError):
        eng.load_state_dict(sd)


def test_hf_written_whisper_directory_loads_and_matches(golden_dir):
    from coral_amd.whisper_setup import WhisperForConditionalGeneration

    z = np.load(golden_dir / "hf_ckpt_whisper.npz")
    model = WhisperForConditionalGeneration.from_pretrained(str(golden_dir / "hf_ckpt_whisper")).eval()
    g = torch.Generator().manual_seed(5)
    for n in (4000, 3300):
        torch.randn(n, generator=g)                      # the generator state the fixture script had reached
    feats = torch.randn(2, 80, 3000, generator=g) * 0.5
    out = model(feats, labels=torch.from_numpy(z["labels"]))
    logits = out["logits"].float().cpu().numpy()
    assert np.abs(logits - z["logits"]).max() <= 5e-2, np.abs(logits - z["logits"]).max()
    assert abs(float(out["loss"]) - float(z["loss"])) <= 1e-2 * float(z["loss"])
    sd = model.engine.state_dict()
    assert "proj_out.weight" not in sd or torch.equal(sd["proj_out.weight"], sd["model.decoder.embed_tokens.weight"])

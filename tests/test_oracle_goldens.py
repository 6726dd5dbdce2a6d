"""Pin the oracle (oracle/wav2vec2_ref.py) to vectors produced by the reference's own library
path (HuggingFace Transformers 5.15.0 CPU; tools/gen_goldens.py).  CPU only."""
import json

import numpy as np
import pytest
import torch

from oracle import wav2vec2_ref as ref


def _tiny_cfg():
    return ref.W2V2Config(hidden_size=128, num_hidden_layers=2, num_attention_heads=4,
                          intermediate_size=256)


def _regen_batch(lens, lab_lens, seed=4242):
    g = torch.Generator().manual_seed(seed)
    waves = []
    for n in lens:
        x = (0.1 * torch.randn(int(n), generator=g)).clamp(-1, 1)
        waves.append((x / x.abs().max()).numpy())
    return waves


def test_feature_extractor_matches_hf(golden_dir):
    z = np.load(golden_dir / "feature_extractor.npz")
    waves = [z[f"wave{i}"] for i in range(4)]
    iv, am = ref.zero_mean_unit_var_norm(waves)
    np.testing.assert_allclose(iv, z["input_values"], rtol=0, atol=2e-6)
    np.testing.assert_array_equal(am, z["attention_mask"])
    iv2, am2 = ref.zero_mean_unit_var_norm(waves, pad_to=2000)
    np.testing.assert_allclose(iv2, z["input_values_max"], rtol=0, atol=2e-6)
    np.testing.assert_array_equal(am2, z["attention_mask_max"])


def test_tokenizer_collapse_matches_hf(golden_dir):
    z = json.loads((golden_dir / "tokenizer_collapse.json").read_text())
    vocab = ref.coral_vocab()
    assert vocab == z["vocab"]
    assert vocab["<pad>"] == 45 and vocab["|"] == 36 and len(vocab) == 46
    for row, text in zip(z["rows"], z["texts"]):
        onehot = np.full((1, len(row), 46), -5.0, dtype=np.float32)
        onehot[0, np.arange(len(row)), row] = 5.0
        ids = ref.greedy_ctc_ids(onehot, blank=45)[0]
        assert ref.ids_to_text(ids, vocab) == text
    assert z["texts"][0] == "hej jj"


def test_ctc_matches_torch_known_answers(golden_dir):
    z = np.load(golden_dir / "ctc_cases.npz")
    n_inf = 0
    for i in range(int(z["n_cases"])):
        logits = torch.tensor(z[f"c{i}_logits"], requires_grad=True)
        tg = [int(c) for c in z[f"c{i}_targets"]]
        V = logits.shape[1]
        lp = torch.log_softmax(logits, -1)
        nll = ref.ctc_nll(lp, tg, int(z[f"c{i}_tin"]), blank=V - 1)
        want = float(z[f"c{i}_loss"])
        if torch.isinf(nll):
            n_inf += 1
            assert want == 0.0  # zero_infinity
            assert np.abs(z[f"c{i}_grad"]).max() == 0.0
            continue
        assert abs(float(nll) - want) <= 1e-4 * max(1.0, abs(want)), (i, float(nll), want)
        nll.backward()
        np.testing.assert_allclose(logits.grad.numpy(), z[f"c{i}_grad"], atol=2e-5, rtol=1e-4)
    assert n_inf >= 2  # the fixture holds infeasible cases
    g = torch.Generator().manual_seed(777)
    logits = torch.randn(499, 46, generator=g).requires_grad_(True)
    tg = torch.randint(0, 42, (120,), generator=g).tolist()
    nll = ref.ctc_nll(torch.log_softmax(logits, -1), tg, 499, blank=45)
    assert abs(float(nll) - float(z["big_loss"])) <= 1e-4 * float(z["big_loss"])
    nll.backward()
    np.testing.assert_allclose(logits.grad[::50].numpy(), z["big_grad_rows"], atol=2e-5, rtol=1e-3)


def test_w2v2_tiny_forward_backward_matches_hf(golden_dir):
    z = np.load(golden_dir / "w2v2_tiny.npz")
    cfg = _tiny_cfg()
    P = {k: v.clone().requires_grad_(True) for k, v in ref.synth_params(cfg).items()}
    waves = _regen_batch(z["lens"], None)
    iv, am = ref.zero_mean_unit_var_norm(waves)
    iv, am = torch.from_numpy(iv), torch.from_numpy(am).long()
    labels = torch.from_numpy(z["labels"])
    col = {}
    loss, logits, _ = ref.forward_loss(iv, am, labels, P, cfg, collect=col)
    np.testing.assert_allclose(col["conv6"].detach().numpy(), z["conv_feats"], atol=2e-5)
    np.testing.assert_allclose(col["posconv"].detach().numpy(), z["hs_first"], atol=2e-5)
    np.testing.assert_allclose(col["layer0"].detach().numpy(), z["hs_l0"], atol=5e-5)
    np.testing.assert_allclose(col["final"].detach().numpy(), z["hs_last"], atol=5e-5)
    np.testing.assert_allclose(logits.detach().numpy(), z["plain_logits"], atol=1e-5)
    assert abs(float(loss) - float(z["plain_loss"])) <= 1e-5 * float(z["plain_loss"])
    loss.backward()
    for key in z.files:
        if key.startswith("grad:"):
            g = P[key[5:]].grad.numpy()
            np.testing.assert_allclose(g, z[key], atol=1e-4 * max(1.0, np.abs(z[key]).max()), err_msg=key)
        elif key.startswith("gradnorm:"):
            g = P[key[9:]].grad
            assert abs(float(g.norm()) - float(z[key])) <= 1e-4 * float(z[key]), key
            np.testing.assert_allclose(g.reshape(-1)[:64].numpy(), z["gradhead:" + key[9:]],
                                       atol=1e-4 * max(1.0, float(np.abs(z["gradhead:" + key[9:]]).max())))
    # SpecAugment with injected time mask
    P2 = ref.synth_params(cfg)
    loss2, logits2, _ = ref.forward_loss(iv, am, labels, P2, cfg, mask_time=torch.from_numpy(z["mask_time"]))
    np.testing.assert_allclose(logits2.numpy(), z["specaug_logits"], atol=1e-5)
    assert abs(float(loss2) - float(z["specaug_loss"])) <= 1e-5 * float(z["specaug_loss"])

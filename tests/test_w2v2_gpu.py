"""End-to-end parity of the HIP Wav2Vec2ForCTC engine against the oracle (oracle/wav2vec2_ref.py,
itself pinned to HF Transformers goldens) on the same seeded inputs.  `-m gpu` only.

Tolerances (bf16 storage / fp32 accumulate vs the fp32 oracle):
  logits  : max-abs <= 6e-2 and cosine >= 0.999      (2e-2 abs / cos 0.999 at
            real scale; the synthetic weights here give logits of magnitude ~3)
  CTC loss: <= 1e-3 rel on identical logits (tests/test_kernels_gpu.py); whole-model loss, which
            also carries the bf16 forward error, <= 1e-2 rel
  gradients: cosine >= 0.99 and norm ratio within 5 % per parameter tensor
  greedy ids: bit-exact on the engine's own fp32 logits (argmax of fp32-accumulated logits).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batch(lens, lab_lens, seed=4242):
    from oracle import wav2vec2_ref as ref

    g = torch.Generator().manual_seed(seed)
    waves = []
    for n in lens:
        x = (0.1 * torch.randn(int(n), generator=g)).clamp(-1, 1)
        waves.append((x / x.abs().max()).numpy())
    Lmax = max(lab_lens)
    labels = torch.full((len(lens), Lmax), -100, dtype=torch.long)
    for b, L in enumerate(lab_lens):
        labels[b, :L] = torch.randint(0, 42, (L,), generator=g)
    iv, am = ref.zero_mean_unit_var_norm(waves)
    return torch.from_numpy(iv), torch.from_numpy(am).long(), labels


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def _run_case(cfg_kw, lens, lab_lens, mask_time=None, check_grads=True):
    from coral_amd.wav2vec2 import Wav2Vec2CTCEngine, Wav2Vec2Shape
    from oracle import wav2vec2_ref as ref

    cfg = ref.W2V2Config(**cfg_kw)
    P = ref.synth_params(cfg)
    iv, am, labels = _batch(lens, lab_lens)
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    kw = {} if mask_time is None else {"mask_time": mask_time}
    loss_ref, logits_ref, nll_ref = ref.forward_loss(iv, am, labels, Pr, cfg, **kw)
    loss_ref.backward()

    eng = Wav2Vec2CTCEngine(Wav2Vec2Shape(**cfg_kw), DEV)
    eng.load_state_dict(P)
    eng.zero_grad()
    out = eng(iv, am, labels, **kw)
    eng.backward()
    torch.cuda.synchronize()
    logits = out.logits.float().cpu()
    assert torch.isfinite(logits).all()
    assert (logits - logits_ref.detach()).abs().max() <= 6e-2, (logits - logits_ref.detach()).abs().max()
    assert _cos(logits, logits_ref.detach()) >= 0.999
    rel = abs(float(out.loss) - float(loss_ref)) / abs(float(loss_ref))
    assert rel <= 1e-2, (float(out.loss), float(loss_ref))
    # greedy ids: bit-exact vs the oracle's decode of the SAME fp32 logits
    ids, raw = eng.greedy_decode()
    want = ref.greedy_ctc_ids(logits.numpy(), cfg.pad_token_id)
    assert ids == want
    assert (raw.cpu().numpy() == logits.numpy().argmax(-1)).all()
    if check_grads:
        bad = []
        for name, g in eng.grad_dict().items():
            gr = Pr[name].grad
            if gr is None:  # parameter unused in this configuration (e.g. masked_spec_embed)
                assert float(g.abs().sum()) == 0.0, name
                continue
            if name.endswith("k_proj.bias"):
                # softmax is invariant to a per-query constant, so d/d(b_k) == 0 exactly; both
                # sides hold rounding noise only.  Bound it against the q-bias gradient instead.
                gq = Pr[name.replace("k_proj", "q_proj")].grad.norm()
                assert float(gr.norm()) <= 1e-3 * float(gq)
                assert float(g.norm()) <= 3e-2 * float(gq), (name, float(g.norm()), float(gq))
                continue
            c = _cos(g.cpu(), gr)
            ratio = float(g.norm().cpu() / (gr.norm() + 1e-30))
            if not (c >= 0.99 and 0.95 <= ratio <= 1.05):
                bad.append((name, round(c, 4), round(ratio, 4)))
        assert not bad, bad
    return eng


def test_tiny_ragged_forward_backward():
    _run_case(dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256),
              [4000, 3400, 2800], [5, 3, 4])


def test_tiny_specaugment_time_mask():
    T = 12
    mt = torch.zeros(3, T, dtype=torch.bool)
    mt[0, 2:5] = True
    mt[1, 0:2] = True
    mt[2, 7] = True
    _run_case(dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256),
              [4000, 3400, 2800], [5, 3, 4], mask_time=mt)


def test_headdim_120_and_80_shapes():
    """head_dim 120 (XLS-R-2B) and 80 (XLS-R-1B): d/16 groups of 120 / 80 channels."""
    _run_case(dict(hidden_size=1920, num_hidden_layers=1, num_attention_heads=16, intermediate_size=512),
              [16000, 12000], [12, 7])
    _run_case(dict(hidden_size=1280, num_hidden_layers=1, num_attention_heads=16, intermediate_size=512),
              [9000, 16000], [4, 10])


def test_golden_logits_from_hf(golden_dir):
    """The engine against the HF-generated fixture directly (not only through the oracle)."""
    from coral_amd.wav2vec2 import Wav2Vec2CTCEngine, Wav2Vec2Shape
    from oracle import wav2vec2_ref as ref

    z = np.load(golden_dir / "w2v2_tiny.npz")
    kw = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256)
    iv, am, _ = _batch(z["lens"], [5, 3, 4])
    eng = Wav2Vec2CTCEngine(Wav2Vec2Shape(**kw), DEV)
    eng.load_state_dict(ref.synth_params(ref.W2V2Config(**kw)))
    out = eng(iv, am, torch.from_numpy(z["labels"]))
    assert (out.logits.float().cpu().numpy() - z["plain_logits"]).__abs__().max() <= 6e-2
    assert abs(float(out.loss) - float(z["plain_loss"])) <= 1e-2 * float(z["plain_loss"])


def test_layerdrop_and_frozen_base():
    from coral_amd.wav2vec2 import Wav2Vec2CTCEngine, Wav2Vec2Shape
    from oracle import wav2vec2_ref as ref

    kw = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256)
    cfg = ref.W2V2Config(**kw)
    P = ref.synth_params(cfg)
    iv, am, labels = _batch([4000, 3000], [4, 3])
    # dropping layer 1 == a 1-layer model with the same weights
    cfg1 = ref.W2V2Config(**{**kw, "num_hidden_layers": 1})
    loss_ref, logits_ref, _ = ref.forward_loss(iv, am, labels, P, cfg1)
    eng = Wav2Vec2CTCEngine(Wav2Vec2Shape(**kw), DEV, freeze_base=True)
    eng.load_state_dict(P)
    eng.zero_grad()
    out = eng(iv, am, labels, layer_keep=[True, False])
    assert (out.logits.float().cpu() - logits_ref).abs().max() <= 6e-2
    eng.backward()
    g = eng.grad_dict()
    assert float(g["lm_head.weight"].abs().sum()) > 0
    assert float(g["wav2vec2.encoder.layers.0.attention.q_proj.weight"].abs().sum()) == 0


def test_activation_dropout_is_consistent_between_fwd_and_bwd():
    """With dropout on, loss(w + eps*dir) - loss(w) ~ eps * <grad, dir> for the same mask seed."""
    from coral_amd.wav2vec2 import Wav2Vec2CTCEngine, Wav2Vec2Shape
    from oracle import wav2vec2_ref as ref

    kw = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256)
    P = ref.synth_params(ref.W2V2Config(**kw))
    iv, am, labels = _batch([4000, 3000], [4, 3])
    eng = Wav2Vec2CTCEngine(Wav2Vec2Shape(**kw, activation_dropout=0.3), DEV).train()
    eng.load_state_dict(P)
    eng.step_seed = 11
    eng.zero_grad()
    l0 = float(eng(iv, am, labels).loss)
    eng.backward()
    name = "wav2vec2.encoder.layers.1.feed_forward.output_dense.bias"
    g = eng.grad_dict()[name].clone()
    l_eval = float(eng.eval()(iv, am, labels).loss)
    assert abs(l_eval - l0) > 1e-3  # dropout really changes the forward
    eng.train()
    P2 = {k: v.clone() for k, v in P.items()}
    direction = torch.sign(g.cpu())
    P2[name] = P2[name] + 0.05 * direction
    eng.load_state_dict(P2)
    l1 = float(eng(iv, am, labels).loss)
    pred = 0.05 * float((g.cpu() * direction).sum())
    assert abs((l1 - l0) - pred) <= 0.25 * abs(pred) + 0.05, (l1 - l0, pred)

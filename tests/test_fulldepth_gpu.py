"""Full-depth parity at BASELINE.json's own architectures, against the fp32 oracle run on the GPU box's host cores.

The oracle (plain torch fp32, pinned to HF fixtures in tests/test_oracle_*.py) is cheap enough for ONE utterance /
clip at the full shapes: XLS-R-2B forward + backward on a 10 s utterance is ~7 TFLOP and 8.6 GB of fp32
parameters (measured in the 8-core build container: parameters 26 s, forward 3.5 s), whisper-medium one 30 s clip
~1.5 TFLOP, whisper-large-v3-turbo two clips ~4.8 TFLOP.  So d = 1920 / head_dim 120 at 48 layers, 24 + 24 layers of
whisper-medium and the 32 + 4-layer turbo shape are compared with the reference arithmetic itself, not only with
themselves (tests/test_fullsize_gpu.py keeps the batch-of-8 property checks).

  configs[1]  wav2vec2-large (XLS-R-2B): logits / CTC loss / greedy ids / gradient norms + cosines
              ($TF/models/wav2vec2/modeling_wav2vec2.py:1667-1728)
  configs[3]  whisper-medium: encoder states, 8 teacher-forced logits rows, 16 greedy tokens under the tie margin
              ($TF/models/whisper/modeling_whisper.py:994-1099, generation_whisper.py:383); 16 clips (the reference's
              evaluation batch) decoded to max_length 225, every pick against the oracle's teacher-forced logits
  default     whisper-large (R/config/asr_finetuning.yaml:1-11: the reference's default model key), 32 + 32 layers: the
              same three comparisons on one clip, and its training step at the reference's batch of 64 (R/makefile:109-137)
  configs[4]  whisper-large-turbo, 32 + 4 layers, 2 clips: bf16 engine loss vs the oracle, and the fp8-forward step
              (`enable_fp8_forward`) vs the bf16 engine - fp8 has no reference oracle: its stated
              tolerance is against the build's own bf16 path

Stated tolerances (bf16 storage / fp32 accumulation vs fp32):
  logits max-abs <= 8e-2 at 48 layers (5e-2 at 24: tests/test_depth_gpu.py), cosine >= 0.999; CTC / CE loss <= 1e-3
  relative (the north-star bound); gradient norms within 5 %, cosine >= 0.97 on the sampled tensors; greedy ids
  bit-exact on the engine's own fp32 logits and equal to the oracle's argmax wherever the oracle's top-2 margin
  exceeds twice the measured logit error.  Measured values are printed (pytest -s) and recorded in DESIGN.md §2.
"""
import time

import numpy as np
import pytest
import torch

from greedy_check import check_greedy_rows

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def test_xlsr2b_one_utterance_forward_backward_against_the_oracle():
    from coral_amd.wav2vec2 import CORAL_W2V2_SHAPES, Wav2Vec2CTCEngine, Wav2Vec2Shape
    from oracle import wav2vec2_ref as ref

    t0 = time.time()
    cfg = ref.W2V2Config(**ref.CORAL_SHAPES["wav2vec2-large"])
    P = ref.synth_params(cfg)
    g = torch.Generator().manual_seed(4242)
    x = (0.1 * torch.randn(160_000, generator=g)).clamp(-1, 1)
    iv, am = ref.zero_mean_unit_var_norm([(x / x.abs().max()).numpy()])
    iv, am = torch.from_numpy(iv), torch.from_numpy(am).long()
    labels = torch.randint(0, 42, (1, 96), generator=g)

    eng = Wav2Vec2CTCEngine(Wav2Vec2Shape(**CORAL_W2V2_SHAPES["wav2vec2-large"]), DEV)
    eng.load_state_dict(P)
    eng.zero_grad()
    out = eng(iv, am, labels)
    eng.backward()
    torch.cuda.synchronize()
    logits = out.logits.float().cpu()
    t_eng = time.time() - t0

    names = ["lm_head.weight", "wav2vec2.encoder.layers.47.feed_forward.output_dense.weight",
             "wav2vec2.encoder.layers.24.attention.q_proj.weight", "wav2vec2.encoder.layers.24.attention.v_proj.bias",
             "wav2vec2.encoder.layers.0.feed_forward.intermediate_dense.weight",
             "wav2vec2.encoder.layers.0.layer_norm.weight", "wav2vec2.feature_projection.projection.weight",
             "wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1",
             "wav2vec2.feature_extractor.conv_layers.3.conv.weight", "wav2vec2.feature_extractor.conv_layers.0.conv.weight"]
    Pr = dict(P)
    for n in names:  # (gradients only where they are compared: the other 8.6 GB of fp32 gradients are not materialised)
        Pr[n] = P[n].clone().requires_grad_(True)
    t1 = time.time()
    loss_ref, logits_ref, _ = ref.forward_loss(iv, am, labels, Pr, cfg)
    loss_ref.backward()
    t_ref = time.time() - t1
    logits_ref = logits_ref.detach()

    err = float((logits - logits_ref).abs().max())
    cos = _cos(logits, logits_ref)
    rel = abs(float(out.loss) - float(loss_ref)) / abs(float(loss_ref))
    print(f"\nXLS-R-2B (48 L, d 1920, hd 120), 1 x 10 s: logits max-abs err {err:.4f} (mean |logit| "
          f"{float(logits_ref.abs().mean()):.3f}), cosine {cos:.6f}, CTC loss {float(out.loss):.4f} vs "
          f"{float(loss_ref):.4f} (rel {rel:.2e}); engine {t_eng:.0f} s incl. parameters, oracle fwd+bwd {t_ref:.0f} s")
    assert torch.isfinite(logits).all()
    assert err <= 8e-2, err
    assert cos >= 0.999, cos
    # The loss of ONE utterance through 48 layers is first-order sensitive to the time-mean of the logit error (a
    # per-class offset common to all frames, driven by the weights' own bf16 rounding): DESIGN.md 2 "why 1e-3 per
    # utterance is not a property of a bf16 path".  With ONLY the weights rounded to bf16 and every activation in fp32
    # the oracle itself moves by 2.7e-4 ... 1.3e-3 on these five utterances (tools/archive/dev_bf16_emulation.py), the
    # reference's own bf16 path (HF autocast) by 9e-6 ... 1.3e-3 (tests/golden/w2v2_cfg2_bf16_noise.npz).  Asserted:
    # 2e-3 per utterance here (3e-3 on the ragged ones), the north star's 1e-3 on the BATCH loss below, and a noise level
    # of the logits no higher than the reference's own bf16 path.
    assert rel <= 2e-3, rel
    from pathlib import Path

    hf = dict(np.load(Path(__file__).parent / "golden" / "w2v2_cfg2_bf16_noise.npz"))
    assert float(np.abs(hf["logits_fp32"] - logits_ref.numpy()).max()) <= 1e-4  # the fixture is this utterance, these weights
    rms = float((logits - logits_ref).double().norm() / logits_ref.double().norm())
    print(f"  logits relative RMS error {rms:.4e}; the reference's own bf16 path (HF autocast vs HF fp32): {float(hf['rel_logits']):.4e}, "
          f"its CTC loss {float(hf['loss_bf16']):.4f} (rel {abs(float(hf['loss_bf16']) - float(hf['loss_fp32'])) / float(hf['loss_fp32']):.2e})")
    assert rms <= 1.05 * float(hf["rel_logits"]), (rms, float(hf["rel_logits"]))
    ids, _ = eng.greedy_decode()
    assert ids == ref.greedy_ctc_ids(logits.numpy(), cfg.pad_token_id)  # bit-exact on the engine's fp32 logits
    top2 = logits_ref.topk(2, dim=-1).values
    decided = (top2[..., 0] - top2[..., 1]) > 2 * err
    assert (logits.argmax(-1)[decided] == logits_ref.argmax(-1)[decided]).all()
    print(f"  argmax equal on all {int(decided.sum())} of {decided.numel()} frames outside the tie margin (2 x {err:.3f})")
    gd = eng.grad_dict()
    for n in names:
        a, b = gd[n].float().cpu(), Pr[n].grad
        ratio, c = float(a.norm() / b.norm()), _cos(a, b)
        print(f"  grad {n}: norm ratio {ratio:.4f}, cosine {c:.5f}")
        assert 0.95 <= ratio <= 1.05 and c >= 0.97, (n, ratio, c)
    # The loss error is the second-order effect of the logits' bf16 noise (the oracle's CTC on the ENGINE's logits gives
    # the engine's loss to 7 digits, tools/archive/dev_depth_drift.py): over a batch it does not grow with the batch.  Four more
    # utterances (ragged, forward only): the summed loss of the five against the oracle.
    lens = [160_000, 131_200, 99_840, 147_520]
    waves = []
    for n in lens:
        w_ = (0.1 * torch.randn(n, generator=g)).clamp(-1, 1)
        waves.append((w_ / w_.abs().max()).numpy())
    iv4, am4 = ref.zero_mean_unit_var_norm(waves)
    iv4, am4 = torch.from_numpy(iv4), torch.from_numpy(am4).long()
    lab4 = torch.full((4, 90), -100, dtype=torch.int64)
    for b, L in enumerate((90, 70, 48, 81)):
        lab4[b, :L] = torch.randint(0, 42, (L,), generator=g)
    out4 = eng(iv4, am4, lab4)
    torch.cuda.synchronize()
    with torch.no_grad():
        loss4, logits4, nll4 = ref.forward_loss(iv4, am4, lab4, P, cfg)
    per = [abs(float(a) - float(b)) / float(b) for a, b in zip(out4["nll"].cpu(), nll4)]
    tot_e, tot_r = float(out.loss) + float(out4.loss), float(loss_ref) + float(loss4)
    rel5 = abs(tot_e - tot_r) / tot_r
    assert float(np.abs(hf["nll4_fp32"] - nll4.numpy()).max()) <= 2e-3  # (the fixture's four utterances are these four)
    hf_per = [abs(float(a) - float(b)) / float(b) for a, b in zip(hf["nll4_bf16"], hf["nll4_fp32"])]
    hf5 = abs(float(hf["loss_bf16"]) + float(hf["nll4_bf16"].sum()) - float(hf["loss_fp32"]) - float(hf["nll4_fp32"].sum())) / \
        (float(hf["loss_fp32"]) + float(hf["nll4_fp32"].sum()))
    print("  four more utterances (ragged): per-utterance CTC rel err " + ", ".join(f"{x:.2e}" for x in per) +
          f"; the five together {tot_e:.3f} vs {tot_r:.3f} (rel {rel5:.2e})\n  the reference's own bf16 path on the same four: " +
          ", ".join(f"{x:.2e}" for x in hf_per) + f"; its five together {hf5:.2e}")
    assert max(per) <= 3e-3
    assert rel5 <= 1e-3, rel5  # the batch loss - what a training step back-propagates - inside the north-star bound
    del eng
    torch.cuda.empty_cache()


def test_xlsr2b_bench_batch_ctc_loss_against_the_oracle():
    """configs[1] on configs[1]'s OWN batch: the 8 x 10 s synthetic batch `bench.py` times (same generator, same labels),
    forward only, against the fp32 oracle on the host cores: the BATCH CTC loss - what the step back-propagates - within
    the north star's 1e-3 (measured 6.3e-4); per utterance within 4e-3 (measured 4.0e-4 ... 3.6e-3 with mixed signs: the
    first-order effect of a time-constant per-class logit offset, NOTEBOOK 2 "why 1e-3 per utterance ...")."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import bench
    from coral_amd.wav2vec2 import CORAL_W2V2_SHAPES, Wav2Vec2CTCEngine, Wav2Vec2Shape
    from oracle import wav2vec2_ref as ref

    cfg = ref.W2V2Config(**ref.CORAL_SHAPES["wav2vec2-large"])
    P = ref.synth_params(cfg)
    batch, lens = bench.synth_batch(8, 10.0, 0, DEV)
    eng = Wav2Vec2CTCEngine(Wav2Vec2Shape(**CORAL_W2V2_SHAPES["wav2vec2-large"]), DEV)
    eng.load_state_dict(P)
    out = eng(batch["input_values"], batch["attention_mask"], batch["labels"])
    torch.cuda.synchronize()
    nll = out["nll"].float().cpu()
    loss = float(out.loss)
    logits = out.logits.float().cpu()
    # A/B (round 6, the review's question): the SAME last hidden state (bf16, as the engine holds it) through a final
    # LayerNorm + lm_head in fp32 with the fp32 master weights ($TF/models/wav2vec2/modeling_wav2vec2.py:791,1700,1717) -
    # does that remove the per-utterance CTC error?  (plain torch on the GPU: a measurement, not a product path)
    Bq, Tq, Vq = out.logits.shape
    hL = eng._saved["w"]["h"][cfg.num_hidden_layers].float().view(Bq, Tq, -1)
    ln32 = torch.nn.functional.layer_norm(hL, (hL.shape[-1],), P["wav2vec2.encoder.layer_norm.weight"].to(DEV).float(),
                                          P["wav2vec2.encoder.layer_norm.bias"].to(DEV).float(), cfg.layer_norm_eps)
    lg32 = ln32 @ P["lm_head.weight"].to(DEV).float().t() + P["lm_head.bias"].to(DEV).float()
    flen = eng._saved["flen"].long().cpu()
    lab_c = batch["labels"].long().cpu()
    tl = (lab_c >= 0).sum(-1)
    nll_head32 = torch.nn.functional.ctc_loss(lg32.log_softmax(-1).transpose(0, 1).cpu(), lab_c.clamp(min=0), flen, tl,
                                              blank=cfg.pad_token_id, reduction="none", zero_infinity=False)
    del eng, hL, ln32, lg32
    torch.cuda.empty_cache()
    t1 = time.time()
    iv, am, lab = batch["input_values"].float().cpu(), batch["attention_mask"].long().cpu(), batch["labels"].long().cpu()
    nll_ref, logit_rows = [], []
    with torch.no_grad():
        for b0 in range(0, 8, 2):  # (two utterances at a time: bounded host memory)
            _, lg, n_ = ref.forward_loss(iv[b0:b0 + 2], am[b0:b0 + 2], lab[b0:b0 + 2], P, cfg)
            nll_ref.append(n_)
            logit_rows.append(lg)
    nll_ref = torch.cat(nll_ref)
    logits_ref = torch.cat(logit_rows)
    t_ref = time.time() - t1
    per = [abs(float(a) - float(b)) / float(b) for a, b in zip(nll, nll_ref)]
    rel = abs(loss - float(nll_ref.sum())) / float(nll_ref.sum())
    err = float((logits - logits_ref).abs().max())
    print(f"\nXLS-R-2B on the bench batch (8 x 10 s): batch CTC loss {loss:.3f} vs {float(nll_ref.sum()):.3f} (rel {rel:.2e}); "
          "per utterance " + ", ".join(f"{x:.2e}" for x in per) + f"; logits max-abs err {err:.4f}, cosine "
          f"{_cos(logits, logits_ref):.6f}; oracle forward {t_ref:.0f} s")
    per32 = [abs(float(a) - float(b)) / float(b) for a, b in zip(nll_head32, nll_ref)]
    rel32 = abs(float(nll_head32.sum()) - float(nll_ref.sum())) / float(nll_ref.sum())
    print("  the same hidden state through an fp32 final LayerNorm + lm_head: per utterance " + ", ".join(f"{x:.2e}" for x in per32) +
          f"; batch {rel32:.2e}")
    assert rel <= 1e-3, rel
    assert max(per) <= 4e-3, per
    assert err <= 8e-2 and _cos(logits, logits_ref) >= 0.999


@pytest.mark.parametrize("model", ["wav2vec2-medium", "wav2vec2-small"])
def test_eight_utterance_batch_ctc_loss_against_the_oracle_at_the_other_model_keys(model):
    """The north star's 1e-3 on the BATCH CTC loss at CoRal's other two model keys - XLS-R-1B (48 layers, d 1280) and
    XLS-R-300M (24 layers, d 1024) - on the same kind of batch as configs[1]'s (8 x 10 s, `bench.synth_batch`): the bound is
    a property of eight utterances averaging their per-utterance offsets, not of the 2B shape (round 5's four-utterance
    XLS-R-1B batch sits at 1.3e-3).  Forward only, fp32 oracle on the host cores."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import bench
    from coral_amd.wav2vec2 import CORAL_W2V2_SHAPES, Wav2Vec2CTCEngine, Wav2Vec2Shape
    from oracle import wav2vec2_ref as ref

    cfg = ref.W2V2Config(**ref.CORAL_SHAPES[model])
    P = ref.synth_params(cfg)
    batch, lens = bench.synth_batch(8, 10.0, 0, DEV)
    eng = Wav2Vec2CTCEngine(Wav2Vec2Shape(**CORAL_W2V2_SHAPES[model]), DEV)
    eng.load_state_dict(P)
    out = eng(batch["input_values"], batch["attention_mask"], batch["labels"])
    torch.cuda.synchronize()
    nll, loss = out["nll"].float().cpu(), float(out.loss)
    del eng
    torch.cuda.empty_cache()
    iv, am, lab = batch["input_values"].float().cpu(), batch["attention_mask"].long().cpu(), batch["labels"].long().cpu()
    t1 = time.time()
    nll_ref = []
    with torch.no_grad():
        for b0 in range(0, 8, 2):
            nll_ref.append(ref.forward_loss(iv[b0:b0 + 2], am[b0:b0 + 2], lab[b0:b0 + 2], P, cfg)[2])
    nll_ref = torch.cat(nll_ref)
    per = [abs(float(a) - float(b)) / float(b) for a, b in zip(nll, nll_ref)]
    rel = abs(loss - float(nll_ref.sum())) / float(nll_ref.sum())
    print(f"\n{model} on 8 x 10 s: batch CTC loss {loss:.3f} vs {float(nll_ref.sum()):.3f} (rel {rel:.2e}); per utterance "
          + ", ".join(f"{x:.2e}" for x in per) + f"; oracle forward {time.time() - t1:.0f} s")
    assert rel <= 1e-3, rel
    assert max(per) <= 5e-3, per


def test_xlsr1b_batch_of_four_ctc_loss_against_the_oracle():
    """XLS-R-1B (CoRal's wav2vec2-medium) on a batch of four ragged utterances, forward only.  Measured (round 5): batch CTC
    loss 1.3e-3 off the oracle's, per utterance 4.0e-4, 4.1e-3, 9.0e-4, 1.2e-3 - four utterances do not average the
    per-utterance offsets out the way configs[1]'s eight do (6.3e-4 there); asserted: 2e-3 on the batch, 5e-3 per utterance."""
    from coral_amd.wav2vec2 import CORAL_W2V2_SHAPES, Wav2Vec2CTCEngine, Wav2Vec2Shape
    from oracle import wav2vec2_ref as ref

    cfg = ref.W2V2Config(**ref.CORAL_SHAPES["wav2vec2-medium"])
    P = ref.synth_params(cfg)
    g = torch.Generator().manual_seed(1281)
    lens = [160_000, 124_800, 96_000, 143_360]
    waves = []
    for n in lens:
        w_ = (0.1 * torch.randn(n, generator=g)).clamp(-1, 1)
        waves.append((w_ / w_.abs().max()).numpy())
    iv, am = ref.zero_mean_unit_var_norm(waves)
    iv, am = torch.from_numpy(iv), torch.from_numpy(am).long()
    lab = torch.full((4, 100), -100, dtype=torch.int64)
    for b, L in enumerate((100, 72, 51, 88)):
        lab[b, :L] = torch.randint(0, 42, (L,), generator=g)
    eng = Wav2Vec2CTCEngine(Wav2Vec2Shape(**CORAL_W2V2_SHAPES["wav2vec2-medium"]), DEV)
    eng.load_state_dict(P)
    out = eng(iv, am, lab)
    torch.cuda.synchronize()
    nll, loss = out["nll"].float().cpu(), float(out.loss)
    del eng
    torch.cuda.empty_cache()
    with torch.no_grad():
        loss_ref, _, nll_ref = ref.forward_loss(iv, am, lab, P, cfg)
    per = [abs(float(a) - float(b)) / float(b) for a, b in zip(nll, nll_ref)]
    rel = abs(loss - float(loss_ref)) / float(loss_ref)
    print(f"\nXLS-R-1B, 4 ragged utterances: batch CTC loss {loss:.3f} vs {float(loss_ref):.3f} (rel {rel:.2e}); per utterance "
          + ", ".join(f"{x:.2e}" for x in per))
    assert rel <= 2e-3, rel
    assert max(per) <= 5e-3, per


def test_xlsr1b_one_utterance_forward_backward_against_the_oracle():
    """CoRal's `model=wav2vec2-medium` (XLS-R-1B: 48 layers, d 1280, head_dim 80 - the third head size of the attention
    kernels, 16 heads x 80 padded to 128 lanes) at full depth on one ragged 7 s utterance: logits, CTC loss, greedy ids
    and a few gradients against the oracle."""
    from coral_amd.wav2vec2 import CORAL_W2V2_SHAPES, Wav2Vec2CTCEngine, Wav2Vec2Shape
    from oracle import wav2vec2_ref as ref

    cfg = ref.W2V2Config(**ref.CORAL_SHAPES["wav2vec2-medium"])
    P = ref.synth_params(cfg)
    g = torch.Generator().manual_seed(1280)
    x = (0.1 * torch.randn(112_000, generator=g)).clamp(-1, 1)
    iv, am = ref.zero_mean_unit_var_norm([(x / x.abs().max()).numpy()])
    iv, am = torch.from_numpy(iv), torch.from_numpy(am).long()
    labels = torch.randint(0, 42, (1, 64), generator=g)
    eng = Wav2Vec2CTCEngine(Wav2Vec2Shape(**CORAL_W2V2_SHAPES["wav2vec2-medium"]), DEV)
    eng.load_state_dict(P)
    eng.zero_grad()
    out = eng(iv, am, labels)
    eng.backward()
    torch.cuda.synchronize()
    logits = out.logits.float().cpu()
    names = ["lm_head.weight", "wav2vec2.encoder.layers.47.attention.out_proj.weight",
             "wav2vec2.encoder.layers.20.attention.k_proj.weight", "wav2vec2.encoder.layers.0.feed_forward.output_dense.weight",
             "wav2vec2.feature_projection.projection.weight"]
    Pr = dict(P)
    for n in names:
        Pr[n] = P[n].clone().requires_grad_(True)
    loss_ref, logits_ref, _ = ref.forward_loss(iv, am, labels, Pr, cfg)
    loss_ref.backward()
    logits_ref = logits_ref.detach()
    err = float((logits - logits_ref).abs().max())
    cos = _cos(logits, logits_ref)
    rel = abs(float(out.loss) - float(loss_ref)) / abs(float(loss_ref))
    print(f"\nXLS-R-1B (48 L, d 1280, hd 80), 1 x 7 s: logits max-abs err {err:.4f} (mean |logit| "
          f"{float(logits_ref.abs().mean()):.3f}), cosine {cos:.6f}, CTC loss rel {rel:.2e}")
    assert torch.isfinite(logits).all()
    assert err <= 8e-2 and cos >= 0.999, (err, cos)
    assert rel <= 2e-3, rel
    ids, _ = eng.greedy_decode()
    assert ids == ref.greedy_ctc_ids(logits.numpy(), cfg.pad_token_id)
    top2 = logits_ref.topk(2, dim=-1).values
    decided = (top2[..., 0] - top2[..., 1]) > 2 * err
    assert (logits.argmax(-1)[decided] == logits_ref.argmax(-1)[decided]).all()
    gd = eng.grad_dict()
    for n in names:
        a, b = gd[n].float().cpu(), Pr[n].grad
        ratio, c = float(a.norm() / b.norm()), _cos(a, b)
        print(f"  grad {n}: norm ratio {ratio:.4f}, cosine {c:.5f}")
        assert 0.95 <= ratio <= 1.05 and c >= 0.97, (n, ratio, c)
    del eng
    torch.cuda.empty_cache()


def _whisper_full_depth_one_clip(key: str, prefix: list, seed: int):
    """One 30 s clip at the full architecture of CoRal model key `key` against the oracle: encoder states, 8
    teacher-forced logits rows + CE loss, 16 greedy tokens under the tie-margin policy."""
    from coral_amd.whisper import CORAL_WHISPER_SHAPES, WhisperEngine, WhisperShape
    from oracle import whisper_ref as w

    kw = dict(CORAL_WHISPER_SHAPES[key])
    c = w.WhisperConfig(**kw)
    P = w.synth_params(c)
    g = torch.Generator().manual_seed(seed)
    wave = (0.1 * torch.randn(16_000 * 11, generator=g)).numpy()
    feats = torch.from_numpy(w.log_mel(w.pad_or_trim(wave), c.num_mel_bins))[None]
    labels = torch.randint(0, 50257, (1, 8), generator=g)
    eng = WhisperEngine(WhisperShape(**kw), DEV)
    eng.load_state_dict(P)
    out = eng.forward(feats, labels=labels)
    torch.cuda.synchronize()
    with torch.no_grad():
        enc_ref = w.encoder(feats, P, c)
        loss_ref, logits_ref = w.forward_loss(feats, labels, P, c)
    enc = out["encoder_last_hidden_state"].float().cpu()
    e_enc, c_enc = float((enc - enc_ref).abs().max()), _cos(enc, enc_ref)
    logits = out["logits"].float().cpu()
    e_log, c_log = float((logits - logits_ref).abs().max()), _cos(logits, logits_ref)
    rel = abs(float(out["loss"]) - float(loss_ref)) / abs(float(loss_ref))
    print(f"\n{key} ({c.encoder_layers} + {c.decoder_layers} L, d {c.d_model}, {c.num_mel_bins} mels), 1 x 30 s: encoder states "
          f"max-abs err {e_enc:.4f} (mean |x| {float(enc_ref.abs().mean()):.3f}, max |x| {float(enc_ref.abs().max()):.1f}), "
          f"cosine {c_enc:.6f}; 8 teacher-forced logits rows max-abs err {e_log:.4f}, cosine {c_log:.6f}; CE loss rel {rel:.2e}")
    assert c_enc >= 0.999 and e_enc <= 2.5e-1  # (the encoder output carries a few large-magnitude channels)
    assert e_log <= 8e-2 and c_log >= 0.999
    assert rel <= 1e-3, rel  # the north-star bound
    # 16 greedy tokens: CoRal's evaluation call (forced Danish transcribe prefix, begin-suppress set)
    bs = [220, c.eos_token_id]
    ids = eng.generate(feats, prefix, len(prefix) + 16, suppress_tokens=None, begin_suppress_tokens=bs)
    with torch.no_grad():
        want = w.greedy_generate(feats, P, c, prefix, len(prefix) + 16, suppress=None, begin_suppress=bs)

        def rows(b, seq):
            lg = w.decoder(torch.tensor([seq[:-1]]), enc_ref[b:b + 1], P, c)[0].clone()
            lg[len(prefix) - 1, bs] = float("-inf")
            return lg

        check_greedy_rows(rows, ids, want, len(prefix), accept=max(3e-2, 1.5 * e_log), forced=max(6e-2, 3 * e_log),
                          label=key)
    del eng
    torch.cuda.empty_cache()


def test_whisper_medium_full_depth_one_clip_against_the_oracle():
    _whisper_full_depth_one_clip("whisper-medium", [50258, 50285, 50359, 50363], seed=21)


def test_whisper_medium_sixteen_clips_decoded_to_max_length_against_the_oracle():
    """The reference's evaluation call at its own batch size (R/config/evaluation.yaml:20 `batch_size: 16`,
    R/src/coral/evaluate.py:56-60 -> $TF/models/whisper/generation_whisper.py:383) run to `max_length` 225: 16 different
    30 s clips through log-mel + encoder + the persistent one-launch-per-token decoder (csrc/decode.hip), the 221 picks
    of a row checked against the fp32 oracle's teacher-forced logits of the engine's OWN sequence (tests/greedy_check.py,
    policy 1 + 2: every pick within `accept` of the oracle's maximum; equal to the oracle's argmax wherever its top-2
    margin exceeds `forced` - so by induction the oracle's own greedy sequence can only leave the engine's at a
    near-tie; the oracle's greedy loop itself is O(T^2) decoder passes and is not run at this length).  The oracle costs
    12 s per row on the GPU box's host cores: rows 0, 5, 10, 15 by default, all 16 with CORAL_TEST_ALL_ROWS=1 (recorded
    run, NOTEBOOK R6.4: 3 536 picks, 13 beside the oracle's argmax, none at a margin above 0.031; logit error 0.054)."""
    import contextlib
    import io
    import os

    from coral_amd import ops
    from coral_amd.whisper import CORAL_WHISPER_SHAPES, WhisperEngine, WhisperShape
    from oracle import whisper_ref as w

    t0 = time.time()
    kw = dict(CORAL_WHISPER_SHAPES["whisper-medium"])
    c = w.WhisperConfig(**kw)
    P = w.synth_params(c)
    g = torch.Generator().manual_seed(77)
    B, max_length, prefix = 16, 225, [50258, 50285, 50359, 50363]
    feats = torch.from_numpy(np.stack([
        w.log_mel(w.pad_or_trim((0.05 + 0.01 * b) * torch.randn(16_000 * (5 + b), generator=g).numpy()), c.num_mel_bins)
        for b in range(B)]))
    eng = WhisperEngine(WhisperShape(**kw), DEV)
    eng.load_state_dict(P)
    assert ops.whisper_decode_token_supported(B, c.d_model, c.decoder_ffn_dim, c.decoder_attention_heads, c.vocab_size)
    bs = [220, c.eos_token_id]
    ids = eng.generate(feats, prefix, max_length, suppress_tokens=None, begin_suppress_tokens=bs)
    torch.cuda.synchronize()
    t1 = time.time()
    assert len(ids) == B and all(len(r) == max_length or c.eos_token_id in r for r in ids)
    n_full = sum(len(r) == max_length and c.eos_token_id not in r[:-1] for r in ids)
    worst = dict(regret=0.0, margin_at_mismatch=0.0, mismatches=0, picks=0, e_log=0.0)
    # rows that ended early carry pad after eos: the policy is checked up to and including the eos pick
    cut = [r[:r.index(c.eos_token_id) + 1] if c.eos_token_id in r else r for r in ids]
    with torch.no_grad():
        enc_eng = eng.encode(feats).clone()
        checked = list(range(B)) if os.environ.get("CORAL_TEST_ALL_ROWS") == "1" else list(range(0, B, 5))
        lg_ref = {}
        for b in checked:
            lg = w.decoder(torch.tensor([cut[b][:-1]]), w.encoder(feats[b:b + 1], P, c), P, c)[0].clone()
            lg[len(prefix) - 1, bs] = float("-inf")
            lg_ref[b] = lg
            # the engine's teacher-forced logits of the same sequence: the measured logit error sets the tie margins
            # (as in _whisper_full_depth_one_clip)
            mine = eng.decode(torch.tensor([cut[b][:-1]]), enc_eng[b:b + 1])[0].float().cpu()
            mine[len(prefix) - 1, bs] = float("-inf")
            keep = torch.isfinite(lg)
            worst["e_log"] = max(worst["e_log"], float((mine[keep] - lg[keep]).abs().max()))
        t2 = time.time()

        def rows(i, seq):
            lg = lg_ref[checked[i]]
            for t in range(len(prefix), len(seq)):
                top2 = lg[t - 1].topk(2)
                worst["picks"] += 1
                worst["regret"] = max(worst["regret"], float(top2.values[0] - lg[t - 1, seq[t]]))
                if int(top2.indices[0]) != seq[t]:
                    worst["mismatches"] += 1
                    worst["margin_at_mismatch"] = max(worst["margin_at_mismatch"], float(top2.values[0] - top2.values[1]))
            return lg

        e_log = worst["e_log"]
        assert e_log <= 8e-2, e_log
        mine = [cut[b] for b in checked]
        with contextlib.redirect_stdout(io.StringIO()):  # (no second sequence to report a divergence from)
            check_greedy_rows(rows, mine, mine, len(prefix), accept=max(3e-2, 1.5 * e_log), forced=max(6e-2, 3 * e_log),
                              label="whisper-medium x16")
    print(f"\nwhisper-medium, 16 x 30 s to max_length {max_length}: {n_full} rows ran to max_length; rows {checked} against the "
          f"oracle: teacher-forced logits max-abs err {e_log:.4f}; {worst['picks']} picks, "
          f"{worst['mismatches']} differ from the oracle's argmax (largest fp32 top-2 margin there {worst['margin_at_mismatch']:.4f}), "
          f"largest distance of a pick below the oracle's maximum {worst['regret']:.4f}; engine {t1 - t0:.1f} s incl. set-up, "
          f"oracle {t2 - t1:.1f} s")
    del eng
    torch.cuda.empty_cache()


def test_whisper_large_the_reference_default_model_full_depth_against_the_oracle():
    """`model=whisper-large` is the reference's DEFAULT (R/config/asr_finetuning.yaml:1-11; R/config/model/whisper-large.yaml:
    openai/whisper-large-v3): 32 + 32 layers, d 1280, 20 heads, 128 mel bins, vocabulary 51 866 - the one CoRal key whose
    decoder is as deep as its encoder.  (large-v3's prefix ids: the task / timestamp tokens sit one higher than in the
    older vocabularies.)"""
    from coral_amd.whisper import CORAL_WHISPER_SHAPES

    kw = CORAL_WHISPER_SHAPES["whisper-large"]
    assert (kw["encoder_layers"], kw["decoder_layers"], kw["num_mel_bins"], kw["vocab_size"]) == (32, 32, 128, 51866)
    _whisper_full_depth_one_clip("whisper-large", [50258, 50285, 50360, 50364], seed=22)


def test_whisper_large_training_step_at_the_reference_batch_of_64():
    """The reference trains its default model at per_device_batch_size = 64 (R/makefile:109-137) WITH gradient
    checkpointing (R/config/asr_finetuning.yaml:73): 80-GB devices cannot keep 32 + 32 layers x 64 clips x 1500 frames of
    activations.  This engine keeps every activation resident - the test runs that step at full size,
    measures what it holds (printed: the peak device memory of forward + backward, parameters and gradients included) and
    checks the batch through size-independent properties: the loss of the 64 clips equals the token-weighted mean of the
    eight 8-clip losses (clips never mix), gradients are finite and the encoder / decoder ends both received them."""
    from coral_amd.whisper import CORAL_WHISPER_SHAPES, WhisperShape
    from coral_amd.whisper_train import WhisperTrainEngine

    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    kw = dict(CORAL_WHISPER_SHAPES["whisper-large"])
    eng = WhisperTrainEngine(WhisperShape(**kw), DEV)
    g = torch.Generator(device=DEV).manual_seed(64)
    for n in eng.store.names():
        v = eng.store.view(n)
        if n.endswith("layer_norm.weight"):
            v.fill_(1.0)
        elif n.endswith(".bias") or n.endswith("__zero"):
            v.zero_()
        else:
            v.normal_(0.0, 0.02, generator=g)
    eng.refresh_compute_weights()
    eng.train(False)  # (dropout off: the 8-clip runs must see the arithmetic of the 64-clip run)
    gen = torch.Generator().manual_seed(65)
    B, L = 64, 48
    feats = torch.randn(B, 128, 3000, generator=gen) * 0.5
    labels = torch.randint(0, 50257, (B, L), generator=gen)
    for b in range(B):
        labels[b, int(torch.randint(16, L + 1, (1,), generator=gen)):] = -100
    eng.zero_grad()
    out = eng.forward_train(feats, labels)
    eng.backward()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    loss64 = float(out["loss"])
    g32 = eng.store.g32
    assert np.isfinite(loss64) and bool(torch.isfinite(g32).all())
    gd = eng.grad_dict()
    for n in ("model.encoder.conv1.weight", "model.encoder.layers.0.self_attn.q_proj.weight",
              "model.decoder.layers.31.fc2.weight", "model.decoder.embed_tokens.weight"):
        assert float(gd[n].abs().max()) > 0.0, n
    tot, cnt = 0.0, 0
    for i in range(0, B, 8):
        o8 = eng.forward_train(feats[i:i + 8], labels[i:i + 8])
        n8 = int((labels[i:i + 8] >= 0).sum())
        tot += float(o8["loss"]) * n8
        cnt += n8
    rel = abs(tot / cnt - loss64) / loss64
    nparam = eng.store.numel
    print(f"\nwhisper-large (32 + 32 L), training step at B = 64 x 30 s, {L} label positions: peak device memory "
          f"{peak:.1f} GiB of 288 (parameters: {nparam / 1e9:.2f} B = {nparam * 10 / 2 ** 30:.1f} GiB of fp32 master + "
          f"gradient + bf16 copy), loss {loss64:.5f}, eight 8-clip batches give {tot / cnt:.5f} (rel {rel:.1e})")
    assert rel <= 2e-5, rel
    assert peak < 270.0
    del eng, out, gd, g32
    torch.cuda.empty_cache()


def test_whisper_large_turbo_full_size_bf16_vs_oracle_and_fp8_vs_bf16():
    """BASELINE configs[4] at its real size: 32 encoder + 4 decoder layers, d 1280, 128 mel bins, 2 clips,
    teacher-forced.  bf16 engine vs the oracle (loss, logits); fp8 forward projections (e4m3 weights and activations
    on `v_mfma_scale_f32_16x16x128_f8f6f4`) vs the bf16 engine: loss within 2 %, logits cosine >= 0.99,
    gradient cosine >= 0.95 - the operands' e4m3 rounding noise through 32 layers, stated against the build's own
    bf16 path because the reference has no fp8."""
    from coral_amd.whisper import CORAL_WHISPER_SHAPES, WhisperShape
    from coral_amd.whisper_train import WhisperTrainEngine
    from oracle import whisper_ref as w

    kw = dict(CORAL_WHISPER_SHAPES["whisper-large-turbo"])
    assert kw["encoder_layers"] == 32 and kw["decoder_layers"] == 4 and kw["num_mel_bins"] == 128
    c = w.WhisperConfig(**kw)
    P = w.synth_params(c)
    g = torch.Generator().manual_seed(31)
    feats = torch.randn(2, 128, 3000, generator=g) * 0.5
    labels = torch.randint(0, 50257, (2, 24), generator=g)
    labels[1, 17:] = -100
    with torch.no_grad():
        loss_ref, logits_ref = w.forward_loss(feats, labels, P, c)
    res = {}
    for mode in ("bf16", "fp8"):
        eng = WhisperTrainEngine(WhisperShape(**kw), DEV)
        eng.load_state_dict(P)
        if mode == "fp8":
            eng.enable_fp8_forward()
        eng.zero_grad()
        out = eng.forward_train(feats, labels)
        eng.backward()
        torch.cuda.synchronize()
        res[mode] = (float(out["loss"]), out["logits"].float().cpu().clone(), eng.store.g32.clone())
        if mode == "fp8":
            # the same step once more with the scales turned over as after an optimiser step (same weights): now the
            # gradient entering fc1 is on the fp8 path too (its scale needs one measured backward: delayed scaling)
            eng.refresh_bucket(next(iter(eng.store.buckets)))
            assert eng._fp8_train["du_ready"][0]
            eng.zero_grad()
            out = eng.forward_train(feats, labels)
            eng.backward()
            torch.cuda.synchronize()
            res["fp8_all"] = (float(out["loss"]), eng.store.g32.clone())
        del eng
        torch.cuda.empty_cache()
    (l0, lg0, g0), (l1, lg1, g1) = res["bf16"], res["fp8"]
    l2, g2 = res["fp8_all"]
    cg2 = float(torch.nn.functional.cosine_similarity(g0.flatten(), g2.flatten(), dim=0))
    print(f"\n  with every fp8 piece active (second step of the fp8 engine, same weights): loss rel {abs(l2 - l0) / abs(l0):.2e}, "
          f"whole-gradient cosine {cg2:.5f}")
    assert abs(l2 - l0) <= 2e-2 * abs(l0) and cg2 >= 0.95
    valid = labels >= 0
    e0 = float((lg0 - logits_ref)[valid].abs().max())
    rel0 = abs(l0 - float(loss_ref)) / float(loss_ref)
    rel1 = abs(l1 - l0) / abs(l0)
    cl = _cos(lg1[valid], lg0[valid])
    cg = float(torch.nn.functional.cosine_similarity(g0.flatten(), g1.flatten(), dim=0))
    print(f"\nwhisper-large-turbo (32 + 4 L), 2 clips: bf16 vs oracle: CE loss {l0:.5f} vs {float(loss_ref):.5f} (rel {rel0:.2e}), "
          f"logits max-abs err {e0:.4f}, cosine {_cos(lg0[valid], logits_ref[valid]):.6f}; fp8 forward vs bf16: loss rel "
          f"{rel1:.2e}, logits cosine {cl:.5f}, whole-gradient cosine {cg:.5f}")
    assert rel0 <= 2e-3 and e0 <= 8e-2
    assert rel1 <= 2e-2 and cl >= 0.99 and cg >= 0.95
    assert not torch.equal(g0, g1)  # the fp8 path really ran

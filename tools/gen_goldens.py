"""Generate tests/golden/* from the reference's own library path (HuggingFace Transformers CPU).

Runs ONLY in the build container (imports `transformers`, which never travels to the GPU box);
its outputs — small input/expected-output vectors — are committed.  CoRal itself pins nothing
numerically on this path, so these vectors are the parity anchor:
Wav2Vec2ForCTC / Wav2Vec2FeatureExtractor / Wav2Vec2CTCTokenizer / F.ctc_loss /
WhisperFeatureExtractor / WhisperForConditionalGeneration as instantiated by
R/src/coral/wav2vec2.py:91-126 and R/src/coral/whisper.py:51-85.

Weights: no checkpoints exist offline, so every parameter is overwritten with
oracle.wav2vec2_ref.synth_params (name-keyed seeded values) — the tests regenerate the same
weights instead of storing them.

usage: python tools/gen_goldens.py [w2v2_tiny ctc featext tokenizer w2v2_cfg1 logmel whisper_tiny whisper_mid hf_ckpt trainer_traj]
"""

from __future__ import annotations

import json
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
GOLD = ROOT / "tests" / "golden"
GOLD.mkdir(parents=True, exist_ok=True)

from oracle import wav2vec2_ref as ref  # noqa: E402


def hf_w2v2(cfg: ref.W2V2Config, apply_spec_augment=False):
    from transformers import Wav2Vec2Config, Wav2Vec2ForCTC

    hc = Wav2Vec2Config(
        hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
        conv_dim=list(cfg.conv_dim), conv_kernel=list(cfg.conv_kernel),
        conv_stride=list(cfg.conv_stride), feat_extract_norm="layer", conv_bias=True,
        do_stable_layer_norm=True, num_conv_pos_embeddings=cfg.num_conv_pos_embeddings,
        num_conv_pos_embedding_groups=cfg.num_conv_pos_embedding_groups,
        vocab_size=cfg.vocab_size, pad_token_id=cfg.pad_token_id,
        ctc_loss_reduction=cfg.ctc_loss_reduction, ctc_zero_infinity=cfg.ctc_zero_infinity,
        layerdrop=0.0, hidden_dropout=0.0, activation_dropout=0.0, attention_dropout=0.0,
        feat_proj_dropout=0.0, final_dropout=0.0, apply_spec_augment=apply_spec_augment,
        mask_time_prob=0.05, mask_feature_prob=0.0, attn_implementation="eager",
    )
    model = Wav2Vec2ForCTC(hc)
    P = ref.synth_params(cfg)
    sd = model.state_dict()
    missing = [k for k in P if k not in sd]
    assert not missing, missing
    with torch.no_grad():
        for k, v in P.items():
            assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
            sd[k].copy_(v)
    model.train()  # all dropouts are 0; train() exercises the SpecAugment branch when enabled
    return model


def synth_batch(B, n_max, lens, lab_lens, seed=4242):
    """Ragged peak-normalised 0.1*randn waveforms + uniform labels (SURVEY.md §8d recipe)."""
    g = torch.Generator().manual_seed(seed)
    waves = []
    for n in lens:
        x = (0.1 * torch.randn(n, generator=g)).clamp(-1, 1)
        waves.append((x / x.abs().max()).numpy())
    Lmax = max(lab_lens)
    labels = torch.full((B, Lmax), -100, dtype=torch.long)
    for b, L in enumerate(lab_lens):
        labels[b, :L] = torch.randint(0, 42, (L,), generator=g)
    return waves, labels


def gen_w2v2_tiny():
    cfg = ref.W2V2Config(hidden_size=128, num_hidden_layers=2, num_attention_heads=4,
                         intermediate_size=256)
    lens, lab_lens = [4000, 3400, 2800], [5, 3, 4]
    waves, labels = synth_batch(3, 4000, lens, lab_lens)
    iv, am = ref.zero_mean_unit_var_norm(waves)
    iv_t, am_t = torch.from_numpy(iv), torch.from_numpy(am).long()
    out = {"lens": np.array(lens), "labels": labels.numpy()}
    for variant in ("plain", "specaug"):
        model = hf_w2v2(cfg, apply_spec_augment=(variant == "specaug"))
        kw = {}
        if variant == "specaug":
            T = int(ref.feat_extract_output_lengths(torch.tensor([4000]), cfg)[0])
            mt = torch.zeros(3, T, dtype=torch.bool)
            mt[0, 2:5] = True
            mt[1, 0:2] = True
            mt[2, 7] = True
            kw["mask_time_indices"] = mt
            out["mask_time"] = mt.numpy()
        if variant == "specaug":
            # Wav2Vec2ForCTC.forward does not take mask_time_indices; drive the base model with
            # the injected mask and apply the CTC tail exactly as modeling_wav2vec2.py:1697-1728.
            hidden = model.wav2vec2(iv_t, attention_mask=am_t, mask_time_indices=kw["mask_time_indices"])[0]
            logits = model.lm_head(model.dropout(hidden))
            in_len = model._get_feat_extract_output_lengths(am_t.sum(-1)).to(torch.long)
            lm = labels >= 0
            lp = torch.nn.functional.log_softmax(logits, dim=-1, dtype=torch.float32).transpose(0, 1)
            loss = torch.nn.functional.ctc_loss(lp, labels.masked_select(lm), in_len, lm.sum(-1),
                                                blank=cfg.pad_token_id, reduction="sum",
                                                zero_infinity=True)
            out["specaug_loss"] = loss.detach().numpy()
            out["specaug_logits"] = logits.detach().numpy()
            continue
        res = model(input_values=iv_t, attention_mask=am_t, labels=labels, output_hidden_states=True)
        res.loss.backward()
        out[f"{variant}_loss"] = res.loss.detach().numpy()
        out[f"{variant}_logits"] = res.logits.detach().numpy()
        if variant == "plain":
            hs = res.hidden_states
            out["hs_first"] = hs[0].detach().numpy()  # after pos-conv add
            out["hs_l0"] = hs[1].detach().numpy()
            out["hs_last"] = hs[-1].detach().numpy()  # after final LN
            feats = model.wav2vec2.feature_extractor(iv_t).transpose(1, 2)
            out["conv_feats"] = feats.detach().numpy()
            sd = dict(model.named_parameters())
            for name in [
                "lm_head.weight", "lm_head.bias",
                "wav2vec2.feature_extractor.conv_layers.0.conv.weight",
                "wav2vec2.feature_extractor.conv_layers.0.layer_norm.weight",
                "wav2vec2.feature_extractor.conv_layers.6.conv.bias",
                "wav2vec2.encoder.layers.0.attention.q_proj.weight",
                "wav2vec2.encoder.layers.1.feed_forward.output_dense.bias",
                "wav2vec2.encoder.layers.1.final_layer_norm.weight",
                "wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original0",
                "wav2vec2.encoder.pos_conv_embed.conv.bias",
                "wav2vec2.feature_projection.projection.bias",
                "wav2vec2.encoder.layer_norm.bias",
            ]:
                out["grad:" + name] = sd[name].grad.detach().numpy()
            for name in [
                "wav2vec2.feature_extractor.conv_layers.1.conv.weight",
                "wav2vec2.feature_extractor.conv_layers.6.conv.weight",
                "wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1",
                "wav2vec2.feature_projection.projection.weight",
                "wav2vec2.encoder.layers.0.feed_forward.intermediate_dense.weight",
            ]:
                gr = sd[name].grad.detach()
                out["gradnorm:" + name] = gr.norm().numpy()
                out["gradhead:" + name] = gr.reshape(-1)[:64].numpy()
    np.savez_compressed(GOLD / "w2v2_tiny.npz", **out)
    print("w2v2_tiny:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if "loss" in k},
          out["plain_loss"], out["specaug_loss"])


def gen_ctc():
    """F.ctc_loss known answers incl. repeats, empty targets and infeasible cases."""
    cases = []
    rng = np.random.RandomState(4242)
    specs = [(12, 5, 3), (20, 6, 0), (8, 4, 8), (8, 4, 5), (30, 12, 10), (16, 7, 7), (5, 3, 6),
             (40, 46, 12), (25, 9, 12), (3, 5, 1), (1, 4, 1), (50, 46, 20), (10, 3, 4), (64, 46, 31),
             (33, 11, 16), (18, 5, 9), (7, 6, 3), (45, 46, 2), (22, 8, 11), (60, 46, 25)]
    out = {}
    for i, (T, V, L) in enumerate(specs):
        blank = V - 1
        g = torch.Generator().manual_seed(1000 + i)
        logits = torch.randn(T, V, generator=g) * 2.0
        if i % 3 == 0 and L >= 2:  # force repeats
            tg = torch.randint(0, max(1, V - 1), (1,), generator=g).repeat(L)
        else:
            tg = torch.randint(0, max(1, V - 1), (L,), generator=g)
        t_in = T if i % 4 else max(1, T - 2)
        lg = logits.clone().requires_grad_(True)
        lp = torch.log_softmax(lg, -1)
        loss = torch.nn.functional.ctc_loss(lp[:, None, :], tg[None, :], torch.tensor([t_in]),
                                            torch.tensor([L]), blank=blank, reduction="sum",
                                            zero_infinity=True)
        loss.backward()
        out[f"c{i}_logits"] = logits.numpy()
        out[f"c{i}_targets"] = tg.numpy()
        out[f"c{i}_tin"] = np.array(t_in)
        out[f"c{i}_loss"] = loss.detach().numpy()
        out[f"c{i}_grad"] = lg.grad.numpy()
        cases.append((T, V, L, float(loss)))
    # one BASELINE-sized case: regenerated from its seed in the test, checksums stored
    T, V, L = 499, 46, 120
    g = torch.Generator().manual_seed(777)
    logits = torch.randn(T, V, generator=g)
    tg = torch.randint(0, 42, (L,), generator=g)
    lg = logits.clone().requires_grad_(True)
    loss = torch.nn.functional.ctc_loss(torch.log_softmax(lg, -1)[:, None, :], tg[None, :],
                                        torch.tensor([T]), torch.tensor([L]), blank=45,
                                        reduction="sum", zero_infinity=True)
    loss.backward()
    out["big_loss"] = loss.detach().numpy()
    out["big_grad_rows"] = lg.grad[::50].numpy()
    out["big_grad_abs_sum"] = lg.grad.abs().sum().numpy()
    out["n_cases"] = np.array(len(specs))
    np.savez_compressed(GOLD / "ctc_cases.npz", **out)
    print("ctc:", cases[:6], float(out["big_loss"]))


def gen_featext():
    from transformers import Wav2Vec2FeatureExtractor

    fe = Wav2Vec2FeatureExtractor(feature_size=1, sampling_rate=16000, padding_value=0.0,
                                  do_normalize=True, return_attention_mask=True)
    g = np.random.RandomState(7)
    waves = [(0.1 * g.randn(n) + off).astype(np.float32) for n, off in
             [(1600, 0.0), (900, 0.02), (1234, -0.01), (400, 0.0)]]
    res = fe(waves, sampling_rate=16000, padding="longest", return_tensors="np")
    res2 = fe(waves, sampling_rate=16000, padding="max_length", max_length=2000, return_tensors="np")
    np.savez_compressed(GOLD / "feature_extractor.npz", **{f"wave{i}": w for i, w in enumerate(waves)},
                        input_values=res["input_values"], attention_mask=res["attention_mask"],
                        input_values_max=res2["input_values"], attention_mask_max=res2["attention_mask"])
    print("featext:", res["input_values"].shape, res2["input_values"].shape)


def gen_tokenizer():
    from transformers import Wav2Vec2CTCTokenizer

    vocab = ref.coral_vocab()
    with tempfile.TemporaryDirectory() as td:
        vf = Path(td) / "vocab.json"
        vf.write_text(json.dumps(vocab))
        tok = Wav2Vec2CTCTokenizer(str(vf), unk_token="<unk>", pad_token="<pad>", bos_token="<s>",
                                   eos_token="</s>", word_delimiter_token="|")
        rng = np.random.RandomState(11)
        rows = [[vocab[c] for c in "hh"] + [45] + [vocab["e"], vocab["j"], vocab["j"], vocab["|"],
                                                  vocab["|"], 45, vocab["j"], 45, vocab["j"]]]
        for _ in range(12):
            n = rng.randint(5, 60)
            r = rng.choice(list(range(42)) + [45] * 20 + [vocab["|"]] * 6, size=n)
            r = np.repeat(r, rng.randint(1, 4, size=n))
            rows.append([int(x) for x in r])
        texts = [tok.decode(r) for r in rows]
    (GOLD / "tokenizer_collapse.json").write_text(
        json.dumps({"vocab": vocab, "rows": rows, "texts": texts}, ensure_ascii=False, indent=0))
    print("tokenizer:", texts[:3])


def gen_w2v2_cfg1():
    """BASELINE.json configs[0]: XLS-R-300M shape, 4 x 5 s, fp32 CPU, fwd+bwd with CTC."""
    cfg = ref.W2V2Config(**ref.CORAL_SHAPES["wav2vec2-small"])
    lens, lab_lens = [80000, 80000, 72000, 56000], [60, 45, 38, 25]
    waves, labels = synth_batch(4, 80000, lens, lab_lens)
    iv, am = ref.zero_mean_unit_var_norm(waves)
    model = hf_w2v2(cfg)
    res = model(input_values=torch.from_numpy(iv), attention_mask=torch.from_numpy(am).long(),
                labels=labels)
    res.loss.backward()
    sd = dict(model.named_parameters())
    out = {"lens": np.array(lens), "labels": labels.numpy(), "loss": res.loss.detach().numpy(),
           "logits_slice": res.logits.detach()[:, ::16, :].numpy(),
           "logits_abs_mean": res.logits.detach().abs().mean().numpy()}
    for name in ["lm_head.weight", "wav2vec2.encoder.layers.23.feed_forward.output_dense.weight",
                 "wav2vec2.encoder.layers.0.attention.q_proj.weight",
                 "wav2vec2.feature_extractor.conv_layers.0.conv.weight",
                 "wav2vec2.feature_extractor.conv_layers.3.conv.weight",
                 "wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1"]:
        out["gradnorm:" + name] = sd[name].grad.norm().numpy()
    # for the GPU test at depth (tests/test_depth_gpu.py): the whole logits tensor, the norm of EVERY gradient tensor and
    # 512 evenly spaced elements of each (direction check) -- the fp32 reference does not have to be re-run on the GPU box
    out["logits_full"] = res.logits.detach().numpy()
    names = [n for n, p_ in sd.items() if p_.grad is not None]
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array([float(sd[n].grad.norm()) for n in names], dtype=np.float32)
    def spread(t):  # 512 evenly spaced elements of the flattened tensor
        t = t.flatten()
        return t[torch.linspace(0, t.numel() - 1, 512).long()].numpy()

    out["grad_samples"] = np.stack([spread(sd[n].grad) for n in names]).astype(np.float32)
    np.savez_compressed(GOLD / "w2v2_cfg1.npz", **out)
    print("w2v2_cfg1 loss", out["loss"], {k: float(v) for k, v in out.items() if k.startswith("gradnorm")})


def gen_collator():
    """DataCollatorCTCWithPadding semantics (R/src/coral/data_collators.py:62-95) from the HF
    building blocks it calls: Wav2Vec2Processor.pad(audio) + pad(labels) + -100 fill."""
    from transformers import Wav2Vec2CTCTokenizer, Wav2Vec2FeatureExtractor, Wav2Vec2Processor

    vocab = {k: v for k, v in ref.coral_vocab().items() if not k.startswith("<")}
    with tempfile.TemporaryDirectory() as td:
        (Path(td) / "vocab.json").write_text(json.dumps(vocab))
        tok = Wav2Vec2CTCTokenizer(str(Path(td) / "vocab.json"), unk_token="<unk>", pad_token="<pad>",
                                   bos_token="<s>", eos_token="</s>", word_delimiter_token="|")
        fe = Wav2Vec2FeatureExtractor(feature_size=1, sampling_rate=16000, padding_value=0.0,
                                      do_normalize=True, return_attention_mask=True)
        proc = Wav2Vec2Processor(feature_extractor=fe, tokenizer=tok)
        g = np.random.RandomState(3)
        texts = ["hej med dig", "æøå 123", "a", "det er en længere sætning end de andre"]
        feats = []
        for i, n in enumerate([800, 1200, 500, 1600]):
            wave = (0.1 * g.randn(n)).astype(np.float32)
            iv = proc(wave, sampling_rate=16000).input_values[0]
            ids = proc(text=texts[i], truncation=True).input_ids
            feats.append(dict(input_values=iv, labels=ids))
        out = {f"iv{i}": f["input_values"] for i, f in enumerate(feats)}
        for i, f in enumerate(feats):
            out[f"lab{i}"] = np.array(f["labels"])
        for padding, key in (("longest", "longest"), ("max_length", "max")):
            batch = proc.pad([dict(input_values=f["input_values"]) for f in feats], padding=padding,
                             return_tensors="np", max_length=2000)
            lb = proc.pad(labels=[dict(input_ids=f["labels"]) for f in feats], padding=padding,
                          return_tensors="np", max_length=min(tok.model_max_length, 512))
            labels = np.where(lb["attention_mask"] == 1, lb["input_ids"], -100)
            out[f"{key}_input_values"] = batch["input_values"]
            out[f"{key}_attention_mask"] = batch["attention_mask"]
            out[f"{key}_labels"] = labels
        (GOLD / "collator_texts.json").write_text(json.dumps(texts, ensure_ascii=False))
        np.savez_compressed(GOLD / "collator.npz", **out)
    print("collator:", out["longest_labels"].shape, out["max_labels"].shape)


def gen_specaug():
    """_compute_mask_indices under fixed np.random seeds (host RNG stream parity)."""
    from transformers.models.wav2vec2.modeling_wav2vec2 import _compute_mask_indices

    cases = [(1, (8, 499), 0.5, 10, None, 2), (2, (4, 249), 0.5, 10, [249, 200, 120, 30], 2),
             (3, (8, 1920), 0.5, 64, None, 0), (4, (3, 12), 0.05, 10, [12, 10, 8], 2), (5, (2, 20), 0.9, 5, [20, 3], 2),
             (6, (8, 1024), 0.5, 64, None, 0)]
    out = {"n": np.array(len(cases))}
    for i, (seed, shape, p, ml, lens, mm) in enumerate(cases):
        np.random.seed(seed)
        am = None
        if lens is not None:
            am = torch.zeros(shape, dtype=torch.long)
            for r, n in enumerate(lens):
                am[r, :n] = 1
        out[f"m{i}"] = np.packbits(_compute_mask_indices(shape, p, ml, attention_mask=am, min_masks=mm))
        out[f"spec{i}"] = np.array([seed, shape[0], shape[1], int(p * 1000), ml, mm] + (lens or []))
    np.savez_compressed(GOLD / "specaugment.npz", **out)
    print("specaug ok")


def gen_logmel():
    """WhisperFeatureExtractor (torch STFT path, the one used when torch is installed) on seeded clips."""
    from transformers import WhisperFeatureExtractor

    rng = np.random.RandomState(5)
    t = np.arange(59_200) / 16000.0
    clips = [(0.3 * np.sin(2 * np.pi * 440 * t) + 0.05 * rng.randn(len(t))).astype(np.float32),
             (0.1 * rng.randn(480_000)).astype(np.float32)]
    out = {"clip0_len": np.array(len(clips[0]))}
    for mels in (80, 128):
        fe = WhisperFeatureExtractor(feature_size=mels)
        out[f"filters{mels}"] = fe.mel_filters.astype(np.float32)
        feats = fe(clips, sampling_rate=16000, return_tensors="np")["input_features"]
        assert feats.shape == (2, mels, 3000)
        out[f"feat{mels}_sub"] = feats[:, :, ::25].astype(np.float32)       # every 25th frame
        out[f"feat{mels}_head"] = feats[:, :, :40].astype(np.float32)       # first frames (reflect pad)
        out[f"feat{mels}_stats"] = np.array([[f.mean(), f.std(), f.min(), f.max()] for f in feats], dtype=np.float64)
    np.savez_compressed(GOLD / "logmel.npz", **out)
    print("logmel:", out["feat80_sub"].shape, out["feat80_stats"])


def gen_whisper_tiny():
    """WhisperForConditionalGeneration with a small architecture: encoder states, teacher-forced
    logits, CE loss and a manual greedy loop over the HF forward."""
    from transformers import WhisperConfig, WhisperForConditionalGeneration

    from oracle import whisper_ref as wref

    c = wref.WhisperConfig(d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                           decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128, num_mel_bins=80,
                           vocab_size=200, max_target_positions=64, pad_token_id=150, decoder_start_token_id=151,
                           eos_token_id=150)
    hc = WhisperConfig(d_model=c.d_model, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                       decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128, num_mel_bins=80,
                       vocab_size=200, max_source_positions=1500, max_target_positions=64, pad_token_id=150,
                       bos_token_id=150, eos_token_id=150, decoder_start_token_id=151, dropout=0.0,
                       attention_dropout=0.0, activation_dropout=0.0, encoder_layerdrop=0.0, decoder_layerdrop=0.0,
                       apply_spec_augment=False, attn_implementation="eager", suppress_tokens=[],
                       begin_suppress_tokens=[])
    model = WhisperForConditionalGeneration(hc)
    P = wref.synth_params(c)
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in P.items():
            assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
            sd[k].copy_(v)
    model.eval()
    g = torch.Generator().manual_seed(9)
    feats = torch.randn(2, 80, 3000, generator=g) * 0.5
    labels = torch.randint(0, 150, (2, 9), generator=g)
    labels[1, 6:] = -100
    res = model(input_features=feats, labels=labels)
    res.loss.backward()
    enc = model.model.encoder(feats).last_hidden_state
    out = {"labels": labels.numpy(), "loss": res.loss.detach().numpy(), "logits": res.logits.detach().numpy(),
           "enc_slice": enc.detach()[:, ::100, :].numpy(),
           "grad_fc1": dict(model.named_parameters())["model.decoder.layers.1.fc1.weight"].grad.detach().numpy(),
           "gradnorm_embed": dict(model.named_parameters())["model.decoder.embed_tokens.weight"].grad.norm().detach().numpy(),
           "gradnorm_conv1": dict(model.named_parameters())["model.encoder.conv1.weight"].grad.norm().detach().numpy()}
    prefix = [151, 160, 161, 162]
    ids = torch.tensor([prefix, prefix])
    done = torch.zeros(2, dtype=torch.bool)
    with torch.no_grad():
        while ids.shape[1] < 24 and not bool(done.all()):
            lg = model(input_features=feats, decoder_input_ids=ids).logits[:, -1].clone()
            lg[:, [170, 171]] = float("-inf")
            if ids.shape[1] == len(prefix):
                lg[:, [20, 150]] = float("-inf")
            nxt = lg.argmax(-1)
            nxt = torch.where(done, torch.full_like(nxt, 150), nxt)
            ids = torch.cat([ids, nxt[:, None]], 1)
            done |= nxt == 150
    out["greedy_ids"] = ids.numpy()
    np.savez_compressed(GOLD / "whisper_tiny.npz", **out)
    print("whisper_tiny: loss", out["loss"], "greedy", ids.tolist())


def gen_whisper_mid():
    """A medium-depth Whisper (6 + 6 layers, d = 512, 8 heads of 64, ffn 2048): the fixture behind the GPU test of bf16
    error accumulation through a deeper encoder / decoder stack (tests/test_depth_gpu.py)."""
    from transformers import WhisperConfig, WhisperForConditionalGeneration

    from oracle import whisper_ref as wref

    kw = dict(d_model=512, encoder_layers=6, decoder_layers=6, encoder_attention_heads=8, decoder_attention_heads=8,
              encoder_ffn_dim=2048, decoder_ffn_dim=2048, num_mel_bins=80, vocab_size=2000, max_target_positions=64,
              pad_token_id=1950, decoder_start_token_id=1951, eos_token_id=1950)
    c = wref.WhisperConfig(**kw)
    hc = WhisperConfig(d_model=512, encoder_layers=6, decoder_layers=6, encoder_attention_heads=8,
                       decoder_attention_heads=8, encoder_ffn_dim=2048, decoder_ffn_dim=2048, num_mel_bins=80,
                       vocab_size=2000, max_source_positions=1500, max_target_positions=64, pad_token_id=1950,
                       bos_token_id=1950, eos_token_id=1950, decoder_start_token_id=1951, dropout=0.0,
                       attention_dropout=0.0, activation_dropout=0.0, encoder_layerdrop=0.0, decoder_layerdrop=0.0,
                       apply_spec_augment=False, attn_implementation="eager", suppress_tokens=[],
                       begin_suppress_tokens=[])
    model = WhisperForConditionalGeneration(hc)
    P = wref.synth_params(c)
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in P.items():
            assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
            sd[k].copy_(v)
    model.eval()
    g = torch.Generator().manual_seed(11)
    feats = torch.randn(2, 80, 3000, generator=g) * 0.5
    labels = torch.randint(0, 1950, (2, 14), generator=g)
    labels[1, 10:] = -100
    res = model(input_features=feats, labels=labels)
    res.loss.backward()
    enc = model.model.encoder(feats).last_hidden_state
    named = dict(model.named_parameters())
    out = {"labels": labels.numpy(), "loss": res.loss.detach().numpy(), "logits": res.logits.detach().numpy(),
           "enc_slice": enc.detach()[:, ::50, :].numpy(), "enc_abs_mean": enc.detach().abs().mean().numpy()}
    for name in ["model.decoder.embed_tokens.weight", "model.encoder.conv1.weight",
                 "model.encoder.layers.0.self_attn.q_proj.weight", "model.encoder.layers.5.fc2.weight",
                 "model.decoder.layers.0.encoder_attn.k_proj.weight", "model.decoder.layers.5.fc1.weight"]:
        out["gradnorm:" + name] = named[name].grad.norm().detach().numpy()
    prefix = [1951, 1960, 1961, 1962]
    ids = torch.tensor([prefix, prefix])
    done = torch.zeros(2, dtype=torch.bool)
    with torch.no_grad():
        while ids.shape[1] < 24 and not bool(done.all()):
            lg = model(input_features=feats, decoder_input_ids=ids).logits[:, -1].clone()
            lg[:, [1970, 1971]] = float("-inf")
            if ids.shape[1] == len(prefix):
                lg[:, [20, 1950]] = float("-inf")
            nxt = lg.argmax(-1)
            nxt = torch.where(done, torch.full_like(nxt, 1950), nxt)
            ids = torch.cat([ids, nxt[:, None]], 1)
            done |= nxt == 1950
    out["greedy_ids"] = ids.numpy()
    np.savez_compressed(GOLD / "whisper_mid.npz", **out)
    print("whisper_mid: loss", out["loss"], "greedy", ids.tolist())


TINY_CKPT_W2V2 = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                      conv_dim=(512, 32, 32, 32, 32, 32, 32), num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)


def gen_hf_ckpt():
    """N2: model directories WRITTEN BY TRANSFORMERS, loaded by the engine's `from_pretrained` in
    tests/test_ckpt_gpu.py (R/src/coral/wav2vec2.py:253-305, R/src/coral/whisper.py:234-267 read exactly such
    directories).  Weights are HF's own random init (torch.manual_seed), rounded to fp16 so the files stay small; the
    golden outputs are computed in fp32 from the same rounded weights.
      hf_ckpt_w2v2/           Wav2Vec2ForCTC.save_pretrained (safetensors)
      hf_ckpt_w2v2_pretrain/  a head-less Wav2Vec2ForPreTraining state dict as `pytorch_model.bin` with the
                              pre-parametrize weight-norm names (weight_g / weight_v) and quantizer / project_* extras
      hf_ckpt_whisper/        WhisperForConditionalGeneration.save_pretrained (tied proj_out)"""
    import shutil

    from transformers import (Wav2Vec2Config, Wav2Vec2ForCTC, Wav2Vec2ForPreTraining, WhisperConfig,
                              WhisperForConditionalGeneration)

    torch.manual_seed(77)
    kw = dict(TINY_CKPT_W2V2)
    hc = Wav2Vec2Config(**{**kw, "conv_dim": list(kw["conv_dim"])}, feat_extract_norm="layer", conv_bias=True,
                        do_stable_layer_norm=True, vocab_size=46, pad_token_id=45, ctc_loss_reduction="sum",
                        ctc_zero_infinity=True, layerdrop=0.0, hidden_dropout=0.0, activation_dropout=0.0,
                        attention_dropout=0.0, feat_proj_dropout=0.0, final_dropout=0.0, apply_spec_augment=False,
                        attn_implementation="eager")
    model = Wav2Vec2ForCTC(hc).half().float().eval()
    with torch.no_grad():  # HF initialises lm_head / biases to values that make the comparison weak: perturb
        for n, p_ in model.named_parameters():
            if n.endswith(".bias") or n.endswith("layer_norm.weight"):
                p_.add_(0.05 * torch.randn_like(p_))
        model.half().float()
    g = torch.Generator().manual_seed(5)
    waves = [(0.1 * torch.randn(n, generator=g)).numpy() for n in (4000, 3300)]
    iv, am = ref.zero_mean_unit_var_norm(waves)
    labels = torch.tensor([[3, 7, 7, 1], [9, 2, -100, -100]])
    out = model(input_values=torch.from_numpy(iv), attention_mask=torch.from_numpy(am).long(), labels=labels,
                output_hidden_states=True)
    d = GOLD / "hf_ckpt_w2v2"
    shutil.rmtree(d, ignore_errors=True)
    model.half().save_pretrained(d)
    model.float()
    np.savez_compressed(GOLD / "hf_ckpt_w2v2.npz", logits=out.logits.detach().numpy(), loss=out.loss.detach().numpy(),
                        last_hidden=out.hidden_states[-1].detach().numpy(), lens=np.array([4000, 3300]),
                        labels=labels.numpy())
    # head-less pretraining checkpoint, legacy file and names
    torch.manual_seed(78)
    pre = Wav2Vec2ForPreTraining(hc).half().float().eval()
    with torch.no_grad():
        hid = pre.wav2vec2(input_values=torch.from_numpy(iv), attention_mask=torch.from_numpy(am).long()).last_hidden_state
    sd = {}
    for k, v in pre.state_dict().items():
        k = k.replace("parametrizations.weight.original0", "weight_g").replace("parametrizations.weight.original1", "weight_v")
        sd[k] = v.half().clone()
    d = GOLD / "hf_ckpt_w2v2_pretrain"
    shutil.rmtree(d, ignore_errors=True)
    d.mkdir(parents=True)
    torch.save(sd, d / "pytorch_model.bin")
    cfgd = json.loads((GOLD / "hf_ckpt_w2v2" / "config.json").read_text())
    cfgd["architectures"] = ["Wav2Vec2ForPreTraining"]
    cfgd.pop("vocab_size", None)
    cfgd["vocab_size"] = 32  # the pretraining config's placeholder: the finetune overrides it (R/src/coral/wav2vec2.py:124)
    (d / "config.json").write_text(json.dumps(cfgd, indent=1))
    np.savez_compressed(GOLD / "hf_ckpt_w2v2_pretrain.npz", last_hidden=hid.numpy(),
                        extra_keys=np.array(sorted(k for k in sd if not k.startswith("wav2vec2."))))
    # Whisper
    torch.manual_seed(79)
    wc = WhisperConfig(d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                       decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128, num_mel_bins=80,
                       vocab_size=200, max_source_positions=1500, max_target_positions=64, pad_token_id=150,
                       bos_token_id=150, eos_token_id=150, decoder_start_token_id=151, dropout=0.0,
                       attention_dropout=0.0, activation_dropout=0.0, encoder_layerdrop=0.0, decoder_layerdrop=0.0,
                       apply_spec_augment=False, attn_implementation="eager", suppress_tokens=[],
                       begin_suppress_tokens=[])
    wm = WhisperForConditionalGeneration(wc).half().float().eval()
    with torch.no_grad():
        for n, p_ in wm.named_parameters():
            if p_.requires_grad and (n.endswith(".bias") or n.endswith("layer_norm.weight")):
                p_.add_(0.05 * torch.randn_like(p_))
        wm.half().float()
    feats = torch.randn(2, 80, 3000, generator=g) * 0.5
    wl = torch.randint(0, 150, (2, 9), generator=g)
    res = wm(input_features=feats, labels=wl)
    d = GOLD / "hf_ckpt_whisper"
    shutil.rmtree(d, ignore_errors=True)
    wm.half().save_pretrained(d)
    np.savez_compressed(GOLD / "hf_ckpt_whisper.npz", logits=res.logits.detach().numpy(), loss=res.loss.detach().numpy(),
                        labels=wl.numpy(), feats_seed=np.array(5))
    for dd in ("hf_ckpt_w2v2", "hf_ckpt_w2v2_pretrain", "hf_ckpt_whisper"):
        print(dd, {f.name: f.stat().st_size for f in (GOLD / dd).iterdir()})


def _trajectory(model, examples, collate, B, accum, steps, lr, warmup, watch):
    """Drive `transformers.Trainer` - the reference's own loop (R/src/coral/finetune.py:60-79; the optimiser, schedule
    and clipping of R/src/coral/wav2vec2.py:156-251 / whisper.py) - over a fixed example stream on the CPU in fp32 and
    return what it logged per optimiser step plus the watched parameters afterwards."""
    from torch.utils.data import IterableDataset
    from transformers import Trainer, TrainerCallback, TrainingArguments

    class Stream(IterableDataset):  # (an IterableDataset, as the reference streams: Trainer does not shuffle it)
        def __iter__(self):
            return iter(examples)

    logs = []

    class Rec(TrainerCallback):
        def on_log(self, args, state, control, logs=None, **kw):
            if logs and "loss" in logs:
                logs_ = dict(logs)
                logs_["step"] = state.global_step
                logs.update({})
                rec.append(logs_)

    rec = logs
    with tempfile.TemporaryDirectory() as td:
        args = TrainingArguments(
            output_dir=td, per_device_train_batch_size=B, gradient_accumulation_steps=accum, max_steps=steps,
            learning_rate=lr, warmup_steps=warmup, lr_scheduler_type="cosine", adam_beta1=0.9, adam_beta2=0.98,
            adam_epsilon=1e-8, weight_decay=0.0, max_grad_norm=1.0, logging_steps=1, logging_first_step=True,
            save_strategy="no", report_to=[], use_cpu=True, seed=4242, dataloader_num_workers=0,
            remove_unused_columns=False, disable_tqdm=True, logging_nan_inf_filter=False)
        tr = Trainer(model=model, args=args, data_collator=collate, train_dataset=Stream(), callbacks=[Rec()])
        accepts = bool(tr.model_accepts_loss_kwargs)
        tr.train()
    sd = dict(model.named_parameters())
    out = {"loss": np.array([r["loss"] for r in rec], dtype=np.float64),
           "grad_norm": np.array([r["grad_norm"] for r in rec], dtype=np.float64),
           "lr_logged": np.array([r["learning_rate"] for r in rec], dtype=np.float64),
           "step": np.array([r["step"] for r in rec]), "model_accepts_loss_kwargs": np.array(accepts)}
    for name in watch:
        out["final:" + name] = sd[name].detach().numpy().copy()
    return out


def gen_trainer_traj():
    """Per-step loss / gradient norm / learning rate and the final head weights of SIX optimiser steps of
    `transformers.Trainer` (gradient_accumulation_steps 2, max_grad_norm 1.0, cosine schedule with 2 warm-up steps,
    AdamW 0.9 / 0.98, dropout 0, fp32 CPU) on the tiny wav2vec2 and the tiny Whisper: the pin of SURVEY.md row A11 -
    accumulation scaling, clipping, warm-up and AdamW TOGETHER ($TF/trainer.py:1892-1963,1778-1796)."""
    from transformers import WhisperConfig, WhisperForConditionalGeneration

    from oracle import whisper_ref as wref

    out = {}
    # ---- wav2vec2 (CTC) -------------------------------------------------------------------------------------
    cfg = ref.W2V2Config(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256)
    B, accum, steps = 3, 2, 6
    g = torch.Generator().manual_seed(1234)
    lens = [int(x) for x in torch.randint(2400, 4001, (B * accum * steps,), generator=g)]
    lab_lens = [int(x) for x in torch.randint(2, 6, (B * accum * steps,), generator=g)]
    waves, labels = synth_batch(len(lens), 4000, lens, lab_lens, seed=77)
    examples = []
    for w, lab, L in zip(waves, labels, lab_lens):
        iv, _ = ref.zero_mean_unit_var_norm([w])
        examples.append({"input_values": iv[0].astype(np.float32), "labels": [int(t) for t in lab[:L]]})

    def collate_ctc(feats):  # DataCollatorCTCWithPadding, padding="longest" (R/src/coral/data_collators.py:62-95)
        n = max(len(f["input_values"]) for f in feats)
        iv = torch.zeros(len(feats), n)
        am = torch.zeros(len(feats), n, dtype=torch.long)
        Lm = max(len(f["labels"]) for f in feats)
        lab = torch.full((len(feats), Lm), -100, dtype=torch.long)
        for i, f in enumerate(feats):
            k = len(f["input_values"])
            iv[i, :k] = torch.from_numpy(f["input_values"])
            am[i, :k] = 1
            lab[i, :len(f["labels"])] = torch.tensor(f["labels"])
        return {"input_values": iv, "attention_mask": am, "labels": lab}

    model = hf_w2v2(cfg)
    watch = ["lm_head.weight", "lm_head.bias", "wav2vec2.encoder.layers.1.feed_forward.output_dense.weight",
             "wav2vec2.feature_extractor.conv_layers.0.conv.weight",
             "wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original0"]
    t = _trajectory(model, examples, collate_ctc, B, accum, steps, 2e-3, 2, watch)
    out.update({"w2v2:" + k: v for k, v in t.items()})
    out["w2v2:lens"] = np.array(lens)
    out["w2v2:lab_lens"] = np.array(lab_lens)
    out["w2v2:labels"] = labels.numpy()
    out["w2v2:hparams"] = np.array([B, accum, steps, 2e-3, 2])
    print("trainer_traj w2v2: accepts_loss_kwargs", t["model_accepts_loss_kwargs"], "loss", t["loss"], "gnorm", t["grad_norm"],
          "lr", t["lr_logged"])
    # ---- Whisper (teacher-forced CE) ------------------------------------------------------------------------
    c = wref.WhisperConfig(d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                           decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128, num_mel_bins=80,
                           vocab_size=200, max_target_positions=64, pad_token_id=150, decoder_start_token_id=151,
                           eos_token_id=150)
    hc = WhisperConfig(d_model=c.d_model, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                       decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128, num_mel_bins=80,
                       vocab_size=200, max_source_positions=1500, max_target_positions=64, pad_token_id=150,
                       bos_token_id=150, eos_token_id=150, decoder_start_token_id=151, dropout=0.0,
                       attention_dropout=0.0, activation_dropout=0.0, encoder_layerdrop=0.0, decoder_layerdrop=0.0,
                       apply_spec_augment=False, attn_implementation="eager", suppress_tokens=[],
                       begin_suppress_tokens=[], use_cache=False)
    wm = WhisperForConditionalGeneration(hc)
    P = wref.synth_params(c)
    sd = wm.state_dict()
    with torch.no_grad():
        for k, v in P.items():
            sd[k].copy_(v)
    wm.train()
    Bw = 2
    nw = Bw * accum * steps
    g = torch.Generator().manual_seed(4321)
    feats = (torch.randn(nw, 80, 3000, generator=g) * 0.5).to(torch.float16)  # (fp16 keeps the fixture small; exact in fp32)
    wl = [int(x) for x in torch.randint(4, 10, (nw,), generator=g)]
    wlab = torch.full((nw, 9), -100, dtype=torch.long)
    for i, L in enumerate(wl):
        wlab[i, :L] = torch.randint(0, 150, (L,), generator=g)
    wex = [{"input_features": feats[i].float().numpy(), "labels": [int(t) for t in wlab[i, :wl[i]]]} for i in range(nw)]

    def collate_s2s(fs):  # DataCollatorSpeechSeq2SeqWithPadding (R/src/coral/data_collators.py:145-187), no BOS to cut
        Lm = max(len(f["labels"]) for f in fs)
        lab = torch.full((len(fs), Lm), -100, dtype=torch.long)
        for i, f in enumerate(fs):
            lab[i, :len(f["labels"])] = torch.tensor(f["labels"])
        return {"input_features": torch.stack([torch.from_numpy(f["input_features"]) for f in fs]), "labels": lab}

    wwatch = ["model.decoder.embed_tokens.weight", "model.decoder.layers.1.fc1.weight", "model.encoder.conv1.weight",
              "model.decoder.embed_positions.weight"]
    t = _trajectory(wm, wex, collate_s2s, Bw, accum, steps, 2e-3, 2, wwatch)
    out.update({"whisper:" + k: v for k, v in t.items()})
    out["whisper:feats_seed"] = np.array(4321)
    out["whisper:lab_lens"] = np.array(wl)
    out["whisper:labels"] = wlab.numpy()
    out["whisper:hparams"] = np.array([Bw, accum, steps, 2e-3, 2])
    print("trainer_traj whisper: accepts_loss_kwargs", t["model_accepts_loss_kwargs"], "loss", t["loss"], "gnorm", t["grad_norm"],
          "lr", t["lr_logged"])
    np.savez_compressed(GOLD / "trainer_traj.npz", **out)
    print("trainer_traj.npz", (GOLD / "trainer_traj.npz").stat().st_size, "bytes")


if __name__ == "__main__":
    todo = sys.argv[1:] or ["w2v2_tiny", "ctc", "featext", "tokenizer", "collator", "specaug"]
    torch.manual_seed(4242)
    for name in todo:
        globals()["gen_" + name]()

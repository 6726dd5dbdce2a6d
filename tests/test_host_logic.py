"""CPU tests of the host-side mirror of the reference interface: config surface, processor /
collator / SpecAugment parity with HF-generated fixtures, metrics, parameter layout, and that the
C-ABI library loads and exports every declared symbol (no compute calls: there is no GPU here)."""
import json
import re

import numpy as np
import pytest
import torch


def test_library_loads_and_exports_every_declared_symbol():
    from pathlib import Path

    from coral_amd import _lib

    lib = _lib.load()
    header = (Path(__file__).resolve().parents[1] / "include" / "coral_amd.h").read_text()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(ca_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 40
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.ca_version() >= 100
    # argument validation works without a GPU and reports through ca_last_error
    rc = lib.ca_layernorm_fwd(None, None, None, None, None, 4, 512, 1e-5, 0, None)
    assert rc == -1 and b"null" in lib.ca_last_error()


def test_product_path_fails_loudly_without_gpu():
    from coral_amd import ops
    from coral_amd.wav2vec2 import Wav2Vec2CTCEngine, Wav2Vec2Shape

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ops.CoralAmdError):
        Wav2Vec2CTCEngine(Wav2Vec2Shape(hidden_size=128, num_hidden_layers=1, num_attention_heads=4,
                                        intermediate_size=256))
    with pytest.raises(ops.CoralAmdError):
        ops.cast_f32_bf16(torch.zeros(8), torch.zeros(8, dtype=torch.bfloat16), 8)  # CPU tensors


def test_config_surface():
    from coral_amd.config import load_config

    c = load_config("asr_finetuning", ["model=wav2vec2-large", "datasets=synthetic", "max_steps=3",
                                       "model.learning_rate=3e-5", "per_device_batch_size=2", "total_batch_size=16"])
    assert c.model.name == "wav2vec2-large" and c.model.type == "wav2vec2"
    assert c.model.pretrained_model_id == "facebook/wav2vec2-xls-r-2b"
    assert c.model.learning_rate == 3e-5 and isinstance(c.model.learning_rate, float)
    assert c.model.mask_time_prob == 0.5 and c.model.mask_feature_length == 64 and c.model.layerdrop == 0.1
    assert c.model.ctc_loss_reduction == "sum" and c.model.activation_dropout == 0.1
    assert c.model_id.startswith("wav2vec2-large-20") and c.model_dir == f"models/{c.model_id}"
    assert c.seed == 4242 and c.adam_second_momentum == 0.98 and c.max_grad_norm == 1.0
    assert c.padding == "longest" and c.max_seconds_per_example == 10.0 and c.warmup_steps == 1000
    assert list(c.datasets.keys()) == ["synthetic"] and c.max_steps == 3
    d = load_config("asr_finetuning", [])  # reference defaults: whisper-large on the two CoRal sets
    assert d.model.name == "whisper-large" and list(d.datasets.keys()) == ["coral_read_aloud", "coral_conversation"]
    for key in ["wav2vec2-small", "wav2vec2-medium", "wav2vec2-large", "whisper-xxsmall", "whisper-xsmall",
                "whisper-small", "whisper-medium", "whisper-large", "whisper-large-turbo", "test-wav2vec2", "test-whisper"]:
        m = load_config("asr_finetuning", [f"model={key}"]).model
        assert m.name == key and m.sampling_rate == 16000
    e = load_config("evaluation", ["model_id=foo", "batch_size=4"])
    assert (e.model_id, e.batch_size, e.eval_split_name, e.no_lm) == ("foo", 4, "test", False)
    with pytest.raises(ValueError):
        load_config("asr_finetuning", ["oops"])


def test_model_setup_training_arguments_and_vocab(tmp_path):
    from coral_amd.config import load_config
    from coral_amd.model_setup import Wav2Vec2ModelSetup, load_model_setup

    c = load_config("asr_finetuning", ["model=test-wav2vec2", "datasets=synthetic", f"models_dir={tmp_path}",
                                       "total_batch_size=64", "per_device_batch_size=8"])
    setup = load_model_setup(c)
    assert isinstance(setup, Wav2Vec2ModelSetup)
    proc = setup.load_processor()
    vocab = json.loads((tmp_path / c.model_id / "vocab.json").read_text())
    assert len(vocab) == 42 and vocab["|"] == 36
    tok = proc.tokenizer
    assert (tok.pad_token_id, tok.unk_token_id, tok.eos_token_id, tok.bos_token_id, len(tok)) == (45, 44, 43, 42, 46)
    args = setup.load_training_arguments()
    assert args.gradient_accumulation_steps == 8  # 64 // 1 device // 8
    assert args.metric_for_best_model == "val_coral_v3_read_aloud_cer"
    assert (args.adam_beta1, args.adam_beta2, args.lr_scheduler_type, args.save_strategy) == (0.9, 0.98, "cosine", "no")
    with pytest.raises(ValueError):
        c.model.type = "other"
        load_model_setup(c)


def test_collator_matches_hf(golden_dir):
    from coral_amd.data_collators import DataCollatorCTCWithPadding
    from coral_amd.processor import CTCTokenizer, Wav2Vec2Processor, WaveformFeatureExtractor
    from oracle import wav2vec2_ref as ref

    z = np.load(golden_dir / "collator.npz")
    texts = json.loads((golden_dir / "collator_texts.json").read_text())
    tok = CTCTokenizer({k: v for k, v in ref.coral_vocab().items() if not k.startswith("<")})
    proc = Wav2Vec2Processor(WaveformFeatureExtractor(), tok)
    feats = []
    for i, t in enumerate(texts):
        ids = proc(text=t)["input_ids"]
        assert ids == z[f"lab{i}"].tolist()  # tokenizer encode parity
        feats.append(dict(input_values=z[f"iv{i}"], labels=ids))
    for padding, key in (("longest", "longest"), ("max_length", "max")):
        batch = DataCollatorCTCWithPadding(proc, 16000, 2000 / 16000, padding)(feats)
        np.testing.assert_array_equal(batch["input_values"].numpy(), z[f"{key}_input_values"])
        np.testing.assert_array_equal(batch["attention_mask"].numpy(), z[f"{key}_attention_mask"])
        np.testing.assert_array_equal(batch["labels"].numpy(), z[f"{key}_labels"])
    with pytest.raises(ValueError):
        DataCollatorCTCWithPadding(proc, 16000, 1.0)([{"foo": 1}])


def test_feature_extractor_and_tokenizer_match_hf(golden_dir):
    from coral_amd.processor import CTCTokenizer, WaveformFeatureExtractor

    z = np.load(golden_dir / "feature_extractor.npz")
    fe = WaveformFeatureExtractor()
    feats = [fe(z[f"wave{i}"], 16000) for i in range(4)]
    b = fe.pad(feats, "longest")
    np.testing.assert_allclose(b["input_values"], z["input_values"], atol=2e-6)
    np.testing.assert_array_equal(b["attention_mask"], z["attention_mask"])
    b = fe.pad(feats, "max_length", 2000)
    np.testing.assert_allclose(b["input_values"], z["input_values_max"], atol=2e-6)
    t = json.loads((golden_dir / "tokenizer_collapse.json").read_text())
    tok = CTCTokenizer({k: v for k, v in t["vocab"].items() if not k.startswith("<")})
    assert [tok.decode(r) for r in t["rows"]] == t["texts"]
    with pytest.raises(ValueError):
        fe(np.zeros(10), sampling_rate=8000)


def test_specaugment_bit_exact_with_hf_rng_stream(golden_dir):
    from coral_amd.specaugment import compute_mask_indices

    z = np.load(golden_dir / "specaugment.npz")
    for i in range(int(z["n"])):
        spec = z[f"spec{i}"].tolist()
        seed, b, n, p1000, ml, mm = spec[:6]
        lens = spec[6:] or None
        np.random.seed(seed)
        m = compute_mask_indices((b, n), p1000 / 1000, ml, lens, mm)
        want = np.unpackbits(z[f"m{i}"])[: b * n].reshape(b, n).astype(bool)
        np.testing.assert_array_equal(m, want)
    with pytest.raises(ValueError):
        compute_mask_indices((2, 5), 0.5, 10)


def test_metrics_and_compute_metrics():
    from coral_amd.compute_metrics import compute_error_rate_metrics
    from coral_amd.metrics import cer, wer
    from coral_amd.processor import CTCTokenizer, Wav2Vec2Processor, WaveformFeatureExtractor
    from oracle import wav2vec2_ref as ref

    assert cer(["hej"], ["hej"]) == 0 and abs(cer(["hej mod dig"], ["hej med dig"]) - 1 / 11) < 1e-12
    assert abs(wer(["hej dig"], ["hej med dig"]) - 1 / 3) < 1e-12
    tok = CTCTokenizer({k: v for k, v in ref.coral_vocab().items() if not k.startswith("<")})
    proc = Wav2Vec2Processor(WaveformFeatureExtractor(), tok)
    ids = tok.encode("hej du")
    T, V = 20, 46
    logits = np.full((2, T, V), -5.0, dtype=np.float32)
    seq = [ids[0], ids[0], 45, ids[1], ids[2], 45, ids[3], ids[3], ids[4], ids[5]]
    logits[0, np.arange(len(seq)), seq] = 5
    logits[0, len(seq):, 45] = 5
    logits[1] = -100.0  # pad_across_processes filler row (compute_metrics.py:63-66)
    labels = np.full((2, 8), -100)
    labels[0, :6] = ids
    m = compute_error_rate_metrics(logits, labels, proc)
    assert m["cer"] == 0.0 and m["wer"] == 0.0
    with pytest.raises(ValueError):
        compute_error_rate_metrics(np.zeros(3), labels, proc)


def test_param_store_layout_is_bucketed_and_aligned():
    from coral_amd.wav2vec2 import Wav2Vec2Shape, _r8, w2v2_param_list
    from oracle import wav2vec2_ref as ref

    s = Wav2Vec2Shape(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256)
    plist = w2v2_param_list(s)
    names = [n for n, _, _ in plist]
    assert set(names) == set(ref.param_shapes(ref.W2V2Config(hidden_size=128, num_hidden_layers=2,
                                                             num_attention_heads=4, intermediate_size=256)))
    buckets = [b for _, _, b in plist]
    order = []
    for b in buckets:
        if not order or order[-1] != b:
            order.append(b)
    assert order == ["front", "layer0", "layer1", "head"]  # contiguous buckets in storage order
    # q,k,v weights (and biases) are adjacent so the fused [3d, d] view is one slice
    i = names.index("wav2vec2.encoder.layers.0.attention.q_proj.weight")
    assert names[i + 1].endswith("k_proj.weight") and names[i + 2].endswith("v_proj.weight")
    assert _r8(46) == 48


def test_weight_gradient_launch_plans_of_the_model_shapes():
    """ops.wgrad_plan (host logic of the grouped weight-gradient launches): which problems of a layer get their own
    launch of the 256x256 kernel, which share one, which stay on the split-K path - at the shapes the engines run."""
    from coral_amd import ops

    def plan(shapes):
        solo, groups, fb = ops.wgrad_plan([dict(M=m, N=n, K=k) for m, n, k in shapes])
        return len(solo), sorted(len(g) for g in groups), len(fb)

    def enc(d, f, rows):
        return [(3 * d, d, rows), (d, d, rows), (f, d, rows), (d, f, rows)]

    def dec(d, f, rows):
        return [(3 * d, d, rows), (d, d, rows), (d, d, rows), (d, d, rows), (f, d, rows), (d, f, rows)]

    assert plan(enc(1920, 7680, 3992)) == (2, [2], 0)       # XLS-R-2B: fc1 / fc2 alone (240 tiles each), q|k|v + out together
    assert plan(enc(1024, 4096, 3992)) == (0, [4], 0)       # XLS-R-300M: one launch of 192 tiles
    assert plan(enc(1280, 5120, 3992)) == (0, [3], 1)       # XLS-R-1B: 200 tiles together, one matrix on the 256x128 kernel
    assert plan(dec(1024, 4096, 904)) == (0, [6], 0)        # whisper-medium decoder layer: six problems, 224 tiles, one launch
    assert plan(dec(1280, 5120, 904)) == (0, [6], 0)        # whisper-large-turbo decoder layer: 350 tiles, one launch
    assert plan(enc(768, 3072, 1000)) == (0, [], 4)         # nothing fills the chip: the general path
    assert plan([(1024, 1024, 200)] * 3) == (0, [], 3)      # short contraction: never grouped
    assert ops.GROUP_MAX == 8


def test_import_asks_for_kernel_arguments_in_device_memory():
    """`import coral_amd` sets HIP_FORCE_DEV_KERNARG=1 unless the environment already says otherwise (the HIP runtime
    reads it when it initialises: INTEGRATION.md), in a fresh interpreter."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = str(Path(__file__).resolve().parents[1])
    code = "import os, coral_amd; print(os.environ.get('HIP_FORCE_DEV_KERNARG'))"
    for preset, want in ((None, "1"), ("0", "0")):
        env = {k: v for k, v in os.environ.items() if k != "HIP_FORCE_DEV_KERNARG"}
        if preset is not None:
            env["HIP_FORCE_DEV_KERNARG"] = preset
        env["PYTHONPATH"] = root
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr[-400:]
        assert out.stdout.strip().splitlines()[-1] == want

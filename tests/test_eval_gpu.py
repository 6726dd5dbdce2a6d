"""`scripts/evaluate_model.py` end to end for both model types (BASELINE.json configs[3] is
"whisper ... (evaluate_model.py path)"): finetune the reference's test model keys for two steps, save in HF layout,
then evaluate the saved directory through the script's `main` — the saved architecture selects greedy CTC or
log-mel + greedy generation with language=danish / task=transcribe (R/src/coral/evaluate.py:56-60,123-158;
R/src/scripts/evaluate_model.py:29-65)."""
import csv
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "scripts"))


def test_whisper_finetune_save_then_evaluate_model_script(tmp_path, monkeypatch):
    import evaluate_model
    import finetune_asr_model

    from coral_amd.whisper_setup import prefix_ids

    monkeypatch.chdir(tmp_path)  # the script writes <model>.<dataset>.csv into the working directory
    res = finetune_asr_model.main(["model=test-whisper", "datasets=synthetic", f"models_dir={tmp_path}",
                                   "model_id=wev", "max_steps=2", "total_batch_size=2", "per_device_batch_size=2",
                                   "max_seconds_per_example=2.0", "min_seconds_per_example=1.0", "logging_steps=1",
                                   "eval_steps=2", "model.max_length=12"])
    shape = res["model"].shape
    mdir = tmp_path / "wev"
    scores = evaluate_model.main([f"model_id={mdir}", "batch_size=3", "generation_max_length=10", "dataset=synthetic"])
    assert scores["model_type"] == "whisper" and scores["n"] == 6
    rows = scores["token_ids"]
    assert len(rows) == 6
    for r in rows:
        assert r[:4] == prefix_ids(shape) and 4 < len(r) <= 10   # forced <|sot|><|da|><|transcribe|><|notimestamps|>
        assert r[4] not in (220, shape.eos_token_id)              # begin-suppress set stays on (CoRal clears suppress_tokens only)
    out = list(csv.reader(open(scores["csv"])))
    assert out[0] == ["prediction", "label", "token_ids"] and len(out) == 7
    # the same call again gives the same ids (greedy, no sampling)
    again = evaluate_model.main([f"model_id={mdir}", "batch_size=3", "generation_max_length=10", "dataset=synthetic",
                                 "store_results=false"])
    assert again["token_ids"] == rows


def test_wav2vec2_evaluate_model_script(tmp_path, monkeypatch):
    import evaluate_model
    import finetune_asr_model

    from coral_amd import modeling

    monkeypatch.chdir(tmp_path)
    monkeypatch.setitem(modeling.HUB_SHAPES, "facebook/wav2vec2-xls-r-300m",
                        dict(hidden_size=128, num_hidden_layers=2, intermediate_size=256, num_attention_heads=4))
    finetune_asr_model.main(["model=test-wav2vec2", "datasets=synthetic", f"models_dir={tmp_path}", "model_id=cev",
                             "max_steps=2", "total_batch_size=2", "per_device_batch_size=2",
                             "max_seconds_per_example=2.0", "min_seconds_per_example=1.0", "logging_steps=1",
                             "eval_steps=2"])
    scores = evaluate_model.main([f"model_id={tmp_path / 'cev'}", "batch_size=4", "dataset=synthetic"])
    assert scores["model_type"] == "wav2vec2" and scores["n"] == 8 and 0.0 <= scores["cer"]
    out = list(csv.reader(open(scores["csv"])))
    assert out[0] == ["prediction", "label"] and len(out) == 9

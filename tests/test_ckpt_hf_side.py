"""N2, transformers side (CPU): the tiny checkpoints the engine saved on the GPU box (tests/golden/engine_ckpt_*,
produced by tools/make_engine_ckpt.py) load into HuggingFace Transformers with identical names / shapes, tied
`proj_out`, parametrized weight-norm, and HF's fp32 forward reproduces the engine's outputs
(tools/check_ckpt_with_hf.py).  Skipped where `transformers` is not installed."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_engine_saved_checkpoints_load_in_transformers(golden_dir):
    pytest.importorskip("transformers")
    if not (golden_dir / "engine_ckpt_w2v2").exists():
        pytest.skip("engine-saved fixture not committed yet")
    sys.path.insert(0, str(ROOT / "tools"))
    import check_ckpt_with_hf

    res = check_ckpt_with_hf.main(golden_dir)
    assert set(res) == {"wav2vec2", "whisper"}
    assert res["wav2vec2"]["logits_max_abs_err"] <= 5e-2 and res["whisper"]["logits_max_abs_err"] <= 5e-2

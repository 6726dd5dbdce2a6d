#!/usr/bin/env python3
"""Evaluate a speech model (wav2vec2: greedy CTC; Whisper: greedy generation, language=danish task=transcribe) — key=value surface of R/src/scripts/evaluate_model.py:
    python scripts/evaluate_model.py model_id=models/wav2vec2-small-2026-10-02 batch_size=8
"""
import logging
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from coral_amd.config import load_config  # noqa: E402
from coral_amd.evaluate import evaluate  # noqa: E402


def main(argv=None):
    logging.basicConfig(level=logging.INFO)
    config = load_config("evaluation", list(argv if argv is not None else sys.argv[1:]))
    if config.model_id is None:
        raise SystemExit("model_id=<local model directory> is required")
    scores = evaluate(config)
    print(scores)
    return scores


if __name__ == "__main__":
    main()

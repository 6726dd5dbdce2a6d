"""Character / word error rates (R/src/coral/metrics.py:8-61 uses `jiwer`, unavailable here): total
Levenshtein distance over all pairs divided by the total reference length."""

from __future__ import annotations


def _edit_distance(ref: list, hyp: list) -> int:
    prev = list(range(len(hyp) + 1))
    for i, r in enumerate(ref, 1):
        cur = [i] + [0] * len(hyp)
        for j, h in enumerate(hyp, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (r != h))
        prev = cur
    return prev[-1]


def cer(predictions: list[str], labels: list[str]) -> float:
    errs = sum(_edit_distance(list(lab), list(p)) for p, lab in zip(predictions, labels))
    total = sum(len(lab) for lab in labels)
    return errs / max(1, total)


def wer(predictions: list[str], labels: list[str]) -> float:
    errs = sum(_edit_distance(lab.split(), p.split()) for p, lab in zip(predictions, labels))
    total = sum(len(lab.split()) for lab in labels)
    return errs / max(1, total)

"""Shared check for Whisper greedy decoding against the fp32 oracle under a stated tie-margin policy.

Greedy ids are "bit-exact on fp32-accumulated logits up to ties" (north star): bf16 storage moves a
logit by up to `accept`, so a position whose fp32 top-2 margin is below that noise may legitimately pick either
token - and from there on the two sequences see different contexts.  The policy checked for EVERY row:

  1. every generated token is within `accept` of the oracle's maximum for the engine's OWN prefix (a valid greedy path);
  2. wherever the oracle's top-2 margin exceeds `forced`, the engine's token IS the oracle's argmax (bit-exact);
  3. against the reference sequence (`want`, HF fp32 greedy ids from the fixture / the oracle's own greedy run):
     the first position where the two differ is reported, and there the oracle's top-2 margin - on the common
     prefix - must be at most `forced`: a divergence may only ever START at a near-tie.
"""
import torch


def check_greedy_rows(decoder_rows, ids, want, prefix_len, accept, forced, label=""):
    """decoder_rows(b, seq) -> fp32 oracle logits [len(seq) - 1, V] for sequence `seq` of row b with the suppress
    masks already applied (row t - 1 scores token t).  -> list of (first_divergence or None, margin there)."""
    report = []
    for b, seq in enumerate(ids):
        seq = list(seq)
        lg = decoder_rows(b, seq)
        for t in range(prefix_len, len(seq)):
            row = lg[t - 1]
            top2 = row.topk(2).values
            assert float(row[seq[t]]) >= float(top2[0]) - accept, (label, b, t, seq[t], int(row.argmax()))
            if float(top2[0] - top2[1]) > forced:
                assert seq[t] == int(row.argmax()), (label, b, t, seq[t], int(row.argmax()))
        ref = list(want[b])
        n = min(len(seq), len(ref))
        div = next((t for t in range(n) if seq[t] != ref[t]), None if len(seq) == len(ref) else n)
        margin = None
        if div is not None:
            assert div >= prefix_len, (label, b, "the forced prefix differs", seq[:prefix_len], ref[:prefix_len])
            if div < len(seq):
                top2 = lg[div - 1].topk(2).values  # seq[:div] == ref[:div]: the common prefix's scores
                margin = float(top2[0] - top2[1])
                assert margin <= forced, (label, b, f"sequences part at position {div} where the fp32 top-2 margin "
                                                    f"is {margin:.4f} > {forced}")
        report.append((div, margin))
        print(f"  {label} row {b}: " + ("identical to the reference ids" if div is None else
                                         f"first divergence at token {div} of {len(seq)}, fp32 top-2 margin there "
                                         f"{margin if margin is None else round(margin, 5)}"))
    return report

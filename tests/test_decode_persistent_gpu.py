"""Greedy decoding with ONE persistent launch per token (ca_whisper_decode_token, coral_amd/csrc/decode.hip) against the launch
sequence it replaces (WhisperEngine._token_step_launches: ~7 launches per layer, itself checked against the oracle in
test_whisper_gpu.py / test_fulldepth_gpu.py): per token the logits, the picked tokens, the K|V cache rows written and the
step's bookkeeping are BIT-identical, and whole generations to max_length 225 (`R/config/evaluation.yaml`,
$TF/models/whisper/generation_whisper.py:383) give identical ids at 1, 8 and 16 clips."""
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PREFIX = [50258, 50285, 50359, 50363]


def _engine(model, B):
    import bench

    eng, shape, waves, _ = bench.whisper_setup_engine(model, torch.device(DEV), 0, B)
    return eng, shape, waves


def _state(eng, shape, kv, B, Lmax, sup):
    from coral_amd import ops

    cache = eng.new_decode_cache(B, Lmax)
    g = eng._graph_state(cache, kv, shape.pad_token_id, shape.eos_token_id)
    ids0 = torch.tensor([PREFIX] * B, dtype=torch.int64, device=DEV)
    base = eng.decode_step(ids0, kv, cache).contiguous()
    ops.argmax_masked(base, sup, g["nxt"], B, shape.vocab_size, shape.vocab_size)
    g["tok"].copy_(g["nxt"])
    g["pos"].fill_(len(PREFIX))
    g["klen"].fill_(len(PREFIX) + 1)
    return cache, g


@pytest.mark.parametrize("model,B,ntok", [("whisper-xxsmall", 1, 6), ("whisper-xxsmall", 8, 6), ("whisper-xxsmall", 16, 6),
                                         ("whisper-xsmall", 3, 4), ("whisper-small", 16, 4), ("whisper-medium", 16, 4),
                                         ("whisper-medium", 8, 3), ("whisper-large-turbo", 5, 4), ("whisper-large-turbo", 16, 3)])
def test_persistent_token_step_is_bit_identical_to_the_launch_sequence(model, B, ntok):
    from coral_amd import ops

    eng, shape, waves = _engine(model, B)
    kv = eng.cross_kv(eng.encode(eng.log_mel(waves)))
    V = shape.vocab_size
    sup = torch.zeros(V, dtype=torch.uint8, device=DEV)
    sup[torch.randint(0, V, (500,), device=DEV)] = 1
    ca, ga = _state(eng, shape, kv, B, 4 + 40, sup)
    cb, gb = _state(eng, shape, kv, B, 4 + 40, sup)
    ps = eng._persistent_state(cb, gb, sup)
    assert ps is not None, "ca_whisper_decode_token must take the Whisper family at up to 16 clips on a 256-CU device"
    for t in range(ntok):
        eng._token_step_launches(ca, ga, sup)
        ops.whisper_decode_token(ps["desc"])
        torch.cuda.synchronize()
        assert ps["status"].tolist() == [0, 0, 0, 0], f"the launch gave up (token {t})"
        la, lb = ga["logits"][:, :V], gb["logits"][:, :V]
        assert torch.equal(la.view(torch.int32), lb.view(torch.int32)), f"logits differ at token {t}: {int((la != lb).sum())} values"
        for k in ("nxt", "tok", "pos", "klen", "done", "out"):
            assert torch.equal(ga[k], gb[k]), f"{k} differs at token {t}"
        for li, (x, y) in enumerate(zip(ca["kv"], cb["kv"])):
            assert torch.equal(x, y), f"self-attention cache of layer {li} differs at token {t}"


@pytest.mark.parametrize("B", [1, 8, 16])
def test_generation_to_max_length_gives_the_same_ids_with_one_launch_per_token(B, monkeypatch):
    eng, shape, waves = _engine("whisper-medium", B)
    feats = eng.log_mel(waves)
    monkeypatch.setenv("CA_DECODE_PERSISTENT", "0")
    ref = eng.generate(feats, PREFIX, 225)
    monkeypatch.setenv("CA_DECODE_PERSISTENT", "1")
    got = eng.generate(feats, PREFIX, 225)
    assert len(ref) == B and all(len(r) > 8 for r in ref)
    assert got == ref


def test_the_persistent_step_is_what_generate_runs_up_to_16_clips_and_larger_batches_keep_the_launches(monkeypatch):
    from coral_amd import ops

    calls = []
    real = ops.whisper_decode_token
    monkeypatch.setattr(ops, "whisper_decode_token", lambda d: (calls.append(1), real(d))[1])
    eng, shape, waves = _engine("whisper-xxsmall", 16)
    eng.generate(eng.log_mel(waves), PREFIX, 12)
    assert calls, "generate at 16 clips must go through ca_whisper_decode_token"
    calls.clear()
    eng, shape, waves = _engine("whisper-xxsmall", 17)
    eng.generate(eng.log_mel(waves), PREFIX, 12)
    assert not calls


def test_a_launch_that_cannot_have_every_cu_comes_back_and_says_so():
    """16 idle workgroups holding 96 KiB of LDS each (ca_debug_cu_hog on a side stream, a chain of 2-s holds) sit on 16 CUs
    while the persistent launch starts: its spins are bounded, so it comes back - either having given up (status word set,
    NO token written, nothing advanced) or, if the CUs came free in a gap of the chain, having decoded normally.  With the
    chip free again the same state decodes."""
    import time

    from coral_amd import ops

    B = 8
    eng, shape, waves = _engine("whisper-xxsmall", B)
    kv = eng.cross_kv(eng.encode(eng.log_mel(waves)))
    sup = torch.zeros(shape.vocab_size, dtype=torch.uint8, device=DEV)
    cache, g = _state(eng, shape, kv, B, 4 + 40, sup)
    ps = eng._persistent_state(cache, g, sup)
    ops.whisper_decode_token(ps["desc"])
    torch.cuda.synchronize()
    assert ps["status"].tolist() == [0, 0, 0, 0] and g["pos"].tolist() == [5] * B
    side = torch.cuda.Stream()
    for _ in range(3):
        ops.check(ops.lib().ca_debug_cu_hog(16, 256, 96 * 1024, 2000.0, side.cuda_stream), "ca_debug_cu_hog")
    time.sleep(0.2)
    t0 = time.time()
    ops.whisper_decode_token(ps["desc"])
    torch.cuda.current_stream().synchronize()
    dt = time.time() - t0
    code, pos = int(ps["status"][0]), g["pos"].tolist()
    print(f"\n16 CUs held: the launch came back after {dt:.2f} s with status {code}, pos {pos[:2]}")
    assert dt < 30.0
    assert (code != 0 and pos == [5] * B) or (code == 0 and pos == [6] * B)
    side.synchronize()
    gave_up = code != 0
    ps["status"].zero_()
    ops.whisper_decode_token(ps["desc"])
    torch.cuda.synchronize()
    assert ps["status"].tolist() == [0, 0, 0, 0] and g["pos"].tolist() == [6 if gave_up else 7] * B


def test_finished_rows_take_pad_in_both_paths():
    """The bookkeeping of finished rows (ca_argmax_advance: a finished row records pad, eos sets `done`): with everything
    but eos suppressed every row finishes at the first step; the following steps must record pad, keep `done`, and leave
    the two paths' state identical."""
    from coral_amd import ops

    B = 5
    eng, shape, waves = _engine("whisper-xxsmall", B)
    kv = eng.cross_kv(eng.encode(eng.log_mel(waves)))
    V = shape.vocab_size
    sup = torch.ones(V, dtype=torch.uint8, device=DEV)
    sup[shape.eos_token_id] = 0
    free = torch.zeros(V, dtype=torch.uint8, device=DEV)
    ca, ga = _state(eng, shape, kv, B, 4 + 12, free)
    cb, gb = _state(eng, shape, kv, B, 4 + 12, free)
    ps = eng._persistent_state(cb, gb, sup)
    for t in range(4):
        eng._token_step_launches(ca, ga, sup)
        ops.whisper_decode_token(ps["desc"])
        torch.cuda.synchronize()
        assert ps["status"].tolist() == [0, 0, 0, 0]
        # (from the second step on every clip had finished BEFORE the launch: the persistent launch then does the step's
        # bookkeeping and nothing else - the raw argmax `nxt` of logits nobody computed is not part of it)
        for k in ("nxt", "tok", "pos", "klen", "done", "out")[(1 if t else 0):]:
            assert torch.equal(ga[k], gb[k]), f"{k} differs at token {t}"
    assert bool(gb["done"].all())
    out = gb["out"][:, 5:9].tolist()
    assert all(row[0] == shape.eos_token_id and row[1:] == [shape.pad_token_id] * 3 for row in out), out


def test_generate_decodes_again_as_a_launch_sequence_when_a_persistent_launch_gave_up(monkeypatch):
    """`generate` reads the status word at the end: a launch that gave up (every CU was not free) has written no token,
    so the batch is decoded again through the launch sequence - the same bits - with a warning, and the engine keeps
    that path; CA_DECODE_STRICT=1 raises instead.  The give-up is injected: the status word of a fresh decode state is
    raised before its first launch (the word is sticky)."""
    from coral_amd import ops

    eng, shape, waves = _engine("whisper-xxsmall", 8)
    feats = eng.log_mel(waves)
    monkeypatch.setenv("CA_DECODE_PERSISTENT", "0")
    ref = eng.generate(feats, PREFIX, 40)
    monkeypatch.setenv("CA_DECODE_PERSISTENT", "1")
    real = eng._persistent_state

    def spoiled(cache, g, suppress):
        fresh = "persist" not in g
        ps = real(cache, g, suppress)
        if fresh and ps is not None:
            ps["status"][0] = 5
        return ps

    monkeypatch.setattr(eng, "_persistent_state", spoiled)
    monkeypatch.setenv("CA_DECODE_STRICT", "1")
    with pytest.raises(ops.CoralAmdError):
        eng.generate(feats, PREFIX, 40)
    monkeypatch.delenv("CA_DECODE_STRICT")
    with pytest.warns(UserWarning, match="gave up"):
        got = eng.generate(feats, PREFIX, 40)
    assert got == ref and eng._persistent_off
    calls = []
    monkeypatch.setattr(ops, "whisper_decode_token", lambda d: calls.append(1))
    assert eng.generate(feats, PREFIX, 40) == ref and not calls  # the engine stays on the launch sequence

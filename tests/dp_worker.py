"""One rank of the 2-process engine test (tests/test_dp_gpu.py).  Started through torch.distributed.run; both ranks
share cuda:0 (the GPU box has one device), the process group is gloo.  Runs real `DataParallelTrainer` steps on the
wav2vec2 engine with its shard of the global batch and writes {losses, parameters, gradient norm} for the test."""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def build_case():
    """Engine + global batch shared by the worker and the 1-rank comparison in the test."""
    from coral_amd.wav2vec2 import Wav2Vec2CTCEngine, Wav2Vec2Shape
    from oracle import wav2vec2_ref as ref  # test infrastructure: the seeded parameters and the host featuriser only

    kw = dict(hidden_size=128, num_hidden_layers=3, num_attention_heads=4, intermediate_size=256)
    cfg = ref.W2V2Config(**kw)
    eng = Wav2Vec2CTCEngine(Wav2Vec2Shape(**kw), "cuda:0")
    eng.load_state_dict(ref.synth_params(cfg))
    g = torch.Generator().manual_seed(99)
    lens, lab_lens = [4000, 3600, 3200, 4000], [5, 4, 3, 6]
    waves = [(0.1 * torch.randn(n, generator=g)).numpy() for n in lens]
    labels = torch.full((4, 6), -100, dtype=torch.long)
    for b, L in enumerate(lab_lens):
        labels[b, :L] = torch.randint(0, 42, (L,), generator=g)

    def shard(idx):
        iv, am = ref.zero_mean_unit_var_norm([waves[i] for i in idx])
        return dict(input_values=torch.from_numpy(iv), attention_mask=torch.from_numpy(am).long(), labels=labels[idx])

    return eng, shard


def build_case_whisper():
    """A 2 + 2-layer Whisper (teacher-forced finetune step) and a global batch of four 30-s feature matrices."""
    from coral_amd.whisper import WhisperShape
    from coral_amd.whisper_train import WhisperTrainEngine
    from oracle import whisper_ref as w  # test infrastructure: the seeded parameters only

    kw = dict(d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4, decoder_attention_heads=4,
              encoder_ffn_dim=128, decoder_ffn_dim=128, num_mel_bins=80, vocab_size=200, max_target_positions=32,
              pad_token_id=150, decoder_start_token_id=151, eos_token_id=150)
    eng = WhisperTrainEngine(WhisperShape(**kw), "cuda:0")
    eng.load_state_dict(w.synth_params(w.WhisperConfig(**kw)))
    g = torch.Generator().manual_seed(77)
    feats = torch.randn(4, 80, 3000, generator=g) * 0.5
    labels = torch.randint(0, 150, (4, 9), generator=g)
    labels[1, 6:] = -100
    labels[3, 4:] = -100

    def shard(idx):
        return dict(input_features=feats[idx], labels=labels[idx])

    return eng, shard


def main():
    out_dir, wire, steps = Path(sys.argv[1]), sys.argv[2], int(sys.argv[3])
    zero = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    kind = sys.argv[5] if len(sys.argv) > 5 else "wav2vec2"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo")
    from coral_amd.trainer import DataParallelTrainer, shard_indices

    eng, shard = build_case_whisper() if kind == "whisper" else build_case()
    tr = DataParallelTrainer(eng, learning_rate=1e-3, warmup_steps=0, max_steps=100, max_grad_norm=1.0,
                             compress_grads=(wire == "bf16"), zero_stage=zero)
    assert tr.world == world and tr.overlap and tr.zero == bool(zero)
    if zero:  # the moments: everything replicated except the layers' weight matrices, of which this rank holds 1/world
        sharded = sum(hi - lo for lo, hi in eng.shard_ranges().values())
        assert sharded > 0 and tr.m.numel() == eng.store.numel - sharded + sharded // world
    mb = shard(shard_indices(4, rank, world))
    losses, norms = [], []
    for _ in range(steps):
        losses.append(float(tr.train_step([mb])))
        norms.append(tr.grad_norm())
    tr.finish()
    m_full, v_full = tr.consolidate()  # (sharded: gathers the master parameters and moments; replicated: a no-op)
    torch.cuda.synchronize()
    torch.save(dict(losses=losses, norms=norms, p32=eng.store.p32.cpu(), p16=eng.store.p16.float().cpu(), m=m_full.cpu(),
                    v=v_full.cpu()),
               out_dir / f"rank{rank}.pt")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

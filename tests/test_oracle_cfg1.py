"""BASELINE.json configs[0] (XLS-R-300M shape, 4 x 5 s, fp32 CPU, fwd+bwd incl. CTC): the oracle
against the HF Transformers fixture at the real shape (tests/golden/w2v2_cfg1.npz).  ~40 s of CPU."""
import numpy as np
import pytest
import torch

from oracle import wav2vec2_ref as ref


@pytest.mark.slow
def test_oracle_matches_hf_at_cfg1(golden_dir):
    z = np.load(golden_dir / "w2v2_cfg1.npz")
    cfg = ref.W2V2Config(**ref.CORAL_SHAPES["wav2vec2-small"])
    P = {k: v.requires_grad_(True) for k, v in ref.synth_params(cfg).items()}
    g = torch.Generator().manual_seed(4242)
    waves = []
    for n in z["lens"]:
        x = (0.1 * torch.randn(int(n), generator=g)).clamp(-1, 1)
        waves.append((x / x.abs().max()).numpy())
    iv, am = ref.zero_mean_unit_var_norm(waves)
    torch.set_num_threads(8)
    loss, logits, _ = ref.forward_loss(torch.from_numpy(iv), torch.from_numpy(am).long(),
                                       torch.from_numpy(z["labels"]), P, cfg)
    assert abs(float(loss) - float(z["loss"])) <= 1e-4 * float(z["loss"])
    np.testing.assert_allclose(logits.detach()[:, ::16].numpy(), z["logits_slice"], atol=2e-4)
    loss.backward()
    for key in z.files:
        if key.startswith("gradnorm:"):
            got = float(P[key[9:]].grad.norm())
            assert abs(got - float(z[key])) <= 2e-3 * float(z[key]), (key, got, float(z[key]))

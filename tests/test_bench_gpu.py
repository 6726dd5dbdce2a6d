"""`python bench.py --gpus N` as the driver calls it without a launcher: the parent process starts the N ranks itself
(the reference's one-command launch is `accelerate launch ... finetune_asr_model.py`,
R/src/scripts/finetune_asr_model.py:9-12) and relays their exit code; the JSON line reports the real number of ranks
and the gradient wire format (fp32 = what the reference's DDP reduces).  The GPU box has one device, so the two ranks
share it through the gloo backend (CA_BENCH_SHARE_GPU=1): everything but the RCCL transport is the production path."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("zero", [None, 0, 2])
def test_bench_spawns_its_ranks_and_reports_them(zero):
    """zero = None: the N > 1 default - the headline is BASELINE configs[2]'s replicated DDP (gradient all-reduce) and the
    sharded optimiser (the reference's `--zero-stage 2` launch) follows in the same run as config.also_zero2; 0 / 2: one
    of them alone."""
    env = dict(os.environ, CA_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--model", "wav2vec2-small",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--check-replicas"] + ([] if zero is None else ["--zero-stage", str(zero)])
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    main = [d for d in lines if "metric" in d]
    assert len(main) == 1, r.stdout[-1500:]              # rank 0 prints ONE line
    d = main[0]
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["backend"] == "gloo" and d["grad_wire"] == "fp32" and d["scaling"] == "weak"
    assert d["rccl_ranks"] == 1  # (no RCCL rank ran: the field counts ranks of backend "nccl" only)
    assert d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["steps"] == 2 and d["warmup"] == 1
    assert ("sharded optimiser" in d["config"]["workload"]) == (zero == 2)
    assert ("gradient all-reduce" in d["config"]["workload"]) == (zero != 2)
    z2 = d["config"].get("also_zero2")
    assert (z2 is not None) == (zero is None)
    if z2 is not None:
        assert z2["value"] > 0 and z2["zero_stage_in_effect"] == 2
    spread = [x for x in lines if "replica_param_spread" in x]
    assert len(spread) == (2 if zero is None else 1)
    for sp_line in spread:  # DDP invariant: identical replicas - parameters, bf16 copy AND the AdamW moments
        assert sp_line["replica_param_spread"] == 0.0
        sp = sp_line["replica_spreads"]
        assert set(sp) >= {"p32", "p16", "m", "v"} and all(x == 0.0 for x in sp.values()), sp
    assert ("p16_vs_master" in spread[-1]["replica_spreads"]) == (zero != 0)


def test_the_headline_line_survives_a_secondary_run_that_does_not_finish():
    """N > 1 default: the zero_stage 2 run follows the headline in the same process group.  If it does not finish
    (CA_BENCH_ZERO2_TIMEOUT; here: at once) rank 0 still prints the ONE line, with the headline and an error field."""
    env = dict(os.environ, CA_BENCH_SHARE_GPU="1", CA_BENCH_ZERO2_TIMEOUT="0.001")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--model", "wav2vec2-small",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    main = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{") and "metric" in l]
    assert len(main) == 1, r.stdout[-1500:] + r.stderr[-1500:]
    assert main[0]["value"] > 0 and main[0]["ranks"] == 2
    assert "error" in main[0]["config"]["also_zero2"]


def test_two_ranks_at_the_full_xlsr_2b_shape_with_the_sharded_optimizer():
    """BASELINE configs[2]'s model (wav2vec2-large = XLS-R-2B, 8 x 10 s per rank) on two ranks with the N > 1 default
    (reduce-scatter of the 48 layers' 44 M-element matrix parts, AdamW on 1/2, bf16 all-gather): the real bucket
    sizes, offsets and slice alignments; after three steps both ranks hold the same bf16 compute copy bit for bit.
    (Two processes share the one GPU over gloo: ~82 GB of device memory, ~40 s.)"""
    env = dict(os.environ, CA_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-also", "--no-fwd-bwd", "--check-replicas", "--zero-stage", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    d = [x for x in lines if "metric" in x][0]
    assert d["n_gpus"] == 2 and "sharded optimiser" in d["config"]["workload"] and "wav2vec2-large" in d["metric"]
    assert d["config"]["loss"] > 0 and d["config"]["loss"] == d["config"]["loss"]  # finite
    spread = [x for x in lines if "replica_param_spread" in x]
    assert spread and spread[0]["replica_param_spread"] == 0.0


def test_bench_refuses_more_ranks_than_devices():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("CA_BENCH_SHARE_GPU", None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "64"], env=env, capture_output=True, text=True,
                       timeout=300, cwd=str(ROOT))
    assert r.returncode != 0 and "only" in r.stderr   # never a silent 1-GPU run


def test_bench_line_of_a_single_rank_and_the_one_rank_rccl_diagnostic():
    """The N = 1 line: every field the contract names, the forward+backward-only time beside the whole step, the
    roofline object; and the labelled diagnostic that runs the N > 1 exchange path over an RCCL group of one rank."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    base = [sys.executable, str(ROOT / "bench.py"), "--model", "wav2vec2-small", "--steps", "2", "--warmup", "1",
            "--no-cpu-baseline"]
    r = subprocess.run(base, env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")][-1]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["grad_wire"] is None and d["vs_baseline"] is None
    fb = d["config"]["fwd_bwd"]
    assert 0 < fb["ms_per_step"] < d["ms_per_step"] and fb["value"] > d["value"]   # the optimiser is inside `value`
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    r = subprocess.run(base + ["--one-rank-exchange", "--no-fwd-bwd"], env=env, capture_output=True, text=True,
                       timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    e = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert "DIAGNOSTIC" in e["config"]["workload"] and e["n_gpus"] == 1 and e["value"] > 0
    assert abs(e["config"]["loss"] - d["config"]["loss"]) <= 1e-3 * abs(d["config"]["loss"])  # identity exchange

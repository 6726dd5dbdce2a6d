"""pytest configuration: the `gpu` marker, collection order and shared fixtures."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLDEN = ROOT / "tests" / "golden"

# Hot-path parity first (SURVEY.md §8 rows a), then full-size properties and the finetune entry points, then the
# "next" rows (f) and the opt-in fp8 path: with `-x` a failure in a later row can never hide the rows in front of it.
_ORDER = ["test_kernels_gpu", "test_w2v2_gpu", "test_depth_gpu", "test_trainer_traj_gpu", "test_whisper_gpu",
          "test_decode_persistent_gpu", "test_fulldepth_gpu", "test_fullsize_gpu",
          "test_finetune_gpu", "test_dp_gpu", "test_bench_gpu", "test_eval_gpu", "test_ckpt_gpu", "test_input_pipeline_gpu",
          "test_augment_gpu", "test_fp8_gpu"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i for i, name in enumerate(_ORDER)}

    def key(item):
        return rank.get(Path(str(item.fspath)).stem, len(_ORDER))

    items.sort(key=key)  # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

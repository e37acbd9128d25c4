import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of CPU oracle work next to the GPU run (still part of -m gpu)")


# collection order of the GPU files: parity against the oracle first, the launcher / bench-contract subprocess tests last, so a launcher
# problem can never again stop `-x` before the parity tests of SURVEY.md 8(a)/(f) have run (round 5 lost its whole record that way)
GPU_FILE_ORDER = ("test_gpu_parity", "test_gpu_config_sized", "test_gpu_training", "test_gpu_train_loss", "test_gpu_range")
LAST_FILES = ("test_gpu_bench",)


def _file_rank(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    if name in GPU_FILE_ORDER:
        return GPU_FILE_ORDER.index(name)
    if name in LAST_FILES:
        return len(GPU_FILE_ORDER) + 1 + LAST_FILES.index(name)
    return len(GPU_FILE_ORDER)


def pytest_collection_modifyitems(config, items):
    """order (see GPU_FILE_ORDER; stable, so the order inside a file is kept); and a plain `pytest tests` on a host without a GPU
    skips the gpu tests instead of failing them"""
    items.sort(key=_file_rank)
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="needs an MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def synthetic_sd():
    from oracle import weights
    return weights.synthetic_state_dict(13, 9, seed=0)

import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests must fail loudly, not skip, when no GPU / no HIP extension is present on a GPU box;
    # without `-m gpu` (CPU container) they are deselected by the marker expression.
    pass


@pytest.fixture(scope="session")
def device():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")

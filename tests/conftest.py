import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def libprl():
    """libprl.so, built on demand (hipcc cross-compiles without a GPU)."""
    from pipelinerl_amd import _lib
    from pipelinerl_amd.build import LIB_PATH, build

    if not LIB_PATH.exists():
        build()
    return _lib.load()


@pytest.fixture(scope="session")
def cuda_device():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu but no HIP device is visible")
    return torch.device("cuda", 0)

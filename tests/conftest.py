import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLD = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from cryovit_amd import _lib

    _lib.load()
    arch = _lib.device_arch()
    assert arch.startswith("gfx950"), f"expected an MI355X (gfx950), found {arch!r}"
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def gold():
    import numpy as np

    def load(name):
        return np.load(GOLD / name)

    return load

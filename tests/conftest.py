import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    """Initialises the engine on cuda:0; GPU tests must run the HIP path or fail (no fallback)."""
    import torch
    import helpers
    assert torch.cuda.is_available(), "GPU test on a box without a GPU"
    p = helpers.pkg()
    arch = p.gpu_init(0)
    assert arch.startswith("gfx950"), arch
    return p

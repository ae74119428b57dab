import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def ag_knobs(monkeypatch):
    """set experiment knobs of the HIP library (AG_GEMM_* ...) for one test: the library caches them, so every change is
    followed by ag_reload_knobs(), also when the test's environment is restored."""
    from autognothi_amd import ops

    def set_(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, str(v))
        ops.reload_knobs()
    yield set_
    monkeypatch.undo()
    ops.reload_knobs()


@pytest.fixture(scope="session")
def cuda_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _oracle_uses_device_rope_freqs():
    """On a GPU box the CPU oracle takes theta_j = powf(theta, -2j/128) from the device, as the reference
    evaluates it there (oracle/kvq_oracle.c: kvqo_rope_freq); on CPU-only runs it keeps the host libm."""
    try:
        import torch
        if torch.cuda.is_available():
            from tests import util
            util.sync_oracle_freqs(10000.0)
    except Exception as e:   # a missing library must fail in the tests that need it, not here
        print("conftest: oracle keeps host powf (%s)" % e)
    yield

import os
import sys

import pytest

# torch bundles its own HIP runtime: load it BEFORE libsmcx.so pulls one in, so that the
# process holds a single libamdhip64 (two runtimes in one process cannot both see the GPU)
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def S():
    import smcx_loader
    return smcx_loader.load()

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The product .so is git-ignored: build it (hipcc cross-compiles without a GPU) if a fresh checkout lacks it."""
    import polystokes_amd
    if not os.path.exists(polystokes_amd.LIB_PATH):
        polystokes_amd.build()


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import ps_oracle
    ps_oracle.build()
    return ps_oracle

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement of the reference hot path (oracle/gravomg_oracle.c), built on demand."""
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def cabi():
    from gravo_mg_amd import cabi as c
    if not os.path.exists(c.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return c

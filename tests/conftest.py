import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The multi-rank tests put several ranks on the ONE device of the test box (functional checks of the N > 1 path); gmg_p2p_connect refuses that
# unless told it is meant (include/gravomg_hip.h).  Spawned workers inherit the variable; test_gpu_p2p.py removes it in one test.
os.environ.setdefault("GMG_P2P_SHARED_DEVICE", "1")


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

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def co():
    """The C oracle (checker)."""
    from oracle import coracle

    coracle.lib()
    return coracle


@pytest.fixture(scope="session")
def pr():
    from oracle import pyref

    return pyref


@pytest.fixture(scope="session")
def ps_api():
    """The product's host mirror; importing it loads libplaysnark_hip.so (fails loudly if absent)."""
    from playsnark_amd import api

    return api


@pytest.fixture(scope="session")
def ctx(ps_api):
    c = ps_api.Context(0)
    yield c
    c.close()

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.bindings import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def engine8():
    from x265_amd.engine import Engine
    e = Engine(0, 8)   # raises when the HIP library or the GPU is missing: no fallback
    yield e
    e.close()


@pytest.fixture(scope="session")
def engine10():
    from x265_amd.engine import Engine
    e = Engine(0, 10)
    yield e
    e.close()

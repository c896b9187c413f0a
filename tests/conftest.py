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
    """The CPU oracle module (builds oracle/liboracle.so with gcc on first use)."""
    from oracle import oracle_py

    oracle_py.load()
    return oracle_py


@pytest.fixture(scope="session")
def knh():
    """The product library; built in-tree by `python -m knaster_amd.build` / __graft_entry__.build()."""
    import knaster_amd
    from knaster_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        from knaster_amd import build

        build.build()
    _lib.load()
    return knaster_amd

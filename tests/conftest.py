import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def agslib():
    """The product library; built in-tree if the .so is stale (hipcc cross-compiles)."""
    from active_gs_amd import build
    build.build()
    from active_gs_amd import _lib
    return _lib.load()

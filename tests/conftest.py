"""pytest configuration: the `gpu` marker and the import path shared by every test file."""
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
def orc():
    """The CPU oracle (oracle/liboracle.so)."""
    import orc as _orc
    return _orc


@pytest.fixture(scope="session")
def rta():
    """The product package (librtx.so)."""
    import __graft_entry__ as entry
    return entry.load_package()

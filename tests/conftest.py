import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build libcsolve_hip.so and liboracle.so if they are not there yet."""
    import __graft_entry__ as g
    from csolve_amd import _lib
    if not os.path.exists(_lib.LIB_PATH) or not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        g.build()
    yield


def golden(*parts):
    return os.path.join(GOLDEN, *parts)

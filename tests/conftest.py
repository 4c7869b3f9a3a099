import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")


def pytest_collection_modifyitems(config, items):
    """A GPU test that hangs (a kernel that never returns) must not sit on the box until the caller's limit: every `gpu` test carries a
    10-minute pytest-timeout with the THREAD method (a timer thread ends the process; the signal method cannot interrupt a host thread that is
    blocked inside hipStreamSynchronize).  The slowest GPU test takes under a minute."""
    try:
        import pytest_timeout  # noqa: F401
    except ImportError:
        return
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600, method="thread"))

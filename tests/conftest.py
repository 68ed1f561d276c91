import os
import sys

import pytest

# torch ships its own copy of the HIP runtime: it has to be the first one loaded into the process, otherwise a
# test file that loads libx265_amd.so before touching torch ends up with two runtimes and torch sees no device
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "missing_prerequisite(reason): set by tests/prereq.py under XMI_REQUIRE_REF=1")


def pytest_runtest_setup(item):
    m = item.get_closest_marker("missing_prerequisite")
    if m:
        pytest.fail(f"XMI_REQUIRE_REF=1 and {m.args[0]}", pytrace=False)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")

import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def apply_test_options():
    """PRHF_TEST_OPTIONS="name=value,..." (set by the tests that re-run a subset under other launch settings):
    applied through the library's explicit option call - the library itself reads no environment variable."""
    spec = os.environ.get("PRHF_TEST_OPTIONS", "")
    if spec:
        from pyrayhf_amd import library
        for item in spec.split(","):
            name, value = item.split("=")
            library.set_option(name.strip(), float(value))


@pytest.fixture(scope="session", autouse=True)
def _test_options():
    if os.environ.get("PRHF_TEST_OPTIONS"):
        apply_test_options()
    yield


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def same_bits(a, b):
    """Bitwise equality with NaN == NaN."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))

"""pytest configuration: registers the `gpu` marker and shared fixtures.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol checks,
gloo world_size-2 tests.  `-m gpu`: parity tests proper, through the C ABI.
"""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def kats():
    with open(os.path.join(GOLDEN, "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.build()
    return orc

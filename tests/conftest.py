"""pytest configuration: marker registration and import paths.

CPU suite:  python -m pytest tests/ -x -q -m "not gpu"
GPU suite:  python -m pytest tests/ -x -q -m gpu     (needs an MI355X and the built .so)
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "send-slam_amd")
for p in (ROOT, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import orb_oracle
    orb_oracle.lib()
    return orb_oracle


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

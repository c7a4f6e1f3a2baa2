"""pytest config: registers the ``gpu`` marker and makes the repo root importable.

``-m "not gpu"`` runs here (no GPU): oracle vs golden vectors, host logic, C-ABI symbol checks.
``-m gpu`` runs on an MI355X box: HIP-vs-oracle parity through the C-ABI.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

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


def experimental_library() -> bool:
    """True when paroquant_amd/_lib/libparo_mi355x.so was built with ``make EXPERIMENTAL=1`` (the persistent decode engines,
    include/paro_abi_experimental.h).  They are outside the default library and the default test run."""
    try:
        from paroquant_amd import _native
        return _native.has_experimental()
    except Exception:
        return False


needs_experimental = pytest.mark.skipif(not experimental_library(), reason="needs `make -C paroquant_amd/csrc EXPERIMENTAL=1` (persistent engines)")

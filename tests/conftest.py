import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("lap-time-optimization_amd")


@pytest.fixture(scope="session")
def tables(pkg):
    return pkg.TrackTables.load_npz(os.path.join(GOLDEN, "tables_buckmore_mx5_curvature.npz"))


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def oracle(orc, tables):
    return orc.Oracle(tables.packed())


@pytest.fixture(scope="session")
def gpu_lib(pkg):
    """Build if needed (no-op on the GPU box when the .so travelled with the snapshot) and load the HIP library."""
    pkg.build_library()
    return importlib.import_module("lap-time-optimization_amd._lib").lib()

import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pcp():
    return importlib.import_module("point-cloud-process_amd")


@pytest.fixture(scope="session")
def oracle():
    return importlib.import_module("oracle.oracle_np")


@pytest.fixture(scope="session")
def syn():
    return importlib.import_module("point-cloud-process_amd.synthetic")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def ctx(pcp):
    """A libpcr context on device 0; GPU tests only."""
    return pcp.default_context(0)

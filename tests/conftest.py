import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def msm_pkg():
    """The product package (directory name has hyphens, so import it through importlib)."""
    return importlib.import_module("metal-msm-gpu-acceleration_amd")


@pytest.fixture(scope="session")
def cfg(msm_pkg):
    """One MsmConfig (≙ setup_metal_state(), msm.rs:77) for the whole GPU session."""
    c = msm_pkg.setup_metal_state()
    yield c
    c.close()

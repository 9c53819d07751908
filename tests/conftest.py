import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build the oracle and the host simulation (CPU artefacts) once per session."""
    import subprocess

    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hostsim")])
    return True


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "golden_real.npz"))


@pytest.fixture(scope="session")
def sim_engine(built):
    """TEST-ONLY serial simulation of the device code (tests/hostsim): lets the CPU tier exercise the device
    state machines.  The package itself never loads this library."""
    import psd_amd

    return psd_amd.Engine(libpath=os.path.join(ROOT, "tests", "hostsim", "_build", "libpsd_hostsim.so"))


@pytest.fixture(scope="session")
def gpu_engine():
    # torch bundles its own HIP runtime: when torch is used in the same process (device-resident test), it has to
    # initialise first, otherwise it reports "No HIP GPUs are available" ("Mixing with PyTorch").
    import torch

    torch.cuda.init()
    import psd_amd

    return psd_amd.Engine(device=0)

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DM_ALLOW_EMULATOR"] = "1"     # the CPU emulator build is loadable from the test harness only (deepmimic_amd/core.py)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle_built():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return True


@pytest.fixture(scope="session")
def emu_lib(oracle_built):
    """CPU fiber-emulator build of the product's host+device sources (test infrastructure)."""
    d = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-s", "-C", d], stderr=subprocess.DEVNULL)
    return os.path.join(d, "libdm_emu.so")


@pytest.fixture(scope="session")
def hip_lib(oracle_built):
    """The product library; GPU tests fail loudly when it is missing or no device is visible."""
    p = os.path.join(ROOT, "deepmimic_amd", "csrc", "libdm_hip.so")
    assert os.path.exists(p), "libdm_hip.so missing: run __graft_entry__.build()"
    # torch and libdm_hip.so share one HIP runtime in this process; bring torch's device context up first (the order bench.py
    # uses) so that tests mixing torch tensors with the C-ABI (policy, closed loop) do not depend on which test ran before them
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda")
    except ImportError:
        pass
    return p

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): compiled on demand with gcc."""
    from oracle import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def gpu():
    """The product library; fails loudly when the HIP extension or the device is missing."""
    from orb_slam2_map_amd import lib
    lib.lib()
    assert lib.device_count() >= 1, "no HIP device visible: GPU tests cannot run (there is no CPU fallback)"
    return lib


@pytest.fixture(scope="session")
def stream640():
    from orb_slam2_map_amd.synth import Stream
    return Stream(640, 480, 1234)


@pytest.fixture(scope="session")
def stream1280():
    from orb_slam2_map_amd.synth import Stream
    return Stream(1280, 960, 1234)


def corners_to_array(c):
    if len(c) == 0:
        return np.zeros((0, 3), np.int32)
    return np.stack([c["x"], c["y"], c["response"]], 1).astype(np.int32)

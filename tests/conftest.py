import os
import sys

import pytest

# PyTorch-ROCm bundles its own libamdhip64 / libhsa-runtime64.  Import it before anything dlopens liblivesgpu.so so
# the process ends up with ONE HIP runtime (two runtimes in one process cannot both own the device).
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    from oracle import pyoracle
    return pyoracle.oracle()


@pytest.fixture(scope="session")
def gpu():
    """initialised GPU library; fails (does not skip) if the HIP extension is missing on a GPU box"""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test on a box without a GPU"
    from lives_amd import lib, ops
    lib.load()
    ops.init(0)
    return ops


@pytest.fixture
def tune():
    """set launch-shape switches of the library for one test (lgpu_tuning_set); every switch touched is restored afterwards"""
    from lives_amd import ops
    saved = {}

    def set_(name, value):
        old = ops.tuning(name, value)
        saved.setdefault(name, old)
    yield set_
    for name, old in saved.items():
        ops.tuning(name, old)

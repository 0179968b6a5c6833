import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="module")
def engine_factory():
    """Engines through the C ABI.  On a GPU box a missing libnbe.so is a hard failure, never a skip.
    (Module scope: an engine keeps its workspace -- tens of GB after a 224^3 sub-box -- until it is closed, and the full-size
    tests of later modules plan their tiles into the memory that is free.)"""
    from jax_nbody_emulator_with_dj_amd.engine import Engine
    made = []

    def make(**kw):
        e = Engine(**kw)
        made.append(e)
        return e

    yield make
    for e in made:
        e.close()


@pytest.fixture(autouse=True, scope="module")
def _release_cached_engines():
    """The API shim caches one engine per (device, architecture, arithmetic) with its workspace -- up to ~276 GB after a
    full-size box.  Hand the card back after every test module so that later tests plan their tiles on a free card."""
    yield
    try:
        from jax_nbody_emulator_with_dj_amd import models
        models.release_engines()
    except Exception:
        pass


@pytest.fixture
def direct_kernels(monkeypatch):
    """Tests that compare two SCHEDULES of the same box bit for bit (tile merging, roll equivariance) run on the direct gauged
    kernel: conv_h3w_kernel (Winograd F(2,3) along z, the default for the blocks' conv_0 layers) pairs planes from the first
    plane of a launch and applies to an even number of planes only, so a voxel's rounding depends on where a tile or slab
    starts.  The schedules themselves do not depend on the kernel; tests/test_gpu_api.py::
    test_winograd_schedules_agree_to_rounding holds the default against them at float32 tolerances."""
    monkeypatch.setenv("NBE_WINO", "0")


def rel_l2(a, b):
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def max_over_rms(a, b):
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.sqrt(np.mean(b * b)), 1e-300))

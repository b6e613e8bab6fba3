import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def hip_lib():
    """the product library; GPU tests fail loudly (never skip to a CPU path) if it is missing"""
    from picles_amd import _capi
    return _capi.load()


@pytest.fixture(scope="session", autouse=True)
def _natives_built():
    """a fresh clone has no binaries: build the checker (oracle) and the product library (hipcc cross-compiles gfx950
    without a GPU; make is a no-op when they are current).  Building is all that happens here — the product has no CPU
    path and the GPU tests fail loudly without a device."""
    import subprocess
    import _oracle
    _oracle.build()
    subprocess.run(["make", "-s", "-C", str(ROOT / "picles_amd" / "csrc"), "libpicles_hip.so"], check=True)

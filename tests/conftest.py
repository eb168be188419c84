import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # before torch loads the HIP runtime (markushgrapher_amd/__init__.py)
import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU tier (`-m "not gpu"`: the SIMT emulator runs the kernels thread by thread) on a few worker processes when pytest-xdist is there:
    14 minutes in one process, 4.5 on four.  Never for the GPU tier (one GPU), never when the caller passed -n itself; MG_TESTS_WORKERS=0 / =N
    overrides.  The emulator library is built once behind a file lock (csrc/build.py), so the workers may all ask for it at once."""
    if os.environ.get("PYTEST_XDIST_WORKER") or hasattr(config, "workerinput") or not config.pluginmanager.hasplugin("xdist"):
        return None
    if "not gpu" not in (getattr(config.option, "markexpr", "") or "") or getattr(config.option, "numprocesses", None) is not None:
        return None                                   # (-n given, even -n 0: the caller chose)
    want = os.environ.get("MG_TESTS_WORKERS")
    n = int(want) if want is not None else min(4, os.cpu_count() or 1)
    if n > 1:
        config.option.numprocesses = n
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` via gpurun)")
    config.addinivalue_line("markers", "slow: long CPU test")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden

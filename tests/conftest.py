import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_addoption(parser):
    parser.addoption("--emu", action="store_true", default=False,
                     help="development aid: run the tests (also the -m gpu ones) against the host-interpreted "
                          "build of the kernel sources in tests/emu instead of a GPU (slow; small sizes only)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    if config.getoption("--emu"):
        from _pytest.monkeypatch import MonkeyPatch
        sys.path.insert(0, os.path.join(REPO, "tests", "emu"))
        import emulated
        config._emu_patch = MonkeyPatch()
        emulated.install(config._emu_patch)


def pytest_unconfigure(config):
    patch = getattr(config, "_emu_patch", None)
    if patch is not None:
        patch.undo()


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load

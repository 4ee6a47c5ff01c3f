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
def shdr():
    """The product package (directory name has a hyphen, so import it by string)."""
    return importlib.import_module("singlehdr-tf2_amd")


def _reload_switches():
    """libshdr caches its SHDR_* environment switches per process (shdr_config_reload, include/shdr.h)"""
    try:
        importlib.import_module("singlehdr-tf2_amd")._lib.load().shdr_config_reload()
    except Exception:       # the library is not built: the tests that need it fail on their own
        pass


@pytest.fixture(autouse=True)
def _fresh_switches():
    """every test starts from the environment as it is NOW (the previous test's monkeypatched switches are undone after its teardown)"""
    _reload_switches()
    yield


@pytest.fixture
def monkeypatch(monkeypatch):
    """the built-in fixture, with setenv / delenv followed by a reload of the library's cached switches"""
    class _Patch:
        def __getattr__(self, name):
            return getattr(monkeypatch, name)

        def setenv(self, *a, **kw):
            monkeypatch.setenv(*a, **kw)
            _reload_switches()

        def delenv(self, *a, **kw):
            monkeypatch.delenv(*a, **kw)
            _reload_switches()
    return _Patch()


@pytest.fixture(params=[False, True], ids=["planned", "split_forced"])
def split_forced(request, monkeypatch, shdr):
    """Run the test a second time with the split-operand kernels (plans x3 / x3n) FORCED onto layers that are too small for them to
    pay: the plan's fill-the-chip thresholds are for speed only, and at the 64 x 64 fixtures of the oracle tests a layer has 16 - 48
    blocks, so by default these tests exercise the Winograd / LDS-DMA kernels while the benchmark runs x3 / x3n."""
    if request.param:
        monkeypatch.setenv("SHDR_X3_MIN_BLOCKS", "1")
        K = shdr._ops
        assert K.conv2d_plan((1, 64, 64, 64), (3, 3, 64, 64)) == "x3" and K.conv2d_plan((1, 64, 64, 16), (7, 7, 16, 16)) == "x3n"
    return request.param


@pytest.fixture(scope="session")
def emor_table():
    return np.load(os.path.join(ROOT, "singlehdr-tf2_amd", "data", "invemor_g0_hinv11.npy"))


def quantised_image(rng, shape):
    """SURVEY.md section 8d input law: round(U[0,1)*255)/255."""
    return np.round(rng.random(shape) * 255.0) / 255.0


def rel_err(a, b):
    """max |a-b| / max |b| (tensor-scale relative error)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))

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

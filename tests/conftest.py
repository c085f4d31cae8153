import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)

PKG = "ofa-for-super-resolution_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def amd(sub=None):
    """import the product package (its directory name has hyphens, so go through importlib)."""
    return importlib.import_module(PKG if sub is None else "%s.%s" % (PKG, sub))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
        return cache[name]

    return load


@pytest.fixture(scope="session")
def ora():
    from oracle import oracle
    oracle.build()
    return oracle


def assert_close(a, b, rtol=1e-5, atol=1e-6, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, "%s shape %s vs %s" % (what, a.shape, b.shape)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    if not np.all(err <= tol):
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError("%s: max violation at %s: got %r expected %r (|err|=%g, tol=%g); max|err|=%g"
                             % (what, i, a[i], b[i], err[i], tol[i], err.max()))

import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no binaries (they are not in the history): build them once, as __graft_entry__.build() does."""
    needed = [os.path.join(ROOT, "wurblpt_amd", "lib", "libwurblpt_hip.so"), os.path.join(ROOT, "wurblpt_amd", "lib", "libwurblpt_host.so"),
              os.path.join(ROOT, "oracle", "liboracle.so"), os.path.join(ROOT, "oracle", "liboracle_libm.so")]
    if not all(os.path.exists(p) for p in needed):
        import __graft_entry__
        __graft_entry__.build()


def _bits_to_f32(seq):
    return np.array([int(s, 16) for s in seq], dtype=np.uint32).view(np.float32)


class Golden:
    """tests/golden/ref_golden.json: vectors produced by the reference's own headers
    (oracle/ref_probe.cpp).  Floats are stored as hex bit patterns."""

    def __init__(self, path):
        with open(path) as f:
            self.raw = json.load(f)

    def f32(self, key):
        return _bits_to_f32(self.raw[key])

    def i64(self, key):
        return np.array(self.raw[key], dtype=np.int64)

    def has(self, key):
        return key in self.raw


@pytest.fixture(scope="session")
def golden():
    return Golden(os.path.join(ROOT, "tests", "golden", "ref_golden.json"))


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_loader
    return oracle_loader.load("portable")


@pytest.fixture(scope="session")
def oracle_libm():
    from tests import oracle_loader
    return oracle_loader.load("libm")

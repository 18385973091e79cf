import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The HIP library is git-ignored (it ships to the GPU box with the snapshot): build it for gfx950 if this
    checkout does not have it yet (hipcc cross-compiles without a GPU, about a minute).  The product code itself never
    builds on demand -- it fails loudly when the library is missing."""
    so = os.path.join(ROOT, "dps_ttc_amd", "lib", "libdpsx.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    yield


class Fixture:
    """Lazy view over one tests/golden/*.npz (allow_pickle=False: data only)."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    def __getitem__(self, key):
        return self._z[key]

    def keys(self):
        return list(self._z.keys())


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Fixture(name + ".npz")
        return cache[name]

    return get


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.build()
    return orc

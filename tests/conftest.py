import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the library honours its test hooks (tinyimgcodec_amd/csrc/tic_hooks.h) only if this is set when it is first used
os.environ.setdefault("TIC_TEST_HOOKS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rand_frame(seed, h, w):
    """The seeded synthetic frame every golden and bench config uses."""
    return np.random.default_rng(seed).integers(0, 256, (h, w), dtype=np.uint8)


@pytest.fixture(scope="session")
def manifest():
    import json

    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)["entries"]


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
        return cache[name]

    return load


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle

    pyoracle.build()
    return pyoracle

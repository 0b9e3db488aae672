import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "autostyle-tts_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def real_bank():
    import numpy as np

    return np.load(os.path.join(GOLDEN, "style_bank_130x6144.f16.npy"))


@pytest.fixture(scope="session")
def kats():
    import json

    with open(os.path.join(GOLDEN, "knn_kats.json")) as f:
        return json.load(f)

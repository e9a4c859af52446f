import importlib
import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")
PKG = "pytorch-human-pose_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module(PKG)


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module(PKG + ".synth")


@pytest.fixture(scope="session")
def decode_golden():
    meta = json.load(open(os.path.join(GOLDEN, "decode_meta.json")))
    data = np.load(os.path.join(GOLDEN, "decode.npz"))
    return meta, data


@pytest.fixture(scope="session")
def net_golden():
    return np.load(os.path.join(GOLDEN, "net_forward.npz"))

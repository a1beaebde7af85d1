import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def qp_fixtures():
    with open(os.path.join(ROOT, "tests", "golden", "qp_fixtures.json")) as f:
        fx = json.load(f)
    for d in fx:
        for k in ("P", "A", "q", "l", "u", "x", "y"):
            if d[k] is not None:
                d[k] = np.array(d[k], float)
    return {d["name"]: d for d in fx}


@pytest.fixture(scope="session")
def builder_kats():
    with open(os.path.join(ROOT, "tests", "golden", "constraint_builder_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def gpu_available():
    import torch
    return torch.cuda.is_available()

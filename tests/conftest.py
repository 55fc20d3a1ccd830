import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("a test marked `gpu` ran without a GPU; select with -m gpu only on the GPU box")
    torch.backends.cuda.matmul.allow_tf32 = False
    return torch.device("cuda:0")


def load_golden(name):
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))

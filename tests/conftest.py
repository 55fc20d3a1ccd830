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


def golden_batch(case, g):
    """(input_ids, attention_mask, token_type_ids, labels, images) CPU tensors of a full-model golden case.  Compact fixtures
    (full-size cases, e.g. BASELINE configs[0]) do not store the images: they are regenerated from the case's seed with the
    generator that made them (oracle.d2r_oracle.synthetic_batch) and checked against the stored sums and a strided probe."""
    import numpy as np
    import torch
    keys = ("input_ids", "attention_mask", "token_type_ids", "labels")
    head = [torch.from_numpy(np.asarray(g[k])) for k in keys]
    if "images" in g:
        return head + [torch.from_numpy(np.asarray(g["images"]))]
    from oracle import d2r_oracle as O
    regen = O.synthetic_batch(case.cfg(), case.B, case.L, seed=case.seed)
    for a, b in zip(head, regen[:4]):
        assert torch.equal(a, b), "the synthetic-batch generator drifted from the one that made the fixture"
    images = regen[4]
    assert abs(float(images.double().sum()) - float(g["images_sum"])) < 1e-6 and \
        abs(float(images.double().abs().sum()) - float(g["images_abs_sum"])) < 1e-6, "regenerated images differ from the fixture's"
    assert torch.equal(images[:, :, ::37, ::41], torch.from_numpy(np.asarray(g["images_probe"]))), "regenerated images differ"
    return head + [images]

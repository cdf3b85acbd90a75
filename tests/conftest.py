"""pytest configuration: the ``gpu`` marker and shared fixtures."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _seed():
    torch.manual_seed(333)
    np.random.seed(333)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def _real64(t: torch.Tensor) -> torch.Tensor:
    t = t.detach().cpu()
    if t.is_complex():               # BEFORE any dtype cast: .double() on a complex tensor drops the imaginary part
        t = torch.view_as_real(t.resolve_conj())
    return t.double()


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = _real64(a), _real64(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    den = b.norm().item()
    return (a - b).norm().item() / (den if den > 0 else 1.0)

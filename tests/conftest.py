import os
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=False)
        return cache[name]

    return load


def rel_err(a, b):
    """max|a-b| / max|b| — the fp32 parity measure used throughout (north_star: 1e-4)."""
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    denom = b.abs().max().clamp_min(1e-30)
    return ((a - b).abs().max() / denom).item()


def phase_err(a, b):
    """max |e^{ia} - e^{ib}| — phases compared modulo 2*pi (SURVEY §7 hard part ii)."""
    return (torch.exp(1j * a) - torch.exp(1j * b)).abs().max().item()

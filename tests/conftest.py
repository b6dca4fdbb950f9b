"""pytest configuration: registers the `gpu` marker and shared helpers."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(sub: str = ""):
    """Import the (hyphenated) product package or one of its submodules."""
    name = "llm-speech-summarization_amd" + (("." + sub) if sub else "")
    return importlib.import_module(name)


def golden(name: str):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def t(a):
    return torch.from_numpy(np.asarray(a))


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.double()
    b = b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="session")
def has_gpu():
    return torch.cuda.is_available()

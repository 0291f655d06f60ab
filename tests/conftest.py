"""pytest configuration: `gpu` marker, import paths, golden-fixture loader.

`python -m pytest tests -m "not gpu"` runs on the CPU-only build container (oracle vs golden vectors, host
logic, C-ABI load/exports); `-m gpu` runs the parity tests proper on an MI355X through the C ABI.
"""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        with np.load(GOLDEN / f"{name}.npz") as z:
            return {k: z[k] for k in z.files}

    return load


@pytest.fixture(scope="session")
def dev():
    return torch.device("cuda:0")

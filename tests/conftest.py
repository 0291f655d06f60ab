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


@pytest.fixture(scope="session")
def record():
    """Append one JSON line of measured parity numbers to gpurun_out/parity_numbers.jsonl (merged back from the GPU box; the round's
    copy is committed under profiles/): `pytest -q` prints nothing of a passing test, and a bar only says "below"."""
    import json

    out = ROOT / "gpurun_out"

    def rec(test: str, **kv):
        try:
            out.mkdir(exist_ok=True)
            with open(out / "parity_numbers.jsonl", "a") as f:
                f.write(json.dumps({"test": test, **kv}) + "\n")
        except OSError:
            pass

    return rec

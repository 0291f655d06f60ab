"""Test-harness-only stub of the `lightning` package.

The reference (/root/reference) imports `lightning` at module import time
(lit_llama/utils.py:15, generate.py:9) but the package is not installed in this
image and there is no network.  Only the names touched at import time are
provided; nothing here is used by the product path.
"""
import random

import numpy as np
import torch

from . import fabric  # noqa: F401


def seed_everything(seed: int) -> int:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    return seed


class Fabric:  # placeholder; generate.main() is never called by the harness
    def __init__(self, *a, **k):
        raise RuntimeError("lightning stub: Fabric is not available")

"""Boundary selector and small helpers of the hot path.

This is the ~50-line plug-in selector whose behaviour IS the drop-in contract, so about half of its lines are the same
as /root/reference lit_llama/utils.py — `llama_model_sizes`, `llama_model_lookup`, `find_multiple` (:19-41) and the
`__enter__` / `__exit__` / `__torch_function__` bodies of `EmptyInitOnDevice` (:73-138) and `quantization` (:141-162)
are kept as they are there on purpose (host-side Python, off the hot path, nothing to re-design): while the context is
active `torch.nn.Linear` is rebound to the quantised class, so `LLaMA.from_name(...)` builds every linear through the
plug-in.  What differs: the quantised classes are this package's (HIP kernels, no bitsandbytes / Triton import).
`lazy_load` (:332-344) is re-exported from checkpoint.py (a reader of the torch.save zip format itself);
`incremental_save` is checkpoint-conversion tooling and out of scope.
"""
from __future__ import annotations

import functools
from contextlib import contextmanager

import torch
import torch.utils._device

from .checkpoint import LazyTensor as NotYetLoadedTensor  # noqa: F401  (the reference's name for a lazy checkpoint entry)
from .checkpoint import lazy_load  # noqa: F401  (same name and use as lit_llama.utils.lazy_load)

llama_model_sizes = {
    4096: "7B",  # 7B n_embd=4096
    5120: "13B",  # 13B n_embd=5120
    6656: "30B",  # 30B n_embd=6656
    8192: "65B",  # 65B n_embd=8192
}


def llama_model_lookup(checkpoint: dict) -> str:
    """Model name from the width of the embedding matrix (lit_llama/utils.py:29-35)."""
    embedding_size = checkpoint["transformer.wte.weight"].shape[1]
    return llama_model_sizes[embedding_size]


def find_multiple(n: int, k: int) -> int:
    if n % k == 0:
        return n
    return n + k - (n % k)


def _quantized_linear_cls(mode, device=None):
    if mode == "llm.int8":
        if device is not None and torch.device(device).type != "cuda":
            raise ValueError("Quantization is only supported on the GPU.")
        from .quantization import Linear8bitLt

        return Linear8bitLt
    if mode == "gptq.int4":
        from .quantization import ColBlockQuantizedLinear

        return functools.partial(ColBlockQuantizedLinear, bits=4, tile_cols=-1)
    if mode == "gptq.int8":
        from .quantization import ColBlockQuantizedLinear

        return functools.partial(ColBlockQuantizedLinear, bits=8, tile_cols=-1)
    return None


@contextmanager
def quantization(mode: str = None):
    """Rebind `torch.nn.Linear` for the duration of model construction (lit_llama/utils.py:141-162)."""
    if mode is not None and mode not in ("llm.int8", "gptq.int4", "gptq.int8"):
        raise ValueError(f"Unknown quantization mode: {mode}")
    quantized_linear_cls = _quantized_linear_cls(mode)
    enabled = mode is not None
    torch_linear_cls = torch.nn.Linear
    if enabled:
        torch.nn.Linear = quantized_linear_cls
    try:
        yield
    finally:
        if enabled:
            torch.nn.Linear = torch_linear_cls


class EmptyInitOnDevice(torch.overrides.TorchFunctionMode):
    """Create tensors directly on `device` / in `dtype`, skip `torch.nn.init.*`, optionally swap the linear
    class (lit_llama/utils.py:73-138).

        with EmptyInitOnDevice("cuda", dtype=torch.bfloat16, quantization_mode="gptq.int4"):
            model = LLaMA.from_name("7B")
        model.load_state_dict(checkpoint)
    """

    def __init__(self, device=None, dtype=None, quantization_mode=None):
        if quantization_mode is not None and quantization_mode not in ("llm.int8", "gptq.int4", "gptq.int8"):
            raise RuntimeError(f"unknown quantization mode {quantization_mode}")
        self.quantization_mode = quantization_mode
        self.quantized_linear_cls = _quantized_linear_cls(quantization_mode, device)
        self.device = device
        self.dtype = dtype

    def __enter__(self):
        if self.quantized_linear_cls is not None:
            self.torch_linear_cls = torch.nn.Linear
            torch.nn.Linear = self.quantized_linear_cls
        return super().__enter__()

    def __exit__(self, exc_type, exc_val, exc_tb):
        if self.quantized_linear_cls is not None:
            torch.nn.Linear = self.torch_linear_cls
        return super().__exit__(exc_type, exc_val, exc_tb)

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if getattr(func, "__module__", None) == "torch.nn.init":
            if "tensor" in kwargs:
                return kwargs["tensor"]
            return args[0]
        if (
            self.device is not None
            and func in torch.utils._device._device_constructors()
            and kwargs.get("device") is None
        ):
            kwargs["device"] = self.device
        if (
            self.dtype is not None
            and func in torch.utils._device._device_constructors()
            and kwargs.get("dtype") is None
        ):
            kwargs["dtype"] = self.dtype
        return func(*args, **kwargs)

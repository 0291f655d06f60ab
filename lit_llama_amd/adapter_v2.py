"""LLaMA-Adapter v2 inference variant (/root/reference lit_llama/adapter_v2.py, generate/adapter_v2.py:63-78): on top of the
v1 prefix attention (lit_llama_amd/adapter.py) every linear of the model gets a learned per-output `adapter_scale` and
`adapter_bias`,  y = adapter_scale * (W x + adapter_bias)  (:29-32).

The parameters are registered on the existing `nn.Linear` modules under the reference's names, so a v2 adapter checkpoint
loads with `strict=False` exactly as in generate/adapter_v2.py.  The matrix product stays the native linear; the scale /
bias pair is applied by `lit_llama_amd.model._linear` as one elementwise epilogue on the device."""
from __future__ import annotations

import torch
import torch.nn as nn

from .adapter import LLaMA  # noqa: F401  (generate/adapter_v2.py imports the v1 model class)


def get_adapter_substrings():
    """Parameter-name fragments of everything adapter v2 trains (adapter_v2.py:10-14)."""
    return ["adapter_wte", "gating_factor", "adapter_scale", "adapter_bias", "rms_1", "rms_2", "ln_f"]


def mark_only_adapter_v2_as_trainable(model: nn.Module) -> None:
    for name, param in model.named_parameters():
        param.requires_grad = any(s in name for s in get_adapter_substrings())


def adapter_v2_state_from_state_dict(state_dict: dict) -> dict:
    return {name: param for name, param in state_dict.items() if any(s in name for s in get_adapter_substrings())}


def adapter_v2_new_forward(self, input: torch.Tensor) -> torch.Tensor:
    """`adapter_scale * (W x + bias + adapter_bias)` (adapter_v2.py:29-32) as a function of the layer: the native linear plus
    the elementwise epilogue of `lit_llama_amd.model._linear`."""
    from .model import _linear

    if type(self) is nn.Linear or getattr(self, "_mi355_plain_weight", False):
        return _linear(self, input)  # native linear + epilogue
    # a plug-in linear (Linear8bitLt, an unmerged LoRA layer) BOUND to this function the way the reference binds it
    # (adapter_v2.py:39, `adapter_v2_new_forward.__get__(layer, layer.__class__)`): `_linear` would call `self(input)`, i.e. this
    # function again.  Call the class's own forward, then the epilogue.
    y = type(self).forward(self, input)
    return self.adapter_scale.detach().to(y.dtype) * (y + self.adapter_bias.detach().to(y.dtype))


def adapter_v2_linear_with_bias_and_scale(layer: nn.Linear) -> nn.Linear:
    """Identity at initialisation: bias 0, scale 1 (adapter_v2.py:35-40)."""
    w = layer.weight
    # (a quantised plug-in keeps an integer weight: Linear8bitLt's int8 rows — the pair is float there, in the default dtype as the
    # reference's `torch.zeros(layer.weight.shape[0])` is; the epilogue casts it to the output's dtype)
    dt = w.dtype if w.is_floating_point() else torch.get_default_dtype()
    layer.adapter_bias = torch.nn.Parameter(torch.zeros(w.shape[0], device=w.device, dtype=dt), requires_grad=True)
    layer.adapter_scale = torch.nn.Parameter(torch.ones(w.shape[0], device=w.device, dtype=dt), requires_grad=True)
    return layer


def add_adapter_v2_parameters_to_linear_layers(model: nn.Module) -> None:
    for module in model.modules():
        if isinstance(module, nn.Linear):
            adapter_v2_linear_with_bias_and_scale(module)

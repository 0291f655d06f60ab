"""LoRA inference variant: the fine-tuned low-rank update of the fused q / k / v projection is merged
into the weight when the model is put in eval mode, after which the model is a plain LLaMA and decodes through the same
native engine as any bf16 checkpoint.

Mirrors the inference-side API of /root/reference lit_llama/lora.py: `lora(r, alpha, dropout, enabled)` (:430-478) makes
`LLaMA(...)` build its attention blocks with `MergedLinear` (:92-326) as `c_attn` (enable_lora = [True, False, True]:
queries and values, :101-110), whose parameters `lora_A` [2 r, n_embd] / `lora_B` [2 n_embd, r] carry the reference's
state-dict keys, so generate/lora.py:71-83 works unchanged:

    with lazy_load(pretrained) as ckpt, lazy_load(lora_path) as lora_ckpt, lora(r=8, alpha=16, dropout=0.05):
        model = LLaMA.from_name(name)
        model.load_state_dict(ckpt, strict=False); model.load_state_dict(lora_ckpt, strict=False)
    model.eval()      # merges: W += zero_pad((B A) * alpha / r)

Training (dropout, gradient flow through A / B, `mark_only_lora_as_trainable`) is out of scope; the unmerged forward is
kept for completeness and follows the reference arithmetic with torch ops.
"""
from __future__ import annotations

import math
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Dict, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import model as llama


class MergedLinear(nn.Linear):
    """`nn.Linear` for fused projections with a low-rank update on some of them (lit_llama/lora.py:92-195)."""

    def __init__(self, in_features: int, out_features: int, r: int = 0, lora_alpha: int = 1, lora_dropout: float = 0.0,
                 enable_lora: Sequence[bool] = (False,), fan_in_fan_out: bool = False, merge_weights: bool = True, **kwargs):
        super().__init__(in_features, out_features, **kwargs)
        assert out_features % len(enable_lora) == 0, "The length of enable_lora must divide out_features"
        assert not fan_in_fan_out, "fan_in_fan_out layouts (GPT-2 Conv1D) do not occur in LLaMA"
        self.r, self.lora_alpha, self.enable_lora = r, lora_alpha, list(enable_lora)
        self.lora_dropout_p = lora_dropout
        self.merge_weights = merge_weights
        self.merged = False
        if r > 0 and any(enable_lora):
            n_on = sum(self.enable_lora)
            self.lora_A = nn.Parameter(self.weight.new_zeros((r * n_on, in_features)))
            self.lora_B = nn.Parameter(self.weight.new_zeros((out_features // len(enable_lora) * n_on, r)))
            self.scaling = self.lora_alpha / self.r
            self.weight.requires_grad = False
            nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))  # lora.py:196-202: A random, B zero
            nn.init.zeros_(self.lora_B)

    @property
    def _mi355_plain_weight(self) -> bool:
        """True when `weight` alone is the whole operator (merged, or no LoRA): the model's fast linear path and the
        native engine then treat the module as a stock nn.Linear."""
        return self.merged or self.r == 0 or not any(self.enable_lora)

    def delta(self) -> torch.Tensor:
        """zero_pad(B A) (lora.py:272-279, :205-241) in the parameters' dtype: rows of the enabled projections only."""
        n = len(self.enable_lora)
        rows = self.out_features // n
        out = self.lora_A.new_zeros((self.out_features, self.in_features))
        g = 0
        for j, on in enumerate(self.enable_lora):
            if on:
                out[j * rows:(j + 1) * rows] = self.lora_B.data[g * rows:(g + 1) * rows] @ self.lora_A.data[g * self.r:(g + 1) * self.r]
                g += 1
        return out

    def train(self, mode: bool = True):
        """eval(): merge the update into `weight`; train(): take it out again (lora.py:243-280)."""
        nn.Linear.train(self, mode)
        should = self.merged if mode else not self.merged
        if self.merge_weights and should:
            if self.r > 0 and any(self.enable_lora):
                with torch.no_grad():
                    upd = self.delta() * self.scaling
                    # in place through the parameter (not `.data`): the version counter moves, so an engine that packed
                    # the old weight is rebuilt (LLaMA.engine() fingerprint)
                    self.weight.add_(-upd if mode else upd)
            self.merged = not mode
        return self

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self._mi355_plain_weight:
            return llama._linear(self, x)
        # separate LoRA branch (training-time form, lora.py:308-326); dropout only in training mode
        y = F.linear(x, self.weight, self.bias)
        xa = F.dropout(x, self.lora_dropout_p, self.training) if self.lora_dropout_p > 0 else x
        after_a = F.linear(xa, self.lora_A)
        n = len(self.enable_lora)
        rows = self.out_features // n
        g = 0
        for j, on in enumerate(self.enable_lora):
            if on:
                y[..., j * rows:(j + 1) * rows] += F.linear(after_a[..., g * self.r:(g + 1) * self.r],
                                                           self.lora_B[g * rows:(g + 1) * rows]) * self.scaling
                g += 1
        return y


def lora_state_dict(model: nn.Module, bias: str = "none") -> Dict[str, torch.Tensor]:
    """The LoRA matrices of a model (lora.py:364-395; only bias = 'none' occurs: LLaMA has no biases)."""
    if bias != "none":
        raise NotImplementedError(bias)
    sd = model.state_dict()
    return {k: v for k, v in sd.items() if "lora_" in k}


@dataclass
class LoRAConfig:
    r: float = 0.0
    alpha: float = 1.0
    dropout: float = 0.0


class CausalSelfAttention(llama.CausalSelfAttention):
    """The attention block with `MergedLinear` as c_attn (lora.py:405-427); everything else is inherited."""

    lora_config = None

    def __init__(self, config: llama.LLaMAConfig) -> None:
        super().__init__(config)
        assert llama._tp(config) == 1, "LoRA checkpoints are merged before tensor-parallel sharding"
        cfg = self.lora_config
        self.c_attn = MergedLinear(config.n_embd, 3 * config.n_embd, r=cfg.r, lora_alpha=cfg.alpha, lora_dropout=cfg.dropout,
                                   enable_lora=[True, False, True], fan_in_fan_out=False, merge_weights=True, bias=False,
                                   device=self.c_attn.weight.device, dtype=self.c_attn.weight.dtype)


@contextmanager
def lora(r, alpha, dropout, enabled: bool = True):
    """While active, `LLaMA(...)` builds LoRA attention blocks (lora.py:430-478)."""
    if not enabled:
        yield
        return
    CausalSelfAttention.lora_config = LoRAConfig(r=r, alpha=alpha, dropout=dropout)
    saved = llama.CausalSelfAttention
    llama.CausalSelfAttention = CausalSelfAttention
    try:
        yield
    finally:
        llama.CausalSelfAttention = saved
        CausalSelfAttention.lora_config = None

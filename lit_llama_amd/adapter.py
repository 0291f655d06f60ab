"""LLaMA-Adapter inference variant: /root/reference lit_llama/adapter.py.

From layer `adapter_start_layer` on, every attention block owns `adapter_prompt_length` learned prefix rows
(`adapter_wte`) and a per-head `gating_factor`; its output becomes

    y + gating_factor * softmax(q ak^T / sqrt(hs)) av          (adapter.py:134-151)

with ak / av the k / v projections of the prefix rows through the block's own c_attn (no RoPE, computed once and kept as
the `adapter_kv_cache`) and q the RoPE'd query of the token.  Class names, constructor signatures, state-dict keys
(`transformer.h.{i}.attn.adapter_wte.weight`, `...attn.gating_factor`) and the forward / cache contract follow the
reference so that generate/adapter.py:67-95 runs unchanged.

The linears, RMSNorm, RoPE, the KV cache and the causal attention are the native kernels (lit_llama_amd/ops.py); the prefix
term — ten rows per head, own softmax, gate — is computed inside the decode attention kernel (every flash-decoding split
adds its share to its partial record; `mi355_attn_args.adapter_*`, csrc/attention.hip) and by `mi355_adapter_prefix` behind the
many-token flash kernel, in the op-by-op path and in the whole-forward engine alike (csrc/engine.hip), so an adapter model
decodes under the same hipGraph as a plain one (not through the persistent fused step, which does not know the prefix term).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import _native as nat
from . import model as llama
from . import ops
from .model import KVCache, MaskCache, RMSNorm, RoPECache, _linear


@dataclass
class LLaMAConfig(llama.LLaMAConfig):
    adapter_prompt_length: int = 10
    adapter_start_layer: int = 2


class CausalSelfAttention(llama.CausalSelfAttention):
    """Self-attention plus the gated cross-attention over the adaption prompt (adapter.py:62-171)."""

    def __init__(self, config: LLaMAConfig, block_idx: int) -> None:
        super().__init__(config)
        if block_idx >= config.adapter_start_layer:
            self.adapter_wte = nn.Embedding(config.adapter_prompt_length, config.n_embd)
            # zero at initialisation: an untrained adapter leaves the pretrained model unchanged (adapter.py:83-85)
            self.gating_factor = torch.nn.Parameter(torch.zeros(1, config.n_head, 1, 1))
        self.block_idx = block_idx
        self.adapter_prompt_length = config.adapter_prompt_length
        self.adapter_start_layer = config.adapter_start_layer

    def forward(
        self,
        x: torch.Tensor,
        rope: RoPECache,
        mask: MaskCache,
        max_seq_length: int,
        input_pos: Optional[torch.Tensor] = None,
        kv_cache: Optional[KVCache] = None,
        adapter_kv_cache: Optional[KVCache] = None,
    ) -> Tuple[torch.Tensor, Optional[KVCache], Optional[KVCache]]:
        B, T, C = x.shape
        nh, hs = self.n_head, C // self.n_head
        qkv = _linear(self.c_attn, x)
        rope_f = rope.float().contiguous()
        adapter = None
        if self.block_idx >= self.adapter_start_layer:
            if adapter_kv_cache is None:
                adapter_kv_cache = self.adapter_prefix_kv(x.dtype)[:2]
            # y += gate * softmax(rope(q) ak^T / sqrt(hs)) av, inside the attention op (csrc/attention.hip)
            adapter = (*adapter_kv_cache, self.gating_factor.detach().float().reshape(-1).contiguous())
        if kv_cache is not None:
            assert input_pos is not None
            k, v = kv_cache
            if int(input_pos[-1]) >= max_seq_length:  # the reference's host decision (adapter.py:119)
                ops.kv_roll(k, v)
            y = ops.attention(qkv, rope_f, nh, pos=input_pos, kv_cache=(k, v), rope_gathered=True, adapter=adapter)
        else:
            y = ops.attention(qkv, rope_f, nh, rope_gathered=True, adapter=adapter)
        return _linear(self.c_proj, y), kv_cache, adapter_kv_cache

    def adapter_prefix_kv(self, dtype: torch.dtype):
        """(ak, av, gate): the k / v projections of the adaption prompt through this block's c_attn, f32 [n_head, aT, hs]
        (no RoPE, adapter.py:136-141), and the per-head gate, f32 [n_head]."""
        C, nh = self.n_embd, self.n_head
        hs = C // nh
        prefix = self.adapter_wte.weight.detach().reshape(1, self.adapter_prompt_length, C)
        akv = _linear(self.c_attn, prefix.to(dtype))                                     # [1, aT, 3 C]
        ak = akv[0, :, C:2 * C].reshape(-1, nh, hs).transpose(0, 1).float().contiguous()  # [nh, aT, hs]
        av = akv[0, :, 2 * C:].reshape(-1, nh, hs).transpose(0, 1).float().contiguous()
        return ak, av, self.gating_factor.detach().float().reshape(-1).contiguous()

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """Old checkpoints hold ONE gating value for all heads (adapter.py:173-183)."""
        name = prefix + "gating_factor"
        if name in state_dict:
            t = state_dict[name]
            t = t._load_tensor() if hasattr(t, "_load_tensor") else t
            if t.dim() < 4:
                state_dict[name] = t.reshape(1, 1, 1, 1).repeat(1, self.n_head, 1, 1)
            else:
                state_dict[name] = t
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class Block(nn.Module):
    """`model.Block` with the adapter attention (adapter.py:186-216)."""

    def __init__(self, config: LLaMAConfig, block_idx: int) -> None:
        super().__init__()
        self.rms_1 = RMSNorm(config.n_embd)
        self.attn = CausalSelfAttention(config, block_idx)
        self.rms_2 = RMSNorm(config.n_embd)
        self.mlp = llama.MLP(config)

    def forward(self, x, rope, mask, max_seq_length, input_pos=None, kv_cache=None, adapter_kv_cache=None):
        h, new_kv, new_akv = self.attn(self.rms_1(x), rope, mask, max_seq_length, input_pos, kv_cache, adapter_kv_cache)
        x = ops.add(x, h)
        x = ops.add(x, self.mlp(self.rms_2(x)))
        return x, new_kv, new_akv


class LLaMA(llama.LLaMA):
    """`model.LLaMA` whose blocks know their index (adapter.py:219-303).  As in the reference the embedding and the
    head are sized by `config.vocab_size`, not by the padded size."""

    def __init__(self, config: LLaMAConfig) -> None:
        nn.Module.__init__(self)
        assert config.vocab_size is not None and config.block_size is not None
        self.config = config
        self.lm_head = nn.Linear(config.n_embd, config.vocab_size, bias=False)
        self.transformer = nn.ModuleDict(
            dict(
                wte=nn.Embedding(config.vocab_size, config.n_embd),
                h=nn.ModuleList(Block(config, i) for i in range(config.n_layer)),
                ln_f=RMSNorm(config.n_embd),
            )
        )
        self.rope_cache: Optional[RoPECache] = None
        self.mask_cache: Optional[MaskCache] = None
        self.kv_caches: List[KVCache] = []
        self.adapter_kv_caches: List[Optional[KVCache]] = []
        self._engine = None
        self._engine_failed = None
        self._engine_failed_fp = None
        self.use_engine = True

    @classmethod
    def from_name(cls, name: str):
        return cls(LLaMAConfig.from_name(name))

    def reset_cache(self) -> None:
        super().reset_cache()
        self.adapter_kv_caches.clear()

    def _adapter_fingerprint(self) -> tuple:
        out = []
        for blk in self.transformer.h:
            attn = blk.attn
            if hasattr(attn, "adapter_wte"):
                for t in (attn.adapter_wte.weight, attn.gating_factor):
                    out.append((t.data_ptr(), t._version))
        return tuple(out)

    def forward(self, idx: torch.Tensor, max_seq_length: Optional[int] = None,
                input_pos: Optional[torch.Tensor] = None) -> torch.Tensor:
        B, T = idx.size()
        block_size = self.config.block_size
        if max_seq_length is None:
            max_seq_length = block_size
        assert T <= max_seq_length, f"Cannot forward sequence of length {T}, max seq length is only {max_seq_length}"
        assert max_seq_length <= block_size, f"Cannot attend to {max_seq_length}, block size is only {block_size}"
        assert T <= block_size, f"Cannot forward sequence of length {T}, block size is only {block_size}"
        nat.require_gpu(idx, "adapter.LLaMA.forward")
        if self.rope_cache is None:
            self.rope_cache = self.build_rope_cache(idx)
        if self.mask_cache is None:
            self.mask_cache = self.build_mask_cache(idx)
        if input_pos is not None and B == 1 and self.use_engine:
            # the engine snapshots the prefix k / v (and the gates) of every adapter block when it is built; the per-token path
            # skips the full fingerprint walk (check=False), so at least an in-place edit of the adapter's own parameters
            # (adapter_wte.weight.copy_(...), gating_factor.fill_(...)) is caught here: ~60 (data_ptr, version) pairs
            afp = self._adapter_fingerprint()
            if self._engine is not None and getattr(self._engine, "_adapter_fp", afp) != afp:
                self._drop_engine()
            eng = self.engine(check=False)  # (None for LLaMA-Adapter v2 linears: op by op below)
            if eng is not None:
                eng._adapter_fp = afp
                out = eng.forward(idx, max_seq_length, input_pos)
                if out is not None:
                    return out
        rope = self.rope_cache.index_select(0, input_pos) if input_pos is not None else self.rope_cache[:T]
        x = ops.embedding(idx, self.transformer.wte.weight.detach())
        if input_pos is None:  # no cache (adapter.py:277-279)
            for block in self.transformer.h:
                x, *_ = block(x, rope, None, max_seq_length)
        else:
            if not self.kv_caches:
                head_size = self.config.n_embd // self.config.n_head
                shape = (B, self.config.n_head, max_seq_length, head_size)
                self.kv_caches = [(torch.zeros(shape, device=x.device, dtype=x.dtype),
                                   torch.zeros(shape, device=x.device, dtype=x.dtype)) for _ in range(self.config.n_layer)]
            if not self.adapter_kv_caches:
                self.adapter_kv_caches = [None for _ in range(self.config.n_layer)]
            for i, block in enumerate(self.transformer.h):
                x, self.kv_caches[i], self.adapter_kv_caches[i] = block(
                    x, rope, None, max_seq_length, input_pos, self.kv_caches[i], self.adapter_kv_caches[i])
        x = self.transformer.ln_f(x)
        return _linear(self.lm_head, x).float()


def mark_only_adapter_as_trainable(model: LLaMA) -> None:
    """adapter.py:306-309 (kept for API completeness; fine-tuning itself is out of scope)."""
    for name, param in model.named_parameters():
        param.requires_grad = "adapter_wte" in name or "gating_factor" in name


def adapter_state_from_state_dict(state_dict: dict) -> dict:
    """Only the adapter weights of a model state dict (adapter.py:312-315)."""
    return {name: param for name, param in state_dict.items() if "adapter_wte" in name or "gating_factor" in name}

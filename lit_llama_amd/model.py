"""LLaMA decoder with the module API of /root/reference lit_llama/model.py, computed by gfx950 HIP kernels.

What is kept (the drop-in contract): `LLaMAConfig`, `LLaMA(config)`, `LLaMA.from_name`,
`forward(idx, max_seq_length=None, input_pos=None) -> logits`, `reset_cache()`, the attributes `config`,
`transformer.{wte,h,ln_f}`, `lm_head`, `rope_cache`, `mask_cache`, `kv_caches`, the state-dict key names of
scripts/convert_checkpoint.py:24-53, `Block` / `CausalSelfAttention` / `MLP` / `RMSNorm` call signatures,
`build_rope_cache`, `apply_rope`.

What is different underneath:
  * two execution paths.  `LLaMA.forward` with B == 1 and a KV cache on a bf16 model enters the native
    engine once per call (`engine.DecodeEngine`: ~5 launches per layer, hipGraph for T == 1).  Everything else
    (no cache, B > 1, f32 "plumbing" models, direct calls of sub-modules) runs op by op through the generic
    kernels, following the reference arithmetic.
  * the KV cache is updated in place (the reference copies both caches per layer per token, model.py:219-220),
    attention reads rows [0, pos] only, and no device->host sync is taken per layer (model.py:214).
  * logits are returned in float32 (the reference returns the model dtype); greedy argmax on bf16 logits
    ties far too often to be comparable with the fp32 CPU path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Tuple, Union

import torch
import torch.nn as nn
from typing_extensions import Self

from . import _native as nat
from . import ops
from .utils import find_multiple

MaskCache = torch.Tensor
RoPECache = torch.Tensor
KVCache = Tuple[torch.Tensor, torch.Tensor]


@dataclass
class LLaMAConfig:
    block_size: int = 2048
    vocab_size: int = 32000
    padded_vocab_size: Optional[int] = None
    n_layer: int = 32
    n_head: int = 32
    n_embd: int = 4096

    def __post_init__(self):
        if self.padded_vocab_size is None:
            self.padded_vocab_size = find_multiple(self.vocab_size, 64)

    @classmethod
    def from_name(cls, name: str) -> Self:
        return cls(**llama_configs[name])

    @property
    def head_size(self) -> int:
        return self.n_embd // self.n_head

    @property
    def n_hidden(self) -> int:
        # lit_llama/model.py:243-245
        return find_multiple(int(2 * (4 * self.n_embd) / 3), 256)


llama_configs = {
    "7B": dict(n_layer=32, n_head=32, n_embd=4096),
    "13B": dict(n_layer=40, n_head=40, n_embd=5120),
    "30B": dict(n_layer=60, n_head=52, n_embd=6656),
    "65B": dict(n_layer=80, n_head=64, n_embd=8192),
}


def _tp(config) -> int:
    """Tensor-parallel degree of a rank-local model built by tp.build_local_model (1 for ordinary models)."""
    return int(getattr(config, "tp_world", 1))


def _linear(mod: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """Apply one of the hot-path linears.  Quantised plug-ins carry their own kernels; a stock `nn.Linear`
    (no-quant configs) goes to the bf16 weight-streaming kernel for skinny inputs, to the MFMA GEMM over the same
    stream for wide bf16 inputs, and to the exact f32 kernel for f32 models."""
    if type(mod) is not nn.Linear and not getattr(mod, "_mi355_plain_weight", False):
        y = mod(x)  # (lora.MergedLinear sets _mi355_plain_weight once its update is merged: then it IS a plain linear)
        if getattr(mod, "adapter_scale", None) is not None:
            # LLaMA-Adapter v2 on a quantised plug-in (generate/adapter_v2.py accepts --quantize llm.int8; Linear8bitLt
            # subclasses nn.Linear, so add_adapter_v2_parameters_to_linear_layers attaches the pair to it as well)
            y = mod.adapter_scale.detach().to(y.dtype) * (y + mod.adapter_bias.detach().to(y.dtype))
        return y
    nat.require_gpu(x, "Linear.forward")
    x2d = x.reshape(-1, x.shape[-1])
    if x2d.stride(-1) != 1:
        x2d = x2d.contiguous()
    M = x2d.shape[0]
    w = mod.weight
    if x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16:
        key = (w.data_ptr(), w._version)
        cached = getattr(mod, "_mi355_stream", None)
        if cached is None or cached[0] != key:
            cached = (key, ops.repack_bf16(w.detach(), None, 1))
            mod._mi355_stream = cached
        if M <= 64 or w.shape[0] % 4 != 0:
            y = ops.linear_fast(x2d, cached[1], nat.W_BF16, 1, w.shape[0], w.shape[1], bias=mod.bias, out_dtype=x.dtype)
        else:
            # wide bf16 inputs: the same LDS-tiled MFMA GEMM as the int4 models, over the BF16 stream (round 2 handed this
            # case to rocBLAS: a library GEMM on the product path of BASELINE configs[1])
            y = ops.linear_gemm(x2d, cached[1], 1, w.shape[0], w.shape[1], out_dtype=x.dtype, fmt=nat.W_BF16)
            if mod.bias is not None:
                y = y + mod.bias.to(y.dtype)
    else:
        y = ops.linear_dense(x2d, w.detach().to(x.dtype), mod.bias)
    y = y.view(*x.shape[:-1], w.shape[0])
    if getattr(mod, "adapter_scale", None) is not None:
        # LLaMA-Adapter v2 (lit_llama/adapter_v2.py:29-32): scale * (W x + bias), one learned pair per output feature
        y = mod.adapter_scale.detach().to(y.dtype) * (y + mod.adapter_bias.detach().to(y.dtype))
    # this path bypasses nn.Module.__call__; forward hooks (GPTQ calibration statistics, lit_llama_amd/gptq.py)
    # still see (module, inputs, output)
    for hook in list(mod._forward_hooks.values()):
        r = hook(mod, (x,), y)
        if r is not None:
            y = r
    return y


class RMSNorm(nn.Module):
    """`scale * x * rsqrt(mean(x^2, -1) + eps)` (lit_llama/model.py:257-277); one fused kernel."""

    def __init__(self, size: int, dim: int = -1, eps: float = 1e-5) -> None:
        super().__init__()
        self.scale = nn.Parameter(torch.ones(size))
        self.eps = eps
        self.dim = dim

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.dim not in (-1, x.dim() - 1):
            raise NotImplementedError("RMSNorm kernel normalises over the last dimension")
        return ops.rmsnorm(x, self.scale.detach(), self.eps)


class MLP(nn.Module):
    def __init__(self, config: LLaMAConfig) -> None:
        super().__init__()
        n_hidden = config.n_hidden // _tp(config)  # rank-local rows under tensor parallelism (tp.py)
        self.c_fc1 = nn.Linear(config.n_embd, n_hidden, bias=False)
        self.c_fc2 = nn.Linear(config.n_embd, n_hidden, bias=False)
        self.c_proj = nn.Linear(n_hidden, config.n_embd, bias=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        gate = ops.swiglu(_linear(self.c_fc1, x), _linear(self.c_fc2, x))
        return _linear(self.c_proj, gate)


class CausalSelfAttention(nn.Module):
    def __init__(self, config: LLaMAConfig) -> None:
        super().__init__()
        assert config.n_embd % config.n_head == 0
        tp = _tp(config)
        self.c_attn = nn.Linear(config.n_embd, 3 * config.n_embd // tp, bias=False)  # [Q; K; V] stacked on dim 0
        self.c_proj = nn.Linear(config.n_embd // tp, config.n_embd, bias=False)
        self.n_head = config.n_head // tp
        self.n_embd = config.n_embd
        self.block_size = config.block_size

    def forward(
        self,
        x: torch.Tensor,
        rope: RoPECache,
        mask: MaskCache,
        max_seq_length: int,
        input_pos: Optional[torch.Tensor] = None,
        kv_cache: Optional[KVCache] = None,
    ) -> Tuple[torch.Tensor, Optional[KVCache]]:
        """`rope` holds the rows for these T tokens (as `LLaMA.forward` selects them).  `mask` is accepted for
        signature compatibility; the kernel applies the causal rule of `build_mask_cache` from the positions."""
        qkv = _linear(self.c_attn, x)
        if kv_cache is not None:
            assert input_pos is not None
            k, v = kv_cache
            # the reference takes the same host decision (`if input_pos[-1] >= max_seq_length`, model.py:214)
            if int(input_pos[-1]) >= max_seq_length:
                ops.kv_roll(k, v)
            y = ops.attention(qkv, rope.float().contiguous(), self.n_head, pos=input_pos, kv_cache=(k, v),
                              rope_gathered=True)
        else:
            y = ops.attention(qkv, rope.float().contiguous(), self.n_head, rope_gathered=True)
        return _linear(self.c_proj, y), kv_cache


class Block(nn.Module):
    def __init__(self, config: LLaMAConfig) -> None:
        super().__init__()
        self.rms_1 = RMSNorm(config.n_embd)
        self.attn = CausalSelfAttention(config)
        self.rms_2 = RMSNorm(config.n_embd)
        self.mlp = MLP(config)

    def forward(
        self,
        x: torch.Tensor,
        rope: RoPECache,
        mask: MaskCache,
        max_seq_length: int,
        input_pos: Optional[torch.Tensor] = None,
        kv_cache: Optional[KVCache] = None,
    ) -> Tuple[torch.Tensor, Optional[KVCache]]:
        h, new_kv_cache = self.attn(self.rms_1(x), rope, mask, max_seq_length, input_pos, kv_cache)
        x = ops.add(x, h)
        x = ops.add(x, self.mlp(self.rms_2(x)))
        return x, new_kv_cache


class LLaMA(nn.Module):
    def __init__(self, config: LLaMAConfig) -> None:
        super().__init__()
        assert config.padded_vocab_size is not None
        self.config = config
        self.lm_head = nn.Linear(config.n_embd, config.padded_vocab_size // _tp(config), bias=False)
        self.transformer = nn.ModuleDict(
            dict(
                wte=nn.Embedding(config.padded_vocab_size, config.n_embd),
                h=nn.ModuleList(Block(config) for _ in range(config.n_layer)),
                ln_f=RMSNorm(config.n_embd),
            )
        )
        self.rope_cache: Optional[RoPECache] = None
        self.mask_cache: Optional[MaskCache] = None
        self.kv_caches: List[KVCache] = []
        self._engine = None           # engine.DecodeEngine, built lazily
        self._engine_failed = None    # reason string once the engine turned out to be inapplicable
        self._engine_failed_fp = None
        self.use_engine = True

    def _init_weights(self, module: nn.Module) -> None:
        std = 0.02 / math.sqrt(2 * self.config.n_layer)
        if isinstance(module, (nn.Linear, nn.Embedding)):
            torch.nn.init.normal_(module.weight, mean=0.0, std=std)

    @classmethod
    def from_name(cls, name: str) -> Self:
        return cls(LLaMAConfig.from_name(name))

    # ---- caches --------------------------------------------------------------------------------------
    def build_rope_cache(self, idx: torch.Tensor) -> RoPECache:
        return build_rope_cache(
            seq_len=self.config.block_size,
            n_elem=self.config.n_embd // self.config.n_head,
            dtype=idx.dtype,
            device=idx.device,
        )

    def build_mask_cache(self, idx: torch.Tensor) -> MaskCache:
        ones = torch.ones((self.config.block_size, self.config.block_size), device=idx.device, dtype=torch.bool)
        return torch.tril(ones).unsqueeze(0).unsqueeze(0)

    def reset_cache(self) -> None:
        self.kv_caches.clear()
        if self._engine is not None:
            self._engine.reset_cache()

    # ---- engine --------------------------------------------------------------------------------------
    def engine(self, check: bool = True):
        """The native whole-forward engine for this model, or None (with the reason in `_engine_failed`).

        The engine holds repacked copies of the weights and raw device pointers.  `load_state_dict`, `.to()` /
        `.bfloat16()` drop it (overrides below); with `check` the (data_ptr, version, dtype) fingerprint of every
        parameter and buffer is compared as well, which also catches in-place edits and re-quantisation
        (`LLaMA.forward` skips that walk on its per-token path; `generate`, tests and bench.py take it)."""
        from .engine import DecodeEngine, EngineUnavailable, model_fingerprint

        fp = None
        if check and (self._engine is not None or self._engine_failed is not None):
            fp = model_fingerprint(self)
            if self._engine is not None and self._engine.fingerprint != fp:
                self._drop_engine()
            if self._engine_failed is not None and self._engine_failed_fp != fp:
                self._engine_failed = None
        if self._engine is None and self._engine_failed is None:
            try:
                self._engine = DecodeEngine(self)
            except EngineUnavailable as e:
                self._engine_failed = str(e)
                self._engine_failed_fp = fp if fp is not None else model_fingerprint(self)
        return self._engine

    def _drop_engine(self) -> None:
        self._engine = None
        self._engine_failed = None
        self.kv_caches = []

    def _apply(self, fn, *args, **kwargs):  # .to() / .cuda() / .bfloat16() / .float(): new storages
        self._drop_engine()
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._drop_engine()
        return super().load_state_dict(*args, **kwargs)

    # ---- forward -------------------------------------------------------------------------------------
    def forward(
        self, idx: torch.Tensor, max_seq_length: Optional[int] = None, input_pos: Optional[torch.Tensor] = None
    ) -> torch.Tensor:
        B, T = idx.size()
        block_size = self.config.block_size
        if max_seq_length is None:
            max_seq_length = block_size
        assert T <= max_seq_length, f"Cannot forward sequence of length {T}, max seq length is only {max_seq_length}"
        assert max_seq_length <= block_size, f"Cannot attend to {max_seq_length}, block size is only {block_size}"
        assert T <= block_size, f"Cannot forward sequence of length {T}, block size is only {block_size}"
        nat.require_gpu(idx, "LLaMA.forward")

        if self.rope_cache is None:
            self.rope_cache = self.build_rope_cache(idx)
        if self.mask_cache is None:
            self.mask_cache = self.build_mask_cache(idx)

        if _tp(self.config) > 1:
            raise nat.NativeError("a rank-local tensor-parallel model is driven through tp.TPDecoder / tp.tp_forward")
        if input_pos is not None and B == 1 and self.use_engine:
            eng = self.engine(check=False)
            if eng is not None:
                out = eng.forward(idx, max_seq_length, input_pos)
                if out is not None:
                    return out

        # ---- op-by-op path (reference structure, model.py:93-122)
        if input_pos is not None:
            rope = self.rope_cache.index_select(0, input_pos)
            mask = None
        else:
            rope = self.rope_cache[:T]
            mask = None
        x = ops.embedding(idx, self.transformer.wte.weight.detach())
        if input_pos is None:
            for block in self.transformer.h:
                x, _ = block(x, rope, mask, max_seq_length)
        else:
            if not self.kv_caches:
                head_size = self.config.n_embd // self.config.n_head
                cache_shape = (B, self.config.n_head, max_seq_length, head_size)
                self.kv_caches = [
                    (torch.zeros(cache_shape, device=x.device, dtype=x.dtype),
                     torch.zeros(cache_shape, device=x.device, dtype=x.dtype))
                    for _ in range(self.config.n_layer)
                ]
            for i, block in enumerate(self.transformer.h):
                x, self.kv_caches[i] = block(x, rope, mask, max_seq_length, input_pos, self.kv_caches[i])
        x = self.transformer.ln_f(x)
        return _linear(self.lm_head, x).float()


def build_rope_cache(
    seq_len: int, n_elem: int, dtype: torch.dtype, device: torch.device, base: int = 10000
) -> RoPECache:
    """[seq_len, n_elem / 2, 2] table of (cos, sin)(pos * base^(-2i / n_elem)) (lit_llama/model.py:280-303).

    Built once on the host with the same torch expressions as the reference's CPU path, so the table is
    bit-identical to the oracle's; integer `dtype` (what `LLaMA.build_rope_cache` passes) yields float32."""
    theta = 1.0 / (base ** (torch.arange(0, n_elem, 2, dtype=dtype) / n_elem))
    seq_idx = torch.arange(seq_len, dtype=dtype)
    idx_theta = torch.outer(seq_idx, theta).float()
    cache = torch.stack([torch.cos(idx_theta), torch.sin(idx_theta)], dim=-1)
    if dtype in (torch.float16, torch.bfloat16, torch.int8):
        cache = cache.half()
    return cache.to(device)


def apply_rope(x: torch.Tensor, rope_cache: RoPECache) -> torch.Tensor:
    """Rotate interleaved pairs of x [B, T, n_head, hs] by rope_cache[:T] in f32 (lit_llama/model.py:306-323)."""
    return ops.apply_rope(x, rope_cache)

"""CPU restatement of lit-llama's inference hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg may import this module; the
product (`lit_llama_amd`) never does — it has no CPU fallback.

Every function restates, op for op and in the same order, the code of /root/reference cited next to it, in
torch CPU tensors (float32 unless the caller passes another dtype), so that on the same inputs it reproduces the
reference's CPU path ("32-true", generate.py:123).  It is PINNED by tests/golden/*.npz, which
oracle/gen_golden.py produced by importing and running the unmodified reference in the build container
(tests/test_oracle_golden.py re-checks the pin on every CPU run).

Exception — parity unpinned: `llm_int8_linear` / `int8_quant_rows`.  The arithmetic of `Linear8bitLt.forward`
lives in bitsandbytes (third party, unpinned in pyproject.toml:19, not installed, CUDA only), and the reference
has no test or golden vector for it.  The restatement follows the published LLM.int8() algorithm (Dettmers et
al. 2022, vector-wise absmax int8 for both operands, int32 accumulation, 1/127^2 dequantisation, fp16 outlier
decomposition at |x| >= 6) as bitsandbytes' MatMul8bitLt implements it, anchored on the reference call sites
lit_llama/quantization.py:38-77.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

StateDict = Dict[str, torch.Tensor]
KVCache = Tuple[torch.Tensor, torch.Tensor]


# ------------------------------------------------------------------------------------------ small ops
def find_multiple(n: int, k: int) -> int:
    """lit_llama/utils.py:38-41"""
    return n if n % k == 0 else n + k - (n % k)


def rmsnorm(x: torch.Tensor, scale: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """lit_llama/model.py:270-277"""
    norm_x = torch.mean(x * x, dim=-1, keepdim=True)
    x_normed = x * torch.rsqrt(norm_x + eps)
    return scale * x_normed


def build_rope_cache(seq_len: int, n_elem: int, dtype: torch.dtype = torch.int64, base: int = 10000) -> torch.Tensor:
    """lit_llama/model.py:280-303 (the model passes the integer dtype of `idx`, :132)"""
    theta = 1.0 / (base ** (torch.arange(0, n_elem, 2, dtype=dtype) / n_elem))
    seq_idx = torch.arange(seq_len, dtype=dtype)
    idx_theta = torch.outer(seq_idx, theta).float()
    cache = torch.stack([torch.cos(idx_theta), torch.sin(idx_theta)], dim=-1)
    if dtype in (torch.float16, torch.bfloat16, torch.int8):
        cache = cache.half()
    return cache


def apply_rope(x: torch.Tensor, rope_cache: torch.Tensor) -> torch.Tensor:
    """lit_llama/model.py:306-323; x [B, T, n_head, hs], rope_cache [>=T, hs/2, 2]"""
    T = x.size(1)
    rope_cache = rope_cache[:T]
    xshaped = x.float().reshape(*x.shape[:-1], -1, 2)
    rope_cache = rope_cache.view(1, xshaped.size(1), 1, xshaped.size(3), 2)
    x_out2 = torch.stack(
        [
            xshaped[..., 0] * rope_cache[..., 0] - xshaped[..., 1] * rope_cache[..., 1],
            xshaped[..., 1] * rope_cache[..., 0] + xshaped[..., 0] * rope_cache[..., 1],
        ],
        -1,
    )
    return x_out2.flatten(3).type_as(x)


# ------------------------------------------------------------------------------------------ ColBlock (GPTQ) linear
def colblock_pack(weight: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor, bits: int, tile_cols: int):
    """ColBlockQuantizedLinear.pack_weight, lit_llama/quantization.py:376-390 (truncating uint8 conversion).
    Returns quant_weight [N, K * bits / 8] uint8 (values only; the column-major storage is a layout detail)."""
    weight = weight.clone().float()
    for j in range(scales.size(1)):
        sl = slice(j * tile_cols, (j + 1) * tile_cols)
        weight[:, sl] /= scales[:, j : j + 1]
        weight[:, sl] += zeros[:, j : j + 1]
    weight = weight.clamp_(min=0, max=2**bits - 1).to(dtype=torch.uint8)
    epb = 8 // bits
    q = torch.zeros((weight.shape[0], weight.shape[1] // epb), dtype=torch.uint8)
    for nr in range(epb):
        q += weight[:, nr::epb] << (nr * bits)
    return q


def colblock_get_weight(quant_weight, scales, zeros, bits: int, tile_cols: int, dtype=torch.float32):
    """ColBlockQuantizedLinear.get_weight, lit_llama/quantization.py:392-411"""
    epb = 8 // bits
    N, K = quant_weight.shape[0], quant_weight.shape[1] * epb
    weight = torch.empty((N, K), dtype=dtype)
    mask = (1 << bits) - 1
    for nr in range(epb):
        weight[:, nr::epb] = ((quant_weight >> (nr * bits)) & mask).float()
    for j in range(scales.size(1)):
        sl = slice(j * tile_cols, (j + 1) * tile_cols)
        weight[:, sl] -= zeros[:, j : j + 1]
        weight[:, sl] *= scales[:, j : j + 1]
    return weight


def colblock_linear(inp, quant_weight, scales, zeros, bits: int, tile_cols: int, bias=None):
    """ColBlockQuantizedLinear.forward off the Triton branch, lit_llama/quantization.py:422-423: the whole weight
    is dequantised in the activation dtype on every call, then F.linear."""
    weight = colblock_get_weight(quant_weight, scales.to(inp.dtype), zeros.to(inp.dtype), bits, tile_cols,
                                 dtype=inp.dtype)
    return F.linear(inp, weight, bias)


# ------------------------------------------------------------------------------------------ LoRA (inference variant)
def lora_delta(lora_A: torch.Tensor, lora_B: torch.Tensor, enable_lora=(True, False, True)) -> torch.Tensor:
    """The weight update of lit_llama/lora.py `MergedLinear` before scaling, zero-padded to the rows of the fused
    q / k / v weight (lora.py:272-279 with zero_pad :205-241): lora_A [r * n_on, in], lora_B [out / n * n_on, r];
    group g of the enabled projections gets B_g @ A_g (the reference spells it as a grouped 1x1 conv1d), the disabled
    ones zeros.  Computed in the parameters' dtype, like the reference."""
    n, n_on = len(enable_lora), sum(enable_lora)
    r = lora_B.shape[1]
    rows = lora_B.shape[0] // n_on
    out = lora_A.new_zeros((rows * n, lora_A.shape[1]))
    g = 0
    for j, on in enumerate(enable_lora):
        if on:
            out[j * rows:(j + 1) * rows] = lora_B[g * rows:(g + 1) * rows] @ lora_A[g * r:(g + 1) * r]
            g += 1
    return out


def lora_merge(weight: torch.Tensor, lora_A: torch.Tensor, lora_B: torch.Tensor, alpha: float,
               enable_lora=(True, False, True)) -> torch.Tensor:
    """`MergedLinear.train(False)` (lora.py:243-280): W + zero_pad((B A) * alpha / r), every step rounded in W's dtype."""
    r = lora_B.shape[1]
    return weight + lora_delta(lora_A, lora_B, enable_lora) * (alpha / r)


def lora_forward_unmerged(x, weight, lora_A, lora_B, alpha: float, enable_lora=(True, False, True)):
    """`MergedLinear.forward` with separate LoRA matrices and dropout 0 (lora.py:308-326)."""
    r = lora_B.shape[1]
    n, n_on = len(enable_lora), sum(enable_lora)
    rows = lora_B.shape[0] // n_on
    y = F.linear(x, weight)
    after_a = F.linear(x, lora_A)
    g = 0
    for j, on in enumerate(enable_lora):
        if on:
            y[..., j * rows:(j + 1) * rows] += F.linear(after_a[..., g * r:(g + 1) * r], lora_B[g * rows:(g + 1) * rows]) * (alpha / r)
            g += 1
    return y


# ------------------------------------------------------------------------------------------ LLM.int8 (parity unpinned)
MM_DEQUANT_CONST = 6.200012e-05  # 1 / (127 * 127) as bitsandbytes spells it


def int8_quant_rows(w: torch.Tensor):
    """bnb.functional.double_quant(W.half()) row statistics as used at lit_llama/quantization.py:69-77:
    SCB[n] = max_k |W[n,k]| (f16 values, f32 statistic), CB = rint(W * (127 / SCB))."""
    wh = w.half().float()
    scb = wh.abs().amax(dim=1)
    # IEEE division (a Python float divided by a tensor would be reciprocal-then-multiply: two roundings)
    inv = torch.where(scb > 0, torch.full_like(scb, 127.0) / scb, torch.zeros_like(scb))
    cb = torch.round(wh * inv[:, None]).to(torch.int8)  # torch.round = round half to even = rintf
    return cb, scb


def llm_int8_linear(x: torch.Tensor, cb: torch.Tensor, scb: torch.Tensor, bias=None, threshold: float = 6.0):
    """MatMul8bitLt forward with has_fp16_weights=False (Linear8bitLt, lit_llama/quantization.py:46-47).
    x [..., K] any float dtype; returns x.dtype."""
    shape = x.shape
    xh = x.reshape(-1, shape[-1]).half().float()  # inputs are cast to fp16 (the warning filtered at :13-20)
    absx = xh.abs()
    if threshold > 0:
        outlier_cols = (absx >= threshold).any(dim=0)
        stat = torch.where(absx >= threshold, torch.zeros_like(absx), absx)
    else:
        outlier_cols = torch.zeros(xh.shape[1], dtype=torch.bool)
        stat = absx
    sca = stat.amax(dim=1)  # row absmax over sub-threshold entries
    inv = torch.where(sca > 0, torch.full_like(sca, 127.0) / sca, torch.zeros_like(sca))
    ca = torch.round(xh * inv[:, None])
    ca[:, outlier_cols] = 0  # CA[:, idx] = 0
    acc = (ca.to(torch.float64) @ cb.to(torch.float64).t()).to(torch.int64)  # exact int32 accumulation
    out = ((acc.float() * MM_DEQUANT_CONST) * sca[:, None]) * scb[None, :]
    if bias is not None:
        out = out + bias.float()[None, :]
    out = out.half().float()
    if bool(outlier_cols.any()):
        sub_a = xh[:, outlier_cols]  # fp16 values
        sub_b = (cb[:, outlier_cols].float() * scb[:, None])
        sub_b = (sub_b / torch.full_like(sub_b, 127.0)).half().float()  # [N, n_out]
        mm = (sub_a @ sub_b.t()).half().float()  # fp16 GEMM, fp32 accumulate, rounded once
        out = (out + mm).half().float()
    return out.to(x.dtype).reshape(*shape[:-1], cb.shape[0])


# ------------------------------------------------------------------------------------------ model
class Config:
    """LLaMAConfig, lit_llama/model.py:25-48"""

    def __init__(self, n_layer=32, n_head=32, n_embd=4096, vocab_size=32000, block_size=2048,
                 padded_vocab_size=None):
        self.n_layer, self.n_head, self.n_embd = n_layer, n_head, n_embd
        self.vocab_size, self.block_size = vocab_size, block_size
        self.padded_vocab_size = padded_vocab_size or find_multiple(vocab_size, 64)

    @property
    def n_hidden(self):  # lit_llama/model.py:243-245
        return find_multiple(int(2 * (4 * self.n_embd) / 3), 256)


def linear(sd: StateDict, prefix: str, x: torch.Tensor, mode: Optional[str]) -> torch.Tensor:
    """The L1 plug-in dispatch: nn.Linear / ColBlockQuantizedLinear / Linear8bitLt."""
    if prefix + ".quant_weight" in sd:
        bits = 8 if mode == "gptq.int8" else 4
        qw = sd[prefix + ".quant_weight"]
        K = qw.shape[1] * (8 // bits)
        scales, zeros = sd[prefix + ".scales"], sd[prefix + ".zeros"]
        tile_cols = (K + scales.shape[1] - 1) // scales.shape[1]
        return colblock_linear(x, qw, scales, zeros, bits, tile_cols)
    w = sd[prefix + ".weight"]
    if mode == "llm.int8":
        key = prefix + ".__int8__"
        if key not in sd:  # quantise once, as Linear8bitLt does at load time (:52-67)
            sd[key] = int8_quant_rows(w)
        cb, scb = sd[key]
        return llm_int8_linear(x, cb, scb)
    y = F.linear(x, w.to(x.dtype))
    if prefix + ".adapter_scale" in sd:  # LLaMA-Adapter v2, lit_llama/adapter_v2.py:29-32: scale * (W x + bias)
        y = sd[prefix + ".adapter_scale"].to(x.dtype) * (y + sd[prefix + ".adapter_bias"].to(x.dtype))
    return y


class Model:
    """LLaMA (lit_llama/model.py:51-145) over a flat state dict with the reference's key names."""

    def __init__(self, cfg: Config, sd: StateDict, mode: Optional[str] = None, dtype=torch.float32):
        self.cfg, self.sd, self.mode, self.dtype = cfg, sd, mode, dtype
        self.rope_cache: Optional[torch.Tensor] = None
        self.mask_cache: Optional[torch.Tensor] = None
        self.kv_caches: List[KVCache] = []

    def reset_cache(self):
        self.kv_caches.clear()

    def p(self, key: str) -> torch.Tensor:
        return self.sd[key].to(self.dtype)

    def attention(self, i: int, x, rope, mask, max_seq_length, input_pos, kv_cache):
        """CausalSelfAttention.forward, lit_llama/model.py:185-237"""
        cfg = self.cfg
        pre = f"transformer.h.{i}.attn."
        B, T, C = x.size()
        q, k, v = linear(self.sd, pre + "c_attn", x, self.mode).split(cfg.n_embd, dim=2)
        hs = C // cfg.n_head
        k = k.view(B, T, cfg.n_head, hs)
        q = q.view(B, T, cfg.n_head, hs)
        v = v.view(B, T, cfg.n_head, hs)
        q = apply_rope(q, rope)
        k = apply_rope(k, rope)
        k, q, v = k.transpose(1, 2), q.transpose(1, 2), v.transpose(1, 2)
        if kv_cache is not None:
            cache_k, cache_v = kv_cache
            if input_pos[-1] >= max_seq_length:
                input_pos = torch.tensor(max_seq_length - 1)
                cache_k = torch.roll(cache_k, -1, dims=2)
                cache_v = torch.roll(cache_v, -1, dims=2)
            k = cache_k.index_copy(2, input_pos, k)
            v = cache_v.index_copy(2, input_pos, v)
            kv_cache = k, v
        y = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0)
        y = y.transpose(1, 2).contiguous().view(B, T, C)
        return linear(self.sd, pre + "c_proj", y, self.mode), kv_cache

    def mlp(self, i: int, x):
        """MLP.forward, lit_llama/model.py:251-254"""
        pre = f"transformer.h.{i}.mlp."
        x = F.silu(linear(self.sd, pre + "c_fc1", x, self.mode)) * linear(self.sd, pre + "c_fc2", x, self.mode)
        return linear(self.sd, pre + "c_proj", x, self.mode)

    def block(self, i: int, x, rope, mask, max_seq_length, input_pos=None, kv_cache=None):
        """Block.forward, lit_llama/model.py:165-168"""
        pre = f"transformer.h.{i}."
        h, new_kv = self.attention(i, rmsnorm(x, self.p(pre + "rms_1.scale")), rope, mask, max_seq_length, input_pos,
                                   kv_cache)
        x = x + h
        x = x + self.mlp(i, rmsnorm(x, self.p(pre + "rms_2.scale")))
        return x, new_kv

    def forward(self, idx: torch.Tensor, max_seq_length: Optional[int] = None,
                input_pos: Optional[torch.Tensor] = None) -> torch.Tensor:
        """LLaMA.forward, lit_llama/model.py:76-122"""
        cfg = self.cfg
        B, T = idx.size()
        if max_seq_length is None:
            max_seq_length = cfg.block_size
        assert T <= max_seq_length <= cfg.block_size
        if self.rope_cache is None:
            self.rope_cache = build_rope_cache(cfg.block_size, cfg.n_embd // cfg.n_head, dtype=idx.dtype)
        if self.mask_cache is None:
            ones = torch.ones((cfg.block_size, cfg.block_size), dtype=torch.bool)
            self.mask_cache = torch.tril(ones).unsqueeze(0).unsqueeze(0)
        if input_pos is not None:
            rope = self.rope_cache.index_select(0, input_pos)
            mask = self.mask_cache.index_select(2, input_pos)
            mask = mask[:, :, :, :max_seq_length]
        else:
            rope = self.rope_cache[:T]
            mask = self.mask_cache[:, :, :T, :T]
        x = F.embedding(idx, self.p("transformer.wte.weight"))
        if input_pos is None:
            for i in range(cfg.n_layer):
                x, _ = self.block(i, x, rope, mask, max_seq_length)
        else:
            if not self.kv_caches:
                hs = cfg.n_embd // cfg.n_head
                shape = (B, cfg.n_head, max_seq_length, hs)
                self.kv_caches = [(torch.zeros(shape, dtype=x.dtype), torch.zeros(shape, dtype=x.dtype))
                                  for _ in range(cfg.n_layer)]
            for i in range(cfg.n_layer):
                x, self.kv_caches[i] = self.block(i, x, rope, mask, max_seq_length, input_pos, self.kv_caches[i])
        x = rmsnorm(x, self.p("transformer.ln_f.scale"))
        return linear(self.sd, "lm_head", x, self.mode)

    __call__ = forward


class AdapterModel(Model):
    """LLaMA-Adapter (lit_llama/adapter.py:62-171): from layer `adapter_start_layer` on, the attention output gets
    gating_factor * softmax(q ak^T / sqrt(hs)) av added, where ak / av are the k / v projections (no RoPE) of the
    `adapter_prompt_length` learned prefix rows `adapter_wte.weight` and q is the RoPE'd query of the token."""

    def __init__(self, cfg: Config, sd: StateDict, mode: Optional[str] = None, dtype=torch.float32,
                 adapter_prompt_length: int = 10, adapter_start_layer: int = 2):
        super().__init__(cfg, sd, mode, dtype)
        self.adapter_prompt_length, self.adapter_start_layer = adapter_prompt_length, adapter_start_layer

    def attention(self, i: int, x, rope, mask, max_seq_length, input_pos=None, kv_cache=None):
        cfg = self.cfg
        pre = f"transformer.h.{i}.attn."
        B, T, C = x.size()
        q, k, v = linear(self.sd, pre + "c_attn", x, self.mode).split(cfg.n_embd, dim=2)
        hs = C // cfg.n_head
        k = k.view(B, T, cfg.n_head, hs)
        q = q.view(B, T, cfg.n_head, hs)
        v = v.view(B, T, cfg.n_head, hs)
        q = apply_rope(q, rope)
        k = apply_rope(k, rope)
        k, q, v = k.transpose(1, 2), q.transpose(1, 2), v.transpose(1, 2)
        if kv_cache is not None:
            cache_k, cache_v = kv_cache
            if input_pos[-1] >= max_seq_length:
                input_pos = torch.tensor(max_seq_length - 1)
                cache_k = torch.roll(cache_k, -1, dims=2)
                cache_v = torch.roll(cache_v, -1, dims=2)
            k = cache_k.index_copy(2, input_pos, k)
            v = cache_v.index_copy(2, input_pos, v)
            kv_cache = k, v
        y = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0)
        if i >= self.adapter_start_layer:  # adapter.py:134-151
            prefix = self.p(pre + "adapter_wte.weight").reshape(1, self.adapter_prompt_length, C)
            aT = prefix.size(1)
            _, ak, av = linear(self.sd, pre + "c_attn", prefix, self.mode).split(cfg.n_embd, dim=2)
            ak = ak.view(1, aT, cfg.n_head, hs).repeat(B, 1, 1, 1).transpose(1, 2)
            av = av.view(1, aT, cfg.n_head, hs).repeat(B, 1, 1, 1).transpose(1, 2)
            amask = torch.ones(q.shape[-2], ak.shape[-2], dtype=torch.bool)
            ay = F.scaled_dot_product_attention(q, ak, av, attn_mask=amask, dropout_p=0.0, is_causal=False)
            y = y + self.p(pre + "gating_factor") * ay
        y = y.transpose(1, 2).contiguous().view(B, T, C)
        return linear(self.sd, pre + "c_proj", y, self.mode), kv_cache


@torch.no_grad()
def generate(model: Model, idx: torch.Tensor, max_new_tokens: int, *, max_seq_length: Optional[int] = None,
             temperature: float = 1.0, top_k: Optional[int] = None, eos_id: Optional[int] = None,
             logits_log: Optional[list] = None) -> torch.Tensor:
    """generate(), /root/reference generate.py:20-91.  `logits_log`, if given, receives the last-position logits
    of every step (the quantity the parity tests compare)."""
    T = idx.size(0)
    T_new = T + max_new_tokens
    if max_seq_length is None:
        cfg = model.config if hasattr(model, "config") else model.cfg  # generate.py:42 reads model.config
        max_seq_length = min(T_new, cfg.block_size)
    device, dtype = idx.device, idx.dtype  # generate.py:44: the loop runs on the prompt's device
    empty = torch.empty(T_new, dtype=dtype, device=device)
    empty[:T] = idx
    idx = empty
    input_pos = torch.arange(0, T, device=device)
    for _ in range(max_new_tokens):
        x = idx.index_select(0, input_pos).view(1, -1)
        logits = model(x, max_seq_length, input_pos)
        logits = logits[0, -1] / temperature
        if logits_log is not None:
            logits_log.append(logits.float().clone())
        if top_k is not None:
            v, _ = torch.topk(logits, min(top_k, logits.size(-1)))
            logits = torch.where(logits < v[[-1]], -float("Inf"), logits)
        probs = torch.nn.functional.softmax(logits, dim=-1)
        idx_next = torch.multinomial(probs, num_samples=1).to(dtype=dtype)
        input_pos = input_pos[-1:] + 1
        idx = idx.index_copy(0, input_pos, idx_next)
        if idx_next == eos_id:
            return idx[:input_pos]
    return idx


def sample_from_uniform(logits: torch.Tensor, temperature: float, top_k: Optional[int], u: float):
    """generate.py:68-76 with the multinomial draw replaced by the inverse CDF of a given uniform (torch.multinomial
    consumes its own noise, so a sample cannot be compared bit for bit; kept set and probabilities can):
    returns (token, probs).  token = min {i : cumsum(probs)[i] > u}."""
    logits = logits.float() / temperature
    if top_k is not None:
        v, _ = torch.topk(logits, min(top_k, logits.size(-1)))
        logits = torch.where(logits < v[[-1]], -float("Inf"), logits)
    probs = torch.nn.functional.softmax(logits, dim=-1)
    cdf = torch.cumsum(probs.double(), dim=-1)
    above = (cdf > u).nonzero()
    token = int(above[0]) if above.numel() else int((probs > 0).nonzero()[-1])
    return token, probs


@torch.no_grad()
def teacher_forced_logits(model: Model, tokens: torch.Tensor, prompt_len: int,
                          max_seq_length: Optional[int] = None) -> torch.Tensor:
    """Last-position logits for each decode step when the model is fed `tokens` (a finished generation):
    row j is the distribution that produced tokens[prompt_len + j]."""
    n_steps = tokens.numel() - prompt_len
    if max_seq_length is None:
        max_seq_length = min(tokens.numel(), model.cfg.block_size)
    model.reset_cache()
    out = []
    input_pos = torch.arange(0, prompt_len)
    for j in range(n_steps):
        x = tokens.index_select(0, input_pos).view(1, -1)
        out.append(model(x, max_seq_length, input_pos)[0, -1].float())
        input_pos = input_pos[-1:] + 1
    model.reset_cache()
    return torch.stack(out)

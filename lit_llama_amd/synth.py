"""Deterministic synthetic checkpoints and prompts — there are no LLaMA weights offline.

State dicts use the reference's key names (scripts/convert_checkpoint.py:24-53) and buffer layouts, so they
load into `lit_llama.LLaMA` (the reference, in tests) and `lit_llama_amd.LLaMA` alike:

  fp          : every Linear W ~ N(0, 1/K); embedding ~ N(0, 1); RMSNorm scales 1 + 0.1 N(0, 1)
  gptq.int4   : the fp weights quantised per output row, asymmetric min/max round-to-nearest exactly as
                GPTQQuantizer.find_params_weight / quantize_weight (lit_llama/quantization.py:472-513, perchannel,
                not sym); integer levels written straight into the packed column-major buffer (:350-359,
                :387-390); scales are rounded to bf16 ONCE here, so the bf16 GPU buffers and the f32 oracle
                dequantise identical weights; zeros are small integers (exact in any dtype)
  llm.int8    : fp weights with a few input channels of the embedding / norm scales boosted so activations cross
                the |x| >= 6 outlier threshold

`stats="llama"` (round 5) bends the same random draws towards what a TRAINED LLaMA checkpoint looks like to a kernel that
stages activations in a narrow format (the persistent step's fp8-limb hand-offs): embeddings of std 0.02, RMSNorm scales from
~0.05 (first block) to ~0.5 (last block) with a log-normal spread per channel, a few hidden units of block 1 whose
c_fc1 / c_fc2 rows coincide and are scaled so that their SwiGLU output reaches 10^3..10^4 on the tokens that excite them, and an
mlp.c_proj that routes those units into three fixed residual channels only ("massive activations": hundreds of times the
rms of the stream, carried to the last block).  See `llama_stats_plan`.

`device="cuda"` generates directly in HBM with the torch CUDA generator (bench); the CPU generator (tests) is
what the golden fixtures were produced from.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .model import LLaMAConfig


def linear_shapes(cfg: LLaMAConfig):
    """(state-dict prefix, out_features, in_features) of every linear, in module order."""
    C, H, V = cfg.n_embd, cfg.n_hidden, cfg.padded_vocab_size
    out = [("lm_head", V, C)]
    for i in range(cfg.n_layer):
        p = f"transformer.h.{i}."
        out += [(p + "attn.c_attn", 3 * C, C), (p + "attn.c_proj", C, C), (p + "mlp.c_fc1", H, C),
                (p + "mlp.c_fc2", H, C), (p + "mlp.c_proj", C, H)]
    return out


def rtn_quantize_rows(w: torch.Tensor, bits: int = 4):
    """Per-row asymmetric round-to-nearest: returns (q uint8 [N, K], scale f32 [N], zero f32 [N])."""
    maxq = 2**bits - 1
    w = w.float()
    zero_t = torch.zeros(w.shape[0], device=w.device)
    xmin = torch.minimum(w.min(1)[0], zero_t)
    xmax = torch.maximum(w.max(1)[0], zero_t)
    flat = (xmin == 0) & (xmax == 0)
    xmin[flat] = -1
    xmax[flat] = +1
    scale = (xmax - xmin) / maxq
    zero = torch.round(-xmin / scale)
    q = torch.clamp(torch.round(w / scale[:, None]) + zero[:, None], 0, maxq).to(torch.uint8)
    return q, scale, zero


def pack_colblock(q: torch.Tensor, bits: int = 4) -> torch.Tensor:
    """q [N, K] uint8 levels -> quant_weight [N, K * bits / 8] uint8 with stride (1, N)."""
    epb = 8 // bits
    N, K = q.shape
    packed = torch.zeros((N, K // epb), dtype=torch.uint8, device=q.device)
    for nr in range(epb):
        packed |= q[:, nr::epb] << (nr * bits)
    return packed.t().contiguous().t()


def _randn(shape, gen, device, std=1.0):
    return torch.randn(shape, generator=gen, device=device, dtype=torch.float32) * std


def llama_stats_plan(cfg: LLaMAConfig) -> dict:
    """Where `stats="llama"` puts its structure: the block, the hidden units with coinciding c_fc1 / c_fc2 rows (SwiGLU output
    a * silu(a) ~ a^2 for a > 0), the residual channels they feed, and the sizes involved.  Kept in one place so that the tests can
    look at the very rows / channels the generator touched."""
    C, H = cfg.n_embd, cfg.n_hidden
    return dict(
        layer=min(1, cfg.n_layer - 1),
        hidden_units=[H * k // 11008 for k in (7003, 4242, 9000, 1234)],
        channels=[C * k // 4096 for k in (1415, 2533, 3431)],
        sigma_a=104.0,       # nominal std of the pre-activation a = b of a massive hidden unit (as measured on the quantised 7B-width rows,
                             # over units AND tokens: ~76; most of it is BETWEEN units — the attention output of block 0 is close to a
                             # running mean, so a unit's a moves by only +-30 % from token to token): the strongest unit's a^2 is
                             # 3000 .. 11000 over the tokens, i.e. on either side of the +-7168 the fp8 hand-off's SwiGLU edge holds
        w_massive=0.004,     # mlp.c_proj weight from a massive hidden unit into a massive channel
        row_shrink=0.25,     # the other entries of those mlp.c_proj rows (keeps w_massive several int4 steps wide)
        emb_std=0.02, norm_lo=0.05, norm_hi=0.5, norm_spread=0.3,
        # trained checkpoints carry SMALL norm weights on their massive channels (the channels act as a constant the blocks read at a
        # chosen gain): without this the three channels dominate every normalised vector behind block 1 — the first version of the
        # fixture generated one token 23 times over
        norm_damp=0.03, ln_f_damp=0.01,
    )


def make_state_dict(
    cfg: LLaMAConfig,
    *,
    seed: int = 0,
    mode: Optional[str] = None,
    dtype: torch.dtype = torch.float32,
    device: str = "cpu",
    outlier_channels: int = 0,
    bf16_exact: bool = True,
    group_cols: int = 0,
    stats: str = "unit",
) -> Dict[str, torch.Tensor]:
    """Synthetic checkpoint.  mode None / "llm.int8": float weights in `dtype`; "gptq.int4" / "gptq.int8": packed buffers with
    scales / zeros in `dtype` (one pair per row, or per row and group of `group_cols` input columns: the
    ColBlockQuantizedLinear layout with tile_cols = group_cols, lit_llama/quantization.py:350-374).  With `bf16_exact` every float value is rounded to bf16 once at generation time
    (and stored in `dtype`), so a bf16 GPU model and the f32 CPU oracle hold identical parameters."""
    assert mode in (None, "gptq.int4", "gptq.int8", "llm.int8") and stats in ("unit", "llama")
    bits = 8 if mode == "gptq.int8" else 4
    plan = llama_stats_plan(cfg) if stats == "llama" else None
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    C = cfg.n_embd
    sd: Dict[str, torch.Tensor] = {}
    wte = _randn((cfg.padded_vocab_size, C), gen, device)
    ln = lambda: 1.0 + 0.1 * _randn((C,), gen, device)  # noqa: E731
    norms = {"transformer.ln_f.scale": ln()}
    for i in range(cfg.n_layer):
        norms[f"transformer.h.{i}.rms_1.scale"] = ln()
        norms[f"transformer.h.{i}.rms_2.scale"] = ln()
    if plan is not None:
        # (the same draws, re-read: d = (scale - 1) / 0.1 is the N(0, 1) sample behind a norm scale)
        wte = wte * plan["emb_std"]
        for i in range(cfg.n_layer):
            base = plan["norm_lo"] * (plan["norm_hi"] / plan["norm_lo"]) ** (i / max(cfg.n_layer - 1, 1))
            for j in (1, 2):
                k = f"transformer.h.{i}.rms_{j}.scale"
                norms[k] = base * torch.exp(plan["norm_spread"] * (norms[k] - 1.0) / 0.1)
                if (i, j) > (plan["layer"], 2):  # every norm that reads the stream behind the block that creates the channels
                    norms[k][plan["channels"]] *= plan["norm_damp"]
        norms["transformer.ln_f.scale"][plan["channels"]] *= plan["ln_f_damp"]
    if outlier_channels:
        # fixed channels x20: after RMSNorm these exceed the LLM.int8 threshold of 6
        ch = torch.arange(outlier_channels, device=device) * (C // max(outlier_channels, 1)) + 3
        for k in norms:
            norms[k][ch] *= 20.0
    rnd = (lambda t: t.to(torch.bfloat16).float()) if bf16_exact else (lambda t: t)  # noqa: E731
    sd["transformer.wte.weight"] = rnd(wte).to(dtype)
    for k, v in norms.items():
        sd[k] = rnd(v).to(dtype)
    fc1_rows = None
    for prefix, N, K in linear_shapes(cfg):
        w = _randn((N, K), gen, device, std=K**-0.5)
        if plan is not None and prefix.startswith(f"transformer.h.{plan['layer']}.mlp."):
            hu = torch.tensor(plan["hidden_units"], device=device)
            if prefix.endswith("c_fc1"):
                # a = row . (g x / rms) has std ~ sqrt(mean g^2) for a N(0, 1/K) row: the gain brings it to sigma_a
                g2 = norms[f"transformer.h.{plan['layer']}.rms_2.scale"]
                w[hu] *= plan["sigma_a"] / float(g2.square().mean().sqrt())
                fc1_rows = w[hu].clone()
            elif prefix.endswith("c_fc2"):
                w[hu] = fc1_rows  # b = a: silu(a) * b = a^2 sigmoid(a)
            else:  # mlp.c_proj: the massive units feed the massive channels and nothing else
                ch = torch.tensor(plan["channels"], device=device)
                w[:, hu] = 0.0
                w[ch] *= plan["row_shrink"]
                w[ch[:, None], hu[None, :]] = plan["w_massive"]
        if mode in ("gptq.int4", "gptq.int8") and 0 < group_cols < K:
            ng = -(-K // group_cols)
            q = torch.empty((N, K), dtype=torch.uint8, device=device)
            scale, zero = torch.empty((N, ng), device=device), torch.empty((N, ng), device=device)
            for j in range(ng):
                sl = slice(j * group_cols, (j + 1) * group_cols)
                q[:, sl], scale[:, j], zero[:, j] = rtn_quantize_rows(w[:, sl], bits)
            sd[prefix + ".quant_weight"] = pack_colblock(q, bits)
            sd[prefix + ".scales"] = scale.to(torch.bfloat16).float().to(dtype)
            sd[prefix + ".zeros"] = zero.to(dtype)
        elif mode in ("gptq.int4", "gptq.int8"):
            q, scale, zero = rtn_quantize_rows(w, bits)
            scale = scale.to(torch.bfloat16).float()  # rounded once; identical on both sides
            sd[prefix + ".quant_weight"] = pack_colblock(q, bits)
            sd[prefix + ".scales"] = scale[:, None].to(dtype)
            sd[prefix + ".zeros"] = zero[:, None].to(dtype)
        else:
            sd[prefix + ".weight"] = rnd(w).to(dtype)
    return sd


def make_adapter_state(cfg, *, seed: int = 0, adapter_prompt_length: int = 10, adapter_start_layer: int = 2,
                       dtype: torch.dtype = torch.float32) -> Dict[str, torch.Tensor]:
    """Seeded LLaMA-Adapter parameters (lit_llama/adapter.py:79-86 key names and shapes): prefix rows and NON-zero gating
    factors (a zero gate, the reference's initial value, would hide the prefix term), bf16-exact values."""
    gen = torch.Generator()
    gen.manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for i in range(adapter_start_layer, cfg.n_layer):
        pre = f"transformer.h.{i}.attn."
        out[pre + "adapter_wte.weight"] = (torch.randn((adapter_prompt_length, cfg.n_embd), generator=gen)
                                           ).to(torch.bfloat16).to(dtype)
        out[pre + "gating_factor"] = (0.5 * torch.randn((1, cfg.n_head, 1, 1), generator=gen)).to(torch.bfloat16).to(dtype)
    return out


def make_adapter_v2_state(sd: Dict[str, torch.Tensor], *, seed: int = 0, dtype: torch.dtype = torch.float32):
    """Seeded LLaMA-Adapter v2 parameters for every linear of `sd` (lit_llama/adapter_v2.py:35-40 names): scales around 1,
    small biases, bf16-exact values."""
    gen = torch.Generator()
    gen.manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for k in sorted(sd):
        if k.endswith(".weight") and sd[k].dim() == 2 and "wte" not in k:
            n = sd[k].shape[0]
            pre = k[: -len("weight")]
            out[pre + "adapter_scale"] = (1.0 + 0.1 * torch.randn(n, generator=gen)).to(torch.bfloat16).to(dtype)
            out[pre + "adapter_bias"] = (0.1 * torch.randn(n, generator=gen)).to(torch.bfloat16).to(dtype)
    return out


def make_prompt(length: int, vocab: int = 32000, seed: int = 1234, device: str = "cpu") -> torch.Tensor:
    """BOS (id 1) followed by uniform ids, int32 1-D (tokenizer.py:43 returns torch.int)."""
    gen = torch.Generator(device="cpu")
    gen.manual_seed(seed)
    ids = torch.randint(0, vocab, (length,), generator=gen, dtype=torch.int64)
    ids[0] = 1
    return ids.to(torch.int32).to(device)


def fill_model_random_int4(model, seed: int = 0, zero: float = 7.5, gain: float = 1.0) -> None:
    """Bench-only shortcut for multi-GB models: fill every ColBlockQuantizedLinear of a GPU model in place with
    uniformly random levels, the zero point `zero` and scales ~ gain x 0.217 / sqrt(K) (uniform levels have a std of 4.61, so a
    row's weights have std ~ gain / sqrt(K)), and the embedding / norm scales as in make_state_dict — without materialising fp32
    weights first.

    Defaults (round 5): zero-MEAN weights of unit gain, i.e. the statistics of make_state_dict's N(0, 1 / K) rows.  Rounds 1-4 used
    `zero=8.0, gain=2.2` (scale = 7.2 / 15 / sqrt(K), the step of a 16-level grid over a Gaussian row, under UNIFORM levels):
    * zero 8 puts -0.5 scale on every weight, a common-mode gain of -0.5 scale K ~ -15 on every linear: the 7B model's residual stream
      is one constant vector that grows by ~6000 per block (mean -1.9 x 10^5 behind block 31, oracle on the CPU), its logits are a
      common-mode term — every path agrees to 0.007 std on them whatever it does to the rest — and the x edge behind block 1 sits at
      ~700 x the previous edge's rms, on either side of the +-448 of an fp8-limb hand-off depending on the token: the bench's own SAMPLED
      generation clipped and was replayed one rung down (403 instead of ~1000 tok/s) while its greedy run happened not to;
    * zero 7.5 with gain 2.2 is a CHAOTIC random network: 2 blocks sit 0.06 std from the oracle, 32 blocks 0.3-1.1 std from each other
      on every pair of paths (tests/diag_bench_model_parity.py, profiles/r05_bench_model_zero_point.txt).
    Timing does not see the values (same box, alternating: 900.7 / 899.2 us per step with zero 8, 907.8 / 901.3 with 7.5).
    tests/test_fused_step_gpu.py keeps `zero=8.0, gain=2.2` as the stress case it always was."""
    from .quantization import ColBlockQuantizedLinear

    dev = model.transformer.wte.weight.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    with torch.no_grad():
        model.transformer.wte.weight.copy_(_randn(tuple(model.transformer.wte.weight.shape), gen, dev))
        for name, mod in model.named_modules():
            if isinstance(mod, ColBlockQuantizedLinear):
                N, Kb = mod.quant_weight.shape
                raw = torch.randint(0, 256, (Kb, N), generator=gen, device=dev, dtype=torch.uint8)
                mod.quant_weight.copy_(raw.t())
                K = mod.in_features
                maxq = 2 ** mod.bits - 1  # (8-bit ColBlock, `gptq.int8`: a byte is one level, the grid is 16 x finer, its centre 127.5)
                s = (7.2 / maxq) * (gain / 2.2) * K**-0.5 * (1.0 + 0.1 * torch.rand((N, 1), generator=gen, device=dev))
                mod.scales.copy_(s.to(mod.scales.dtype))
                mod.zeros.fill_(zero if mod.bits == 4 else 16.0 * zero + 7.5)
            elif name.endswith(("rms_1", "rms_2", "ln_f")):
                mod.scale.copy_((1.0 + 0.1 * _randn(tuple(mod.scale.shape), gen, dev)).to(mod.scale.dtype))


def fill_tp_shard_random_int4(model, seed: int, rank: int, zero: float = 7.5, gain: float = 1.0) -> None:
    """Bench-only: rank-local shard of a synthetic int4 model (tp.build_local_model) filled in place.  Replicated
    tensors (embedding, norm scales, scale / zero of the row-parallel linears) come from a generator seeded the same
    on every rank; sharded tensors from `seed + 1 + rank`."""
    from .quantization import ColBlockQuantizedLinear

    dev = model.transformer.wte.weight.device
    shared = torch.Generator(device=dev)
    shared.manual_seed(seed)
    local = torch.Generator(device=dev)
    local.manual_seed(seed + 1 + rank)
    with torch.no_grad():
        model.transformer.wte.weight.copy_(_randn(tuple(model.transformer.wte.weight.shape), shared, dev))
        for name, mod in model.named_modules():
            if name.endswith(("rms_1", "rms_2", "ln_f")):
                mod.scale.copy_((1.0 + 0.1 * _randn(tuple(mod.scale.shape), shared, dev)).to(mod.scale.dtype))
        for name, mod in model.named_modules():
            if isinstance(mod, ColBlockQuantizedLinear):
                N, Kb = mod.quant_weight.shape
                raw = torch.randint(0, 256, (Kb, N), generator=local, device=dev, dtype=torch.uint8)
                mod.quant_weight.copy_(raw.t())
                row_parallel = name.endswith("c_proj")  # attn.c_proj / mlp.c_proj: per-row scale / zero replicated
                K_full = mod.in_features * (getattr(model.config, "tp_world", 1) if row_parallel else 1)
                g = shared if row_parallel else local
                sc = (7.2 / 15.0) * (gain / 2.2) * K_full**-0.5 * (1.0 + 0.1 * torch.rand((N, 1), generator=g, device=dev))
                mod.scales.copy_(sc.to(mod.scales.dtype))
                mod.zeros.fill_(zero)  # (zero-mean weights: see fill_model_random_int4)

"""generate/adapter.py:67-95 through the native kernels: lit_llama_amd.adapter.LLaMA — op by op (native linears, RMSNorm,
RoPE + KV cache + causal attention, the ten-row gated prefix attention as mi355_adapter_prefix) in f32, and through the
whole-forward engine (prefix term right after the causal attention, decode under a hipGraph) in bf16 — against the
reference's own run (tests/golden/adapter.npz)."""
import numpy as np
import pytest
import torch

import lit_llama_amd
from lit_llama_amd import adapter as A
from lit_llama_amd.utils import EmptyInitOnDevice
from oracle import oracle
from test_adapter import CFG, adapter_state_dict, adapter_v2_state_dict

pytestmark = pytest.mark.gpu


def _teacher_forced(model, toks, T, S, dev):
    model.reset_cache()
    rows, pos = [], torch.arange(0, T, device=dev)
    for _ in range(toks.numel() - T):
        rows.append(model(toks.index_select(0, pos).view(1, -1), S, pos)[0, -1].float().cpu())
        pos = pos[-1:] + 1
    model.reset_cache()
    return torch.stack(rows)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-3), (torch.bfloat16, 0.05)])
def test_adapter_model_follows_the_reference(dev, golden, dtype, tol):
    g = golden("adapter")
    T, S = int(g["prompt_len"]), int(g["max_seq_length"])
    toks = torch.from_numpy(g["tokens"]).to(dev)
    ref_logits = torch.from_numpy(g["logits"])
    std = float(ref_logits.std(-1).mean())
    with EmptyInitOnDevice(device=dev, dtype=dtype):
        model = A.LLaMA(A.LLaMAConfig(**CFG))
    model.load_state_dict(adapter_state_dict())
    model.eval()
    on_engine = dtype == torch.bfloat16
    eng = model.engine()
    assert (eng is not None) == on_engine, model._engine_failed  # f32 models run op by op through the exact kernels
    if on_engine:
        assert eng.fused is None and eng.layers[2].adapter_len == 10 and eng.layers[0].adapter_len == 0
    margins = g["margin"]
    n = T + 1 + next((i for i, m in enumerate(margins.tolist()) if m <= 2 * tol * std), len(margins))
    for use_engine in ((True, False) if on_engine else (False,)):
        model.use_engine = use_engine
        got = _teacher_forced(model, toks, T, S, dev)
        err = (got - ref_logits).abs().max().item()
        assert err <= tol * std, f"{dtype} engine={use_engine}: adapter logits off by {err:.5f} (std {std:.3f})"
        assert model.adapter_kv_caches == [] and len(model.kv_caches) == 0
        out = lit_llama_amd.generate(model, toks[:T], toks.numel() - T, top_k=1).cpu()
        assert torch.equal(out[:n].long(), toks[:n].cpu().long()), f"engine={use_engine}: {out.tolist()} vs {toks.tolist()}"
        if not use_engine:  # op by op: prefix k / v computed once and kept (adapter.py:136-141)
            assert model.adapter_kv_caches[2] is not None and model.adapter_kv_caches[0] is None


@pytest.mark.parametrize("qdtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("hs,nh,B,T", [(128, 32, 1, 1), (128, 4, 2, 5), (16, 4, 1, 7), (256, 2, 1, 3)])
def test_adapter_prefix_kernel_matches_the_reference_arithmetic(dev, qdtype, hs, nh, B, T):
    """mi355_adapter_prefix against adapter.py:134-151 spelled in torch: softmax(rope(q) ak^T / sqrt(hs)) av, gated, added."""
    import math

    from lit_llama_amd import ops
    from lit_llama_amd.model import build_rope_cache

    gen = torch.Generator().manual_seed(hs + T)
    C, aT = nh * hs, 10
    qkv = torch.randn((B, T, 3 * C), generator=gen).to(qdtype)
    ak, av = torch.randn((nh, aT, hs), generator=gen), torch.randn((nh, aT, hs), generator=gen)
    gate = torch.randn((nh,), generator=gen)
    y0 = torch.randn((B, T, C), generator=gen).to(qdtype)
    pos = torch.arange(3, 3 + T)
    table = build_rope_cache(64, hs, torch.int64, torch.device("cpu")).float()
    rope = table.index_select(0, pos)
    q = qkv[..., :C].float().reshape(B, T, nh, hs)
    qs = q.reshape(B, T, nh, hs // 2, 2)
    rc = rope.view(1, T, 1, hs // 2, 2)
    qr = torch.stack([qs[..., 0] * rc[..., 0] - qs[..., 1] * rc[..., 1],
                      qs[..., 1] * rc[..., 0] + qs[..., 0] * rc[..., 1]], -1).flatten(3)
    att = torch.einsum("bthd,hsd->bhts", qr, ak) / math.sqrt(hs)
    ay = torch.einsum("bhts,hsd->bthd", torch.softmax(att, -1), av)
    ref = y0.float() + (gate.view(1, 1, nh, 1) * ay).reshape(B, T, C)
    for gathered in (True, False):
        y = y0.clone().to(dev)
        ops.adapter_prefix(qkv.to(dev), (rope if gathered else table).contiguous().to(dev), nh, ak.to(dev), av.to(dev),
                           gate.to(dev), y, pos=None if gathered else pos.to(dev), rope_gathered=gathered)
        err = (y.float().cpu() - ref).abs().max().item()
        assert err <= (1e-5 if qdtype == torch.float32 else 2e-2) * ref.abs().max().item(), (gathered, err)


@pytest.mark.parametrize("case", ["decode", "decode_split4", "no_cache_T5", "prefill_T40"])
def test_attention_with_the_prefix_term_folded_in_equals_attention_plus_the_prefix_kernel(dev, case):
    """mi355_attn_args.adapter_*: the decode kernel adds the prefix term itself — every flash-decoding split adds
    l_split * P to its un-normalised record, so the combined result is y + P — and the many-token path runs the prefix
    kernel behind the flash kernel.  Checked against attention() followed by the stand-alone adapter_prefix()."""
    from lit_llama_amd import ops
    from lit_llama_amd.model import build_rope_cache

    nh, hs, aT, S = 8, 128, 10, 64
    C = nh * hs
    gen = torch.Generator().manual_seed(len(case))
    T = {"decode": 1, "decode_split4": 1, "no_cache_T5": 5, "prefill_T40": 40}[case]
    ns = 4 if case == "decode_split4" else 1
    dt = torch.float32 if case == "no_cache_T5" else torch.bfloat16
    qkv = torch.randn((1, T, 3 * C), generator=gen).to(dt).to(dev)
    ak, av = torch.randn((nh, aT, hs), generator=gen).to(dev), torch.randn((nh, aT, hs), generator=gen).to(dev)
    gate = torch.randn((nh,), generator=gen).to(dev)
    table = build_rope_cache(S, hs, torch.int64, dev).float().contiguous()
    if case == "no_cache_T5":
        kw = dict(rope_gathered=True)
        rope = table[:T].contiguous()
        y0 = ops.attention(qkv, rope, nh, **kw)
        y1 = ops.attention(qkv, rope, nh, adapter=(ak, av, gate), **kw)
        ref = ops.adapter_prefix(qkv, rope, nh, ak, av, gate, y0.clone(), rope_gathered=True)
    else:
        start = 20 if T == 1 else 0
        pos = torch.arange(start, start + T, device=dev)
        k0 = torch.randn((1, nh, S, hs), generator=gen).to(torch.bfloat16).to(dev)
        v0 = torch.randn((1, nh, S, hs), generator=gen).to(torch.bfloat16).to(dev)
        kw = dict(pos=pos, n_split=ns)
        y0 = ops.attention(qkv, table, nh, kv_cache=(k0.clone(), v0.clone()), **kw)
        y1 = ops.attention(qkv, table, nh, kv_cache=(k0.clone(), v0.clone()), adapter=(ak, av, gate), **kw)
        ref = ops.adapter_prefix(qkv, table, nh, ak, av, gate, y0.clone(), pos=pos, rope_gathered=False)
    assert (ref.float() - y0.float()).abs().max().item() > 0.05          # the prefix term is not a rounding error
    tol = 1e-5 if dt == torch.float32 else 2e-2                          # bf16 y: one rounding (folded) vs two
    assert (y1.float() - ref.float()).abs().max().item() <= tol * ref.float().abs().max().item()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-3), (torch.bfloat16, 0.05)])
def test_adapter_v2_model_follows_the_reference(dev, golden, dtype, tol):
    """generate/adapter_v2.py:63-78: the adapter model with a learned scale / bias on every linear."""
    from lit_llama_amd import adapter_v2 as V2

    g = golden("adapter_v2")
    T, S = int(g["prompt_len"]), int(g["max_seq_length"])
    toks = torch.from_numpy(g["tokens"]).to(dev)
    ref_logits = torch.from_numpy(g["logits"])
    std = float(ref_logits.std(-1).mean())
    with EmptyInitOnDevice(device=dev, dtype=dtype):
        model = A.LLaMA(A.LLaMAConfig(**CFG))
        V2.add_adapter_v2_parameters_to_linear_layers(model)
    model.load_state_dict(adapter_v2_state_dict())
    model.eval()
    got = _teacher_forced(model, toks, T, S, dev)
    err = (got - ref_logits).abs().max().item()
    assert err <= tol * std, f"{dtype}: adapter v2 logits off by {err:.5f} (std {std:.3f})"
    out = lit_llama_amd.generate(model, toks[:T], toks.numel() - T, top_k=1).cpu()
    margins = g["margin"]
    n = T + 1 + next((i for i, m in enumerate(margins.tolist()) if m <= 2 * tol * std), len(margins))
    assert torch.equal(out[:n].long(), toks[:n].cpu().long()), f"{out.tolist()} vs {toks.tolist()}"


def test_adapter_v2_on_llm_int8_linears_stays_off_the_engine_and_keeps_its_epilogue(dev):
    """generate/adapter_v2.py accepts --quantize llm.int8: Linear8bitLt subclasses nn.Linear, so every quantised linear carries
    adapter_scale / adapter_bias too.  The engine's streams have no such epilogue: the model must decode op by op (advisor r3:
    engine._kind used to answer "i8" before it looked at adapter_scale, and the pair was silently dropped)."""
    from lit_llama_amd import adapter_v2 as V2
    from lit_llama_amd.engine import EngineUnavailable, _kind

    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode="llm.int8"):
        model = A.LLaMA(A.LLaMAConfig(**CFG))
        V2.add_adapter_v2_parameters_to_linear_layers(model)
    sd = {k: v for k, v in adapter_v2_state_dict().items()}
    model.load_state_dict(sd)
    model.eval()
    lin = model.transformer.h[0].attn.c_attn
    assert type(lin).__name__ == "Linear8bitLt" and lin.adapter_scale is not None
    with pytest.raises(EngineUnavailable):
        _kind(lin)
    assert model.engine() is None and "Adapter v2" in model._engine_failed
    toks = torch.arange(3, 3 + 6, device=dev, dtype=torch.int32)
    S = 12
    got = _teacher_forced(model, torch.cat([toks, toks]), 6, S, dev)
    # the same model with the scale / bias pairs at their identity values must differ: the epilogue is applied
    for mod in model.modules():
        if getattr(mod, "adapter_scale", None) is not None:
            mod.adapter_scale.data.fill_(1.0)
            mod.adapter_bias.data.zero_()
    plain = _teacher_forced(model, torch.cat([toks, toks]), 6, S, dev)
    assert torch.isfinite(got).all() and (got - plain).abs().max().item() > 1e-3 * float(plain.std())


def test_adapter_parameter_edit_in_place_rebuilds_the_engine_on_the_token_path(dev):
    """The engine snapshots the prefix k / v at build time; `forward` (check=False on the per-token path) must still notice an
    in-place edit of adapter_wte / gating_factor (advisor r3)."""
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16):
        model = A.LLaMA(A.LLaMAConfig(**CFG))
    model.load_state_dict(adapter_state_dict())
    model.eval()
    toks = torch.arange(3, 3 + 6, device=dev, dtype=torch.int32)
    pos = torch.arange(0, 6, device=dev)
    a = model(toks.view(1, -1), 16, pos)[0, -1].float().cpu()
    assert model._engine is not None
    eng0 = model._engine
    for blk in model.transformer.h:
        if hasattr(blk.attn, "gating_factor"):
            with torch.no_grad():
                blk.attn.gating_factor.mul_(-3.0)  # in place: same storage, new version
    model.reset_cache()
    b = model(toks.view(1, -1), 16, pos)[0, -1].float().cpu()
    assert model._engine is not eng0, "stale engine kept after an in-place edit of the gates"
    model.use_engine = False
    model.reset_cache()
    c = model(toks.view(1, -1), 16, pos)[0, -1].float().cpu()
    std = float(c.std())
    assert (b - c).abs().max().item() <= 0.05 * std and (a - b).abs().max().item() > 0.01 * std

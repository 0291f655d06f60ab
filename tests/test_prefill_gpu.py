"""Wide path (prompt prefill / no-cache evaluation): the LDS-tiled MFMA int4 GEMM (csrc/gemm.hip) + the flash-style
causal attention (csrc/flash_prefill.hip) against the CPU oracle at the 7B width.

Reference: /root/reference lit_llama/model.py:76-122 called with T > 1 (generate.py's prompt pass, and
evaluate/full.py:120-129 with no cache); lit_llama/quantization.py:413-423 for the linears.
Bar: bf16 operands vs the oracle's f32 arithmetic — logits within 0.05 logit-std at every position, argmax equal
wherever the oracle's top-2 margin exceeds twice that.
"""
import numpy as np
import pytest
import torch

import lit_llama_amd
from lit_llama_amd import synth
from lit_llama_amd.model import LLaMA, LLaMAConfig
from lit_llama_amd.utils import EmptyInitOnDevice
from oracle import oracle

pytestmark = pytest.mark.gpu

W7B = dict(n_layer=1, n_head=32, n_embd=4096)


def build(dev, seed=0):
    cfg = LLaMAConfig(**W7B)
    sd = synth.make_state_dict(cfg, seed=seed, mode="gptq.int4")
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode="gptq.int4"):
        model = LLaMA(cfg)
    model.load_state_dict(sd)
    model.eval()
    om = oracle.Model(oracle.Config(**W7B), {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()},
                      mode="gptq.int4")
    return model, om, cfg


def check(got, ref, what):
    std = float(ref.std(-1).mean())
    err = (got - ref).abs().max().item()
    assert err <= 0.05 * std, f"{what}: logits off by {err:.4f} (std {std:.3f})"
    top2 = torch.topk(ref, 2, dim=-1).values
    decisive = (top2[:, 0] - top2[:, 1]) > 0.1 * std
    assert torch.equal(got.argmax(-1)[decisive], ref.argmax(-1)[decisive]), what
    return err / std


@torch.no_grad()
def test_chunked_prefill_through_the_engine_matches_oracle(dev, monkeypatch):
    """600 prompt tokens = one chunk of 512 + one of 88 starting at position 512 (flash attention over the cache rows
    of the first chunk), then one decode step on the fused path on top of that cache."""
    monkeypatch.setenv("MI355_PREFILL_T", "512")
    model, om, cfg = build(dev)
    eng = model.engine()
    assert eng is not None and eng.max_T == 512, model._engine_failed
    T, S = 600, 640
    prompt = synth.make_prompt(T + 1)
    pos = torch.arange(T, device=dev)
    pos._mi355_pos0 = 0
    got = model(prompt[:T].view(1, -1).to(dev), S, pos)[0].float().cpu()
    ref = om(prompt[:T].view(1, -1), S, torch.arange(T))[0].float()
    e1 = check(got, ref, "prefill")
    p1 = torch.tensor([T], device=dev)
    p1._mi355_pos0 = T
    got1 = model(prompt[T:T + 1].view(1, -1).to(dev), S, p1)[0].float().cpu()
    ref1 = om(prompt[T:T + 1].view(1, -1), S, torch.tensor([T]))[0].float()
    e2 = check(got1, ref1, "decode step after the chunked prefill")
    eng.check_status()
    print(f"prefill {T} tokens: max |dlogit| {e1:.4f} std; next decode step {e2:.4f} std")


@torch.no_grad()
def test_no_cache_forward_module_path_matches_oracle(dev):
    """evaluate/full.py:120-129: model(x) without positions or cache, T = 200 — the module path: wide GEMM for every
    ColBlockQuantizedLinear and the flash kernel over the call's own K / V."""
    model, om, cfg = build(dev, seed=2)
    T = 200
    toks = synth.make_prompt(T, seed=77)
    got = model(toks.view(1, -1).long().to(dev))[0].float().cpu()
    ref = om(toks.view(1, -1).long())[0].float()
    e = check(got, ref, "no-cache forward")
    print(f"no-cache forward {T} tokens: max |dlogit| {e:.4f} std")


@torch.no_grad()
def test_generate_with_a_long_prompt_agrees_between_wide_and_skinny_prefill(dev, monkeypatch):
    """The same greedy run with the prompt fed through the wide path and through chunks of the skinny kernel."""
    model, om, cfg = build(dev, seed=3)
    prompt = synth.make_prompt(150, seed=9).to(dev)
    a = lit_llama_amd.generate(model, prompt, 6, top_k=1).cpu()
    la = model(prompt.view(1, -1), 160, _pos(150, dev))[0, -1].float().cpu()
    monkeypatch.setenv("MI355_PREFILL_GEMM", "0")
    model._drop_engine()
    eng = model.engine()
    assert eng.max_T <= 16
    b = lit_llama_amd.generate(model, prompt, 6, top_k=1).cpu()
    lb = model(prompt.view(1, -1), 160, _pos(150, dev))[0, -1].float().cpu()
    std = float(lb.std())
    assert (la - lb).abs().max().item() <= 0.03 * std
    top2 = torch.topk(lb, 2).values
    if float(top2[0] - top2[1]) > 0.06 * std:
        assert int(a[150]) == int(b[150])


def _pos(T, dev):
    p = torch.arange(T, device=dev)
    p._mi355_pos0 = 0
    return p


@torch.no_grad()
def test_evaluate_style_nll_at_block_size_matches_oracle(dev):
    """The reference's own quality check for a quantised model, at its own length (evaluate/full.py:120-129): a NO-CACHE
    forward over block_size = 2048 tokens, then the mean next-token negative log-likelihood.  One 7B-width layer: the
    2048-row GEMMs and the flash kernel over 2048 x 2048 (the shapes bench.py's prefill line times) against the oracle —
    logits at every position, and the NLL itself."""
    model, om, cfg = build(dev, seed=4)
    T = cfg.block_size
    assert T == 2048
    toks = synth.make_prompt(T, seed=31)
    torch.set_num_threads(max(torch.get_num_threads(), 16))
    got = model(toks.view(1, -1).long().to(dev))[0].float().cpu()
    ref = om(toks.view(1, -1).long())[0].float()
    e = check(got, ref, "no-cache forward at T = 2048")
    tgt = toks[1:].long()
    nll_got = torch.nn.functional.cross_entropy(got[:-1], tgt).item()
    nll_ref = torch.nn.functional.cross_entropy(ref[:-1], tgt).item()
    assert abs(nll_got - nll_ref) <= 2e-3 * max(1.0, abs(nll_ref)), (nll_got, nll_ref)
    # per-token NLL: no position is off by more than the logit bar allows (a wrong row would hide in the mean)
    d = (torch.nn.functional.cross_entropy(got[:-1], tgt, reduction="none") -
         torch.nn.functional.cross_entropy(ref[:-1], tgt, reduction="none")).abs().max().item()
    assert d <= 0.1 * float(ref.std(-1).mean()), d
    print(f"T = 2048 no-cache: max |dlogit| {e:.4f} std; NLL {nll_got:.5f} vs oracle {nll_ref:.5f}; max per-token |dNLL| {d:.4f}")


@torch.no_grad()
def test_bf16_model_prompt_takes_the_wide_path_and_matches_oracle(dev):
    """BASELINE configs[1] (no quantisation): a 150-token prompt of a 7B-width bf16 layer through the engine — the MFMA GEMM
    over the BF16 stream + flash attention (round 2 fed such prompts through <= 16-row chunks of the streaming kernel and
    the module path through rocBLAS) — and a no-cache module forward, both against the oracle."""
    cfg = LLaMAConfig(**W7B)
    sd = synth.make_state_dict(cfg, seed=6, mode=None, dtype=torch.bfloat16)
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16):
        model = LLaMA(cfg)
    model.load_state_dict(sd)
    model.eval()
    om = oracle.Model(oracle.Config(**W7B), {k: v.float() for k, v in sd.items()})
    eng = model.engine()
    assert eng is not None and eng.max_T >= 512 and eng.gemm_ws is not None, model._engine_failed
    T, S = 150, 160
    toks = synth.make_prompt(T, seed=12)
    got = model(toks.view(1, -1).to(dev), S, _pos(T, dev))[0].float().cpu()
    ref = om(toks.view(1, -1), S, torch.arange(T))[0].float()
    e1 = check(got, ref, "bf16 prefill through the engine")
    model.reset_cache()
    got2 = model(toks.view(1, -1).long().to(dev))[0].float().cpu()  # module path: _linear -> ops.linear_gemm(fmt = BF16)
    om.reset_cache()
    ref2 = om(toks.view(1, -1).long())[0].float()
    e2 = check(got2, ref2, "bf16 no-cache forward")
    print(f"bf16 7B-width layer, {T} tokens: engine prefill {e1:.4f} std, module path {e2:.4f} std")


@torch.no_grad()
def test_llm_int8_prompt_takes_the_int8_gemm_and_matches_oracle(dev):
    """BASELINE configs[3]: a 100-token prompt of a 7B-width llm.int8 layer through the engine — mi355_linear_int8_gemm (outlier
    columns over the whole prompt, as MatMul8bitLt determines them; the streaming kernel's <= 16-row chunks each had their own
    set) + flash attention — and the module path (Linear8bitLt.forward on 100 rows), against the oracle's restatement
    (PARITY UNPINNED: bitsandbytes is not vendored).  The band is the one of the int8 decode test: int8 rounding of the
    activations, not kernel error, sets it."""
    cfg = LLaMAConfig(**W7B)
    sd = synth.make_state_dict(cfg, seed=8, mode="llm.int8", dtype=torch.bfloat16, outlier_channels=8)
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode="llm.int8"):
        model = LLaMA(cfg)
    model.load_state_dict(sd)
    model.eval()
    om = oracle.Model(oracle.Config(**W7B), {k: v.float() for k, v in sd.items()}, mode="llm.int8")
    eng = model.engine()
    assert eng is not None and eng.gemm_ws is not None and eng.max_T >= 512, model._engine_failed
    T, S = 100, 112
    toks = synth.make_prompt(T, seed=14)
    got = model(toks.view(1, -1).to(dev), S, _pos(T, dev))[0].float().cpu()
    ref = om(toks.view(1, -1), S, torch.arange(T))[0].float()
    std = float(ref.std(-1).mean())
    e1 = (got - ref).abs().max().item() / std
    assert e1 <= 0.15, f"int8 prefill through the engine: {e1:.4f} std"
    model.reset_cache()
    om.reset_cache()
    got2 = model(toks.view(1, -1).long().to(dev))[0].float().cpu()
    ref2 = om(toks.view(1, -1).long())[0].float()
    e2 = (got2 - ref2).abs().max().item() / std
    assert e2 <= 0.15, f"int8 no-cache forward: {e2:.4f} std"
    print(f"llm.int8 7B-width layer, {T} tokens: engine prefill {e1:.4f} std, module path {e2:.4f} std")


@torch.no_grad()
@pytest.mark.parametrize("T", [700, 300, 100])
@pytest.mark.parametrize("mode", ["gptq.int4", None])
def test_fused_prompt_chain_matches_oracle_and_the_staged_chain(dev, monkeypatch, mode, T):
    """Round 4: a prompt chunk wide enough that no GEMM launch is split over K (7B width: more than 640 tokens) runs the layer
    as a producer / consumer chain (csrc/gemm_fuse.h): no staging pass in front of a linear (the residual epilogues emit the
    next operand and its partial sums, the SwiGLU and attention outputs are operands as they are), the c_attn epilogue rotates
    k and writes the K / V cache rows.  Two 7B-width layers (the mlp.c_proj -> next layer's c_attn hand-over included), 700
    prompt tokens through the engine: against the oracle (/root/reference lit_llama/model.py:76-122 with T > 1), against the
    staged chain of rounds 2-3 (MI355_GEMM_FUSE=0) on the same weights, and one decode step on top of each cache.
    T = 700: no launch is split over K, the GEMM epilogues are the producers; T = 100: every launch is split and the reduction of
    the K-slices (splitk_fused_reduce_kernel) produces, the consumers' slices add the shares of their own units; T = 300: mixed
    (c_attn and the pair whole, the N = 4096 linears split)."""
    cfg_kw = dict(n_layer=2, n_head=32, n_embd=4096)
    cfg = LLaMAConfig(**cfg_kw)
    sd = synth.make_state_dict(cfg, seed=11, mode=mode, **(dict(dtype=torch.bfloat16) if mode is None else {}))
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode=mode):
        model = LLaMA(cfg)
    model.load_state_dict(sd)
    model.eval()
    om = oracle.Model(oracle.Config(**cfg_kw), {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}, mode=mode)
    eng = model.engine()
    assert eng is not None and eng.max_T >= 700 and eng.gemm_ws is not None, model._engine_failed
    S = T + 12
    prompt = synth.make_prompt(T + 1, seed=21)
    torch.set_num_threads(max(torch.get_num_threads(), 16))
    ref = om(prompt[:T].view(1, -1), S, torch.arange(T))[0].float()
    ref1 = om(prompt[T:T + 1].view(1, -1), S, torch.tensor([T]))[0].float()
    out = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("MI355_GEMM_FUSE", fuse)
        model.reset_cache()
        got = model(prompt[:T].view(1, -1).to(dev), S, _pos(T, dev))[0].float().cpu()
        p1 = torch.tensor([T], device=dev)
        p1._mi355_pos0 = T
        got1 = model(prompt[T:T + 1].view(1, -1).to(dev), S, p1)[0].float().cpu()
        eng.check_status()
        e = check(got, ref, f"prefill, MI355_GEMM_FUSE={fuse}")
        e1 = check(got1, ref1, f"decode step on that cache, MI355_GEMM_FUSE={fuse}")
        out[fuse] = (got, got1, e, e1)
    std = float(ref.std(-1).mean())
    d = (out["1"][0] - out["0"][0]).abs().max().item() / std
    d1 = (out["1"][1] - out["0"][1]).abs().max().item() / std
    # the two chains round the same operands; they differ in the ORDER of the per-row sums (partial sums per block) and in
    # where k is rotated: f32 noise in front of a bf16 rounding
    assert d <= 0.02 and d1 <= 0.02, (d, d1)
    print(f"{mode} T={T}: fused chain {out['1'][2]:.4f} / {out['1'][3]:.4f} std vs oracle (staged {out['0'][2]:.4f} / {out['0'][3]:.4f}); "
          f"fused vs staged {d:.4f} / {d1:.4f} std")

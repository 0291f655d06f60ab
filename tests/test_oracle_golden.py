"""The oracle (CPU restatement) against the golden vectors produced by the real reference
(oracle/gen_golden.py).  This is the pin that makes the oracle trustworthy as a checker."""
import numpy as np
import pytest
import torch

from lit_llama_amd import synth
from lit_llama_amd.model import LLaMAConfig
from oracle import oracle

CFG1 = dict(n_layer=2, n_head=4, n_embd=256)
TINY = dict(block_size=128, vocab_size=16, n_layer=1, n_head=4, n_embd=8)


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_colblock_pack_dequant_forward_match_reference(golden):
    g = golden("colblock")
    for tag in ("b4_row", "b4_g64", "b8_row"):
        N, K, bits, tc = (int(v) for v in g[f"{tag}_meta"])
        w, scales, zeros = _t(g[f"{tag}_w"]), _t(g[f"{tag}_scales"]), _t(g[f"{tag}_zeros"])
        q = oracle.colblock_pack(w, scales, zeros, bits, tc)
        assert torch.equal(q, _t(g[f"{tag}_q"]))
        assert torch.equal(oracle.colblock_get_weight(q, scales, zeros, bits, tc), _t(g[f"{tag}_wdq"]))
        y = oracle.colblock_linear(_t(g[f"{tag}_x"]), q, scales, zeros, bits, tc)
        assert torch.equal(y, _t(g[f"{tag}_y"]))


def test_rmsnorm_rope_match_reference(golden):
    g = golden("blocks")
    assert torch.equal(oracle.rmsnorm(_t(g["rms_x"]), _t(g["rms_scale"]), 1e-6), _t(g["rms_y"]))
    rc = oracle.build_rope_cache(6, 4, dtype=torch.float32)
    assert torch.equal(rc, _t(g["rope_cache"]))
    assert torch.equal(oracle.apply_rope(_t(g["rope_x"]), rc), _t(g["rope_y"]))
    big = oracle.build_rope_cache(2048, 128, dtype=torch.int64)
    assert big.dtype == torch.float32 and big.shape == (2048, 64, 2)
    assert torch.equal(big[[0, 1, 17, 511, 2047]], _t(g["rope_big_rows"]))


def test_block_forward_and_kv_cache_match_reference(golden):
    g = golden("blocks")
    sd = {k.split("::", 1)[1]: _t(v) for k, v in g.items() if k.startswith("blk_sd::")}
    om = oracle.Model(oracle.Config(block_size=64, vocab_size=100, n_layer=2, n_head=4, n_embd=32), sd)
    idx = _t(g["blk_idx"])
    with torch.no_grad():
        assert torch.allclose(om(idx), _t(g["blk_logits"]), atol=1e-6, rtol=0)
        out = om(idx[:1], 12, torch.arange(9))
    assert torch.allclose(out, _t(g["blk_logits_pos"]), atol=1e-6, rtol=0)
    assert torch.allclose(om.kv_caches[1][0], _t(g["blk_kcache"]), atol=1e-6, rtol=0)
    assert torch.allclose(om.kv_caches[1][1], _t(g["blk_vcache"]), atol=1e-6, rtol=0)


def _run_case(golden, name, cfg_kwargs, mode):
    g = golden(name)
    T = int(g["prompt_len"])
    toks = _t(g["tokens"])
    sd = synth.make_state_dict(LLaMAConfig(**cfg_kwargs), seed=int(g["seed"]), mode=mode)
    om = oracle.Model(oracle.Config(**cfg_kwargs), sd, mode=mode)
    S = int(g["max_seq_length"])
    out = oracle.generate(om, toks[:T], toks.numel() - T, top_k=1, max_seq_length=S)
    assert torch.equal(out, toks), f"{name}: greedy tokens differ from the reference"
    return g, om, toks, T, S


def test_cfg1_fp32_greedy_tokens_and_logits(golden):
    g, om, toks, T, S = _run_case(golden, "cfg1_fp32", CFG1, None)
    logits = oracle.teacher_forced_logits(om, toks, T, S)
    probes = (np.arange(64) * (32000 // 64) + 7) % 32000
    assert np.allclose(logits[:, probes].numpy(), g["probes"], atol=2e-6, rtol=0)
    assert np.array_equal(logits.argmax(-1).numpy().astype(np.int32), g["argmax"])


def test_cfg1_int4_greedy_tokens_and_logits(golden):
    g, om, toks, T, S = _run_case(golden, "cfg1_int4", CFG1, "gptq.int4")
    logits = oracle.teacher_forced_logits(om, toks, T, S)
    probes = (np.arange(64) * (32000 // 64) + 7) % 32000
    assert np.allclose(logits[:, probes].numpy(), g["probes"], atol=2e-6, rtol=0)


def test_cfg1_int8g_greedy_tokens_and_logits(golden):
    """`--quantize gptq.int8` (ColBlockQuantizedLinear, bits = 8, lit_llama/utils.py:100-102) through the reference itself on the CPU."""
    g, om, toks, T, S = _run_case(golden, "cfg1_int8g", CFG1, "gptq.int8")
    logits = oracle.teacher_forced_logits(om, toks, T, S)
    probes = (np.arange(64) * (32000 // 64) + 7) % 32000
    assert np.allclose(logits[:, probes].numpy(), g["probes"], atol=2e-6, rtol=0)
    assert np.array_equal(logits.argmax(-1).numpy().astype(np.int32), g["argmax"])


def test_tiny_model_cache_roll_regime(golden):
    _run_case(golden, "tiny_roll", TINY, None)
    _run_case(golden, "tiny_noroll", TINY, None)


def test_llm_int8_restatement_is_close_to_fp_and_handles_outliers():
    """No reference vector exists for LLM.int8 (parity unpinned): sanity-bound the restatement against the fp
    product it approximates, with and without outlier columns."""
    gen = torch.Generator().manual_seed(3)
    N, K = 96, 512
    w = torch.randn((N, K), generator=gen) * K**-0.5
    cb, scb = oracle.int8_quant_rows(w)
    assert cb.dtype == torch.int8 and int(cb.abs().max()) == 127
    x = torch.randn((4, K), generator=gen)
    ref = x @ w.t()
    y = oracle.llm_int8_linear(x, cb, scb)
    assert (y - ref).abs().max() / ref.abs().max() < 3e-2
    x[:, 5] = 9.0
    x[1, 77] = -7.5
    y2 = oracle.llm_int8_linear(x, cb, scb)
    assert (y2 - x @ w.t()).abs().max() / (x @ w.t()).abs().max() < 3e-2
    # the outlier columns must not saturate the int8 scale: error stays at the no-outlier level
    y3 = oracle.llm_int8_linear(x, cb, scb, threshold=0.0)
    assert (y3 - x @ w.t()).abs().max() > (y2 - x @ w.t()).abs().max()


@pytest.mark.parametrize("name", ["gptq_actorder", "gptq_plain"])
def test_gptq_restatement_matches_reference(golden, name):
    """oracle/gptq.py against the reference GPTQQuantizer's own results (quantization.py:426-616), bit for bit:
    Hessian accumulation, row parameters, the quantised weights handed to pack_weight, the error and the bytes."""
    from oracle import gptq as ogptq

    g = golden(name)
    W = torch.from_numpy(g["weight"])
    hs = ogptq.Hessian(W.shape[1])
    for b in torch.from_numpy(g["batches"]):
        hs.add(b)
    Q, sc, ze, err = ogptq.gptq_quantize(W, hs.H, bits=int(g["bits"]), groupsize=int(g["groupsize"]),
                                         actorder=bool(g["actorder"]))
    assert torch.equal(sc, torch.from_numpy(g["scales"])) and torch.equal(ze, torch.from_numpy(g["zeros"]))
    assert torch.equal(Q, torch.from_numpy(g["Q"]))
    assert err == float(g["error"])
    packed = oracle.colblock_pack(Q, sc, ze, 4, W.shape[1])
    assert torch.equal(packed.contiguous(), torch.from_numpy(g["quant_weight"]))


def test_operand_arithmetic_model_statements():
    """oracle/sim_operand_arith.py (the CPU model that priced the operand arithmetic of round 5) on one small linear: its `exact` statement is
    the oracle's dequantise-then-multiply; the offset statement equals it in exact arithmetic (no massive activation -> only f32 rounding);
    the fp8-limb rounding keeps 12 bits and saturates past 448 x the edge's pre-scale; the ladder variant hands a clipped row to fp16."""
    import importlib.util
    from pathlib import Path

    spec = importlib.util.spec_from_file_location("sim_operand_arith", Path(__file__).resolve().parents[1] / "oracle" / "sim_operand_arith.py")
    sim = importlib.util.module_from_spec(spec)
    import sys

    saved_path = list(sys.path)  # (the script puts oracle/ itself on sys.path; tests that spawn workers must not inherit that)
    try:
        spec.loader.exec_module(sim)
        _operand_arithmetic_statements(sim)
    finally:
        sys.path[:] = saved_path


def _operand_arithmetic_statements(sim):
    cfg = LLaMAConfig(n_layer=1, n_head=4, n_embd=256)
    sd = synth.make_state_dict(cfg, seed=3, mode="gptq.int4")
    prefix = "transformer.h.0.mlp.c_proj"
    K = sd[prefix + ".quant_weight"].shape[1] * 2
    x = torch.randn((1, 5, K), generator=torch.Generator().manual_seed(0))
    ref = oracle.linear(sd, prefix, x, "gptq.int4")
    exact = sim.make_linear("f32", None)(sd, prefix, x, "gptq.int4")
    assert torch.allclose(exact, ref, atol=1e-5, rtol=0)
    off = sim.make_linear("f32", 1024.0)(sd, prefix, x, "gptq.int4")
    assert (off - ref).abs().max() <= 2e-3 * ref.abs().max()     # f32 accumulation of (1024 + q) x: small while x has no massive value
    xm = x.clone()
    xm[0, 2, 7] = 3.0e4                                           # ... and not small with one
    refm = oracle.linear(sd, prefix, xm, "gptq.int4")
    offm = sim.make_linear("f32", 1024.0)(sd, prefix, xm, "gptq.int4")
    cen = sim.make_linear("f32", 8.0)(sd, prefix, xm, "gptq.int4")
    assert (offm - refm)[0, 2].abs().max() > 20 * (cen - refm)[0, 2].abs().max()
    # limbs: 12 bits inside the range, saturation outside, fp16 for the clipped row under the ladder
    sim.LADDER = False
    r = sim.f8_limb_round(x[0], prefix)                           # SwiGLU edge: pre-scale 2^4, exact to 12 bits from 2^-2 up
    big = x[0].abs() >= 0.25
    assert ((r - x[0]).abs()[big] <= x[0].abs()[big] * 2.0 ** -11).all()
    r = sim.f8_limb_round(xm[0], prefix)
    assert float(r[2, 7]) < 8000.0                                # 3e4 saturates at 477.75 x 16
    sim.LADDER = True
    r = sim.f8_limb_round(xm[0], prefix)
    sim.LADDER = False
    assert float(r[2, 7]) == float(torch.tensor(3.0e4).to(torch.float16)) and torch.equal(r[2], xm[0, 2].to(torch.float16).float())

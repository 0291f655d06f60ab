"""GPU parity tests of the whole path (module API, engine, hipGraph, generate) against the golden vectors the
reference produced and against the oracle.

Parity protocol for floating point (SURVEY.md §7 "token-for-token parity is ill-posed at near-ties"):
  * f32 models run through the exact-f32 generic kernels: greedy tokens must EQUAL the reference's; logits
    within 2e-4 * std (summation order only).
  * bf16 / int4 models (bf16 MFMA operands, bf16 KV cache, f32 residual stream): teacher-forced on the
    reference's tokens, per-step logits must stay within TOL = 0.05 * logit std (max abs error over the probe
    columns), and the argmax must equal the reference's wherever the reference's top-2 margin exceeds 2 * TOL;
    free-running greedy tokens must equal the reference's up to the first step whose margin is below that bound.
"""
import os

import numpy as np
import pytest
import torch

import lit_llama_amd
from lit_llama_amd import _native as nat
from lit_llama_amd import synth
from lit_llama_amd.model import LLaMA, LLaMAConfig
from lit_llama_amd.utils import EmptyInitOnDevice, quantization
from oracle import oracle

pytestmark = pytest.mark.gpu

CFG1 = dict(n_layer=2, n_head=4, n_embd=256)
TINY = dict(block_size=128, vocab_size=16, n_layer=1, n_head=4, n_embd=8)
PROBES = (np.arange(64) * (32000 // 64) + 7) % 32000


def _t(a):
    return torch.from_numpy(np.asarray(a))


def build(cfg_kwargs, mode, dtype, dev, seed=0, outliers=0):
    cfg = LLaMAConfig(**cfg_kwargs)
    sd = synth.make_state_dict(cfg, seed=seed, mode=mode, outlier_channels=outliers)
    with EmptyInitOnDevice(device=dev, dtype=dtype, quantization_mode=mode):
        model = LLaMA(cfg)
    model.load_state_dict(sd)
    model.eval()
    return model, sd, cfg


@torch.no_grad()
def teacher_forced(model, toks, T, S, dev):
    model.reset_cache()
    rows = []
    input_pos = torch.arange(0, T, device=dev)
    pos0 = 0
    for _ in range(toks.numel() - T):
        x = toks.index_select(0, input_pos).view(1, -1)
        input_pos._mi355_pos0 = pos0
        rows.append(model(x, S, input_pos)[0, -1].float().cpu())
        pos0 = pos0 + input_pos.numel()
        input_pos = input_pos[-1:] + 1
    model.reset_cache()
    return torch.stack(rows)


# ---------------------------------------------------------------------------------------------- f32 plumbing config
@pytest.mark.parametrize("name,mode", [("cfg1_fp32", None), ("cfg1_int4", "gptq.int4")])
def test_cfg1_f32_tokens_equal_reference(dev, golden, name, mode):
    """BASELINE.json configs[0]: LLaMAConfig(n_layer=2, n_head=4, n_embd=256), greedy, on the GPU in f32."""
    g = golden(name)
    model, _, cfg = build(CFG1, mode, torch.float32, dev)
    T, S = int(g["prompt_len"]), int(g["max_seq_length"])
    toks = _t(g["tokens"]).to(dev)
    out = lit_llama_amd.generate(model, toks[:T], toks.numel() - T, top_k=1)
    assert torch.equal(out.cpu(), _t(g["tokens"])), "greedy tokens differ from the reference CPU path"
    model.reset_cache()
    logits = teacher_forced(model, toks, T, S, dev)
    err = np.abs(logits[:, PROBES].numpy() - g["probes"]).max()
    assert err <= 2e-4 * float(g["std"].mean()), f"logit error {err:.3e}"
    assert np.array_equal(logits.argmax(-1).numpy().astype(np.int32), g["argmax"])
    # no-cache forward over the whole sequence (evaluate-style call)
    full = model(toks[:-1].view(1, -1).long())[0].float().cpu()
    assert np.array_equal(full.argmax(-1).numpy().astype(np.int32), g["nocache_argmax"])
    assert np.abs(full[:, PROBES].numpy() - g["nocache_probes"]).max() <= 2e-4 * float(g["std"].mean())


@pytest.mark.parametrize("name", ["tiny_roll", "tiny_noroll"])
def test_tiny_model_cache_roll_tokens_equal_reference(dev, golden, name):
    """tests/test_generate.py:26-54 of the reference: max_seq_length < T + max_new_tokens rolls the cache."""
    g = golden(name)
    model, _, _ = build(TINY, None, torch.float32, dev)
    T, S = int(g["prompt_len"]), int(g["max_seq_length"])
    toks = _t(g["tokens"])
    out = lit_llama_amd.generate(model, toks[:T].to(dev), toks.numel() - T, top_k=1, max_seq_length=S)
    assert torch.equal(out.cpu(), toks)


def test_module_level_forward_matches_reference_blocks(dev, golden):
    """Block / attention / cache semantics on the shapes of the reference's tests/test_model.py."""
    g = golden("blocks")
    sd = {k.split("::", 1)[1]: _t(v) for k, v in g.items() if k.startswith("blk_sd::")}
    cfg = LLaMAConfig(block_size=64, vocab_size=100, n_layer=2, n_head=4, n_embd=32)
    model = LLaMA(cfg).to(dev)
    model.load_state_dict(sd)
    idx = _t(g["blk_idx"]).to(dev)
    with torch.no_grad():
        logits = model(idx).cpu()
        assert (logits - _t(g["blk_logits"])).abs().max().item() <= 1e-4
        lp = model(idx[:1], 12, torch.arange(9, device=dev)).cpu()
        assert (lp - _t(g["blk_logits_pos"])).abs().max().item() <= 1e-4
        k, v = model.kv_caches[1]
        assert k.shape == (1, 4, 12, 8)
        assert (k.cpu() - _t(g["blk_kcache"])).abs().max().item() <= 1e-5
        assert (v.cpu() - _t(g["blk_vcache"])).abs().max().item() <= 1e-5
        # Block called directly with the reference's positional signature
        x = torch.randn(3, 9, 32, device=dev)
        rope = model.rope_cache[:9]
        y, _ = model.transformer.h[0](x, rope, None, 64)
        om = oracle.Model(oracle.Config(block_size=64, vocab_size=100, n_layer=2, n_head=4, n_embd=32), sd)
        mask = torch.tril(torch.ones(9, 9, dtype=torch.bool))[None, None]
        yo, _ = om.block(0, x.cpu(), oracle.build_rope_cache(64, 8)[:9], mask, 64)
        assert (y.cpu() - yo).abs().max().item() <= 1e-4


# ---------------------------------------------------------------------------------------------- bf16 engine path
def _margin_check(name, logits, g):
    std = float(g["std"].mean())
    tol = 0.05 * std
    err = np.abs(logits[:, PROBES].numpy() - g["probes"]).max()
    assert err <= tol, f"{name}: max |dlogit| {err:.4f} > {tol:.4f} (logit std {std:.3f})"
    am = logits.argmax(-1).numpy().astype(np.int32)
    decisive = g["margin"] > 2 * tol
    assert np.array_equal(am[decisive], g["argmax"][decisive]), f"{name}: argmax differs at a decisive step"
    return err, tol, int(decisive.sum())


def _free_running_check(out, g, tol):
    ref = g["tokens"]
    T = int(g["prompt_len"])
    out = out.cpu().numpy().astype(np.int32)
    assert np.array_equal(out[:T], ref[:T])
    for j in range(len(ref) - T):
        if out[T + j] != ref[T + j]:
            assert g["margin"][j] <= 2 * tol, f"token {j} differs at margin {g['margin'][j]:.4f} > {2 * tol:.4f}"
            return j
    return None


@pytest.mark.parametrize("name,mode", [("cfg1_fp32", None), ("cfg1_int4", "gptq.int4")])
def test_cfg1_bf16_engine_teacher_forced_parity(dev, golden, name, mode):
    g = golden(name)
    model, _, _ = build(CFG1, mode, torch.bfloat16, dev)
    assert model.engine() is not None, model._engine_failed
    T, S = int(g["prompt_len"]), int(g["max_seq_length"])
    toks = _t(g["tokens"]).to(dev)
    logits = teacher_forced(model, toks, T, S, dev)
    err, tol, n = _margin_check(name, logits, g)
    out = lit_llama_amd.generate(model, toks[:T], toks.numel() - T, top_k=1)
    first_div = _free_running_check(out, g, tol)
    print(f"{name} bf16: max|dlogit| {err:.4f} (tol {tol:.4f}), {n} decisive steps, first divergence {first_div}")


def test_engine_graph_equals_eager_and_module_path(dev, golden):
    g = golden("cfg1_int4")
    T, S = int(g["prompt_len"]), int(g["max_seq_length"])
    toks = _t(g["tokens"]).to(dev)
    model, _, _ = build(CFG1, "gptq.int4", torch.bfloat16, dev)
    eng = model.engine()
    assert eng is not None and eng.use_graph
    lg_graph = teacher_forced(model, toks, T, S, dev)
    assert eng._graphs, "decode steps did not go through a hipGraph"
    eng.use_graph = False
    lg_eager = teacher_forced(model, toks, T, S, dev)
    assert torch.equal(lg_graph, lg_eager), "graph replay and eager launches must be bit-identical"
    eng.use_graph = True
    # greedy fast path (device-side argmax chain) vs the reference-style loop through model.forward
    fast = lit_llama_amd.generate(model, toks[:T], toks.numel() - T, top_k=1)
    model.reset_cache()
    model.use_engine = False  # op-by-op module path: different kernels for norm / residual, same math
    lg_mod = teacher_forced(model, toks, T, S, dev)
    model.use_engine = True
    std = float(g["std"].mean())
    assert (lg_mod - lg_graph).abs().max().item() <= 0.05 * std
    slow_logits = lg_graph.argmax(-1)
    # the fast path must reproduce its own teacher-forced argmax chain while it follows the same tokens
    fast_c = fast.cpu()
    for j in range(toks.numel() - T):
        if int(fast_c[T + j]) != int(toks[T + j]):
            break
        assert int(fast_c[T + j]) == int(slow_logits[j]) or g["margin"][j] <= 0.1 * std


def test_engine_attention_split_variants_agree(dev, golden):
    """1 / 4 (fused into the c_proj prologue) / 8 (stand-alone combine: the path wide single-GPU shards take)
    K/V splits per head compute the same attention; logits agree to the bf16-path tolerance."""
    from lit_llama_amd.engine import DecodeEngine

    g = golden("cfg1_int4")
    T, S = int(g["prompt_len"]), int(g["max_seq_length"])
    toks = _t(g["tokens"]).to(dev)
    model, _, _ = build(CFG1, "gptq.int4", torch.bfloat16, dev)
    std = float(g["std"].mean())
    ref = None
    for splits in (4, 1, 8):
        model._engine = DecodeEngine(model, tune={"attn_splits": splits})
        lg = teacher_forced(model, toks, T, S, dev)
        if ref is None:
            ref = lg
        else:
            assert (lg - ref).abs().max().item() <= 0.02 * std, f"attn_splits={splits}"
    model._engine = None


def test_generate_api_sampling_and_eos(dev):
    model, _, cfg = build(CFG1, "gptq.int4", torch.bfloat16, dev)
    prompt = synth.make_prompt(6).to(dev)
    greedy = lit_llama_amd.generate(model, prompt, 12, top_k=1)
    assert greedy.shape == (18,) and greedy.dtype == prompt.dtype and torch.equal(greedy[:6], prompt)
    model.reset_cache()
    again = lit_llama_amd.generate(model, prompt, 12, top_k=1)
    assert torch.equal(greedy, again), "greedy decode must be reproducible after reset_cache()"
    model.reset_cache()
    # EOS: the reference returns idx[:input_pos] (tokens before the EOS position)
    eos = int(greedy[9])
    first = int((greedy[6:] == eos).nonzero()[0]) + 6
    cut = lit_llama_amd.generate(model, prompt, 12, top_k=1, eos_id=eos)
    assert torch.equal(cut, greedy[:first])
    model.reset_cache()
    torch.manual_seed(0)
    sampled = lit_llama_amd.generate(model, prompt, 8, temperature=0.8, top_k=50)
    assert sampled.shape == (14,) and int(sampled.max()) < cfg.padded_vocab_size
    model.reset_cache()
    # prompt of length 1 and a long prompt crossing the engine's chunk size
    one = lit_llama_amd.generate(model, prompt[:1], 4, top_k=1)
    assert one.shape == (5,)
    model.reset_cache()
    long_prompt = synth.make_prompt(37).to(dev)
    out = lit_llama_amd.generate(model, long_prompt, 5, top_k=1)
    assert out.shape == (42,) and torch.equal(out[:37], long_prompt)


def test_llm_int8_model_against_oracle(dev):
    """Config 4 on a small model (parity unpinned: oracle only).  The kernels restate the oracle's LLM.int8
    arithmetic exactly (tests/test_kernels_gpu.py), but a whole model re-quantises its activations at every linear:
    a bf16-level perturbation (KV cache, residual rounding) flips int8 levels and comes out at int8 granularity
    (~1/127).  Hence a wider band than for int4: 0.15 logit-std without outlier channels, and a sanity band with
    the x20 outlier channels switched on."""
    for outliers, band in ((0, 0.15), (4, 0.6)):
        model, sd, cfg = build(CFG1, "llm.int8", torch.bfloat16, dev, outliers=outliers)
        assert model.engine() is not None, model._engine_failed
        prompt = synth.make_prompt(8)
        om = oracle.Model(oracle.Config(**CFG1), {k: v.float() for k, v in sd.items()}, mode="llm.int8")
        ref_toks = oracle.generate(om, prompt, 10, top_k=1)
        om.reset_cache()
        ref_logits = oracle.teacher_forced_logits(om, ref_toks, 8)
        got = teacher_forced(model, ref_toks.to(dev), 8, 18, dev)
        assert torch.isfinite(got).all()
        std = float(ref_logits.std(-1).mean())
        err = (got - ref_logits).abs().max().item()
        assert err <= band * std, f"int8 (outliers={outliers}) logits off by {err:.4f} (std {std:.3f})"
        top2 = torch.topk(ref_logits, 2, dim=-1).values
        decisive = (top2[:, 0] - top2[:, 1]) > 2 * band * std
        assert torch.equal(got.argmax(-1)[decisive], ref_logits.argmax(-1)[decisive])
        print(f"int8 outliers={outliers}: max|dlogit| {err:.3f} (std {std:.3f})")


def test_7b_width_single_layer_engine_matches_oracle(dev):
    """One LLaMA-7B-shaped layer (n_embd 4096, 32 heads, n_hidden 11008, vocab 32000) through the engine:
    the production tile shapes / grids against the oracle at full width (the oracle needs ~1 minute here)."""
    cfgk = dict(n_layer=1, n_head=32, n_embd=4096)
    model, sd, cfg = build(cfgk, "gptq.int4", torch.bfloat16, dev)
    assert model.engine() is not None, model._engine_failed
    prompt = synth.make_prompt(5)
    om = oracle.Model(oracle.Config(**cfgk), {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()},
                      mode="gptq.int4")
    toks = oracle.generate(om, prompt, 3, top_k=1)
    om.reset_cache()
    ref = oracle.teacher_forced_logits(om, toks, 5)
    got = teacher_forced(model, toks.to(dev), 5, 8, dev)
    std = float(ref.std(-1).mean())
    err = (got - ref).abs().max().item()
    assert err <= 0.05 * std, f"7B-width logits off by {err:.4f} (std {std:.3f})"


def test_full_7b_int4_model_size_independent_properties(dev):
    """BASELINE.json configs[2] at FULL size (32 layers, 3.3 GB of int4 weights; the CPU oracle would need ~30 s per
    token, so the checks are properties instead of a golden run): (1) greedy decode is reproducible run to run,
    (2) the chained hipGraph replay, the un-chained graph and eager launches give bit-identical logits / tokens,
    (3) the engine agrees with the op-by-op module path (independent generic kernels, reference arithmetic order)
    to the bf16-path tolerance on teacher-forced steps."""
    from lit_llama_amd.model import LLaMA, LLaMAConfig

    cfg = LLaMAConfig.from_name("7B")
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode="gptq.int4"):
        model = LLaMA(cfg)
    model.eval()
    synth.fill_model_random_int4(model, seed=0)
    eng = model.engine()
    assert eng is not None and eng.use_graph, model._engine_failed
    prompt = synth.make_prompt(9, vocab=cfg.vocab_size, seed=3).to(dev)
    n_new = 12
    a = lit_llama_amd.generate(model, prompt, n_new, top_k=1)            # chained graph replays
    model.reset_cache()
    b = lit_llama_amd.generate(model, prompt, n_new, top_k=1)
    assert torch.equal(a, b), "greedy decode of the full model is not reproducible"
    assert a.shape == (9 + n_new,) and int(a.min()) >= 0 and int(a.max()) < cfg.padded_vocab_size
    # teacher-forced over the generated sequence: graph (un-chained) vs eager, bit for bit; argmax chain == tokens
    S = 9 + n_new
    lg_graph = teacher_forced(model, a, 9, S, dev)
    eng.use_graph = False
    lg_eager = teacher_forced(model, a, 9, S, dev)
    eng.use_graph = True
    assert torch.equal(lg_graph, lg_eager)
    assert torch.equal(lg_graph.argmax(-1).to(a.dtype).cpu(), a[9:].cpu()), \
        "the chained greedy loop and model.forward disagree on the argmax chain"
    # independent implementation: op-by-op module path (3 teacher-forced steps are enough at 32 layers)
    short = a[:12]
    model.use_engine = False
    lg_mod = teacher_forced(model, short, 9, 12, dev)
    model.use_engine = True
    std = float(lg_mod.std(-1).mean())
    err = (lg_mod - lg_graph[:lg_mod.shape[0]]).abs().max().item()
    assert err <= 0.05 * std, f"engine vs module path at 7B: {err:.4f} (std {std:.3f})"

"""Produce tests/golden/*.npz by running the UNMODIFIED reference (/root/reference) in this container.

    python oracle/gen_golden.py            # needs /root/reference; run in the build container only

The reference imports `lightning` at module import time (lit_llama/utils.py:15, generate.py:9); that package is
absent here, so oracle/_stubs/lightning provides the two names touched at import.  Nothing under
/root/reference is modified or copied.  Weights are NOT stored: they are re-created from a seed by
lit_llama_amd/synth.py (torch's CPU generator is deterministic for a given torch build), only inputs and the
reference's outputs are written.  The script also checks oracle/oracle.py (the restatement) against the same
runs and refuses to write fixtures the restatement does not reproduce.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
sys.path[:0] = [str(ROOT / "oracle" / "_stubs"), str(REF), str(ROOT)]

import generate as ref_generate  # noqa: E402  (reference generate.py)
import lit_llama as ref  # noqa: E402
from lit_llama.quantization import ColBlockQuantizedLinear as RefColBlock  # noqa: E402
from lit_llama.utils import quantization as ref_quantization  # noqa: E402

from lit_llama_amd import synth  # noqa: E402
from lit_llama_amd.model import LLaMAConfig as OurConfig  # noqa: E402
from oracle import oracle  # noqa: E402

OUT = ROOT / "tests" / "golden"
PROBES = 64  # logit columns stored per step


def probe_index(vocab: int) -> np.ndarray:
    return (np.arange(PROBES) * (vocab // PROBES) + 7) % vocab


@torch.no_grad()
def ref_teacher_forced(model, tokens, prompt_len, max_seq_length):
    model.reset_cache() if model.mask_cache is not None else None
    out = []
    input_pos = torch.arange(0, prompt_len)
    for _ in range(tokens.numel() - prompt_len):
        x = tokens.index_select(0, input_pos).view(1, -1)
        out.append(model(x, max_seq_length, input_pos)[0, -1].float())
        input_pos = input_pos[-1:] + 1
    model.reset_cache()
    return torch.stack(out)


def summarize(logits: torch.Tensor, vocab: int):
    top2 = torch.topk(logits, 2, dim=-1)
    return dict(
        argmax=top2.indices[:, 0].numpy().astype(np.int32),
        margin=(top2.values[:, 0] - top2.values[:, 1]).numpy().astype(np.float32),
        probes=logits[:, torch.from_numpy(probe_index(vocab))].numpy().astype(np.float32),
        std=logits.std(dim=-1).numpy().astype(np.float32),
        mean=logits.mean(dim=-1).numpy().astype(np.float32),
    )


def model_case(name, cfg_kwargs, mode, prompt_len, new_tokens, max_seq_length=None, seed=0, nocache=True, stats="unit"):
    torch.manual_seed(1234)
    ref_cfg = ref.LLaMAConfig(**cfg_kwargs)
    our_cfg = OurConfig(**cfg_kwargs)
    sd = synth.make_state_dict(our_cfg, seed=seed, mode=mode, stats=stats)
    with ref_quantization(mode):
        model = ref.LLaMA(ref_cfg)
    model.load_state_dict(sd)
    model.eval()
    prompt = synth.make_prompt(prompt_len, vocab=ref_cfg.vocab_size)
    toks = ref_generate.generate(model, prompt, new_tokens, top_k=1, max_seq_length=max_seq_length)
    model.reset_cache()
    S = max_seq_length if max_seq_length is not None else min(prompt_len + new_tokens, ref_cfg.block_size)
    rolled = prompt_len + new_tokens > S
    fix = dict(tokens=toks.numpy().astype(np.int32), prompt_len=np.int32(prompt_len), seed=np.int32(seed),
               max_seq_length=np.int32(S))
    if stats != "unit":
        fix["stats"] = np.array(stats)
    if not rolled:
        logits = ref_teacher_forced(model, toks, prompt_len, S)
        fix.update(summarize(logits, ref_cfg.padded_vocab_size))
        if nocache:
            # no-cache full forward over the finished sequence (evaluate/full.py:120-129 shape of call)
            with torch.no_grad():
                full = model(toks[:-1].view(1, -1).long())[0].float()
            fix["nocache_argmax"] = full.argmax(-1).numpy().astype(np.int32)
            fix["nocache_probes"] = full[:, torch.from_numpy(probe_index(ref_cfg.padded_vocab_size))].numpy().astype(np.float32)
            model.reset_cache()

    # ---- pin the restatement on the very same run
    om = oracle.Model(oracle.Config(**cfg_kwargs), {k: v.clone() for k, v in sd.items()}, mode=mode)
    otoks = oracle.generate(om, prompt, new_tokens, top_k=1, max_seq_length=max_seq_length)
    assert torch.equal(otoks, toks), f"{name}: oracle tokens differ from the reference"
    if not rolled:
        ologits = oracle.teacher_forced_logits(om, toks, prompt_len, S)
        err = (ologits - logits).abs().max().item()
        assert err <= 1e-5 * max(1.0, logits.abs().max().item()), f"{name}: oracle logits off by {err}"
        print(f"  {name}: oracle == reference (tokens equal, max |dlogit| {err:.2e}, min margin {fix['margin'].min():.3e})")
    else:
        print(f"  {name}: oracle == reference (tokens equal, cache-roll regime)")
    np.savez_compressed(OUT / f"{name}.npz", **fix)


def colblock_cases():
    fix = {}
    gen = torch.Generator().manual_seed(7)
    for tag, (N, K, bits, tile_cols) in {"b4_row": (96, 256, 4, -1), "b4_g64": (48, 256, 4, 64),
                                          "b8_row": (32, 128, 8, -1)}.items():
        mod = RefColBlock(K, N, bias=False, bits=bits, tile_cols=tile_cols)
        w = torch.randn((N, K), generator=gen) * K**-0.5
        tc = mod.tile_cols
        G = mod.scales.shape[1]
        maxq = 2**bits - 1
        scales, zeros = torch.empty((N, G)), torch.empty((N, G))
        for g in range(G):
            blk = w[:, g * tc:(g + 1) * tc]
            xmin = torch.minimum(blk.min(1)[0], torch.zeros(N))
            xmax = torch.maximum(blk.max(1)[0], torch.zeros(N))
            scales[:, g] = (xmax - xmin) / maxq
            zeros[:, g] = torch.round(-xmin / scales[:, g])
        mod.scales.copy_(scales)
        mod.zeros.copy_(zeros)
        mod.pack_weight(w)
        x = torch.randn((3, K), generator=gen)
        y = mod(x)
        wdq = mod.get_weight()
        # restatement pin
        oq = oracle.colblock_pack(w, scales, zeros, bits, tc)
        assert torch.equal(oq, mod.quant_weight), f"colblock {tag}: pack differs"
        assert torch.equal(oracle.colblock_get_weight(oq, scales, zeros, bits, tc), wdq), f"colblock {tag}: dequant differs"
        assert torch.equal(oracle.colblock_linear(x, oq, scales, zeros, bits, tc), y), f"colblock {tag}: forward differs"
        fix.update({f"{tag}_w": w.numpy(), f"{tag}_q": mod.quant_weight.contiguous().numpy(),
                    f"{tag}_scales": scales.numpy(), f"{tag}_zeros": zeros.numpy(), f"{tag}_x": x.numpy(),
                    f"{tag}_y": y.numpy(), f"{tag}_wdq": wdq.numpy(),
                    f"{tag}_meta": np.array([N, K, bits, tc], dtype=np.int32),
                    f"{tag}_qstride": np.array(mod.quant_weight.stride(), dtype=np.int64)})
    print("  colblock: oracle == reference (pack / get_weight / forward bit-equal)")
    np.savez_compressed(OUT / "colblock.npz", **fix)


def block_cases():
    """RMSNorm, RoPE, Block forward (no cache, B = 3) and attention with cache on small dims, like the shapes of
    tests/test_model.py:37-102, tests/test_rope.py, tests/test_rmsnorm.py of the reference."""
    torch.manual_seed(11)
    fix = {}
    x = torch.randn(2, 16, 16)
    norm = ref.RMSNorm(16, eps=1e-6)
    norm.scale.data = 1 + 0.1 * torch.randn(16)
    fix.update(rms_x=x.numpy(), rms_scale=norm.scale.detach().numpy(), rms_y=norm(x).detach().numpy())
    assert torch.equal(oracle.rmsnorm(x, norm.scale.detach(), 1e-6), norm(x).detach())

    rc = ref.build_rope_cache(seq_len=6, n_elem=4, dtype=torch.float32, device="cpu")
    xr = torch.randn(1, 6, 2, 4)
    fix.update(rope_cache=rc.numpy(), rope_x=xr.numpy(), rope_y=ref.apply_rope(xr, rc).numpy())
    big = ref.build_rope_cache(seq_len=2048, n_elem=128, dtype=torch.int64, device="cpu")
    assert torch.equal(oracle.build_rope_cache(2048, 128, dtype=torch.int64), big)
    fix.update(rope_big_rows=big[[0, 1, 17, 511, 2047]].numpy())

    cfg = ref.LLaMAConfig(block_size=64, vocab_size=100, n_layer=2, n_head=4, n_embd=32)
    model = ref.LLaMA(cfg)
    model.apply(model._init_weights)
    for p_ in model.parameters():
        p_.data = torch.randn_like(p_) * (0.3 if p_.dim() > 1 else 0.1) + (0 if p_.dim() > 1 else 1)
    model.eval()
    idx = torch.randint(0, 100, (3, 9))
    with torch.no_grad():
        logits = model(idx)
        logits_pos = model(idx[:1], 12, torch.arange(9))
        kc, vc = model.kv_caches[1]
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    om = oracle.Model(oracle.Config(block_size=64, vocab_size=100, n_layer=2, n_head=4, n_embd=32), sd)
    with torch.no_grad():
        assert torch.allclose(om(idx), logits, atol=1e-6), "block: oracle no-cache forward differs"
        ol = om(idx[:1], 12, torch.arange(9))
        assert torch.allclose(ol, logits_pos, atol=1e-6), "block: oracle cached forward differs"
        assert torch.allclose(om.kv_caches[1][0], kc, atol=1e-6)
    fix.update({f"blk_sd::{k}": v.numpy() for k, v in sd.items()})
    fix.update(blk_idx=idx.numpy().astype(np.int64), blk_logits=logits.numpy(), blk_logits_pos=logits_pos.numpy(),
               blk_kcache=kc.numpy(), blk_vcache=vc.numpy())
    print("  rmsnorm / rope / block: oracle == reference")
    np.savez_compressed(OUT / "blocks.npz", **fix)


def lora_cases():
    """lit_llama/lora.py `MergedLinear` (the c_attn of the LoRA attention block, enable_lora = [True, False, True]):
    merged weight after `.eval()` in f32 and bf16, and the unmerged forward, for seeded W / A / B."""
    from lit_llama.lora import MergedLinear as RefMerged

    out = {}
    C, r, alpha = 64, 4, 16
    gen = torch.Generator().manual_seed(11)
    W = torch.randn((3 * C, C), generator=gen) * C**-0.5
    A = torch.randn((2 * r, C), generator=gen) * 0.3
    B = torch.randn((2 * C, r), generator=gen) * 0.3
    x = torch.randn((1, 5, C), generator=gen)  # (batch, T, features): the reference's zero_pad assumes 3-D activations
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        m = RefMerged(C, 3 * C, r=r, lora_alpha=alpha, lora_dropout=0.0, enable_lora=[True, False, True],
                      fan_in_fan_out=False, merge_weights=True, bias=False).to(dt)
        with torch.no_grad():
            m.weight.copy_(W.to(dt))
            m.lora_A.copy_(A.to(dt))
            m.lora_B.copy_(B.to(dt))
        m.train(True)
        y_unmerged = m(x.to(dt)).detach().float()
        m.train(False)
        merged = m.weight.detach().float()
        y_merged = m(x.to(dt)).detach().float()
        # the restatement must reproduce the reference before anything is written
        om = oracle.lora_merge(W.to(dt), A.to(dt), B.to(dt), alpha).float()
        tol = 0.0 if dt == torch.float32 else 2.0**-7
        assert (om - merged).abs().max().item() <= tol * merged.abs().max().item() + (1e-6 if dt == torch.float32 else 0), name
        oy = oracle.lora_forward_unmerged(x.to(dt), W.to(dt), A.to(dt), B.to(dt), alpha).float()
        assert (oy - y_unmerged).abs().max().item() <= (1e-5 if dt == torch.float32 else 0.1), name
        out[f"{name}_merged"] = merged.numpy()
        out[f"{name}_y_unmerged"] = y_unmerged.numpy()
        out[f"{name}_y_merged"] = y_merged.numpy()
    out.update(W=W.numpy(), A=A.numpy(), B=B.numpy(), x=x.numpy(), meta=np.array([C, r, alpha], dtype=np.int64))
    sd_keys = sorted(k for k in RefMerged(C, 3 * C, r=r, lora_alpha=alpha, enable_lora=[True, False, True], bias=False).state_dict())
    out["state_dict_keys"] = np.array(sd_keys)
    np.savez_compressed(OUT / "lora.npz", **out)
    print("lora.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


ADAPTER_CFG = dict(n_layer=3, n_head=4, n_embd=64, vocab_size=128, block_size=64)


def adapter_case():
    """lit_llama/adapter.py LLaMA (adapter_prompt_length 10, adapter_start_layer 2, non-zero gates) through the
    reference's generate(): greedy tokens and teacher-forced logits, f32 on the CPU."""
    import lit_llama.adapter as ref_adapter

    ours = OurConfig(**ADAPTER_CFG)
    sd = synth.make_state_dict(ours, seed=21, mode=None, dtype=torch.float32)
    sd.update(synth.make_adapter_state(ours, seed=22))
    model = ref_adapter.LLaMA(ref_adapter.LLaMAConfig(**ADAPTER_CFG))
    model.load_state_dict(sd)
    model.eval()
    prompt = synth.make_prompt(6, vocab=ADAPTER_CFG["vocab_size"], seed=5)
    T, new = 6, 12
    toks = ref_generate.generate(model, prompt, new, top_k=1)
    logits = ref_teacher_forced(model, toks, T, T + new)
    # the same model WITHOUT the adapter term must differ (the fixture exercises the prefix attention)
    plain = ref.LLaMA(ref.LLaMAConfig(**ADAPTER_CFG))
    plain.load_state_dict({k: v for k, v in sd.items() if "adapter_wte" not in k and "gating_factor" not in k})
    lp = ref_teacher_forced(plain.eval(), toks, T, T + new)
    assert (lp - logits).abs().max().item() > 0.05 * float(logits.std(-1).mean())
    om = oracle.AdapterModel(oracle.Config(**ADAPTER_CFG), sd)
    ot = oracle.generate(om, prompt, new, top_k=1)
    om.reset_cache()
    ol = oracle.teacher_forced_logits(om, toks, T)
    assert torch.equal(ot, toks) and (ol - logits).abs().max().item() <= 1e-4, "oracle.AdapterModel does not reproduce the reference"
    out = dict(tokens=toks.numpy().astype(np.int32), prompt_len=np.int64(T), max_seq_length=np.int64(T + new),
               logits=logits.numpy().astype(np.float32), **summarize(logits, ADAPTER_CFG["vocab_size"]))
    np.savez_compressed(OUT / "adapter.npz", **out)
    print("adapter.npz: tokens", toks.tolist(), "min margin", float(out["margin"].min()), "std", float(out["std"].mean()))


def adapter_v2_case():
    """lit_llama/adapter_v2.py: the adapter model of `adapter_case` plus scale / bias on every linear."""
    import lit_llama.adapter as ref_adapter
    import lit_llama.adapter_v2 as ref_v2

    ours = OurConfig(**ADAPTER_CFG)
    sd = synth.make_state_dict(ours, seed=21, mode=None, dtype=torch.float32)
    sd.update(synth.make_adapter_state(ours, seed=22))
    sd.update(synth.make_adapter_v2_state(sd, seed=23))
    model = ref_adapter.LLaMA(ref_adapter.LLaMAConfig(**ADAPTER_CFG))
    ref_v2.add_adapter_v2_parameters_to_linear_layers(model)
    model.load_state_dict(sd)
    model.eval()
    prompt = synth.make_prompt(6, vocab=ADAPTER_CFG["vocab_size"], seed=5)
    T, new = 6, 12
    toks = ref_generate.generate(model, prompt, new, top_k=1)
    logits = ref_teacher_forced(model, toks, T, T + new)
    om = oracle.AdapterModel(oracle.Config(**ADAPTER_CFG), sd)
    ot = oracle.generate(om, prompt, new, top_k=1)
    om.reset_cache()
    ol = oracle.teacher_forced_logits(om, toks, T)
    assert torch.equal(ot, toks) and (ol - logits).abs().max().item() <= 1e-4, "oracle does not reproduce adapter v2"
    v1 = np.load(OUT / "adapter.npz")
    assert not np.array_equal(v1["tokens"], toks.numpy()) or np.abs(v1["logits"] - logits.numpy()).max() > 0.05
    out = dict(tokens=toks.numpy().astype(np.int32), prompt_len=np.int64(T), max_seq_length=np.int64(T + new),
               logits=logits.numpy().astype(np.float32), state_dict_keys=np.array(sorted(model.state_dict())),
               **summarize(logits, ADAPTER_CFG["vocab_size"]))
    np.savez_compressed(OUT / "adapter_v2.npz", **out)
    print("adapter_v2.npz: tokens", toks.tolist(), "min margin", float(out["margin"].min()), "std", float(out["std"].mean()))


def big_case():
    """BASELINE.json configs[2] at FULL depth: LLaMA-7B (32 layers) gptq.int4 with seeded synthetic weights, prompt of 8,
    six greedy tokens, teacher-forced logits (probes / argmax / margins).  ~25 forwards of the real reference on the
    CPU (every call dequantises 3.3 GB of int4 weights); run with `--big`."""
    torch.set_num_threads(max(8, (torch.get_num_threads() or 8)))
    model_case("cfg2_7b_int4", dict(n_layer=32, n_head=32, n_embd=4096), "gptq.int4", prompt_len=8, new_tokens=6,
               nocache=False)


def big_long_case():
    """The same checkpoint with a 128-token prompt and 16 greedy tokens (VERDICT r2 item 2 ii): the prompt goes through
    the wide path of the engine (GEMM + flash attention) at full depth, the decode steps start at position 128."""
    torch.set_num_threads(max(8, (torch.get_num_threads() or 8)))
    model_case("cfg2_7b_int4_long", dict(n_layer=32, n_head=32, n_embd=4096), "gptq.int4", prompt_len=128, new_tokens=16,
               nocache=False)


def big_s1_case():
    """A SECOND full-depth 7B int4 checkpoint (seed 1: other weights, scales and zero points), prompt of 24, 32 greedy tokens: the
    token-for-token evidence of north_star on more than one checkpoint ("one fixture pair is one fixture pair").
    `--big-s1`; bf16 calibration twin: `--big-bf16 --s1`."""
    torch.set_num_threads(max(8, (torch.get_num_threads() or 8)))
    model_case("cfg2_7b_int4_s1", dict(n_layer=32, n_head=32, n_embd=4096), "gptq.int4", prompt_len=24, new_tokens=32, seed=1,
               nocache=False)


def big_real_case():
    """A full-depth 7B int4 checkpoint with the statistics of a TRAINED LLaMA as a narrow hand-off format sees them (VERDICT r4 item 1:
    `synth.make_state_dict(stats="llama")` — embeddings of std 0.02, norm scales 0.05 .. 0.5, massive-activation channels hundreds of
    times the rms of the residual stream, SwiGLU outputs up to ~10^4), prompt of 24, 24 greedy tokens from the UNMODIFIED reference on
    the CPU.  `--big-real`; bf16 calibration twin: `--big-bf16 --real`."""
    torch.set_num_threads(max(8, (torch.get_num_threads() or 8)))
    model_case("cfg2_7b_int4_real", dict(n_layer=32, n_head=32, n_embd=4096), "gptq.int4", prompt_len=24, new_tokens=24, seed=2,
               nocache=False, stats="llama")


WIDE = dict(n_layer=2, n_head=64, n_embd=8192)  # two blocks at the LLaMA-65B width (lit_llama/model.py:47; n_hidden 22016)


def wide_case():
    """BASELINE.json configs[4]'s WIDTH (LLaMA-65B: n_embd 8192, 64 heads, n_hidden 22016, vocab 32000) at two layers, gptq.int4, seeded
    synthetic weights: prompt of 24, 12 greedy tokens from the UNMODIFIED reference on the CPU — the fixture of the wide-shape persistent
    step (csrc/fused_step_wide.hip, round 6).  `--wide`; bf16 calibration twin: `--big-bf16 --wide`."""
    torch.set_num_threads(max(8, (torch.get_num_threads() or 8)))
    model_case("cfg4_65b_w2_int4", WIDE, "gptq.int4", prompt_len=24, new_tokens=12, seed=3, nocache=False)


def big_real_aux():
    """Adds `oracle_swiglu_absmax` [positions] to cfg2_7b_int4_real.npz: the largest |silu(c_fc1 x) * c_fc2 x| of the block that holds
    the massive hidden units (synth.llama_stats_plan), per position of the fixture's token sequence, from the ORACLE's activations
    (oracle == reference on this fixture, checked when it was written; the first blocks only — seconds).  The GPU test uses it to
    say which decode steps must leave the range of the persistent step's fp8 hand-off (+-7168 on this edge)."""
    import torch.nn.functional as F

    name = "cfg2_7b_int4_real"
    with np.load(OUT / f"{name}.npz") as z:
        fx = {k: z[k] for k in z.files}
    cfg_kwargs = dict(n_layer=32, n_head=32, n_embd=4096)
    our = OurConfig(**cfg_kwargs)
    plan = synth.llama_stats_plan(our)
    L = plan["layer"] + 1
    small = OurConfig(n_layer=L, n_head=32, n_embd=4096)
    sd_full_seed = int(fx["seed"])
    # the first L blocks of the fixture's checkpoint: the generator draws lm_head, then the blocks in order, so a model of L layers
    # built from the same seed shares embedding, norms of those blocks and their linears with the 32-layer one ONLY if the norm draws
    # agree — they do not (all norm scales are drawn before the linears), so build the full state dict and keep what is needed
    sd = synth.make_state_dict(our, seed=sd_full_seed, mode="gptq.int4", stats=str(fx["stats"]))
    keep = {k: v for k, v in sd.items() if not k.startswith("transformer.h.") or int(k.split(".")[2]) < L}
    del sd
    om = oracle.Model(oracle.Config(n_layer=L, n_head=32, n_embd=4096), keep, mode="gptq.int4")
    toks = torch.from_numpy(fx["tokens"].astype(np.int64)).view(1, -1)
    T = toks.shape[1]
    rope = oracle.build_rope_cache(2048, 128)[:T]
    mask = torch.tril(torch.ones(T, T, dtype=torch.bool))[None, None]
    with torch.no_grad():
        x = F.embedding(toks, keep["transformer.wte.weight"].float())
        for i in range(L):
            pre = f"transformer.h.{i}."
            h, _ = om.attention(i, oracle.rmsnorm(x, keep[pre + "rms_1.scale"].float()), rope, mask, 2048, None, None)
            x = x + h
            xn = oracle.rmsnorm(x, keep[pre + "rms_2.scale"].float())
            hh = F.silu(oracle.linear(keep, pre + "mlp.c_fc1", xn, "gptq.int4")) * oracle.linear(keep, pre + "mlp.c_fc2", xn, "gptq.int4")
            if i == plan["layer"]:
                amax = hh[0].abs().amax(-1)
            x = x + oracle.linear(keep, pre + "mlp.c_proj", hh, "gptq.int4")
    fx["oracle_swiglu_absmax"] = amax.numpy().astype(np.float32)
    np.savez_compressed(OUT / f"{name}.npz", **fx)
    Tp = int(fx["prompt_len"])
    print("oracle_swiglu_absmax (decode positions):", np.round(fx["oracle_swiglu_absmax"][Tp:], 0).tolist())
    print("positions past 7168:", [int(p) for p in np.nonzero(fx["oracle_swiglu_absmax"] > 7168)[0]])


def big_p400_case():
    """The same checkpoint with a 400-token prompt and 16 greedy tokens: decode runs at positions
    400..415, i.e. past the fused step's row-split threshold (position 384) at FULL depth — where the reference's own
    usage lives (generate.py:94-155 with real prompts, evaluate/full.py:120-129)."""
    torch.set_num_threads(max(8, (torch.get_num_threads() or 8)))
    model_case("cfg2_7b_int4_p400", dict(n_layer=32, n_head=32, n_embd=4096), "gptq.int4", prompt_len=400, new_tokens=16,
               nocache=False)


def big_none_case():
    """BASELINE.json configs[1] at FULL depth: LLaMA-7B (32 layers) WITHOUT quantisation, seeded synthetic weights (every value a bf16
    number, held in f32: 27 GB), prompt of 8, six greedy tokens, teacher-forced logits — the unmodified reference in f32 on the CPU.
    Built to fit the container: the state dict is ASSIGNED to a meta-device reference model (no second copy) and the oracle reads the
    same tensors.  `--big-none`; its bf16 calibration twin: `--big-none --bf16`."""
    torch.set_num_threads(max(8, (torch.get_num_threads() or 8)))
    name = "cfg1_7b_none"
    cfg_kwargs = dict(n_layer=32, n_head=32, n_embd=4096)
    ref_cfg = ref.LLaMAConfig(**cfg_kwargs)
    prompt_len, new_tokens, seed = 8, 6, 0
    sd = synth.make_state_dict(OurConfig(**cfg_kwargs), seed=seed, mode=None)
    if "--bf16" in sys.argv:
        # the reference's own bf16 run (`--precision bf16-true`) on the tokens of the f32 fixture: the calibration of the GPU test's bar
        fx = np.load(OUT / f"{name}.npz")
        sd = {k: v.to(torch.bfloat16) for k, v in sd.items()}
        with torch.device("meta"):
            model = ref.LLaMA(ref_cfg)
        model.load_state_dict(sd, assign=True)
        model.eval()
        toks = torch.from_numpy(fx["tokens"].astype(np.int32))
        old = torch.get_default_dtype()
        torch.set_default_dtype(torch.bfloat16)
        try:
            logits = ref_teacher_forced(model, toks, prompt_len, int(fx["max_seq_length"]))
        finally:
            torch.set_default_dtype(old)
        probes_bf = logits[:, torch.from_numpy(probe_index(ref_cfg.padded_vocab_size))].float().numpy().astype(np.float32)
        d = np.abs(probes_bf - fx["probes"]).max(axis=1) / fx["std"]
        np.savez_compressed(OUT / f"{name}_bf16ref.npz", probes=probes_bf, argmax=logits.float().argmax(-1).numpy().astype(np.int32),
                            dist_std=d.astype(np.float32), max_dist_std=np.float32(d.max()), source=np.array(name))
        print(f"{name}_bf16ref.npz: reference bf16 vs reference f32, max |dlogit| / std per step:", np.round(d, 4).tolist())
        return
    with torch.device("meta"):
        model = ref.LLaMA(ref_cfg)
    model.load_state_dict(sd, assign=True)
    model.eval()
    prompt = synth.make_prompt(prompt_len, vocab=ref_cfg.vocab_size)
    toks = ref_generate.generate(model, prompt, new_tokens, top_k=1)
    model.reset_cache()
    S = prompt_len + new_tokens
    logits = ref_teacher_forced(model, toks, prompt_len, S)
    fix = dict(tokens=toks.numpy().astype(np.int32), prompt_len=np.int32(prompt_len), seed=np.int32(seed), max_seq_length=np.int32(S))
    fix.update(summarize(logits, ref_cfg.padded_vocab_size))
    om = oracle.Model(oracle.Config(**cfg_kwargs), sd, mode=None)
    otoks = oracle.generate(om, prompt, new_tokens, top_k=1)
    assert torch.equal(otoks, toks), f"{name}: oracle tokens differ from the reference"
    om.reset_cache()
    ologits = oracle.teacher_forced_logits(om, toks, prompt_len, S)
    err = (ologits - logits).abs().max().item()
    assert err <= 1e-5 * max(1.0, logits.abs().max().item()), f"{name}: oracle logits off by {err}"
    print(f"  {name}: oracle == reference (tokens equal, max |dlogit| {err:.2e}, min margin {fix['margin'].min():.3e})")
    np.savez_compressed(OUT / f"{name}.npz", **fix)


def big_bf16_case(name="cfg2_7b_int4"):
    """Calibrates the engine's parity bar: the REFERENCE ITSELF in bf16 on the CPU (what `--precision bf16-true` makes
    of it: parameters, scales / zeros and activations in bf16, generate.py:123-134) on the tokens of the f32 fixture,
    teacher-forced.  Stored: its logit probes / argmax and its distance from the reference's own f32 run, in units of
    the f32 run's logit std.  The reference states one tolerance for this comparison (bf16 device run vs f32 CPU run):
    atol 5e-3 + rtol 1e-3 (tests/test_model.py:133)."""
    torch.set_num_threads(max(8, (torch.get_num_threads() or 8)))
    fx = np.load(OUT / f"{name}.npz")
    cfg_kwargs = WIDE if name.startswith("cfg4_65b_w2") else dict(n_layer=32, n_head=32, n_embd=4096)
    ref_cfg = ref.LLaMAConfig(**cfg_kwargs)
    sd = synth.make_state_dict(OurConfig(**cfg_kwargs), seed=int(fx["seed"]), mode="gptq.int4",
                               stats=str(fx["stats"]) if "stats" in fx.files else "unit")
    with ref_quantization("gptq.int4"):
        model = ref.LLaMA(ref_cfg)
    model.load_state_dict(sd)
    del sd
    model = model.to(torch.bfloat16).eval()  # quant_weight stays uint8; scales / zeros / wte / norms become bf16
    toks = torch.from_numpy(fx["tokens"].astype(np.int32))
    T = int(fx["prompt_len"])
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)  # Fabric's bf16-true runs forward under this default (rope cache dtype)
    try:
        logits = ref_teacher_forced(model, toks, T, int(fx["max_seq_length"]))
    finally:
        torch.set_default_dtype(old)
    probes_bf = logits[:, torch.from_numpy(probe_index(ref_cfg.padded_vocab_size))].numpy().astype(np.float32)
    d = np.abs(probes_bf - fx["probes"]).max(axis=1) / fx["std"]
    out = dict(probes=probes_bf, argmax=logits.argmax(-1).numpy().astype(np.int32), dist_std=d.astype(np.float32),
               max_dist_std=np.float32(d.max()), source=np.array(name))
    np.savez_compressed(OUT / f"{name}_bf16ref.npz", **out)
    print(f"{name}_bf16ref.npz: reference bf16 vs reference f32, max |dlogit| / std per step:", np.round(d, 4).tolist(),
          "argmax equal:", bool((out["argmax"] == fx["argmax"]).all()))


def lazy_load_case():
    """f2: the reference's `lazy_load` (lit_llama/utils.py:166-344) on a small committed checkpoint — keys, dtypes, shapes,
    strides, Parameter-ness and values of what it hands out; tests/test_checkpoint.py holds lit_llama_amd.utils.lazy_load to it."""
    from lit_llama.utils import lazy_load as ref_lazy_load

    gen = torch.Generator().manual_seed(11)
    base = torch.arange(64, dtype=torch.float32).reshape(8, 8)
    sd = {
        "a.weight": torch.randn((5, 7), generator=gen),
        "b.bf16": torch.randn((4, 6), generator=gen).to(torch.bfloat16),
        # the transposed view ColBlockQuantizedLinear registers (lit_llama/quantization.py:350-358)
        "c.quant_weight": torch.randint(0, 255, (6, 10), generator=gen, dtype=torch.uint8).t(),
        "d.view1": base[2:5],      # two views of ONE storage, one of them strided
        "d.view2": base[:, 3],
        "e.scalar": torch.tensor(3.5),
        "f.param": torch.nn.Parameter(torch.randn((3, 3), generator=gen)),
        "g.int": torch.arange(10, dtype=torch.int64),
        "h.half": torch.randn((2, 2, 2), generator=gen).half(),
    }
    ckpt = OUT / "lazy_ckpt.pth"
    torch.save(sd, ckpt)
    out = {"keys": np.array(list(sd))}
    with ref_lazy_load(ckpt) as lz:
        assert list(lz) == list(sd)
        for k, v in lz.items():
            assert type(v).__name__ == "NotYetLoadedTensor", (k, type(v))
            t = v._load_tensor()
            assert torch.equal(t, sd[k])
            raw = t.detach().contiguous()
            raw = raw.view(torch.int16) if raw.dtype in (torch.bfloat16, torch.float16) else raw
            out[k + "/values"] = raw.numpy()
            out[k + "/meta"] = np.array([str(t.dtype), str(tuple(t.shape)), str(tuple(t.stride())),
                                         str(isinstance(t, torch.nn.Parameter))])
    np.savez(OUT / "lazy_load.npz", **out)
    print("lazy_load fixture:", len(sd), "tensors ->", ckpt)


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    if "--lazy-load" in sys.argv:
        print("generating the lazy_load fixture from", REF)
        lazy_load_case()
        return
    if "--big-none" in sys.argv:
        print("generating the full-depth unquantised 7B fixture from", REF)
        big_none_case()
        return
    if "--big-real-aux" in sys.argv:
        big_real_aux()
        return
    if "--wide" in sys.argv and "--big-bf16" not in sys.argv:
        print("generating the two-layer 65B-width fixture from", REF)
        wide_case()
        return
    if "--big-real" in sys.argv:
        print("generating the LLaMA-statistics full-depth 7B fixture from", REF)
        big_real_case()
        return
    if "--big-s1" in sys.argv:
        print("generating the second-checkpoint full-depth 7B fixture from", REF)
        big_s1_case()
        return
    if "--big-p400" in sys.argv:
        print("generating the 400-token-prompt full-depth 7B fixture from", REF)
        big_p400_case()
        return
    if "--big-long" in sys.argv:
        print("generating the long full-depth 7B fixture from", REF)
        big_long_case()
        return
    if "--big-bf16" in sys.argv:
        print("running the reference in bf16 on the full-depth fixture from", REF)
        big_bf16_case("cfg4_65b_w2_int4" if "--wide" in sys.argv else "cfg2_7b_int4_real" if "--real" in sys.argv else "cfg2_7b_int4_s1" if "--s1" in sys.argv else "cfg2_7b_int4_p400" if "--p400" in sys.argv else
                      "cfg2_7b_int4_long" if "--long" in sys.argv else "cfg2_7b_int4")
        return
    if "--adapter-v2" in sys.argv:
        print("generating the LLaMA-Adapter v2 fixture from", REF)
        adapter_v2_case()
        return
    if "--adapter" in sys.argv:
        print("generating the LLaMA-Adapter fixture from", REF)
        adapter_case()
        return
    if "--lora" in sys.argv:
        print("generating the LoRA fixture from", REF)
        lora_cases()
        return
    if "--big" in sys.argv:
        print("generating the full-depth 7B fixture from", REF)
        big_case()
        return
    torch.set_num_threads(8)
    print("generating golden fixtures from", REF)
    colblock_cases()
    block_cases()
    cfg1 = dict(n_layer=2, n_head=4, n_embd=256)
    model_case("cfg1_fp32", cfg1, None, prompt_len=8, new_tokens=24)
    model_case("cfg1_int4", cfg1, "gptq.int4", prompt_len=8, new_tokens=24)
    # `--quantize gptq.int8` (lit_llama/utils.py:100-102, :150-152: ColBlockQuantizedLinear with bits = 8): the reference dequantises
    # the whole matrix on every forward call (quantization.py:413-423)
    model_case("cfg1_int8g", cfg1, "gptq.int8", prompt_len=8, new_tokens=24)
    # the cache-roll regime of tests/test_generate.py:26-54 (max_seq_length < T + max_new_tokens)
    tiny = dict(block_size=128, vocab_size=16, n_layer=1, n_head=4, n_embd=8)
    model_case("tiny_roll", tiny, None, prompt_len=5, new_tokens=20, max_seq_length=10)
    model_case("tiny_noroll", tiny, None, prompt_len=5, new_tokens=20)
    print("done ->", OUT)


if __name__ == "__main__":
    main()

"""LoRA inference variant (lit_llama_amd/lora.py) against the reference's own `MergedLinear` (tests/golden/lora.npz,
generated from /root/reference lit_llama/lora.py by `python oracle/gen_golden.py --lora`) and the oracle restatement.
Host logic only: the merged model then decodes through the ordinary bf16 path (tests/test_lora_gpu.py)."""
import numpy as np
import pytest
import torch

from lit_llama_amd import lora as L
from lit_llama_amd import model as llama
from lit_llama_amd.model import LLaMA, LLaMAConfig
from oracle import oracle


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _module(g, dtype):
    C, r, alpha = (int(v) for v in g["meta"])
    m = L.MergedLinear(C, 3 * C, r=r, lora_alpha=alpha, lora_dropout=0.0, enable_lora=[True, False, True], bias=False).to(dtype)
    with torch.no_grad():
        m.weight.copy_(_t(g["W"]).to(dtype))
        m.lora_A.copy_(_t(g["A"]).to(dtype))
        m.lora_B.copy_(_t(g["B"]).to(dtype))
    return m, alpha


@pytest.mark.parametrize("name,dtype,tol", [("f32", torch.float32, 1e-6), ("bf16", torch.bfloat16, 2.0**-7)])
def test_merge_matches_the_reference_module(golden, name, dtype, tol):
    g = golden("lora")
    ref = _t(g[f"{name}_merged"])
    W, A, B = (_t(g[k]).to(dtype) for k in "WAB")
    alpha = int(g["meta"][2])
    om = oracle.lora_merge(W, A, B, alpha).float()
    assert (om - ref).abs().max().item() <= tol * ref.abs().max().item()
    m, _ = _module(g, dtype)
    assert sorted(m.state_dict()) == [str(k) for k in g["state_dict_keys"]]  # lora_A, lora_B, weight
    assert not m._mi355_plain_weight
    v0 = m.weight._version
    m.eval()
    assert m.merged and m._mi355_plain_weight and m.weight._version > v0  # (an engine fingerprint sees the change)
    assert (m.weight.float() - ref).abs().max().item() <= tol * ref.abs().max().item()
    # the k rows are untouched, the q and v rows changed
    C = m.in_features
    assert torch.equal(m.weight[C:2 * C], W[C:2 * C])
    assert not torch.equal(m.weight[:C], W[:C]) and not torch.equal(m.weight[2 * C:], W[2 * C:])
    m.eval()  # idempotent
    assert (m.weight.float() - ref).abs().max().item() <= tol * ref.abs().max().item()
    m.train()  # takes the update out again
    assert not m.merged
    assert (m.weight.float() - W.float()).abs().max().item() <= 4 * tol * W.float().abs().max().item() + 1e-6


def test_unmerged_forward_matches_the_reference_module(golden):
    g = golden("lora")
    m, alpha = _module(g, torch.float32)
    m.train()
    x = _t(g["x"])
    y = m(x)  # separate LoRA branch: torch ops, no native kernel involved
    ref = _t(g["f32_y_unmerged"])
    assert (y - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    yo = oracle.lora_forward_unmerged(x, _t(g["W"]), _t(g["A"]), _t(g["B"]), alpha)
    assert (yo - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    # merged and unmerged forwards agree (the point of the merge)
    assert (_t(g["f32_y_merged"]) - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


def test_lora_context_builds_the_reference_layout_and_merges_on_eval():
    cfg = LLaMAConfig(n_layer=2, n_head=4, n_embd=64, vocab_size=100)
    plain = LLaMA(cfg)
    assert type(plain.transformer.h[0].attn.c_attn) is torch.nn.Linear
    with L.lora(r=4, alpha=16, dropout=0.05):
        model = LLaMA(cfg)
    assert llama.CausalSelfAttention is not L.CausalSelfAttention  # restored on exit
    assert type(LLaMA(cfg).transformer.h[0].attn.c_attn) is torch.nn.Linear
    with L.lora(r=4, alpha=16, dropout=0.05, enabled=False):
        assert type(LLaMA(cfg).transformer.h[0].attn.c_attn) is torch.nn.Linear
    c = model.transformer.h[1].attn.c_attn
    assert isinstance(c, L.MergedLinear) and c.enable_lora == [True, False, True]
    assert c.lora_A.shape == (8, 64) and c.lora_B.shape == (128, 4) and c.weight.shape == (192, 64)
    keys = set(model.state_dict())
    assert {"transformer.h.0.attn.c_attn.lora_A", "transformer.h.1.attn.c_attn.lora_B",
            "transformer.h.0.attn.c_attn.weight"} <= keys
    assert set(L.lora_state_dict(model)) == {k for k in keys if "lora_" in k}
    # generate/lora.py:75-83: pretrained checkpoint, then the LoRA checkpoint, both strict=False; eval() merges
    pre = plain.state_dict()
    gen = torch.Generator().manual_seed(0)
    lo = {k: torch.randn(v.shape, generator=gen) * 0.2 for k, v in L.lora_state_dict(model).items()}
    missing = model.load_state_dict(pre, strict=False)
    assert all("lora_" in k for k in missing.missing_keys) and not missing.unexpected_keys
    model.load_state_dict(lo, strict=False)
    model.eval()
    for i in range(cfg.n_layer):
        c = model.transformer.h[i].attn.c_attn
        want = oracle.lora_merge(pre[f"transformer.h.{i}.attn.c_attn.weight"], lo[f"transformer.h.{i}.attn.c_attn.lora_A"],
                                 lo[f"transformer.h.{i}.attn.c_attn.lora_B"], 16)
        assert c.merged and (c.weight - want).abs().max().item() <= 1e-6

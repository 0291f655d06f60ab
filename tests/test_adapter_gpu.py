"""generate/adapter.py:67-95 through the native kernels: lit_llama_amd.adapter.LLaMA (op by op: native linears, RMSNorm,
RoPE + KV cache + causal attention; the ten-row prefix attention as tensor ops) against the reference's own run
(tests/golden/adapter.npz) in f32 and against the oracle in bf16."""
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
    got = _teacher_forced(model, toks, T, S, dev)
    err = (got - ref_logits).abs().max().item()
    assert err <= tol * std, f"{dtype}: adapter logits off by {err:.5f} (std {std:.3f})"
    assert model.adapter_kv_caches == [] and len(model.kv_caches) == 0
    out = lit_llama_amd.generate(model, toks[:T], toks.numel() - T, top_k=1).cpu()
    margins = g["margin"]
    n = T + 1 + next((i for i, m in enumerate(margins.tolist()) if m <= 2 * tol * std), len(margins))
    assert torch.equal(out[:n].long(), toks[:n].cpu().long()), f"{out.tolist()} vs {toks.tolist()}"
    assert model.adapter_kv_caches[2] is not None and model.adapter_kv_caches[0] is None  # prefix k / v computed once


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

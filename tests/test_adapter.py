"""LLaMA-Adapter inference variant: the oracle restatement (oracle.AdapterModel) against the reference's own run
(tests/golden/adapter.npz, from /root/reference lit_llama/adapter.py by `python oracle/gen_golden.py --adapter`), and the
host-side contract of lit_llama_amd/adapter.py (state-dict layout, caches, old-checkpoint gating factors)."""
import numpy as np
import torch

from lit_llama_amd import adapter as A
from lit_llama_amd import synth
from lit_llama_amd.model import LLaMAConfig as BaseConfig
from oracle import oracle

CFG = dict(n_layer=3, n_head=4, n_embd=64, vocab_size=128, block_size=64)  # oracle/gen_golden.py ADAPTER_CFG


def adapter_state_dict(dtype=torch.float32):
    base = BaseConfig(**CFG)
    sd = synth.make_state_dict(base, seed=21, mode=None, dtype=dtype)
    sd.update(synth.make_adapter_state(base, seed=22, dtype=dtype))
    return sd


def test_oracle_adapter_model_reproduces_the_reference(golden):
    g = golden("adapter")
    T = int(g["prompt_len"])
    toks = torch.from_numpy(g["tokens"]).long()
    om = oracle.AdapterModel(oracle.Config(**CFG), adapter_state_dict())
    out = oracle.generate(om, toks[:T].int(), toks.numel() - T, top_k=1)
    assert torch.equal(out.long(), toks)
    om.reset_cache()
    logits = oracle.teacher_forced_logits(om, toks.int(), T)
    assert (logits - torch.from_numpy(g["logits"])).abs().max().item() <= 1e-4
    # without the prefix term the logits move by much more than any tolerance used downstream
    plain = oracle.Model(oracle.Config(**CFG), {k: v for k, v in adapter_state_dict().items()
                                                if "adapter_wte" not in k and "gating_factor" not in k})
    lp = oracle.teacher_forced_logits(plain, toks.int(), T)
    assert (lp - logits).abs().max().item() > 0.05 * float(logits.std(-1).mean())


def test_adapter_module_layout_and_cache_contract():
    cfg = A.LLaMAConfig(**CFG)
    assert (cfg.adapter_prompt_length, cfg.adapter_start_layer) == (10, 2)
    model = A.LLaMA(cfg)
    sd = adapter_state_dict()
    assert set(model.state_dict()) == set(sd)  # the reference's key names (adapter.py:79-86, 226-236)
    model.load_state_dict(sd)
    assert not hasattr(model.transformer.h[1].attn, "adapter_wte") and hasattr(model.transformer.h[2].attn, "adapter_wte")
    assert model.transformer.h[2].attn.gating_factor.shape == (1, 4, 1, 1)
    assert model.engine() is None and "cpu" in model._engine_failed  # (on a GPU, bf16: the native engine runs the prefix term)
    assert set(A.adapter_state_from_state_dict(sd)) == {k for k in sd if "adapter_wte" in k or "gating_factor" in k}
    A.mark_only_adapter_as_trainable(model)
    assert {n for n, p_ in model.named_parameters() if p_.requires_grad} == set(A.adapter_state_from_state_dict(sd))
    # checkpoints from before the per-head gate hold one value (adapter.py:173-183)
    old = dict(sd)
    old["transformer.h.2.attn.gating_factor"] = torch.tensor([0.25])
    model.load_state_dict(old)
    assert torch.equal(model.transformer.h[2].attn.gating_factor, torch.full((1, 4, 1, 1), 0.25))
    model.adapter_kv_caches = [None, None, (torch.zeros(1), torch.zeros(1))]
    model.reset_cache()
    assert model.adapter_kv_caches == [] and model.kv_caches == []


def adapter_v2_state_dict(dtype=torch.float32):
    sd = adapter_state_dict(dtype)
    sd.update(synth.make_adapter_v2_state(sd, seed=23, dtype=dtype))
    return sd


def test_oracle_reproduces_adapter_v2_and_module_layout(golden):
    """lit_llama/adapter_v2.py: scale * (W x + bias) on every linear, on top of the v1 prefix attention."""
    from lit_llama_amd import adapter_v2 as V2

    g = golden("adapter_v2")
    T = int(g["prompt_len"])
    toks = torch.from_numpy(g["tokens"]).long()
    sd = adapter_v2_state_dict()
    om = oracle.AdapterModel(oracle.Config(**CFG), sd)
    out = oracle.generate(om, toks[:T].int(), toks.numel() - T, top_k=1)
    assert torch.equal(out.long(), toks)
    om.reset_cache()
    logits = oracle.teacher_forced_logits(om, toks.int(), T)
    assert (logits - torch.from_numpy(g["logits"])).abs().max().item() <= 1e-4
    model = A.LLaMA(A.LLaMAConfig(**CFG))
    V2.add_adapter_v2_parameters_to_linear_layers(model)
    assert sorted(model.state_dict()) == [str(k) for k in g["state_dict_keys"]]
    lin = model.transformer.h[0].mlp.c_fc1
    assert torch.equal(lin.adapter_scale, torch.ones(lin.weight.shape[0])) and float(lin.adapter_bias.detach().abs().max()) == 0.0
    model.load_state_dict(sd)
    assert set(V2.adapter_v2_state_from_state_dict(sd)) == {k for k in sd if any(s in k for s in V2.get_adapter_substrings())}
    V2.mark_only_adapter_v2_as_trainable(model)
    assert model.transformer.ln_f.scale.requires_grad and not model.lm_head.weight.requires_grad
    # a plain model with v2 parameters must not reach the native engine silently
    from lit_llama_amd.engine import EngineUnavailable, _kind

    import pytest
    with pytest.raises(EngineUnavailable, match="Adapter v2"):
        _kind(lin)


def test_adapter_v2_new_forward_bound_to_a_plugin_linear_does_not_recurse():
    """The reference binds the function as the layer's forward (adapter_v2.py:39); on a linear with a forward of its own
    (Linear8bitLt, LoRA) going back through `mod(x)` would never return (advisor r3)."""
    import torch
    import torch.nn as nn

    from lit_llama_amd import adapter_v2 as V2

    class Doubling(nn.Linear):
        def forward(self, x):
            return 2.0 * nn.functional.linear(x, self.weight, self.bias)

    layer = Doubling(4, 3, bias=False)
    V2.adapter_v2_linear_with_bias_and_scale(layer)
    layer.adapter_scale.data.fill_(0.5)
    layer.adapter_bias.data.fill_(1.0)
    x = torch.randn(2, 4)
    want = 0.5 * (2.0 * nn.functional.linear(x, layer.weight) + 1.0)
    layer.forward = V2.adapter_v2_new_forward.__get__(layer, layer.__class__)
    assert torch.allclose(layer(x), want)

"""generate/lora.py:71-95 through the native path: a bf16 model built under `lora()`, pretrained + LoRA checkpoints loaded
with strict=False, `eval()` merges the update into c_attn — from there it is configs[1] (unquantised) decode on the native
engine.  Checked against the CPU oracle run on the merged weights."""
import pytest
import torch

import lit_llama_amd
from lit_llama_amd import lora as L
from lit_llama_amd import synth
from lit_llama_amd.model import LLaMA, LLaMAConfig
from lit_llama_amd.utils import EmptyInitOnDevice
from oracle import oracle

pytestmark = pytest.mark.gpu


def test_lora_model_merges_and_decodes_on_the_engine(dev):
    kw = dict(n_layer=2, n_head=4, n_embd=256)
    cfg = LLaMAConfig(**kw)
    sd = synth.make_state_dict(cfg, seed=5, mode=None)  # bf16-exact float weights
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16), L.lora(r=8, alpha=16, dropout=0.05):
        model = LLaMA(cfg)
    assert isinstance(model.transformer.h[0].attn.c_attn, L.MergedLinear)
    model.load_state_dict(sd, strict=False)
    gen = torch.Generator().manual_seed(3)
    lo = {k: (torch.randn(v.shape, generator=gen) * 0.05).to(torch.bfloat16) for k, v in L.lora_state_dict(model).items()}
    model.load_state_dict(lo, strict=False)
    model.train()
    assert model.engine() is None and "not merged" in model._engine_failed  # the separate LoRA branch is not a hot path
    model.eval()
    eng = model.engine()
    assert eng is not None, model._engine_failed
    # the oracle on the weights the module actually holds after the merge (bf16 values, f32 arithmetic)
    merged = {k: v.float().cpu() for k, v in model.state_dict().items() if "lora_" not in k}
    for i in range(cfg.n_layer):
        k = f"transformer.h.{i}.attn.c_attn"
        want = oracle.lora_merge(sd[k + ".weight"].to(torch.bfloat16), lo[k + ".lora_A"], lo[k + ".lora_B"], 16).float()
        assert (merged[k + ".weight"] - want).abs().max().item() <= 2.0**-7 * want.abs().max().item()
        assert not torch.equal(merged[k + ".weight"], sd[k + ".weight"].float())
    om = oracle.Model(oracle.Config(**kw), merged)
    prompt = synth.make_prompt(7)
    ref = oracle.generate(om, prompt, 8, top_k=1)
    om.reset_cache()
    ref_logits = oracle.teacher_forced_logits(om, ref, 7)
    out = lit_llama_amd.generate(model, prompt.to(dev), 8, top_k=1).cpu()
    model.reset_cache()
    rows, pos, p0, r = [], torch.arange(0, 7, device=dev), 0, ref.to(dev)
    for _ in range(8):
        pos._mi355_pos0 = p0
        rows.append(model(r.index_select(0, pos).view(1, -1), 15, pos)[0, -1].float().cpu())
        p0 += pos.numel()
        pos = pos[-1:] + 1
    got = torch.stack(rows)
    std = float(ref_logits.std(-1).mean())
    assert (got - ref_logits).abs().max().item() <= 0.05 * std
    top2 = torch.topk(ref_logits, 2, dim=-1).values
    decisive = (top2[:, 0] - top2[:, 1]) > 0.1 * std
    assert torch.equal(got.argmax(-1)[decisive], ref_logits.argmax(-1)[decisive])
    n = 7 + 1 + next((i for i, d in enumerate(decisive.tolist()) if not d), 8)
    assert torch.equal(out[:n], ref[:n])

"""Device-side sampling (csrc/sample.hip, `mi355_sample`) against the reference's sampling arithmetic
(/root/reference generate.py:68-76, restated in oracle.sample_from_uniform): kept set of the top-k threshold (ties kept,
exactly as `logits < v[-1]`), probabilities, and the inverse-CDF draw for a given uniform."""
import numpy as np
import pytest
import torch

import lit_llama_amd
from lit_llama_amd import ops, synth
from lit_llama_amd.model import LLaMA, LLaMAConfig
from lit_llama_amd.utils import EmptyInitOnDevice
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("V,temperature,top_k", [(32000, 1.0, None), (32000, 0.8, 200), (32000, 0.7, 1), (32000, 1.3, 5),
                                                 (1000, 0.5, 5000), (4099, 2.0, 37), (16, 1.0, 3)])
def test_sample_kernel_matches_reference_arithmetic(dev, V, temperature, top_k):
    gen = torch.Generator().manual_seed(V + (top_k or 0))
    logits = torch.randn(V, generator=gen) * 3
    if V > 100:
        logits[17] = logits[23] = logits[101]  # ties, possibly at the threshold
    us = [0.0, 1e-7, 0.123, 0.5, 0.77, 0.999, 0.9999999]
    d_logits = logits.to(dev)
    for u in us:
        uni = torch.full((8,), u, device=dev)
        pos = torch.tensor([3], dtype=torch.int32, device=dev)
        tok = torch.zeros(1, dtype=torch.int32, device=dev)
        out = torch.full((8,), -1, dtype=torch.int32, device=dev)
        slot = torch.zeros(1, dtype=torch.int32, device=dev)
        probs = torch.empty(V, device=dev)
        ops.sample(d_logits, temperature, top_k, uni, pos, tok, out_tokens=out, tokens=slot, advance=True, probs_out=probs)
        ref_tok, ref_p = oracle.sample_from_uniform(logits, temperature, top_k, u)
        p = probs.cpu()
        assert torch.equal(p > 0, ref_p > 0), "kept set differs"
        assert (p - ref_p).abs().max().item() <= 2e-6 + 1e-5 * ref_p.max().item()
        got = int(tok.item())
        assert int(out[4]) == got and int(slot.item()) == got and int(pos.item()) == 4
        if got != ref_tok:
            # only legitimate at a CDF boundary: both candidates' cumulative masses straddle u within f32 rounding
            cdf = torch.cumsum(ref_p.double(), 0)
            lo, hi = min(got, ref_tok), max(got, ref_tok)
            assert float(ref_p[got]) > 0 and abs(float(cdf[lo]) - u) <= 1e-5 and \
                float(ref_p[lo + 1:hi + 1].sum()) <= float(ref_p[hi]) + 1e-5, f"u={u}: {got} vs {ref_tok}"


def test_sampled_generate_is_reproducible_and_degenerates_to_greedy(dev):
    cfg = LLaMAConfig(n_layer=2, n_head=32, n_embd=4096)
    sd = synth.make_state_dict(cfg, seed=0, mode="gptq.int4")
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode="gptq.int4"):
        model = LLaMA(cfg)
    model.load_state_dict(sd)
    prompt = synth.make_prompt(12).to(dev)
    torch.manual_seed(7)
    a = lit_llama_amd.generate(model, prompt, 16, temperature=0.8, top_k=200)
    torch.manual_seed(7)
    b = lit_llama_amd.generate(model, prompt, 16, temperature=0.8, top_k=200)
    torch.manual_seed(8)
    c = lit_llama_amd.generate(model, prompt, 16, temperature=0.8, top_k=200)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert a.shape == (28,) and int(a.min()) >= 0 and int(a.max()) < cfg.padded_vocab_size
    # a cold temperature concentrates the mass on the arg-max: the sampled run follows the greedy run
    greedy = lit_llama_amd.generate(model, prompt, 8, top_k=1)
    cold = lit_llama_amd.generate(model, prompt, 8, temperature=1e-3, top_k=50)
    assert torch.equal(greedy, cold)
    # the reference-style loop (torch ops on returned logits) still runs and samples from the same kept set
    torch.manual_seed(7)
    d = lit_llama_amd.generate(model, prompt, 4, temperature=0.8, top_k=200, sample_on_device=False)
    assert d.shape == (16,)

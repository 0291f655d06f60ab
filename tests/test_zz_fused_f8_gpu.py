"""The fp8-operand path of the persistent int4 decode step: mi355_fused_step_args.weight_fmt = 3 (csrc/fused_step_ring.hip
`fused_step_ring_kernel<false, 3>`, what the engine selects for per-row int4 models since round 4): the int4 streams of weight_fmt 0
through fp8 operands — one v_mfma_scale_f32_16x16x128_f8f6f4 per 1-KiB piece, activations published as three E4M3 limbs under 16-bit
tags.  Every other GPU test of an int4 model runs the default path, i.e. this one; here BOTH operand paths run on one engine
(`fused.weight_fmt` toggled, hand-off workspace zeroed in between) against the launch-per-operator step and the oracle.
Primitives: scripts/micro/mx_fp8.hip (int4 bytes as E4M3 subnormals, per-lane block scales, the limb split, 15.2 against 35.5 ns of
matrix pipe per piece); first GPU run of this file and of the full-depth golden test under the path: profiles/r04_f8_operands_tests.txt.
MI355_TEST_F8=0 skips the file.

Reference path: /root/reference generate.py:63-91 -> lit_llama/model.py:76-122 for one token at a time (as tests/test_fused_step_gpu.py).
Bars = those of the fp16-operand step: within 0.03 logit-std of the launch-per-operator step on the same weights, greedy tokens equal
up to the first near tie, within 0.05 std of the oracle, bit-reproducible, no abort, no clipped granule.
"""
import os

import pytest
import torch

import lit_llama_amd
from lit_llama_amd import synth
from lit_llama_amd.model import LLaMA, LLaMAConfig
from lit_llama_amd.utils import EmptyInitOnDevice
from oracle import oracle

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("MI355_TEST_F8", "1") == "0", reason="MI355_TEST_F8=0")]

W7B = dict(n_head=32, n_embd=4096)


def build(n_layer, dev, seed=0):
    cfg = LLaMAConfig(n_layer=n_layer, **W7B)
    sd = synth.make_state_dict(cfg, seed=seed, mode="gptq.int4")
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode="gptq.int4"):
        model = LLaMA(cfg)
    model.load_state_dict(sd)
    model.eval()
    eng = model.engine()
    assert eng is not None, model._engine_failed
    if eng.fused is None:
        pytest.skip("fused decode step not available on this device (needs 256 CUs)")
    assert eng.fused.weight_fmt in (0, 3)
    return model, sd, cfg, eng


def set_fmt(eng, fmt):
    """Switch the operand path of a live engine; hand-off granules of the other tag width must not be mistaken for fresh ones."""
    torch.cuda.synchronize()
    eng.use_fused_format(fmt)
    torch.cuda.synchronize()


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


@torch.no_grad()
def test_f8_operand_step_matches_the_launch_path_and_the_fp16_operand_step(dev):
    model, _, cfg, eng = build(2, dev)
    prompt = synth.make_prompt(20).to(dev)
    outs, logits = {}, {}
    try:
        for key, fused, fmt in (("launch", False, 0), ("f16", True, 0), ("f8", True, 3)):
            eng.fused_enabled = fused
            if fused:
                set_fmt(eng, fmt)
            model.reset_cache()
            outs[key] = lit_llama_amd.generate(model, prompt, 24, top_k=1, max_seq_length=64).cpu()
            logits[key] = teacher_forced(model, outs["launch"].to(dev), 20, 64, dev)
            eng.check_status()
            assert eng.fused_clipped == 0
    finally:
        eng.fused_enabled = True
        set_fmt(eng, 0)
    std = float(logits["launch"].std(-1).mean())
    assert torch.isfinite(logits["f8"]).all()
    err = (logits["f8"] - logits["launch"]).abs().max().item()
    err16 = (logits["f16"] - logits["launch"]).abs().max().item()
    print(f"weight_fmt 3 vs launch path {err / std:.4f} std (weight_fmt 0: {err16 / std:.4f}); 3 vs 0: "
          f"{(logits['f8'] - logits['f16']).abs().max().item() / std:.4f}")
    assert err <= 0.03 * std, f"fp8-operand step vs launch path: {err:.4f} (std {std:.3f})"
    top2 = torch.topk(logits["launch"], 2, dim=-1).values
    margins = (top2[:, 0] - top2[:, 1]).tolist()
    first_tie = next((i for i, m_ in enumerate(margins) if m_ <= 2 * 0.03 * std), len(margins))
    n = 20 + first_tie + 1
    assert torch.equal(outs["f8"][:n], outs["launch"][:n]), f"{outs['f8'].tolist()}\n{outs['launch'].tolist()}"


@torch.no_grad()
def test_f8_operand_step_against_the_oracle_at_7b_width(dev):
    model, sd, cfg, eng = build(1, dev)
    prompt = synth.make_prompt(5)
    om = oracle.Model(oracle.Config(n_layer=1, **W7B), {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()},
                      mode="gptq.int4")
    toks = oracle.generate(om, prompt, 4, top_k=1)
    om.reset_cache()
    ref = oracle.teacher_forced_logits(om, toks, 5)
    try:
        set_fmt(eng, 3)
        got = teacher_forced(model, toks.to(dev), 5, 16, dev)  # row 0 = prefill (launch path), rows 1.. = fused steps
        eng.check_status()
        got2 = teacher_forced(model, toks.to(dev), 5, 16, dev)
        eng.check_status()
    finally:
        set_fmt(eng, 0)
    assert torch.equal(got, got2), "the fp8-operand step is not bit-reproducible"
    std = float(ref.std(-1).mean())
    err = (got - ref).abs().max().item()
    print(f"weight_fmt 3 vs oracle: {err / std:.4f} std")
    assert err <= 0.05 * std, f"fp8-operand 7B-width logits off by {err:.4f} (std {std:.3f})"
    top2 = torch.topk(ref, 2, dim=-1).values
    decisive = (top2[:, 0] - top2[:, 1]) > 0.1 * std
    assert torch.equal(got.argmax(-1)[decisive], ref.argmax(-1)[decisive])


@torch.no_grad()
@pytest.mark.parametrize("T", [257, 700])
def test_f8_operand_step_over_several_cache_blocks_and_the_row_split(dev, T):
    """Positions past one 256-row cache block and past the row-split threshold (384): the attention phase is the fp16 step's, its
    output edge is not."""
    model, _, cfg, eng = build(2, dev, seed=3)
    prompt = synth.make_prompt(T, seed=11).to(dev)
    res = {}
    try:
        eng.fused_enabled = False
        model.reset_cache()
        toks = lit_llama_amd.generate(model, prompt, 6, top_k=1, max_seq_length=T + 8)
        res["launch"] = teacher_forced(model, toks, T, T + 8, dev)
        eng.fused_enabled = True
        set_fmt(eng, 3)
        res["f8"] = teacher_forced(model, toks, T, T + 8, dev)
        eng.check_status()
        assert eng.fused_clipped == 0
    finally:
        eng.fused_enabled = True
        set_fmt(eng, 0)
    std = float(res["launch"].std(-1).mean())
    err = (res["f8"] - res["launch"]).abs().max().item()
    print(f"T = {T}: weight_fmt 3 vs launch path {err / std:.4f} std")
    assert err <= 0.03 * std

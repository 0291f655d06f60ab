"""BASELINE.json configs[2] at FULL depth against the reference itself: LLaMA-7B (32 layers, n_embd 4096), gptq.int4,
seeded synthetic weights (lit_llama_amd/synth.py), prompt of 8, six greedy tokens.  tests/golden/cfg2_7b_int4.npz
holds what the UNMODIFIED /root/reference produced on the CPU for exactly these weights (generate.py:63-91 with
top_k = 1, then teacher-forced logits; oracle/gen_golden.py --big, which also pins oracle/oracle.py to it with
max |dlogit| = 0).  Here the same checkpoint goes through the engine: prefill on the launch path, decode steps on
the fused persistent step.

A second fixture of the same checkpoint, cfg2_7b_int4_long.npz (--big-long), has a 128-token prompt (the wide path of the
engine at full depth: GEMM + flash attention) and 16 decode steps starting at position 128.  A third, cfg2_7b_int4_p400.npz
(--big-p400, round 4), has a 400-token prompt: its 16 decode steps run at positions 400..415, past the fused step's row-split
threshold (position 384) — the regime of the reference's own usage (generate.py:94-155 with real prompts) at FULL depth.

The bar is CALIBRATED on the reference itself: tests/golden/cfg2_7b_int4*_bf16ref.npz (oracle/gen_golden.py --big-bf16)
hold the reference's OWN bf16 run (parameters, scales and activations in bf16, what `--precision bf16-true` makes of
generate.py:123-134) on the same tokens: it sits 0.052-0.086 logit-std (short fixture) and 0.071-0.122 (long) from the
reference's f32 run.  The only tolerance the reference states for "bf16 run vs f32 run" is atol 5e-3 + rtol 1e-3
(tests/test_model.py:133, one block of a model with its init scale); at full depth its own bf16 path is 0.05-0.12 std
away, so that number cannot be the bar for 32 layers.  Bar here (bf16 operands, bf16 KV cache, f32 residual stream vs the
reference's f32 CPU arithmetic): teacher-forced logits on the probe columns within HALF the reference's own bf16 distance
(measured: 0.0235 std on the short fixture, i.e. 27 % of it) and within 0.03 std on the short fixture as a regression
bar; argmax equal wherever the reference's top-2 margin exceeds twice the tolerance; free-running greedy tokens equal up
to the first step inside that margin.
Rebuilding the 3.6 GB checkpoint from its seed takes a minute or two of host time (once for both fixtures).
"""
import numpy as np
import pytest
import torch

import lit_llama_amd
from lit_llama_amd import synth
from lit_llama_amd.model import LLaMA, LLaMAConfig
from lit_llama_amd.utils import EmptyInitOnDevice

pytestmark = pytest.mark.gpu

PROBES = (np.arange(64) * (32000 // 64) + 7) % 32000


# absolute regression bars in logit-std on top of the calibrated one (half the reference's own bf16 distance is
# 0.157 std on the 400-token fixture against a measured 0.025 — a 6x regression would have passed).  Measured in round 4: 0.016 / 0.017 /
# 0.025 / (s1: first GPU run in the driver's suite, no number recorded); the launch-per-operator path 0.024 / 0.024 / 0.025.
ABS_BAR = {"cfg2_7b_int4": 0.03, "cfg2_7b_int4_long": 0.04, "cfg2_7b_int4_p400": 0.04, "cfg2_7b_int4_s1": 0.04, "cfg2_7b_int4_real": 0.07}
# The LLaMA-statistics fixture (round 5) is judged against the reference's own bf16 run AS IT IS (0.0703 std), not against half of it:
# with massive activations next to a median of 10^-3, the 8 significant bits of a bf16 operand cost 0.04-0.054 std at full depth even
# when everything else is exact (oracle/sim_operand_arith.py, `bf16:exact`: the oracle with the inputs of its linears rounded to
# bf16; `f16:exact` 0.004) — that IS the reference's GPU arithmetic (bf16 operands, quantization.py:187-333), and the prompt pass and
# the launch-per-operator step stage their operands in bf16 like it.  On top of the absolute bar the fused rungs may not be worse than
# the launch-per-operator rung by more than 0.015 std (they were, by 0.058, while their fp16 operands still carried a +1024 offset).
REL_BAR = {"cfg2_7b_int4_real": 1.0}
# ... and the DECODE steps of the fused rungs (rows 1.. of the teacher-forced run: the persistent step's own arithmetic on top of the bf16
# prompt pass's cache rows) keep the 0.04 of the other fixtures: measured 0.0355 (fp8 limbs, clipped steps on fp16 operands) / 0.0351 (fp16);
# the CPU model puts the bf16 K / V rows alone at 0.027-0.034 there (profiles/r05_llama_statistics_operand_arithmetic.txt section 4).
DECODE_BAR = {"cfg2_7b_int4_real": 0.04}
FUSED_OVER_LAUNCH = 0.015


@torch.no_grad()
def test_full_depth_7b_int4_against_the_reference_golden_run(dev, golden, record):
    _int4_checkpoint_against_its_fixtures(dev, golden, ("cfg2_7b_int4", "cfg2_7b_int4_long", "cfg2_7b_int4_p400"), record)


def _teacher_forced(model, eng, toks, T, S, dev, on_step=None):
    model.reset_cache()
    rows = []
    input_pos = torch.arange(0, T, device=dev)
    pos0 = 0
    for i in range(toks.numel() - T):
        x = toks.index_select(0, input_pos).view(1, -1)
        input_pos._mi355_pos0 = pos0
        rows.append(model(x, S, input_pos)[0, -1].float().cpu())
        if on_step is not None:
            on_step(i, pos0 + input_pos.numel() - 1)
        pos0 += input_pos.numel()
        input_pos = input_pos[-1:] + 1
    return torch.stack(rows)


def _int4_checkpoint_against_its_fixtures(dev, golden, fixtures, record=None):
    """Every fixture of ONE seeded checkpoint through the three rungs of the engine's ladder: the persistent step with fp8-limb
    operands (weight_fmt 3, the default), with fp16 operands (weight_fmt 0) and the launch-per-operator step."""
    import warnings

    g0 = golden(fixtures[0])
    stats = str(g0["stats"]) if "stats" in g0 else "unit"
    cfg = LLaMAConfig.from_name("7B")
    torch.set_num_threads(max(torch.get_num_threads(), 16))
    sd = synth.make_state_dict(cfg, seed=int(g0["seed"]), mode="gptq.int4", stats=stats)
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode="gptq.int4"):
        model = LLaMA(cfg)
    model.load_state_dict(sd)
    del sd
    model.eval()
    eng = model.engine()
    assert eng is not None, model._engine_failed
    failures = []
    rungs = [("launch", None)]
    if eng.fused is not None:
        top = eng._fused_top_fmt
        rungs = [(f"fused/fmt{top}", top)] + ([("fused/fmt0", 0)] if top != 0 else []) + rungs
    for name in fixtures:
        g, ref_bf16 = golden(name), golden(name + "_bf16ref")
        assert int(g["seed"]) == int(g0["seed"])
        T, S = int(g["prompt_len"]), int(g["max_seq_length"])
        toks = torch.from_numpy(g["tokens"]).to(dev)
        std = float(g["std"].mean())
        ref_dist = float(ref_bf16["max_dist_std"])  # the reference's own bf16 run vs its f32 run, in logit std
        assert 0.02 < ref_dist < 0.5
        tol = min(REL_BAR.get(name, 0.5) * ref_dist, ABS_BAR[name]) * std
        measured = {}
        def set_rung(fmt):
            eng.reset_fused_format()
            if fmt is None:
                eng.fused_enabled = False
            else:
                eng.use_fused_format(fmt)  # (zeroes the hand-off workspace when the tag width of the granules changes)

        for label, fmt in rungs:
            set_rung(fmt)
            # teacher-forced on the reference's tokens.  A step whose activations leave the range of the hand-off format is
            # recomputed one rung down by LLaMA.forward (engine.check_status); on the unit-statistics fixtures that must not happen,
            # on the LLaMA-statistics one (test_zz_golden_7b_real_gpu.py) the engine is put back on the rung under test after
            # every such step, so that every OTHER step is still measured there
            demoted, seen = [], [len(eng.fused_demotions)]

            def on_step(i, pos, fmt=fmt, demoted=demoted, seen=seen):
                if len(eng.fused_demotions) > seen[0]:
                    demoted.append((i, pos, eng.fused_demotions[-1][1]))
                    set_rung(fmt)
                seen[0] = len(eng.fused_demotions)

            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                logits = _teacher_forced(model, eng, toks, T, S, dev, on_step)
            assert eng.check_status() is None
            if stats == "unit":
                assert eng.fused_clipped == 0 and not demoted, (label, demoted)
            per_step = np.abs(logits[:, PROBES].numpy() - g["probes"]).max(axis=1) / std
            err = float(per_step.max()) * std
            if record is not None:
                record("full_depth_7b_int4", fixture=name, path=label, max_dist_std=round(err / std, 5),
                       per_step_dist_std=[round(float(v), 5) for v in per_step], ref_bf16_dist_std=round(ref_dist, 5),
                       bar_std=round(tol / std, 5), recomputed_steps=[[i, p, to] for i, p, to in demoted],
                       clipped_pairs=int(eng.fused_clipped))
            measured[label] = err / std
            if fmt is not None and name in DECODE_BAR and float(per_step[1:].max()) > DECODE_BAR[name]:
                # (the persistent step's own work — every row behind the prompt pass — against the absolute bar of the other fixtures)
                failures.append(f"{name} {label}: decode steps off by {float(per_step[1:].max()):.4f} std (bar {DECODE_BAR[name]} std)")
            if err > tol:  # (collected: every rung's numbers are recorded before the test fails)
                failures.append(f"{name} {label}: 7B logits off by {err:.4f} = {err / std:.4f} std (tol {tol / std:.4f} std)")
                continue
            assert np.abs(logits.std(-1).numpy() - g["std"]).max() <= 0.02 * std
            decisive = g["margin"] > 2 * tol
            assert np.array_equal(logits.argmax(-1).numpy()[decisive], g["argmax"][decisive])
            if "oracle_swiglu_absmax" in g and fmt == 3:
                # the steps that HAD to be recomputed are the ones whose SwiGLU output (position of the step's token) passes the fp8
                # hand-off's +-7168 — known from the oracle's activations (2 % guard band around the limit)
                a = g["oracle_swiglu_absmax"]
                must = {i for i in range(1, logits.shape[0]) if a[T + i - 1] > 7168 * 1.02}
                may = {i for i in range(1, logits.shape[0]) if a[T + i - 1] > 7168 * 0.98}
                got_steps = {i for i, _, _ in demoted}
                assert must <= got_steps <= may, (sorted(must), sorted(got_steps), sorted(may))
                # (one rung down is enough unless a value also passes fp16's +-65504 — 3.3 sigma of the generator's units)
                assert all(to.startswith("fp16") or a[p] > 60000 for _, p, to in demoted), demoted
                assert len(must) >= 1, "the LLaMA-statistics fixture no longer exercises the fp8 hand-off's range limit"
            # free running (generate.py:63-91) on the product's own ladder (per-step demotion with hysteresis, replay from the clipped position):
            # equal up to the first near tie
            eng.fused_demotions.clear()
            eng.fused_clipped = 0
            model.reset_cache()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                out = lit_llama_amd.generate(model, toks[:T], toks.numel() - T, top_k=1).cpu()
            assert eng.check_status() is None
            first_tie = next((i for i, m_ in enumerate(g["margin"]) if m_ <= 2 * tol), len(g["margin"]))
            n = T + first_tie
            assert torch.equal(out[:n], torch.from_numpy(g["tokens"])[:n]), f"{name} {label}: {out.tolist()} vs {g['tokens'].tolist()}"
            if stats == "unit":
                assert not eng.fused_demotions
            print(f"{name} {label}: max |dlogit| {err:.4f} = {err / std:.4f} std (reference bf16 vs f32: {ref_dist:.4f} std); "
                  f"min margin {g['margin'].min():.3f}; recomputed steps {demoted}; generate() demotions {eng.fused_demotions}")
            eng.fused_demotions.clear()
            eng.fused_clipped = 0
        for label, v in measured.items():
            if label != "launch" and v > measured["launch"] + FUSED_OVER_LAUNCH:
                failures.append(f"{name} {label}: {v:.4f} std, the launch-per-operator rung {measured['launch']:.4f}")
    eng.reset_fused_format()
    assert not failures, failures


@torch.no_grad()
def test_full_depth_7b_unquantised_against_the_reference_golden_run(dev, golden):
    """BASELINE.json configs[1] at FULL depth against the reference itself (round 4): LLaMA-7B without quantisation, seeded synthetic
    weights (every value a bf16 number), prompt of 8, six greedy tokens.  tests/golden/cfg1_7b_none.npz holds what the UNMODIFIED
    /root/reference produced in f32 on the CPU (oracle/gen_golden.py --big-none, oracle == reference with max |dlogit| ~ 0), its
    `_bf16ref` twin the reference's own bf16 run on the same tokens.  Here the bf16 model goes through the engine: prefill on the wide path,
    decode steps on the persistent step over BF16 streams (and on the launch-per-operator step).  Bar: within HALF the reference's own bf16
    distance, argmax wherever the reference's margin is decisive, greedy tokens up to the first near tie."""
    g, ref_bf16 = golden("cfg1_7b_none"), golden("cfg1_7b_none_bf16ref")
    cfg = LLaMAConfig.from_name("7B")
    torch.set_num_threads(max(torch.get_num_threads(), 16))
    sd = synth.make_state_dict(cfg, seed=int(g["seed"]), mode=None, dtype=torch.bfloat16)
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16):
        model = LLaMA(cfg)
    model.load_state_dict(sd)
    del sd
    model.eval()
    eng = model.engine()
    assert eng is not None, model._engine_failed
    T, S = int(g["prompt_len"]), int(g["max_seq_length"])
    toks = torch.from_numpy(g["tokens"]).to(dev)
    std = float(g["std"].mean())
    ref_dist = float(ref_bf16["max_dist_std"])
    assert 0.005 < ref_dist < 0.5
    tol = 0.5 * ref_dist * std
    for fused in ([True, False] if eng.fused is not None else [False]):
        eng.fused_enabled = fused
        model.reset_cache()
        rows = []
        input_pos = torch.arange(0, T, device=dev)
        pos0 = 0
        for _ in range(toks.numel() - T):
            x = toks.index_select(0, input_pos).view(1, -1)
            input_pos._mi355_pos0 = pos0
            rows.append(model(x, S, input_pos)[0, -1].float().cpu())
            pos0 += input_pos.numel()
            input_pos = input_pos[-1:] + 1
        logits = torch.stack(rows)
        assert eng.check_status() is None
        err = np.abs(logits[:, PROBES].numpy() - g["probes"]).max()
        assert err <= tol, f"fused={fused}: 7B bf16 logits off by {err:.4f} (std {std:.3f}, tol {tol:.4f})"
        decisive = g["margin"] > 2 * tol
        assert np.array_equal(logits.argmax(-1).numpy()[decisive], g["argmax"][decisive])
        model.reset_cache()
        out = lit_llama_amd.generate(model, toks[:T], toks.numel() - T, top_k=1).cpu()
        first_tie = next((i for i, m_ in enumerate(g["margin"]) if m_ <= 2 * tol), len(g["margin"]))
        n = T + first_tie
        assert torch.equal(out[:n], torch.from_numpy(g["tokens"])[:n]), f"fused={fused}: {out.tolist()} vs {g['tokens'].tolist()}"
        print(f"cfg1_7b_none fused={fused}: max |dlogit| {err:.4f} = {err / std:.4f} std (reference bf16 vs f32: {ref_dist:.4f} std); "
              f"min margin {g['margin'].min():.3f}")
    eng.fused_enabled = True

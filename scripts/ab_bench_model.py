"""Same box, same process, same kernel: does the persistent step's rate depend on the VALUES it streams?  The bench model's zero point
(synth.fill_model_random_int4) moved from 8 to 7.5 in round 5 — zero-mean weights instead of a common-mode gain of -15 per linear
that made the residual stream one growing constant vector.  Bytes, instructions and launch are identical; what could differ is
switching activity (power -> clocks).  Alternates the two fills on ONE model, three blocks of 64 chained greedy steps each
(positions 136..328), two rounds.
    python scripts/ab_bench_model.py
"""
import sys
import warnings
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from lit_llama_amd import synth  # noqa: E402
from lit_llama_amd.model import LLaMA, LLaMAConfig  # noqa: E402
from lit_llama_amd.utils import EmptyInitOnDevice  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = LLaMAConfig.from_name("7B")
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode="gptq.int4"):
        model = LLaMA(cfg)
    model.eval()
    prompt = synth.make_prompt(128).to(dev)
    blocks = 3
    S = 128 + 8 + 64 * blocks + 80
    for rnd in (1, 2):
        for zero in (8.0, 7.5):
            model._drop_engine()
            synth.fill_model_random_int4(model, seed=0, zero=zero, gain=2.2)
            eng = model.engine()
            assert eng is not None and eng.fused is not None, model._engine_failed
            with warnings.catch_warnings(record=True) as wl:
                warnings.simplefilter("always")
                with torch.cuda.stream(eng.stream):
                    model.reset_cache()
                    eng._ensure_cache(S)
                    eng.prefill(prompt, 0, all_logits=False, argmax=True)
                    eng.set_step(None, 1, 128, from_next=True)
                    eng.embed_step()
                    for _ in range(8):
                        eng.run_step(3)
                    evs = [torch.cuda.Event(enable_timing=True) for _ in range(blocks + 1)]
                    evs[0].record(eng.stream)
                    for b in range(blocks):
                        for _ in range(64):
                            eng.run_step(3)
                        evs[b + 1].record(eng.stream)
                evs[-1].synchronize()
                bad = eng.check_status()
            us = [round(evs[b].elapsed_time(evs[b + 1]) / 64 * 1e3, 1) for b in range(blocks)]
            print("AB", {"round": rnd, "zero": zero, "weight_fmt": int(eng.fused.weight_fmt), "us_per_step": us,
                         "tok_s_first_block": round(1e6 / us[0], 1), "clipped_from": bad, "clipped_pairs": int(eng.fused_clipped),
                         "warnings": len(wl)}, flush=True)


if __name__ == "__main__":
    main()

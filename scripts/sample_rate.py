#!/usr/bin/env python
"""Rate of the reference-style sampling loop (generate.py defaults: temperature 0.8, top_k 200) and of the greedy
fast path through `lit_llama_amd.generate` on a synthetic 7B gptq.int4 model (includes the prompt, like
generate.py:146-153)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import lit_llama_amd  # noqa: E402
from bench import build_model  # noqa: E402


class A:
    model, quantize, tune = "7B", "gptq.int4", None


def main():
    dev = torch.device("cuda:0")
    model, cfg = build_model(A, dev)
    from lit_llama_amd import synth

    prompt = synth.make_prompt(16, vocab=cfg.vocab_size, seed=1).to(dev)
    for name, kw in [("greedy top_k=1", dict(top_k=1)), ("sampled T=0.8 top_k=200", dict(temperature=0.8, top_k=200))]:
        for rep in range(2):
            model.reset_cache()
            torch.manual_seed(1234)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            y = lit_llama_amd.generate(model, prompt, 128, **kw)
            torch.cuda.synchronize()
            t = time.perf_counter() - t0
        print(f"{name:26s}: {(y.numel() - prompt.numel()) / t:8.1f} tokens/s incl. prompt ({t * 1e3:.1f} ms for 128 new tokens)")


def kernel_time():
    """Average duration of one mi355_sample launch (V = 32000, temperature 0.8, top_k 200 / none), back to back."""
    from lit_llama_amd import ops

    dev = torch.device("cuda:0")
    logits = (torch.randn(32000, device=dev) * 3).contiguous()
    uni = torch.rand(64, device=dev)
    pos = torch.tensor([3], dtype=torch.int32, device=dev)
    tok = torch.zeros(1, dtype=torch.int32, device=dev)
    for top_k in (200, None, 1):
        for _ in range(3):
            ops.sample(logits, 0.8, top_k, uni, pos, tok)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            ops.sample(logits, 0.8, top_k, uni, pos, tok)
        e1.record()
        torch.cuda.synchronize()
        print(f"mi355_sample V=32000 top_k={top_k}: {e0.elapsed_time(e1) * 1e3 / 200:.1f} us per launch (back to back)")


def loop_times():
    """Where does a sampled token's time go?  Enqueue (host) and device time of N decode steps, alone and with the
    sampling launch after each."""
    from lit_llama_amd import ops, synth

    dev = torch.device("cuda:0")
    model, cfg = build_model(A, dev)
    eng = model.engine()
    prompt = synth.make_prompt(16, vocab=cfg.vocab_size, seed=1).to(dev)
    lit_llama_amd.generate(model, prompt, 4, top_k=1, max_seq_length=512)
    uni = torch.rand(1024, device=dev)
    row = eng.logits[0, : eng.m.lm_head.N]
    n = 128
    for name in ("step(mode 0)", "step(mode 0) + sample", "step(mode 3, chained)"):
        with torch.cuda.stream(eng.stream):
            eng.set_step(prompt[-1:], 1, 16)
            if "chained" in name:
                eng.embed_step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(n):
                eng.run_step(3 if "chained" in name else 0)
                if "sample" in name:
                    ops.sample(row, 0.8, 200, uni, eng.pos, eng.next_token, out_tokens=eng.out_tokens, tokens=eng.tokens,
                               advance=True)
            e1.record()
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            eng.check_status()
        print(f"{name:24s}: device {e0.elapsed_time(e1) * 1e3 / n:7.1f} us / token, host enqueue {t_host * 1e6 / n:6.1f} us / token")


if __name__ == "__main__":
    kernel_time()
    loop_times()
    main()

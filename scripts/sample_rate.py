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


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Launch time of the int4 streaming kernel with one (scale, zero) pair per row vs per row and group of 128 columns,
7B decode shapes, M = 1, weights rotated through 8 copies (no Infinity-Cache reuse).  Prints us per launch and the
achieved stream rate (weight bytes + scale tables)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from lit_llama_amd import _native as nat  # noqa: E402
from lit_llama_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(0)
    shapes = [("attn.c_attn", 12288, 4096, 2), ("attn.c_proj", 4096, 4096, 1), ("mlp.c_proj", 4096, 11008, 1),
              ("lm_head", 32000, 4096, 2)]
    for name, N, K, R in shapes:
        x = torch.randn((1, K), device=dev, generator=gen).to(torch.bfloat16)
        nbytes = ops.packed_bytes(nat.W_Q4, N, K, R, False)
        streams = [torch.randint(0, 255, (nbytes,), device=dev, dtype=torch.uint8, generator=gen) for _ in range(8)]
        for g in (0, 128):
            ng = 1 if g == 0 else K // g
            sc = (torch.rand((N * ng,), device=dev, generator=gen) * 0.01 + 0.005).to(torch.bfloat16)
            ze = torch.randint(0, 16, (N * ng,), device=dev, generator=gen).to(torch.bfloat16)
            y = torch.empty((1, N), device=dev, dtype=torch.bfloat16)
            for i in range(8):
                ops.linear_fast(x, streams[i], nat.W_Q4, R, N, K, scales=sc, zeros=ze, out=y, group_cols=g)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 400
            e0.record()
            for i in range(n):
                ops.linear_fast(x, streams[i & 7], nat.W_Q4, R, N, K, scales=sc, zeros=ze, out=y, group_cols=g)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            total = nbytes + 4 * N * ng
            print(f"{name:12s} N={N:6d} K={K:6d} group={g or 'row':>4}: {us:7.2f} us / launch (back to back), "
                  f"{total / us / 1e3:7.1f} GB/s of {total / 1e6:6.2f} MB")


if __name__ == "__main__":
    main()

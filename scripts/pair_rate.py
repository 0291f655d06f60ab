#!/usr/bin/env python
"""Launch time of the c_fc1 / c_fc2 SwiGLU pair (int4 streaming kernel, M = 1, fused RMSNorm) for the 7B / 13B / 65B
widths, weights rotated through 6 copies, timed with the dispatch's own timestamps."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from lit_llama_amd import _native as nat  # noqa: E402
from lit_llama_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(0)
    for name, N, K in (("7B", 11008, 4096), ("13B", 13824, 5120), ("65B", 22016, 8192), ("65B/8", 2752, 8192)):
        x = torch.randn((1, K), device=dev, generator=gen)
        ns = torch.ones(K, device=dev, dtype=torch.bfloat16)
        nbytes = ops.packed_bytes(nat.W_Q4, N, K, 2, True)
        streams = [torch.randint(0, 255, (nbytes,), device=dev, dtype=torch.uint8, generator=gen) for _ in range(6)]
        sc = (torch.rand((N,), device=dev, generator=gen) * 0.01 + 0.005).to(torch.bfloat16)
        ze = torch.randint(0, 16, (N,), device=dev, generator=gen).to(torch.bfloat16)
        y = torch.empty((1, N), device=dev, dtype=torch.bfloat16)
        kw = dict(scales=sc, zeros=ze, scales2=sc, zeros2=ze, norm_scale=ns, epi=nat.EPI_SWIGLU, out=y)
        for i in range(6):
            ops.linear_fast(x, streams[i], nat.W_Q4, 2, N, K, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot, n = 0.0, 60
        for i in range(n):
            e0.record()
            ops.linear_fast(x, streams[i % 6], nat.W_Q4, 2, N, K, **kw)
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        print(f"{name:6s} pair N={N:6d} K={K:5d}: {tot * 1e3 / n:7.2f} us per launch (events around one launch), {nbytes / 1e6:6.1f} MB")


if __name__ == "__main__":
    main()

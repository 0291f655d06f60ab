"""Timing of the wide int4 linear (mi355_linear_gemm) on the 7B prefill shapes, T = 2048: TFLOP/s vs the 2.5 PFLOP/s
dense bf16 MFMA peak.    python scripts/bench_gemm.py [--M 2048] [--group 128]"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from lit_llama_amd import _native as nat  # noqa: E402
from lit_llama_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=2048)
    ap.add_argument("--group", type=int, default=0, help="grouped scales: input columns per (scale, zero) pair")
    ap.add_argument("--fmt", default="q4", choices=["q4", "bf16"], help="bf16: the unquantised stream (BASELINE configs[1])")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M = a.M
    total = 0.0
    for name, N, K, R, epi in [("c_attn", 12288, 4096, 1, nat.EPI_STORE), ("c_proj", 4096, 4096, 1, nat.EPI_ACCUM),
                               ("fc pair", 11008, 4096, 2, nat.EPI_SWIGLU), ("mlp.c_proj", 4096, 11008, 1, nat.EPI_ACCUM)]:
        fmt = nat.W_BF16 if a.fmt == "bf16" else nat.W_Q4
        nbytes = ops.packed_bytes(fmt, N, K, R, R == 2)
        if a.fmt == "bf16":  # finite values (random bytes would hold NaNs); the layout does not matter to the clock
            stream = (torch.randn(nbytes // 2, device=dev) * 0.02).to(torch.bfloat16).view(torch.uint8)
        else:
            stream = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device=dev)
        ng = -(-K // a.group) if a.group else 1
        sc = (torch.rand(N * ng, device=dev) * 0.01 + 0.005).to(torch.bfloat16)
        z = torch.full((N * ng,), 8.0, device=dev, dtype=torch.bfloat16)
        norm = R == 2 or name == "c_attn"
        x = torch.randn((M, K), device=dev, dtype=torch.float32 if norm else torch.bfloat16)
        g = torch.ones(K, device=dev, dtype=torch.bfloat16) if norm else None
        out = torch.zeros((M, N), device=dev, dtype=torch.bfloat16 if R == 2 else torch.float32)
        kw = dict(scales=sc, zeros=z, norm_scale=g, epi=epi, out=out, group_cols=a.group, fmt=fmt)
        if R == 2:
            kw.update(scales2=sc, zeros2=z)
        for _ in range(3):
            ops.linear_gemm(x, stream, R, N, K, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 10
        for _ in range(n):
            ops.linear_gemm(x, stream, R, N, K, **kw)
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        flops = 2.0 * M * N * K * (2 if R == 2 else 1)
        total += us
        print(f"{name:12s} M={M} N={N} K={K}: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s ({flops / us / 1e6 / 2500 * 100:4.1f} % of 2.5 PF)")
    print(f"one layer's linears: {total:.1f} us -> 32 layers {32 * total / 1e3:.1f} ms")


if __name__ == "__main__":
    main()

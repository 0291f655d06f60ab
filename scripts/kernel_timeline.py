#!/usr/bin/env python
"""In-kernel phase timeline of the weight-streaming linear (wall-clock stamps written by wave 0 of every workgroup,
see mi355_linear_args.debug_stamps).  Prints, per 7B shape, min / median / max over workgroups of each stamp
relative to the first workgroup's entry, in microseconds."""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from lit_llama_amd import _native as nat  # noqa: E402
from lit_llama_amd import ops  # noqa: E402
from scripts.sweep_gemv import SHAPES_7B  # noqa: E402

NAMES = ["entry", "ring issued", "x staged", "tile0 loop", "tile0 epilogue", "exit"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--waves", type=int, default=8)
    ap.add_argument("--prefetch", type=int, default=4)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--group", type=int, default=0, help="grouped scales: input columns per (scale, zero) pair")
    ap.add_argument("--shapes", default=",".join(SHAPES_7B))
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(0)
    for name in args.shapes.split(","):
        N, K, R, epi = SHAPES_7B[name]
        pair = epi == nat.EPI_SWIGLU
        nbytes = ops.packed_bytes(nat.W_Q4, N, K, R, pair)
        n_buf = max(2, int(600e6 // nbytes) + 1)
        streams = [torch.randint(0, 256, (nbytes,), generator=gen, device=dev, dtype=torch.uint8) for _ in range(n_buf)]
        ng = -(-K // args.group) if args.group else 1
        sc = (0.005 + 0.005 * torch.rand(N * ng, generator=gen, device=dev)).to(torch.bfloat16)
        ze = torch.full((N * ng,), 8.0, device=dev, dtype=torch.bfloat16)
        x = torch.randn((1, K), generator=gen, device=dev).to(torch.bfloat16 if epi == nat.EPI_ACCUM else torch.float32)
        norm = None if epi == nat.EPI_ACCUM else (1 + 0.1 * torch.randn(K, generator=gen, device=dev)).to(torch.bfloat16)
        out = torch.zeros((1, N), device=dev, dtype=torch.float32 if epi != nat.EPI_SWIGLU else torch.bfloat16)
        n_tiles = (N + (16 if pair else 16 * R) - 1) // (16 if pair else 16 * R)
        grid = min(args.grid or 2 * nat.num_cus(), n_tiles)
        stamps = torch.zeros((grid, 8), dtype=torch.int64, device=dev)

        def make(stream, dbg):
            a = nat.LinearArgs()
            a.fmt, a.R, a.w, a.N, a.K = nat.W_Q4, R, stream.data_ptr(), N, K
            a.x, a.x_dtype, a.M, a.ldx = x.data_ptr(), nat.dtype_code(x.dtype), 1, K
            a.norm_scale = None if norm is None else norm.data_ptr()
            a.norm_dtype, a.eps = nat.BF16, 1e-5
            a.scales, a.zeros = sc.data_ptr(), ze.data_ptr()
            if pair:
                a.scales2, a.zeros2 = sc.data_ptr(), ze.data_ptr()
            a.sz_dtype, a.epi = nat.BF16, epi
            a.group_cols = args.group
            a.y, a.y_dtype, a.ldy = out.data_ptr(), nat.dtype_code(out.dtype), N
            a.waves, a.grid, a.prefetch, a.flags = args.waves, grid, args.prefetch, args.flags
            a.debug_stamps = stamps.data_ptr() if dbg else None
            return a

        arr = (nat.LinearArgs * n_buf)()
        for i in range(n_buf):
            arr[i] = make(streams[i], i == n_buf - 1)
        nat.check(nat.lib().mi355_linear_fast_batch(arr, n_buf, nat.stream_ptr()), "batch")
        torch.cuda.synchronize()
        st = stamps.cpu().numpy()[:, :6].astype(np.float64) / 100.0  # 100 MHz -> us
        t0 = st[:, 0].min()
        print(f"{name}: N={N} K={K} R={R} grid={grid} waves={args.waves} prefetch={args.prefetch} ({nbytes / 1e6:.1f} MB)")
        for i, nm in enumerate(NAMES):
            col = st[:, i]
            col = col[col > 0] - t0
            if col.size:
                print(f"   {nm:15s} min {col.min():6.2f}  med {np.median(col):6.2f}  max {col.max():6.2f}  (n={col.size})")
        del streams
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""In-kernel phase timeline + launch durations of the LLM.int8 linear at the 7B decode shapes (wall-clock stamps
written by thread 0 of every workgroup, see mi355_int8_args.debug_stamps)."""
import argparse
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from lit_llama_amd import _native as nat  # noqa: E402
from lit_llama_amd import ops  # noqa: E402
from scripts.sweep_gemv import SHAPES_7B  # noqa: E402

NAMES = ["entry", "ring issued", "rows staged", "quantised", "tile0 streamed", "tile0 stored", "exit"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--waves", type=int, default=8)
    ap.add_argument("--prefetch", default="0,4")
    ap.add_argument("--outliers", type=int, default=0)
    ap.add_argument("--shapes", default="attn,proj,fc,mproj")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(0)
    for name in args.shapes.split(","):
        N, K, R, epi = SHAPES_7B[name]
        R = 2 if epi == nat.EPI_SWIGLU else 1
        pair = epi == nat.EPI_SWIGLU
        nbytes = ops.packed_bytes(nat.W_I8, N, K, R, pair)
        n_buf = max(2, int(600e6 // nbytes) + 1)
        streams = [torch.randint(0, 256, (nbytes,), generator=gen, device=dev, dtype=torch.uint8) for _ in range(n_buf)]
        scb = (0.05 + 0.05 * torch.rand(N, generator=gen, device=dev)).float()
        x = torch.randn((1, K), generator=gen, device=dev)
        for i in range(args.outliers):
            x[0, (37 * i + 5) % K] = 7.0 + i
        x = x.to(torch.bfloat16 if epi == nat.EPI_ACCUM else torch.float32)
        norm = None if epi == nat.EPI_ACCUM else torch.ones(K, device=dev).to(torch.bfloat16)
        out = torch.zeros((1, N), device=dev, dtype=torch.float32 if epi != nat.EPI_SWIGLU else torch.bfloat16)
        n_tiles = (N + (16 if pair else 16 * R) - 1) // (16 if pair else 16 * R)
        grid = min(args.grid or nat.num_cus(), n_tiles)
        for pf in [int(v) for v in args.prefetch.split(",")]:
            stamps = torch.zeros((grid, 8), dtype=torch.int64, device=dev)

            def make(stream, dbg):
                a = nat.Int8Args()
                a.w, a.scb, a.N, a.K = stream.data_ptr(), scb.data_ptr(), N, K
                a.x, a.x_dtype, a.M, a.ldx = x.data_ptr(), nat.dtype_code(x.dtype), 1, K
                a.norm_scale = None if norm is None else norm.data_ptr()
                a.norm_dtype, a.eps, a.threshold, a.R = nat.BF16, 1e-5, 6.0, R
                a.epi = epi
                a.scb2 = scb.data_ptr() if pair else None
                a.y, a.y_dtype, a.ldy = out.data_ptr(), nat.dtype_code(out.dtype), N
                a.waves, a.grid, a.prefetch = args.waves, grid, pf
                a.debug_stamps = stamps.data_ptr() if dbg else None
                return a

            sp = nat.stream_ptr()
            argv = [make(streams[i], False) for i in range(n_buf)]
            for a in argv[:2]:
                nat.check(nat.lib().mi355_linear_int8(C.byref(a), sp), "warm")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for rep in range(3):
                for a in argv:
                    nat.check(nat.lib().mi355_linear_int8(C.byref(a), sp), "run")
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (3 * n_buf)
            a = make(streams[0], True)
            nat.check(nat.lib().mi355_linear_int8(C.byref(a), sp), "stamped")
            torch.cuda.synchronize()
            st = stamps.cpu().numpy()[:, :7].astype(np.float64) / 100.0
            t0 = st[:, 0].min()
            print(f"{name}: N={N} K={K} R={R} grid={grid} prefetch={pf} ({nbytes / 1e6:.1f} MB)  "
                  f"{us:.2f} us/launch back-to-back = {nbytes / us / 1e3:.0f} GB/s")
            for i, nm in enumerate(NAMES):
                col = st[:, i]
                col = col[col > 0] - t0
                if col.size:
                    print(f"   {nm:15s} min {col.min():6.2f}  med {np.median(col):6.2f}  max {col.max():6.2f}")
        del streams
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

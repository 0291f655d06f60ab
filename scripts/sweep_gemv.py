#!/usr/bin/env python
"""On-box tuning sweep of the weight-streaming linear: per-shape launch duration over (waves, grid, prefetch, nt),
cycling through enough distinct weight buffers that nothing is served from the 256 MiB Infinity Cache.

    python scripts/sweep_gemv.py [--fmt q4] [--quick]   ->  one line per configuration + a best-of table
"""
import argparse
import itertools
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from lit_llama_amd import _native as nat  # noqa: E402
from lit_llama_amd import ops  # noqa: E402

SHAPES_7B = {
    # name: (N, K, R, epi)
    "attn": (12288, 4096, 2, nat.EPI_STORE),
    "proj": (4096, 4096, 1, nat.EPI_ACCUM),
    "fc": (11008, 4096, 2, nat.EPI_SWIGLU),
    "mproj": (4096, 11008, 1, nat.EPI_ACCUM),
    "lm_head": (32000, 4096, 2, nat.EPI_STORE),
    "attn1": (12288, 4096, 1, nat.EPI_STORE),
    "lm_head1": (32000, 4096, 1, nat.EPI_STORE),
    # 13B (n_embd 5120, n_hidden 13824), as the engine packs them (R = 1 except the c_fc1 / c_fc2 pair)
    "attn13": (15360, 5120, 1, nat.EPI_STORE),
    "proj13": (5120, 5120, 1, nat.EPI_ACCUM),
    "fc13": (13824, 5120, 2, nat.EPI_SWIGLU),
    "mproj13": (5120, 13824, 1, nat.EPI_ACCUM),
    "lm_head13": (32000, 5120, 1, nat.EPI_STORE),
    # 65B (n_embd 8192, n_hidden 22016): the model behind configs[4] and bench.py's TP = 1 leg
    "attn65": (24576, 8192, 1, nat.EPI_STORE),
    "proj65": (8192, 8192, 1, nat.EPI_ACCUM),
    "fc65": (22016, 8192, 2, nat.EPI_SWIGLU),
    "mproj65": (8192, 22016, 1, nat.EPI_ACCUM),
    "lm_head65": (32000, 8192, 1, nat.EPI_STORE),
    # one rank's shard of 65B at TP = 8 (scripts/convert_checkpoint.py:57-65: c_attn / c_fc rows, c_proj columns)
    "attn_tp8": (3072, 8192, 1, nat.EPI_STORE),
    "proj_tp8": (8192, 1024, 1, nat.EPI_STORE),
    "fc_tp8": (2752, 8192, 2, nat.EPI_SWIGLU),
    "mproj_tp8": (8192, 2752, 1, nat.EPI_STORE),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--shapes", default="attn,proj,fc,mproj,lm_head")
    ap.add_argument("--out", default=None)
    ap.add_argument("--group", type=int, default=0, help="grouped scales: input columns per (scale, zero) pair")
    ap.add_argument("--bufs", type=int, default=0, help="distinct weight buffers in rotation (0: enough to defeat "
                    "the 256 MiB Infinity Cache; 1: one buffer, i.e. cache-resident weights)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(0)
    cus = nat.num_cus()
    results = {}
    for name in args.shapes.split(","):
        N, K, R, epi = SHAPES_7B[name]
        pair = epi == nat.EPI_SWIGLU
        nbytes = ops.packed_bytes(nat.W_Q4, N, K, R, pair)
        n_buf = args.bufs if args.bufs > 0 else max(2, int(600e6 // nbytes) + 1)  # > 2x the Infinity Cache
        streams = [torch.randint(0, 256, (nbytes,), generator=gen, device=dev, dtype=torch.uint8) for _ in range(n_buf)]
        ng = -(-K // args.group) if args.group else 1
        sc = (0.005 + 0.005 * torch.rand(N * ng, generator=gen, device=dev)).to(torch.bfloat16)
        ze = torch.full((N * ng,), 8.0, device=dev, dtype=torch.bfloat16)
        x = torch.randn((1, K), generator=gen, device=dev).to(torch.bfloat16 if epi == nat.EPI_ACCUM else torch.float32)
        norm = None if epi == nat.EPI_ACCUM else (1 + 0.1 * torch.randn(K, generator=gen, device=dev)).to(torch.bfloat16)
        out = torch.zeros((1, N), device=dev, dtype=torch.float32 if epi != nat.EPI_SWIGLU else torch.bfloat16)
        n_tiles = (N + (16 if pair else 16 * R) - 1) // (16 if pair else 16 * R)
        grids = sorted({g for g in (n_tiles, cus, 2 * cus, 3 * cus, 4 * cus, 8 * cus) if g <= n_tiles})
        waves_l, pf_l, nt_l = (8, 16), (4,), (0,)
        if args.quick:
            grids, waves_l, pf_l, nt_l = [g for g in grids if g in (cus, 2 * cus)] or grids[-1:], (8,), (4,), (0,)
        best = None
        import ctypes as C

        def make_args(stream, waves, grid, pf, flags):
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
            a.waves, a.grid, a.prefetch, a.flags = waves, grid, pf, flags
            return a

        reps = 4 if args.bufs == 0 else max(4, 64 // n_buf)
        for waves, grid, pf, flags in itertools.product(waves_l, grids, pf_l, nt_l):
            arr = (nat.LinearArgs * (reps * n_buf))()
            for i in range(reps * n_buf):
                arr[i] = make_args(streams[i % n_buf], waves, grid, pf, flags)
            sp = nat.stream_ptr()
            nat.check(nat.lib().mi355_linear_fast_batch(arr, n_buf, sp), "warm-up")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            nat.check(nat.lib().mi355_linear_fast_batch(arr, reps * n_buf, sp), "batch")
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (reps * n_buf)
            gbs = nbytes / us / 1e3
            rec = dict(shape=name, waves=waves, grid=grid, prefetch=pf, us=round(us, 2), GBps=round(gbs, 1))
            print(json.dumps(rec), flush=True)
            if best is None or us < best["us"]:
                best = rec
        results[name] = best
        del streams
        torch.cuda.empty_cache()
    print("BEST " + json.dumps(results), flush=True)
    if args.out:
        Path(args.out).write_text(json.dumps(results, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""In-kernel phase timeline of the decode attention kernel (7B geometry) + per-launch time vs n_split."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from lit_llama_amd import _native as nat  # noqa: E402
from oracle import oracle  # noqa: E402

import argparse  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--S", type=int, default=401)
ap.add_argument("--pos", default="150,399")
ap.add_argument("--splits", default="1,2,4,8")
cli = ap.parse_args()
NAMES = ["entry", "kv issued", "q staged", "rows done", "exit"]
dev = torch.device("cuda:0")
n_head, hs, S = 32, 128, cli.S
C_ = n_head * hs
gen = torch.Generator(device=dev).manual_seed(0)
rope = oracle.build_rope_cache(2048, hs, dtype=torch.int64).to(dev)
L = 32  # rotate over 32 layers' caches like the model does
ks = [torch.randn((1, n_head, S, hs), generator=gen, device=dev).to(torch.bfloat16) for _ in range(L)]
vs = [torch.randn((1, n_head, S, hs), generator=gen, device=dev).to(torch.bfloat16) for _ in range(L)]
qkv = torch.randn((1, 1, 3 * C_), generator=gen, device=dev)
y = torch.zeros((1, 1, C_), dtype=torch.bfloat16, device=dev)
for pos_v in [int(v) for v in cli.pos.split(",")]:
    pos = torch.tensor([pos_v], dtype=torch.int32, device=dev)
    for ns in [int(v) for v in cli.splits.split(",")]:
        parts = torch.zeros((1, n_head, ns, hs + 4), dtype=torch.float32, device=dev)
        nblk = n_head * ns
        stamps = torch.zeros((nblk, 8), dtype=torch.int64, device=dev)

        def args(l, dbg):
            a = nat.AttnArgs()
            a.qkv, a.qkv_dtype, a.B, a.ld_qkv = qkv.data_ptr(), nat.F32, 1, 3 * C_
            a.rope, a.pos = rope.data_ptr(), pos.data_ptr()
            a.kcache, a.vcache, a.cache_dtype = ks[l].data_ptr(), vs[l].data_ptr(), nat.BF16
            a.T, a.n_head, a.hs, a.S = 1, n_head, hs, S
            a.y, a.y_dtype, a.ldy = y.data_ptr(), nat.BF16, C_
            a.n_split = ns
            a.partials = parts.data_ptr() if ns > 1 else None
            a.debug_stamps = stamps.data_ptr() if dbg else None
            return a

        s = nat.stream_ptr()
        for l in range(L):
            nat.check(nat.lib().mi355_attention(C.byref(args(l, False)), s))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        arr = [args(l % L, False) for l in range(4 * L)]
        e0.record()
        for a in arr:
            nat.check(nat.lib().mi355_attention(C.byref(a), s))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / len(arr)
        nat.check(nat.lib().mi355_attention(C.byref(args(0, True)), s))
        torch.cuda.synchronize()
        st = stamps.cpu().numpy()[:, :5].astype(np.float64) / 100.0
        t0 = st[:, 0].min()
        line = "  ".join(f"{nm} {np.median(st[:, i] - t0):5.2f}/{(st[:, i] - t0).max():5.2f}" for i, nm in enumerate(NAMES))
        print(f"pos {pos_v} n_split {ns}: {us:6.2f} us/launch (host-paced eager)   stamps med/max: {line}")

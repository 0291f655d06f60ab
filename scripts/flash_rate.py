"""Time of the prefill attention (rope_kv_write + flash_prefill_kernel) for T query tokens of a 7B layer (32 heads of 128).
    python scripts/flash_rate.py [T]
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from lit_llama_amd import ops  # noqa: E402
from lit_llama_amd.model import build_rope_cache  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
nh, hs = 32, 128
gen = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn((1, T, 3 * nh * hs), generator=gen, device=dev, dtype=torch.float32)
rope = build_rope_cache(T, hs, torch.int64, dev).float().contiguous()
k = torch.zeros((1, nh, T, hs), dtype=torch.bfloat16, device=dev)
v = torch.zeros_like(k)
pos = torch.arange(T, device=dev, dtype=torch.int32)
for _ in range(3):
    y = ops.attention(qkv, rope, nh, pos=pos, kv_cache=(k, v), out_dtype=torch.bfloat16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    y = ops.attention(qkv, rope, nh, pos=pos, kv_cache=(k, v), out_dtype=torch.bfloat16)
e1.record()
e1.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
flop = 4.0 * nh * hs * T * (T + 1) / 2
print(f"attention T={T}: {us:.1f} us per call (rope_kv_write + flash) = {flop / us / 1e6:.1f} TFLOP/s of causal work; checksum {float(y.float().abs().mean()):.5f}")

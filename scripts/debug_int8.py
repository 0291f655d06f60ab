"""Segment-by-segment comparison of the native engine with the oracle for a small llm.int8 model (debug aid)."""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from lit_llama_amd import synth  # noqa: E402
from lit_llama_amd._native import check, lib  # noqa: E402
from lit_llama_amd.model import LLaMA, LLaMAConfig  # noqa: E402
from lit_llama_amd.utils import EmptyInitOnDevice  # noqa: E402
from oracle import oracle  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "llm.int8"
outl = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
kw = dict(n_layer=2, n_head=4, n_embd=256)
cfg = LLaMAConfig(**kw)
sd = synth.make_state_dict(cfg, seed=0, mode=mode, outlier_channels=outl)
with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode=mode):
    model = LLaMA(cfg)
model.load_state_dict(sd)
eng = model.engine()
assert eng is not None, model._engine_failed
T = 5
prompt = synth.make_prompt(T)
om = oracle.Model(oracle.Config(**kw), {k: v.clone() for k, v in sd.items()}, mode=mode)

# oracle trace
idx = prompt.view(1, -1)
pos = torch.arange(T)
om.rope_cache = oracle.build_rope_cache(cfg.block_size, cfg.n_embd // cfg.n_head)
ones = torch.ones((cfg.block_size, cfg.block_size), dtype=torch.bool)
om.mask_cache = torch.tril(ones).unsqueeze(0).unsqueeze(0)
S = 12
rope = om.rope_cache.index_select(0, pos)
mask = om.mask_cache.index_select(2, pos)[:, :, :, :S]
x = torch.nn.functional.embedding(idx, om.p("transformer.wte.weight"))
hs = cfg.n_embd // cfg.n_head
om.kv_caches = [(torch.zeros(1, cfg.n_head, S, hs), torch.zeros(1, cfg.n_head, S, hs)) for _ in range(cfg.n_layer)]
trace = [("embed", x[0].clone())]
for i in range(cfg.n_layer):
    pre = f"transformer.h.{i}."
    h, om.kv_caches[i] = om.attention(i, oracle.rmsnorm(x, om.p(pre + "rms_1.scale")), rope, mask, S, pos, om.kv_caches[i])
    x = x + h
    trace.append((f"L{i} attn", x[0].clone()))
    x = x + om.mlp(i, oracle.rmsnorm(x, om.p(pre + "rms_2.scale")))
    trace.append((f"L{i} mlp", x[0].clone()))

with torch.cuda.stream(eng.stream):
    eng._ensure_cache(S)
    eng.set_step(prompt.to(dev), T, 0)
    s = eng.stream.cuda_stream
    check(lib().mi355_forward_embed(C.byref(eng.m), T, s))
    got = [("embed", eng.x[:T].clone())]
    for i in range(cfg.n_layer):
        check(lib().mi355_forward_segment(C.byref(eng.m), T, i, 0, 2, s))
        got.append((f"L{i} attn", eng.x[:T].clone()))
        check(lib().mi355_forward_segment(C.byref(eng.m), T, i, 2, 4, s))
        got.append((f"L{i} mlp", eng.x[:T].clone()))
eng.stream.synchronize()
for (n, a), (_, b) in zip(trace, got):
    b = b.float().cpu()
    err = (a - b).abs().max().item()
    print(f"{n:10s} ref rms {a.pow(2).mean().sqrt():8.4f} max|ref| {a.abs().max():8.3f}  max err {err:8.4f}  at {tuple((a - b).abs().argmax().item() // a.shape[1:2][0:1][0] if False else [])}")

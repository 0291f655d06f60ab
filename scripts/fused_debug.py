"""Bring-up check of the fused decode step (csrc/fused_step.hip) against the 162-launch engine step on the same
weights: logits of one step, the KV rows it writes, a chained greedy run, and the rate of both paths.
    python scripts/fused_debug.py [--layers 2] [--steps 32] [--prompt 20]
"""
import argparse
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import lit_llama_amd  # noqa: E402
from lit_llama_amd import synth  # noqa: E402
from lit_llama_amd.model import LLaMA, LLaMAConfig  # noqa: E402
from lit_llama_amd.utils import EmptyInitOnDevice  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--prompt", type=int, default=20)
    ap.add_argument("--S", type=int, default=256)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = LLaMAConfig(n_layer=a.layers, n_head=32, n_embd=4096)
    t0 = time.time()
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode="gptq.int4"):
        model = LLaMA(cfg)
    if a.layers <= 4:
        model.load_state_dict(synth.make_state_dict(cfg, seed=0, mode="gptq.int4"))
    else:
        synth.fill_model_random_int4(model, seed=0)
    model.eval()
    print(f"model built in {time.time() - t0:.1f}s", flush=True)
    eng = model.engine()
    assert eng is not None, model._engine_failed
    print("fused plan:", eng.fused_plan is not None, "fused:", eng.fused is not None, flush=True)
    assert eng.fused is not None
    prompt = synth.make_prompt(a.prompt).to(dev)
    S = a.S

    def one_step(fused: bool):
        eng.fused_enabled = fused
        model.reset_cache()
        with torch.cuda.stream(eng.stream):
            eng._ensure_cache(S)
            eng.prefill(prompt, 0, all_logits=False, argmax=True)
            eng.set_step(None, 1, a.prompt, from_next=True)
            eng.embed_step()
            eng.run_step(1)
        eng.stream.synchronize()
        eng.check_status()
        kv = torch.stack([torch.stack([k[0, :, a.prompt], v[0, :, a.prompt]]) for k, v in model.kv_caches]).float().cpu()
        return eng.logits[0].clone().float().cpu(), int(eng.next_token.item()), kv

    lg_u, tok_u, kv_u = one_step(False)
    lg_f, tok_f, kv_f = one_step(True)
    std = float(lg_u.std())
    print(f"unfused: token {tok_u}  fused: token {tok_f}   logit std {std:.4f}")
    print(f"max |dlogit| {float((lg_u - lg_f).abs().max()):.5f} ({float((lg_u - lg_f).abs().max()) / std:.5f} std)")
    print(f"kv row max |d| {float((kv_u - kv_f).abs().max()):.5f} (max |kv| {float(kv_u.abs().max()):.3f})")
    for l in range(min(a.layers, 4)):
        print(f"  layer {l}: k d {float((kv_u[l, 0] - kv_f[l, 0]).abs().max()):.5f}  v d {float((kv_u[l, 1] - kv_f[l, 1]).abs().max()):.5f}")

    def run(fused: bool, n: int):
        eng.fused_enabled = fused
        model.reset_cache()
        torch.cuda.synchronize()
        t = time.time()
        out = lit_llama_amd.generate(model, prompt, n, max_seq_length=S, top_k=1)
        torch.cuda.synchronize()
        return out.cpu(), time.time() - t

    out_u, _ = run(False, a.steps)
    out_f, _ = run(True, a.steps)
    same = (out_u == out_f)
    print("greedy tokens equal:", bool(same.all()), "first mismatch:", int((~same).nonzero()[0]) if not same.all() else -1)
    print(" unfused:", out_u[a.prompt:a.prompt + 16].tolist())
    print(" fused  :", out_f[a.prompt:a.prompt + 16].tolist())
    # rates (generate includes the prompt; measure the chained steps alone)
    for fused in (False, True):
        eng.fused_enabled = fused
        model.reset_cache()
        with torch.cuda.stream(eng.stream):
            eng._ensure_cache(S)
            eng.prefill(prompt, 0, all_logits=False, argmax=True)
            eng.set_step(None, 1, a.prompt, from_next=True)
            eng.embed_step()
            for _ in range(8):
                eng.run_step(3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(eng.stream)
            n = 64
            for _ in range(n):
                eng.run_step(3)
            e1.record(eng.stream)
        e1.synchronize()
        eng.check_status()
        ms = e0.elapsed_time(e1) / n
        print(f"{'fused' if fused else 'unfused'}: {ms * 1e3:.1f} us/step = {1e3 / ms:.1f} tok/s ({a.layers} layers: "
              f"{ms * 1e3 / a.layers:.2f} us per layer incl. head)")


if __name__ == "__main__":
    main()

"""us per persistent decode step as a function of the position: the 7B int4 bench model, a short prompt, then blocks of 16 chained
steps timed with events on the engine's stream.   python scripts/pos_curve.py [--upto 640] [--heads 32 --layers 32]"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from lit_llama_amd import synth  # noqa: E402
from lit_llama_amd.model import LLaMA, LLaMAConfig  # noqa: E402
from lit_llama_amd.utils import EmptyInitOnDevice  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--upto", type=int, default=640)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--block", type=int, default=16)
    ap.add_argument("--tag", default="default")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = LLaMAConfig(n_layer=a.layers, n_head=a.heads, n_embd=128 * a.heads)
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode="gptq.int4"):
        model = LLaMA(cfg)
    synth.fill_model_random_int4(model, seed=0)
    model.eval()
    eng = model.engine()
    assert eng is not None and eng.fused is not None, model._engine_failed
    P = 16
    prompt = synth.make_prompt(P).to(dev)
    S = a.upto + 64
    out = []
    with torch.cuda.stream(eng.stream):
        for rep in range(2):  # (the first pass warms)
            model.reset_cache()
            eng._ensure_cache(S)
            eng.prefill(prompt, 0, all_logits=False, argmax=True)
            eng.set_step(None, 1, P, from_next=True)
            eng.embed_step()
            pos = P
            out = []
            while pos + a.block <= a.upto:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(eng.stream)
                for _ in range(a.block):
                    eng.run_step(3)
                e1.record(eng.stream)
                e1.synchronize()
                out.append((pos, e0.elapsed_time(e1) * 1e3 / a.block))
                pos += a.block
    assert eng.check_status() is None
    print(f"POS {a.tag} weight_fmt {int(eng.fused.weight_fmt)}: " + " ".join(f"{p}:{t:.0f}" for p, t in out))


if __name__ == "__main__":
    main()

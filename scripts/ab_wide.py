"""A / B driver of the wide-shape persistent decode step (csrc/fused_step_wide.hip): ONE process per library variant
(MI355_LLAMA_LIB selects it), a `--layers`-deep gptq.int4 model at the 65B width (64 heads, n_embd 8192; --heads 32: the 7B shape
through MI355_FUSED_WIDE=1), a 128-token prompt; prints the parity of the persistent step against the launch-per-operator step on the
same weights and the step time over three blocks of chained greedy steps.
    MI355_LLAMA_LIB=lit_llama_amd/_variants/libmi355llama_x.so python scripts/ab_wide.py [--tag x] [--layers 16]
"""
import argparse
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from lit_llama_amd import synth  # noqa: E402
from lit_llama_amd.model import LLaMA, LLaMAConfig  # noqa: E402
from lit_llama_amd.utils import EmptyInitOnDevice  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="default")
    ap.add_argument("--layers", type=int, default=16)
    ap.add_argument("--heads", type=int, default=64)
    ap.add_argument("--prompt", type=int, default=128)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--no-parity", action="store_true")
    a = ap.parse_args()
    if a.heads == 32:
        os.environ["MI355_FUSED_WIDE"] = "1"
    dev = torch.device("cuda:0")
    cfg = LLaMAConfig(n_layer=a.layers, n_head=a.heads, n_embd=128 * a.heads)
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode="gptq.int4"):
        model = LLaMA(cfg)
    synth.fill_model_random_int4(model, seed=0)
    model.eval()
    eng = model.engine()
    assert eng is not None and eng.fused is not None and int(eng.fused.weight_fmt) in (4, 5), model._engine_failed
    a.tag = f"{a.tag}/fmt{int(eng.fused.weight_fmt)}"
    prompt = synth.make_prompt(a.prompt).to(dev)
    S = a.prompt + 8 + 4 * a.steps + 16

    def start():
        model.reset_cache()
        eng._ensure_cache(S)
        eng.prefill(prompt, 0, all_logits=False, argmax=True)
        eng.set_step(None, 1, a.prompt, from_next=True)
        eng.embed_step()

    if not a.no_parity:
        res = {}
        for fused in (False, True):
            eng.fused_enabled = fused
            lgs = []
            with torch.cuda.stream(eng.stream):
                start()
                for _ in range(8):
                    eng.run_step(3)
                    lgs.append(eng.logits[0].clone())
            eng.stream.synchronize()
            assert eng.check_status() is None
            res[fused] = torch.stack(lgs).float().cpu()
        eng.fused_enabled = True
        std = float(res[False].std(-1).mean())
        d = (res[True] - res[False]).abs().amax(-1) / std
        print("AB", a.tag, "PARITY vs launch path (logit-std per step):", [round(float(x), 4) for x in d], "finite", bool(torch.isfinite(res[True]).all()))
    blocks = []
    with torch.cuda.stream(eng.stream):
        start()
        for _ in range(8):
            eng.run_step(3)
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(eng.stream)
            for _ in range(a.steps):
                eng.run_step(3)
            e1.record(eng.stream)
            e1.synchronize()
            blocks.append(e0.elapsed_time(e1) * 1e3 / a.steps)
    assert eng.check_status() is None
    med = sorted(blocks)[1]
    print(f"AB {a.tag} layers {a.layers} heads {a.heads}: us/step {[round(b, 1) for b in blocks]} median {med:.1f} = {med / a.layers:.2f} us per layer "
          f"(incl. 1/{a.layers} of lm_head)")


if __name__ == "__main__":
    main()

"""Diagnostic (round 5): the 7B bench model (synth.fill_model_random_int4) through every decode path on the SAME tokens — persistent step
with fp8-limb operands, with fp16 operands, launch-per-operator step, op-by-op module path (independent generic kernels) — pairwise
distances in logit std per step, for the zero points 8 and 7.5."""
import sys
import warnings
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import lit_llama_amd  # noqa: E402
from lit_llama_amd import synth  # noqa: E402
from lit_llama_amd.model import LLaMA, LLaMAConfig  # noqa: E402
from lit_llama_amd.utils import EmptyInitOnDevice  # noqa: E402


def teacher_forced(model, toks, T, S, dev):
    model.reset_cache()
    rows = []
    input_pos = torch.arange(0, T, device=dev)
    pos0 = 0
    for _ in range(toks.numel() - T):
        x = toks.index_select(0, input_pos).view(1, -1)
        input_pos._mi355_pos0 = pos0
        rows.append(model(x, S, input_pos)[0, -1].float().cpu())
        pos0 = pos0 + input_pos.numel()
        input_pos = input_pos[-1:] + 1
    model.reset_cache()
    return torch.stack(rows)


def main():
    dev = torch.device("cuda:0")
    cfg = LLaMAConfig.from_name("7B")
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode="gptq.int4"):
        model = LLaMA(cfg)
    model.eval()
    T, n_new = 9, 8
    S = T + n_new
    prompt = synth.make_prompt(T, vocab=cfg.vocab_size, seed=3).to(dev)
    for zero, gain in ((8.0, 2.2), (7.5, 2.2), (7.0, 2.2), (7.5, 1.0)):
        model._drop_engine()
        synth.fill_model_random_int4(model, seed=0, zero=zero, gain=gain)
        eng = model.engine()
        eng.fused_enabled = False
        model.reset_cache()
        toks = lit_llama_amd.generate(model, prompt, n_new, top_k=1)
        rows = {}
        with warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter("always")
            for label, fused, fmt in (("launch", False, None), ("fused/fmt3", True, 3), ("fused/fmt0", True, 0)):
                eng.reset_fused_format()
                eng.fused_enabled = fused
                if fused:
                    eng.use_fused_format(fmt)
                rows[label] = teacher_forced(model, toks, T, S, dev)
                bad = eng.check_status()
                print(f"zero {zero} gain {gain} {label}: status {bad}, demotions {eng.fused_demotions}", flush=True)
                eng.fused_demotions.clear()
            eng.reset_fused_format()
            model.use_engine = False
            rows["module"] = teacher_forced(model, toks, T, S, dev)
            model.use_engine = True
        std = float(rows["module"].std(-1).mean())
        names = list(rows)
        for i, a in enumerate(names):
            for b in names[i + 1:]:
                d = (rows[a] - rows[b]).abs().amax(-1) / std
                print(f"zero {zero} gain {gain} {a} vs {b}: max {float(d.max()):.4f} std; per step {[round(float(v), 4) for v in d]}", flush=True)
        print(f"zero {zero} gain {gain}: logit std {std:.3f}, warnings {len(wl)}", flush=True)


if __name__ == "__main__":
    main()

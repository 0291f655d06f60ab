"""Diagnostic (round 5): is a FRACTIONAL zero point (7.5: zero-mean levels) computed the same by every path?  Two 7B-width blocks (no depth
for chaos to develop), uniform random int4 levels, zero 8 / 7.5, scales of gain 1 (0.217 / sqrt(K)) and of the bench model (0.48 / sqrt(K)):
teacher-forced logits of the launch-per-operator step, the persistent step (fp8 limbs, fp16 operands) and the module path against the
CPU oracle (f32) on the same state dict."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "oracle"))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import oracle  # noqa: E402
import lit_llama_amd  # noqa: E402
from lit_llama_amd import synth  # noqa: E402
from lit_llama_amd.model import LLaMA, LLaMAConfig  # noqa: E402
from lit_llama_amd.utils import EmptyInitOnDevice  # noqa: E402
from diag_bench_model_parity import teacher_forced  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    n_layer = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    cfg = LLaMAConfig(n_layer=n_layer, n_head=32, n_embd=4096)
    T, n_new = 9, 6
    S = T + n_new
    prompt = synth.make_prompt(T, vocab=cfg.vocab_size, seed=3).to(dev)
    for zero, gain in ((8.0, 1.0), (7.5, 1.0), (7.5, 2.2), (8.0, 2.2)):
        with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode="gptq.int4"):
            model = LLaMA(cfg)
        model.eval()
        synth.fill_model_random_int4(model, seed=0, zero=zero, gain=gain)
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        eng = model.engine()
        eng.fused_enabled = False
        toks = lit_llama_amd.generate(model, prompt, n_new, top_k=1)
        om = oracle.Model(oracle.Config(n_layer=n_layer, n_head=32, n_embd=4096), {k: v.float() if v.is_floating_point() else v for k, v in sd.items()},
                          mode="gptq.int4")
        ref = oracle.teacher_forced_logits(om, toks.cpu().long(), T) if hasattr(oracle, "teacher_forced_logits") else None
        rows = {}
        for label, fused, fmt in (("launch", False, None), ("fused/fmt3", True, 3), ("fused/fmt0", True, 0)):
            if fused and eng.fused is None:
                continue
            eng.reset_fused_format()
            eng.fused_enabled = fused
            if fused:
                eng.use_fused_format(fmt)
            rows[label] = teacher_forced(model, toks, T, S, dev)
            eng.check_status()
        eng.reset_fused_format()
        model.use_engine = False
        rows["module"] = teacher_forced(model, toks, T, S, dev)
        model.use_engine = True
        ref = torch.as_tensor(ref).float().reshape(rows["module"].shape) if ref is not None else rows["module"]
        std = float(ref.std(-1).mean())
        for k, v in rows.items():
            d = (v - ref).abs().amax(-1) / std
            print(f"layers {n_layer} zero {zero} gain {gain} {k} vs oracle: max {float(d.max()):.4f} std; per step {[round(float(x), 4) for x in d]}", flush=True)
        del model, eng


if __name__ == "__main__":
    main()

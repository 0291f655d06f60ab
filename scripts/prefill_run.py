"""A 2048-token prompt through the 7B int4 engine (for rocprofv3 --kernel-trace --stats and for timing)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from lit_llama_amd import synth  # noqa: E402
from lit_llama_amd.model import LLaMA, LLaMAConfig  # noqa: E402
from lit_llama_amd.utils import EmptyInitOnDevice  # noqa: E402

dev = torch.device("cuda:0")
cfg = LLaMAConfig.from_name("7B")
mode = sys.argv[2] if len(sys.argv) > 2 else "gptq.int4"  # "none": the bf16 model (BASELINE configs[1])
if mode == "none":
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16):
        model = LLaMA(cfg)
    for prm in model.parameters():
        prm.data.normal_(0.0, 0.02)
elif mode == "llm.int8":  # BASELINE configs[3], weights as bench.py draws them
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode="llm.int8"):
        model = LLaMA(cfg)
    gen = torch.Generator(device=dev).manual_seed(0)
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if name.endswith("scale"):
                prm.copy_((1 + 0.1 * torch.randn(prm.shape, generator=gen, device=dev)).to(prm.dtype))
            elif name.endswith("wte.weight"):
                prm.copy_(torch.randn(prm.shape, generator=gen, device=dev).to(prm.dtype))
        for mod in model.modules():
            if isinstance(mod, torch.nn.Linear):
                mod._quantize_weight(torch.randn(mod.weight.shape, generator=gen, device=dev) * mod.in_features**-0.5)
else:  # "gptq.int4", or "g128" / "g64" ...: GPTQ groupsize checkpoints (scales / zeros per row and group of columns)
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode="gptq.int4"):
        model = LLaMA(cfg)
    if mode.startswith("g") and mode[1:].isdigit():
        from lit_llama_amd.quantization import ColBlockQuantizedLinear
        for _, mod in list(model.named_modules()):
            for cname, child in list(mod.named_children()):
                if isinstance(child, ColBlockQuantizedLinear):
                    q = ColBlockQuantizedLinear(child.in_features, child.out_features, bias=False, bits=4, tile_cols=int(mode[1:]))
                    setattr(mod, cname, q.to(device=dev, dtype=torch.bfloat16))
    synth.fill_model_random_int4(model, seed=0)
model.eval()
eng = model.engine()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
prompt = synth.make_prompt(T).to(dev)
with torch.cuda.stream(eng.stream):
    eng._ensure_cache(T + 8)
    for i in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(eng.stream)
        eng.prefill(prompt, 0, all_logits=False, argmax=True)
        e1.record(eng.stream)
        e1.synchronize()
        print(f"prefill {T} tokens (chunk {eng.max_T}): {e0.elapsed_time(e1):.2f} ms")

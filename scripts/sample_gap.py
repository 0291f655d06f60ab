#!/usr/bin/env python
"""Kernel-trace helper: a few decode steps alone, with a tiny kernel after each, and with mi355_sample after each.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap -o gap -- python scripts/sample_gap.py
    python scripts/sample_gap.py --report gpurun_out/gap/.../gap_kernel_trace.csv
prints, per variant, the fused kernel's duration and the idle gaps around it."""
import csv
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


class A:
    model, quantize, tune = "7B", "gptq.int4", None


def run():
    import lit_llama_amd
    from bench import build_model
    from lit_llama_amd import ops, synth

    dev = torch.device("cuda:0")
    model, cfg = build_model(A, dev)
    eng = model.engine()
    prompt = synth.make_prompt(16, vocab=cfg.vocab_size, seed=1).to(dev)
    lit_llama_amd.generate(model, prompt, 4, top_k=1, max_seq_length=512)
    uni = torch.rand(1024, device=dev)
    row = eng.logits[0, : eng.m.lm_head.N]
    one = torch.zeros(1, device=dev)
    n = 24
    for name in ("alone", "tiny", "sample"):
        with torch.cuda.stream(eng.stream):
            eng.set_step(prompt[-1:], 1, 16)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                eng.run_step(0)
                if name == "tiny":
                    one.add_(1.0)
                elif name == "sample":
                    ops.sample(row, 0.8, 200, uni, eng.pos, eng.next_token, out_tokens=eng.out_tokens, tokens=eng.tokens,
                               advance=True)
            e1.record()
            torch.cuda.synchronize()
            eng.check_status()
        print(f"{name:8s}: {e0.elapsed_time(e1) * 1e3 / n:7.1f} us / token")
        time.sleep(0.05)


def report(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ks = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    for i, (nm, s, e) in enumerate(ks):
        if "fused_step" in nm and i + 2 < len(ks) and i > 0:
            prev, nxt = ks[i - 1], ks[i + 1]
            print(f"fused {(e - s) / 1e3:8.1f} us | gap before {(s - prev[2]) / 1e3:7.1f} (after {prev[0][:28]:28s} {(prev[2] - prev[1]) / 1e3:6.1f} us)"
                  f" | gap after {(nxt[1] - e) / 1e3:7.1f} -> {nxt[0][:28]}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        report(sys.argv[2])
    else:
        run()

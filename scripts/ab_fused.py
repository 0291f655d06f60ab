"""A / B driver of the persistent decode step: ONE process per library variant (MI355_LLAMA_LIB selects it), the 32-layer 7B
int4 bench model, a 128-token prompt; prints the parity of the fused step against the launch-per-operator step on the same
weights, the rate over three blocks of 64 chained steps (positions 136..328) and, with --timeline, the phase timeline of one
layer (scripts/fused_timeline.py).
    MI355_LLAMA_LIB=lit_llama_amd/_variants/libmi355llama_x.so python scripts/ab_fused.py [--timeline] [--tag x]
`--f8`: the two operand paths of the int4 step (mi355_fused_step_args.weight_fmt 0 / 3) in ONE process on one engine.
"""
import argparse
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from lit_llama_amd import synth  # noqa: E402
from lit_llama_amd.model import LLaMA, LLaMAConfig  # noqa: E402
from lit_llama_amd.utils import EmptyInitOnDevice  # noqa: E402
from scripts.fused_timeline import NAMES, ORDER, budget  # noqa: E402


def f8_ab(a):
    """weight_fmt 0 against weight_fmt 3 on one engine (same box, same arena, same process)."""
    dev = torch.device("cuda:0")
    cfg = LLaMAConfig(n_layer=a.layers, n_head=32, n_embd=4096)
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode="gptq.int4"):
        model = LLaMA(cfg)
    synth.fill_model_random_int4(model, seed=0)
    model.eval()
    eng = model.engine()
    assert eng is not None and eng.fused is not None and eng.fused.weight_fmt in (0, 3), model._engine_failed
    prompt = synth.make_prompt(a.prompt).to(dev)
    S = a.prompt + 8 + 64 * a.blocks * 2 + 80

    def set_fmt(f):
        eng.use_fused_format(f)  # (zeroes the granules of the other tag width: include/mi355_llama.h, weight_fmt)
        eng.stream.synchronize()

    def start():
        model.reset_cache()
        eng._ensure_cache(S)
        eng.prefill(prompt, 0, all_logits=False, argmax=True)
        eng.set_step(None, 1, a.prompt, from_next=True)
        eng.embed_step()

    def decode12(fused, f):
        eng.fused_enabled = fused
        if fused:
            set_fmt(f)
        toks, lgs = [], []
        with torch.cuda.stream(eng.stream):
            start()
            for _ in range(12):
                eng.run_step(3)
                lgs.append(eng.logits[0].clone())
                toks.append(eng.next_token.clone())
        eng.stream.synchronize()
        try:
            eng.check_status()
            status = "ok"
        except Exception as e:  # noqa: BLE001 (an abort of the experimental path is a result, not a crash)
            status = str(e)[:120]
        eng.fused_enabled = True
        return torch.stack(lgs).float().cpu(), torch.cat(toks).cpu(), status

    base = decode12(False, 0)
    std = float(base[0].std(-1).mean())
    for f in (0, 3):
        import warnings

        with warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter("always")
            lg, tk, status = decode12(True, f)
        d = (lg - base[0]).abs().amax(-1) / std
        print("PARITY", {"weight_fmt": f, "status": status, "finite": bool(torch.isfinite(lg).all()),
                         "dlogit_std_max": round(float(d.max()), 5), "dlogit_std_per_step": [round(float(x), 4) for x in d],
                         "tokens_equal": bool((tk == base[1]).all()), "clipped": int(getattr(eng, "fused_clipped", 0) or 0),
                         "warnings": [str(w.message)[:80] for w in wl]}, flush=True)
    # rates: blocks of 64 chained steps, the two paths alternating inside one chain of positions
    for order in ((0, 3), (3, 0)):
        for f in order:
            set_fmt(f)
            with torch.cuda.stream(eng.stream):
                start()
                for _ in range(8):
                    eng.run_step(3)
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.blocks + 1)]
                evs[0].record(eng.stream)
                for b in range(a.blocks):
                    for _ in range(64):
                        eng.run_step(3)
                    evs[b + 1].record(eng.stream)
            evs[-1].synchronize()
            try:
                eng.check_status()
                status = "ok"
            except Exception as e:  # noqa: BLE001
                status = str(e)[:120]
            us = [round(evs[b].elapsed_time(evs[b + 1]) / 64 * 1e3, 1) for b in range(a.blocks)]
            print("RATE", {"weight_fmt": f, "us_per_step": us, "tok_s_first_block": round(1e6 / us[0], 1), "status": status}, flush=True)
    if a.timeline:
        for f in (0, 3):
            set_fmt(f)
            stamps = torch.zeros((256, 64), dtype=torch.int64, device=dev)
            with torch.cuda.stream(eng.stream):
                start()
                for _ in range(8):
                    eng.run_step(3)
                eng.fused.debug_stamps = stamps.data_ptr()
                eng.fused.reserved0 = a.layer
                eng.run_step(3)
                eng.fused.debug_stamps = None
                eng.fused.reserved0 = 0
            eng.stream.synchronize()
            st = stamps.cpu().numpy().astype(np.float64) / 100.0
            t0 = st[:, 2].min()
            print(f"timeline weight_fmt {f}: layer {a.layer}, position {a.prompt + 8}; whole step {st[:, 1].max() - st[:, 0].min():.1f} us")
            prev = 0.0
            for i in ORDER:
                col = st[:, i] - t0
                print(f"  {NAMES[i]:28s} min {col.min():7.2f}  med {np.median(col):7.2f}  max {col.max():7.2f}   (+{np.median(col) - prev:5.2f})")
                prev = np.median(col)
            budget(st)
    set_fmt(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default=os.environ.get("MI355_LLAMA_LIB", "default"))
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--prompt", type=int, default=128)
    ap.add_argument("--blocks", type=int, default=3)
    ap.add_argument("--timeline", action="store_true")
    ap.add_argument("--layer", type=int, default=10)
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--f8", action="store_true",
                    help="ONE process, both operand paths of the int4 step: weight_fmt 0 (fp16 operands) and 3 (fp8 limbs), toggled on the "
                         "live engine; parity of each against the launch path, alternating rate blocks, timelines")
    a = ap.parse_args()
    if a.f8:
        return f8_ab(a)
    dev = torch.device("cuda:0")
    cfg = LLaMAConfig(n_layer=a.layers, n_head=32, n_embd=4096)
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode="gptq.int4"):
        model = LLaMA(cfg)
    synth.fill_model_random_int4(model, seed=0)
    model.eval()
    eng = model.engine()
    assert eng is not None and eng.fused is not None, model._engine_failed
    prompt = synth.make_prompt(a.prompt).to(dev)
    S = a.prompt + 8 + 64 * a.blocks + 80

    def start():
        model.reset_cache()
        eng._ensure_cache(S)
        eng.prefill(prompt, 0, all_logits=False, argmax=True)
        eng.set_step(None, 1, a.prompt, from_next=True)
        eng.embed_step()

    out = {"tag": a.tag}
    if not a.no_parity:
        res = {}
        for fused in (False, True):
            eng.fused_enabled = fused
            toks, lgs = [], []
            with torch.cuda.stream(eng.stream):
                start()
                for _ in range(12):
                    eng.run_step(3)
                    lgs.append(eng.logits[0].clone())
                    toks.append(eng.next_token.clone())
            eng.stream.synchronize()
            eng.check_status()
            res[fused] = (torch.stack(lgs).float().cpu(), torch.cat(toks).cpu())
        std = float(res[False][0].std(-1).mean())
        d = float((res[False][0] - res[True][0]).abs().max()) / std
        same = bool((res[False][1] == res[True][1]).all())
        out["dlogit_std"] = round(d, 5)
        out["tokens_equal"] = same
        out["finite"] = bool(torch.isfinite(res[True][0]).all())
    eng.fused_enabled = True
    rates = []
    with torch.cuda.stream(eng.stream):
        start()
        for _ in range(8):
            eng.run_step(3)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.blocks + 1)]
        evs[0].record(eng.stream)
        for b in range(a.blocks):
            for _ in range(64):
                eng.run_step(3)
            evs[b + 1].record(eng.stream)
    evs[-1].synchronize()
    eng.check_status()
    for b in range(a.blocks):
        rates.append(evs[b].elapsed_time(evs[b + 1]) / 64 * 1e3)
    out["us_per_step"] = [round(r, 1) for r in rates]
    out["tok_s_first_block"] = round(1e6 / rates[0], 1)
    out["clipped"] = int(getattr(eng, "fused_clipped", 0) or 0)
    print("AB", out, flush=True)

    if a.timeline:
        stamps = torch.zeros((256, 64), dtype=torch.int64, device=dev)
        with torch.cuda.stream(eng.stream):
            start()
            for _ in range(8):
                eng.run_step(3)
            eng.fused.debug_stamps = stamps.data_ptr()
            eng.fused.reserved0 = a.layer
            eng.run_step(3)
            eng.fused.debug_stamps = None
            eng.fused.reserved0 = 0
        eng.stream.synchronize()
        eng.check_status()
        st = stamps.cpu().numpy().astype(np.float64) / 100.0
        t0 = st[:, 2].min()
        print(f"timeline {a.tag}: layer {a.layer}, position {a.prompt + 8}; whole step {st[:, 1].max() - st[:, 0].min():.1f} us")
        prev = 0.0
        for i in ORDER:
            col = st[:, i] - t0
            print(f"  {NAMES[i]:28s} min {col.min():7.2f}  med {np.median(col):7.2f}  max {col.max():7.2f}   (+{np.median(col) - prev:5.2f})")
            prev = np.median(col)
        budget(st)


if __name__ == "__main__":
    main()

"""Phase timeline of one layer inside the fused decode step (in-kernel wall-clock stamps, 100 MHz).
    python scripts/fused_timeline.py [--layer 10] [--prompt 128]
"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from lit_llama_amd import synth  # noqa: E402
from lit_llama_amd.model import LLaMA, LLaMAConfig  # noqa: E402
from lit_llama_amd.utils import EmptyInitOnDevice  # noqa: E402

NAMES = {2: "G  x gathered (c_attn)", 3: "G  q/k/v published", 4: "G  head q gathered", 5: "G  attn partials ready",
         6: "G  attn out published", 7: "G  attn out gathered", 8: "G  c_proj published", 9: "G  x gathered (fc)",
         10: "G  hidden published", 11: "G  hidden gathered", 12: "G  mlp.c_proj published",
         20: "S  c_attn B1", 21: "S  c_attn consumed", 23: "S  attn q staged",          25: "S  attn out done", 26: "S  c_proj B1", 27: "S  c_proj consumed", 28: "S  fc B1", 29: "S  fc consumed",
         30: "S  mproj B1", 31: "S  mproj consumed"}
ORDER = [2, 20, 21, 3, 4, 23, 25, 5, 6, 7, 26, 27, 8, 9, 28, 29, 10, 11, 30, 31, 12]


# the chain of one layer inside ONE workgroup (gatherer 0 and streamer wave 0 share the workgroup's barriers): consecutive events
CHAIN = [(2, 20, "x gathered -> c_attn B1 (streamers past the barrier)"), (20, 21, "c_attn: B1 -> last tile parked"),
         (21, 3, "c_attn: parked -> q/k/v published (epilogue, RoPE, cache row)"), (3, 4, "head-local q/k/v hand-off"),
         (4, 23, "q staged -> streamers past Ba1"), (23, 25, "attention: scores + weighted values"), (25, 5, "-> partials ready (Ba3)"),
         (5, 6, "merge + attention output published"), (6, 7, "attention-output hand-off (all-gather)"), (7, 26, "gathered -> c_proj B1"),
         (26, 27, "attn.c_proj: B1 -> tile parked"), (27, 8, "attn.c_proj: parked -> x published (residual, limbs)"),
         (8, 9, "x hand-off into the MLP (all-gather)"), (9, 28, "gathered -> fc B1"), (28, 29, "c_fc1/c_fc2: B1 -> last tile parked"),
         (29, 10, "pair: parked -> hidden published"), (10, 11, "hidden hand-off (all-gather)"), (11, 30, "gathered -> mproj B1"),
         (30, 31, "mlp.c_proj: B1 -> tile parked"), (31, 12, "mlp.c_proj: parked -> x published"),
         (12, 46, "x hand-off into the next layer (all-gather)")]


def budget(st, out=print):
    """Where a layer's period goes, with a time base that closes: every row is the difference of two
    consecutive events INSIDE one workgroup (same clock, same chain), reduced over the 256 workgroups; slot 46 is 'x gathered' of
    the next layer, so the rows add up to the layer's period in every workgroup and the medians add up to ~ the median period.
    `st`: [256][64] stamps in us."""
    period = st[:, 46] - st[:, 2]
    ok = (st[:, 46] > 0) & (st[:, 2] > 0)
    if not ok.all():
        out(f"  (budget: slot 46 missing in {int((~ok).sum())} workgroups: library without the next-layer stamp)")
        return None
    rows, tot = [], 0.0
    for a, b, what in CHAIN:
        d = st[:, b] - st[:, a]
        rows.append((what, float(np.median(d)), float(d.min()), float(d.max())))
        tot += float(np.median(d))
    # wave skew (fp8-operand kernel): slots 48 / 51 / 52 / 53 = when the LAST streamer wave reached the phase's last tile end
    for slot, first, pub, nm in ((48, 21, 3, "c_attn"), (51, 27, 8, "attn.c_proj"), (52, 29, 10, "c_fc1/c_fc2"), (53, 31, 12, "mlp.c_proj")):
        if (st[:, slot] > 0).all():
            d1, d2 = st[:, slot] - st[:, first], st[:, pub] - st[:, slot]
            rows.append((f"  [{nm}: wave 0 parked -> LAST wave parked]", float(np.median(d1)), float(d1.min()), float(d1.max())))
            rows.append((f"  [{nm}: last wave parked -> published]", float(np.median(d2)), float(d2.min()), float(d2.max())))
    hand = sum(m for w, m, _, _ in rows if "hand-off" in w)
    out(f"  layer period (x gathered -> x gathered of the next layer, per workgroup): median {np.median(period):6.2f} us, "
        f"min {period.min():6.2f}, max {period.max():6.2f}; sum of the row medians {tot:6.2f}; hand-off rows {hand:5.2f}")
    for what, m, lo, hi in rows:
        out(f"    {what:62s} med {m:6.2f}  min {lo:6.2f}  max {hi:6.2f}")
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", type=int, default=10)
    ap.add_argument("--prompt", type=int, default=128)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--heads", type=int, default=32, help="64: the 65B width (n_embd 8192) on the wide-shape kernel (fused_step_wide.hip)")
    ap.add_argument("--group-cols", type=int, default=0, help="GPTQ groupsize model (GRP instantiation of the kernel)")
    ap.add_argument("--quantize", default="gptq.int4", choices=["gptq.int4", "llm.int8", "none"],
                    help="stream format of the persistent step: int4 (default), LLM.int8 or BF16 (round 4)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = LLaMAConfig(n_layer=a.layers, n_head=a.heads, n_embd=128 * a.heads)
    if a.quantize != "gptq.int4":
        import types

        import bench  # the bench's random 7B model of that format (32 layers)

        built = bench.build_model(types.SimpleNamespace(model="7B", quantize=a.quantize, adapter=False, group_cols=0, tune=None), dev)
        model = built[0] if isinstance(built, tuple) else built
        a.layers = model.config.n_layer
    else:
        with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode="gptq.int4"):
            model = LLaMA(cfg)
    if a.group_cols:
        from lit_llama_amd.quantization import ColBlockQuantizedLinear

        for _, mod in list(model.named_modules()):
            for cname, child in list(mod.named_children()):
                if isinstance(child, ColBlockQuantizedLinear):
                    q = ColBlockQuantizedLinear(child.in_features, child.out_features, bias=False, bits=4, tile_cols=a.group_cols)
                    setattr(mod, cname, q.to(device=dev, dtype=torch.bfloat16))
    if a.quantize == "gptq.int4":
        synth.fill_model_random_int4(model, seed=0)
    eng = model.engine()
    assert eng is not None and eng.fused is not None, model._engine_failed
    prompt = synth.make_prompt(a.prompt).to(dev)
    stamps = torch.zeros((256, 64), dtype=torch.int64, device=dev)
    with torch.cuda.stream(eng.stream):
        eng._ensure_cache(a.prompt + 64)
        eng.prefill(prompt, 0, all_logits=False, argmax=True)
        eng.set_step(None, 1, a.prompt, from_next=True)
        eng.embed_step()
        for _ in range(8):
            eng.run_step(3)
        eng.fused.debug_stamps = stamps.data_ptr()
        eng.fused.reserved0 = a.layer
        eng.run_step(3)
        eng.fused.debug_stamps = None
        eng.fused.reserved0 = 0
    eng.stream.synchronize()
    eng.check_status()
    st = stamps.cpu().numpy().astype(np.float64) / 100.0  # us
    t0 = st[:, 2].min()
    print(f"layer {a.layer}, position {a.prompt + 8}; whole step {st[:, 1].max() - st[:, 0].min():.1f} us; "
          f"times in us after the first workgroup has gathered the layer's input")
    prev = 0.0
    for i in ORDER:
        col = st[:, i] - t0
        print(f"  {NAMES[i]:28s} min {col.min():7.2f}  med {np.median(col):7.2f}  max {col.max():7.2f}   (+{np.median(col) - prev:5.2f})")
        prev = np.median(col)
    raw = stamps.cpu().numpy()
    if raw[:, 32].any():
        print("ring state of streamer wave 0 at the start of a phase (quads of 4 KiB; LDS-DMA kernel only):")
        for k3, nm, b1 in ((0, "c_attn", 20), (1, "attn.c_proj", 26), (2, "c_fc1/c_fc2", 28)):
            o = 32 + 4 * k3
            print(f"  {nm:12s} known landed {raw[:, o + 2].mean():.2f}  requested {raw[:, o + 3].mean():.2f}   "
                  f"requesting took {np.median(st[:, o] - st[:, b1]):.2f} us, waiting for the first group "
                  f"{np.median(st[:, o + 1] - st[:, o]):.2f} us")
    budget(st)
    if (st[:, 57] > 0).all() and (st[:, 53] > 0).all():  # wide-shape kernel: gatherer 0 past the Bt of c_attn / mlp.c_proj
        for nm, last, bt, pub in (("c_attn", 48, 56, 3), ("mlp.c_proj", 53, 57, 12)):
            print(f"  {nm}: last wave parked -> gatherer 0 past Bt med {np.median(st[:, bt] - st[:, last]):5.2f} us, "
                  f"past Bt -> published med {np.median(st[:, pub] - st[:, bt]):5.2f} us")


if __name__ == "__main__":
    main()

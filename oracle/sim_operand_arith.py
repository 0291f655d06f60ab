"""CPU model of the int4 linears' OPERAND ARITHMETIC at full depth (round 5): what do the logits of a 7B checkpoint lose when
a kernel computes  y = s (sum_k (off + q_k) r(x_k) - (off + z) sum_k r(x_k))  — the "magic exponent" conversion of an int4 level
with its offset `off` undone by the operand sum — instead of  y = sum_k s (q_k - z) r(x_k)  (the reference: dequantise, then multiply,
lit_llama/quantization.py:376-423), with r = the activation rounding of the path (bf16 / fp16) and f32 accumulation in MFMA-sized steps.

The oracle's `linear` is replaced by the statement of ONE variant and the fixture's 48 tokens go through all 32 blocks in one causal
pass (teacher forced); the distance to the fixture's reference logits is printed per decode step in the unit the GPU tests use
(max |dlogit| over the probe columns / mean logit std).  Nothing here touches the product; it exists so that an arithmetic can be
priced before a kernel is rewritten for it (NOTES round 5, item 68).

    python oracle/sim_operand_arith.py cfg2_7b_int4_real  bf16:128 bf16:16 bf16:exact f16:1024 f16:exact

More variants (round 5, later): `f8:limbs` = the fp8-limb hand-off of the persistent step's default rung — the input of every linear as three
E4M3 limbs under the pre-scale of its edge kind (x edges 2^0 on the normalised vector, attention output 2^2, SwiGLU output 2^4; values past
448 x the pre-scale SATURATE, i.e. what a clipped step would compute if the engine did not recompute it; `f8:ladder` gives the rows that
would clip fp16 operands instead, as the engine's recovery does), weights as the levels themselves, zero point by the operand sum; a `+kv` suffix on any variant rounds q (after RoPE), K and V to bf16 as the engine's cache does.
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/oracle")
import oracle  # noqa: E402
from lit_llama_amd import synth  # noqa: E402
from lit_llama_amd.model import LLaMAConfig  # noqa: E402

PROBES = (np.arange(64) * (32000 // 64) + 7) % 32000
_unpacked = {}


def unpack(sd, prefix):
    qw = sd[prefix + ".quant_weight"]
    N, Kb = qw.shape
    q = torch.empty((N, Kb * 2), dtype=torch.float32)
    q[:, 0::2] = (qw & 15).float()
    q[:, 1::2] = (qw >> 4).float()
    return q, sd[prefix + ".scales"].float()[:, 0], sd[prefix + ".zeros"].float()[:, 0]


def f8_limb_round(x, prefix):
    """x as the consumers of an fp8-limb edge see it: l0 + l1 / 16 + l2 / 256 under the edge's pre-scale (tests/layouts.py states the codec)."""
    sys.path.insert(0, "/root/repo/tests")
    import layouts

    e = 2 if prefix.endswith("attn.c_proj") else F8_EH if prefix.endswith("mlp.c_proj") else 0
    v = x.double().numpy() * 2.0 ** -e
    limbs = layouts.f8_limbs(v)
    rec = layouts.e4m3_decode(limbs[0]) + layouts.e4m3_decode(limbs[1]) / 16.0 + layouts.e4m3_decode(limbs[2]) / 256.0
    out = torch.from_numpy(rec * 2.0 ** e).float()
    if LADDER:  # rows (positions) whose values pass the edge's range take fp16 operands instead — the engine recomputes such a step one rung down
        clipped = x.abs().amax(-1) > 448.0 * 2.0 ** e
        out[clipped] = x[clipped].to(torch.float16).float()
    return out


LADDER = False
F8_EH = int(__import__("os").environ.get("SIM_F8_EH", "4"))  # pre-scale exponent of the SwiGLU edge (csrc/fused_step_ring.hip MI355_F8_EH)


def rnd(x, kind):
    if kind == "bf16":
        return x.to(torch.bfloat16).float()
    if kind == "f16":
        return x.to(torch.float16).float()
    if kind == "f32":
        return x
    raise ValueError(kind)


def make_linear(kind, off, waves=8, step=32):
    def linear(sd, prefix, x, mode):
        q, s, z = unpack(sd, prefix)
        B, T, K = x.shape
        xr = f8_limb_round(x.reshape(T, K).float(), prefix) if kind == "f8" else rnd(x.reshape(T, K).float(), kind)
        if off is None:  # dequantise, then multiply (products exact in f32 up to the accumulation: f64 here)
            w = ((q - z[:, None]) * s[:, None]).double()
            return (xr.double() @ w.t()).float().view(B, T, -1)
        units = K // 128
        per = -(-units // waves)
        tot = torch.zeros((T, q.shape[0]), dtype=torch.float32)
        S = torch.zeros((T, 1), dtype=torch.float32)
        qd = (q + off).double()
        xd = xr.double()
        for w in range(waves):
            acc = torch.zeros_like(tot)
            sw = torch.zeros_like(S)
            for k0 in range(w * per * 128, min((w + 1) * per * 128, K), step):
                p = xd[:, k0:k0 + step] @ qd[:, k0:k0 + step].t()  # exact: small integers x (<= 11-bit) operands in f64
                acc = (acc.double() + p).float()                   # one f32 rounding per MFMA
                sw = (sw.double() + xd[:, k0:k0 + step].sum(1, keepdim=True)).float()
            tot = (tot.double() + acc.double()).float()
            S = (S.double() + sw.double()).float()
        y = s[None, :] * (tot - (off + z)[None, :] * S)
        return y.view(B, T, -1)

    return linear


def main():
    name = sys.argv[1]
    variants = sys.argv[2:]
    fx = np.load(f"/root/repo/tests/golden/{name}.npz")
    torch.set_num_threads(8)
    cfg = LLaMAConfig.from_name("7B")
    stats = str(fx["stats"]) if "stats" in fx.files else "unit"
    t0 = time.time()
    sd = synth.make_state_dict(cfg, seed=int(fx["seed"]), mode="gptq.int4", stats=stats)
    print(f"checkpoint in {time.time() - t0:.0f} s", flush=True)
    toks = torch.from_numpy(fx["tokens"].astype(np.int64)).view(1, -1)
    T = int(fx["prompt_len"])
    std = float(fx["std"].mean())
    om = oracle.Model(oracle.Config(n_layer=cfg.n_layer, n_head=cfg.n_head, n_embd=cfg.n_embd), sd, mode="gptq.int4")
    orig = oracle.linear
    orig_rope = oracle.apply_rope
    orig_sdpa = torch.nn.functional.scaled_dot_product_attention
    bf = lambda t: t.to(torch.bfloat16).to(t.dtype)  # noqa: E731
    for v in variants:
        kv = v.endswith("+kv")
        kind, off = v[:-3].split(":") if kv else v.split(":")
        global LADDER
        LADDER = off == "ladder"  # `f8:ladder`: limbs, except for the rows that would clip (fp16 operands there)
        if LADDER:
            off = "limbs"
        # (`f8:limbs`: operands q x limbs, zero point by the operand sum = offset 0 in the statement above)
        oracle.linear = orig if (kind, off) == ("f32", "ref") else make_linear(kind, None if off == "exact" else 0.0 if off == "limbs" else float(off))
        # +kv: q (after RoPE), K and V as the bf16 numbers the engine's cache rows / attention operands are
        oracle.apply_rope = (lambda x, r: bf(orig_rope(x, r))) if kv else orig_rope
        oracle.F.scaled_dot_product_attention = (lambda q, k, v_, **kw: orig_sdpa(q, k, bf(v_), **kw)) if kv else orig_sdpa
        t0 = time.time()
        with torch.no_grad():
            logits = om(toks[:, :-1])[0]
        rows = logits[T - 1:, PROBES].numpy()
        n = min(rows.shape[0], fx["probes"].shape[0])
        per_step = np.abs(rows[:n] - fx["probes"][:n]).max(axis=1) / std
        print(f"{name} {v}: max {per_step.max():.4f} std, per step {np.round(per_step, 4).tolist()} ({time.time() - t0:.0f} s)",
              flush=True)
    oracle.linear, oracle.apply_rope, oracle.F.scaled_dot_product_attention = orig, orig_rope, orig_sdpa


if __name__ == "__main__":
    main()

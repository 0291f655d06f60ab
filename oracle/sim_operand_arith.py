"""CPU model of the int4 linears' OPERAND ARITHMETIC at full depth (round 5): what do the logits of a 7B checkpoint lose when
a kernel computes  y = s (sum_k (off + q_k) r(x_k) - (off + z) sum_k r(x_k))  — the "magic exponent" conversion of an int4 level
with its offset `off` undone by the operand sum — instead of  y = sum_k s (q_k - z) r(x_k)  (the reference: dequantise, then multiply,
lit_llama/quantization.py:376-423), with r = the activation rounding of the path (bf16 / fp16) and f32 accumulation in MFMA-sized steps.

The oracle's `linear` is replaced by the statement of ONE variant and the fixture's 48 tokens go through all 32 blocks in one causal
pass (teacher forced); the distance to the fixture's reference logits is printed per decode step in the unit the GPU tests use
(max |dlogit| over the probe columns / mean logit std).  Nothing here touches the product; it exists so that an arithmetic can be
priced before a kernel is rewritten for it (NOTES round 5, item 68).

    python oracle/sim_operand_arith.py cfg2_7b_int4_real  bf16:128 bf16:16 bf16:exact f16:1024 f16:exact
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
        xr = rnd(x.reshape(T, K).float(), kind)
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
    for v in variants:
        kind, off = v.split(":")
        oracle.linear = orig if v == "f32:ref" else make_linear(kind, None if off == "exact" else float(off))
        t0 = time.time()
        with torch.no_grad():
            logits = om(toks[:, :-1])[0]
        rows = logits[T - 1:, PROBES].numpy()
        n = min(rows.shape[0], fx["probes"].shape[0])
        per_step = np.abs(rows[:n] - fx["probes"][:n]).max(axis=1) / std
        print(f"{name} {v}: max {per_step.max():.4f} std, per step {np.round(per_step, 4).tolist()} ({time.time() - t0:.0f} s)",
              flush=True)
    oracle.linear = orig


if __name__ == "__main__":
    main()

"""Produce tests/golden/gptq_*.npz by running the UNMODIFIED reference GPTQQuantizer (/root/reference
lit_llama/quantization.py:426-616) in this container on seeded CPU inputs.

    python oracle/gen_golden_gptq.py       # needs /root/reference; run in the build container only

Stored: the f32 weight, the calibration batches, and the reference's results (scales, zeros, packed quant_weight,
quantised weights Q as handed to pack_weight, error).  The script refuses to write a fixture that oracle/gptq.py (the restatement) does not
reproduce bit for bit.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
sys.path[:0] = [str(ROOT / "oracle" / "_stubs"), str(REF), str(ROOT)]

import lit_llama.quantization as refq  # noqa: E402
from lit_llama.quantization import GPTQQuantizer as RefGPTQ  # noqa: E402

from oracle import gptq as ogptq  # noqa: E402
from oracle import oracle  # noqa: E402

OUT = ROOT / "tests" / "golden"

CASES = {
    # name: (out_features, in_features, groupsize, actorder, n_batches, rows per batch)
    "gptq_actorder": (32, 256, -1, True, 4, 16),      # the generate-time format: one group per row, act-order
    "gptq_plain": (24, 192, -1, False, 3, 16),          # same without the act-order permutation
    # (groupsize > 0 cannot be pinned: the reference itself raises at quantization.py:578, assigning a [rows, 1]
    #  tensor into a [rows] column; every inference entry point uses groupsize = -1)
}


def run_case(name, N, K, groupsize, actorder, n_batches, rows):
    gen = torch.Generator().manual_seed(len(name) * 1000 + N + K)
    lin = torch.nn.Linear(K, N, bias=False)
    with torch.no_grad():
        lin.weight.copy_(torch.randn((N, K), generator=gen) * K**-0.5)
    # a few strong input channels so that act-order actually reorders and the Hessian is far from identity
    col_scale = 1.0 + 4.0 * (torch.rand(K, generator=gen) > 0.9).float()
    batches = [torch.randn((1, rows, K), generator=gen) * col_scale for _ in range(n_batches)]
    q = RefGPTQ(lin, bits=4, groupsize=groupsize, actorder=actorder)
    for b in batches:
        q.collect_input_stats(None, (b,), None)
    H_ref = q.H.clone()
    # the reference hands its quantised weights Q straight to pack_weight (:612) and keeps no copy: observe the
    # argument through a wrapper (nothing under /root/reference is modified)
    seen = {}
    orig_pack = refq.ColBlockQuantizedLinear.pack_weight

    def spy(self, weight):
        seen["Q"] = weight.detach().clone()
        return orig_pack(self, weight)

    refq.ColBlockQuantizedLinear.pack_weight = spy
    try:
        qmod, err = q.quantize()
    finally:
        refq.ColBlockQuantizedLinear.pack_weight = orig_pack
    Q_ref = seen["Q"].float()

    # the restatement must reproduce everything bit for bit
    hs = ogptq.Hessian(K)
    for b in batches:
        hs.add(b)
    assert torch.equal(hs.H, H_ref), f"{name}: Hessian restatement differs"
    Q, sc, ze, err_o = ogptq.gptq_quantize(lin.weight, hs.H, bits=4, groupsize=groupsize, actorder=actorder)
    assert torch.equal(sc, qmod.scales) and torch.equal(ze, qmod.zeros), f"{name}: scales / zeros differ"
    assert torch.equal(Q, Q_ref), f"{name}: quantised weights differ"
    assert err_o == err, f"{name}: error {err_o} != {err}"
    packed = oracle.colblock_pack(Q, sc, ze, 4, K if groupsize == -1 else groupsize)
    assert torch.equal(packed, qmod.quant_weight), f"{name}: packed bytes differ"

    np.savez_compressed(
        OUT / f"{name}.npz",
        weight=lin.weight.detach().numpy(), batches=torch.stack(batches).numpy(),
        groupsize=np.int32(groupsize), actorder=np.int32(actorder), bits=np.int32(4),
        scales=qmod.scales.numpy(), zeros=qmod.zeros.numpy(), quant_weight=qmod.quant_weight.contiguous().numpy(),
        Q=Q_ref.numpy(), error=np.float64(err),
    )
    print(f"{name}: N={N} K={K} groupsize={groupsize} actorder={actorder} error {err:.6f} -> {OUT / (name + '.npz')}")


if __name__ == "__main__":
    torch.manual_seed(0)
    for name, spec in CASES.items():
        run_case(name, *spec)

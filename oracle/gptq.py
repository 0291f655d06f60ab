"""CPU restatement of the reference's GPTQ weight quantiser — TEST INFRASTRUCTURE ONLY.

Follows `GPTQQuantizer` of /root/reference lit_llama/quantization.py:426-616 (E. Frantar et al., GPTQ,
arXiv:2210.17323, as adapted by lit-llama) and the blockwise driver of quantize/gptq.py:37-135.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import anything under oracle/; the product path
(lit_llama_amd/gptq.py + csrc/gptq.hip) never does.

Pinned: oracle/gen_golden_gptq.py runs the UNMODIFIED reference class here on seeded inputs and refuses to write
tests/golden/gptq_*.npz unless this restatement reproduces scales, zeros, quantised levels and the reported error
bit for bit; tests/test_oracle_golden.py re-checks the pin on every CPU run.

The arithmetic is written row by row / column by column (numpy-style), which is how the HIP kernel walks it; the
floating-point operations and their order are exactly the reference's:
  * row parameters (`find_params_weight`, :472-513): min / max clamped to include 0, scale = (max - min) / maxq,
    zero = round(-min / scale) (or (maxq + 1) / 2 when symmetric);
  * Hessian (`collect_input_stats`, :515-529): running mean of 2 x x^T over calibration rows, in f32;
  * `quantize` (:531-616): dead columns, optional act-order permutation, damping percdamp * mean(diag H), upper
    Cholesky factor of H^-1, then per 128-column block the sequential loop
        q_i = scale * (clamp(round(w_i / scale) + zero, 0, maxq) - zero);  e_i = (w_i - q_i) / d_i;
        w_j -= e_i * Hinv[i, j]  (j >= i, product rounded, then subtracted)
    and after the block  W[:, later] -= E @ Hinv[block, later].
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch


def row_params(x: torch.Tensor, maxq: int, sym: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-row (scale, zero) of a [rows, cols] slice; quantization.py:472-513 with perchannel=True."""
    zero_row = torch.zeros(x.shape[0])
    lo = torch.minimum(x.min(1).values, zero_row)
    hi = torch.maximum(x.max(1).values, zero_row)
    if sym:
        hi = torch.maximum(lo.abs(), hi)
        neg = lo < 0
        lo = torch.where(neg, -hi, lo)
    flat = (lo == 0) & (hi == 0)
    lo = torch.where(flat, torch.full_like(lo, -1.0), lo)
    hi = torch.where(flat, torch.full_like(hi, 1.0), hi)
    scale = (hi - lo) / maxq
    zero = torch.full_like(scale, (maxq + 1) / 2) if sym else torch.round(-lo / scale)
    return scale, zero


def fake_quant(w: torch.Tensor, scale: torch.Tensor, zero: torch.Tensor, maxq: int) -> torch.Tensor:
    """quantization.py:466-470: the dequantised value of the nearest level."""
    level = torch.clamp(torch.round(w / scale) + zero, 0, maxq)
    return scale * (level - zero)


class Hessian:
    """Running H = mean over calibration rows of 2 x x^T (quantization.py:515-529)."""

    def __init__(self, columns: int):
        self.H = torch.zeros((columns, columns))
        self.n = 0

    def add(self, inp: torch.Tensor) -> None:
        x = inp.reshape(-1, inp.shape[-1]) if inp.dim() == 3 else inp
        batch = 1 if inp.dim() == 2 else inp.shape[0]  # the reference counts samples, not rows
        self.H *= self.n / (self.n + batch)
        self.n += batch
        xt = math.sqrt(2 / self.n) * x.t().float()
        self.H += xt.matmul(xt.t())


def block_loop(W1: torch.Tensor, Hinv1: torch.Tensor, scale_cols: torch.Tensor, zero_cols: torch.Tensor, maxq: int):
    """The sequential inner loop over one block of columns (quantization.py:573-592), one row at a time.

    W1 [rows, count] is updated in place; scale_cols / zero_cols [rows, count] give each column's row parameters.
    Returns (Q1, Err1, Losses1)."""
    rows, count = W1.shape
    Q1 = torch.zeros_like(W1)
    E1 = torch.zeros_like(W1)
    L1 = torch.zeros_like(W1)
    for i in range(count):
        w = W1[:, i].clone()
        d = Hinv1[i, i]
        q = fake_quant(w, scale_cols[:, i], zero_cols[:, i], maxq)
        Q1[:, i] = q
        L1[:, i] = (w - q) ** 2 / d**2
        e = (w - q) / d
        # outer product rounded to f32, then subtracted (two roundings, no fused multiply-add)
        W1[:, i:] -= e.unsqueeze(1) * Hinv1[i, i:].unsqueeze(0)
        E1[:, i] = e
    return Q1, E1, L1


def gptq_quantize(weight: torch.Tensor, H: torch.Tensor, *, bits: int, groupsize: int = -1, actorder: bool = False,
                  blocksize: int = 128, percdamp: float = 0.01, sym: bool = False):
    """quantization.py:531-616 up to (not including) the packing.  Returns (levels-as-dequantised-weights Q [N, K],
    scales [N, G], zeros [N, G], error)."""
    W = weight.detach().to(torch.float32).clone()
    rows, columns = W.shape
    maxq = 2**bits - 1
    tile = columns if groupsize == -1 else groupsize
    n_groups = (columns + tile - 1) // tile
    scales = torch.zeros((rows, n_groups))
    zeros = torch.zeros((rows, n_groups))
    s, z = row_params(W, maxq, sym)
    scales[:] = s[:, None]
    zeros[:] = z[:, None]
    assert not (actorder and groupsize != -1)

    H = H.clone()
    dead = torch.diag(H) == 0
    H[dead, dead] = 1
    W[:, dead] = 0
    perm: Optional[torch.Tensor] = None
    if actorder:
        perm = torch.argsort(torch.diag(H), descending=True)
        W = W[:, perm]
        H = H[perm][:, perm]
    damp = percdamp * torch.mean(torch.diag(H))
    idx = torch.arange(columns)
    H[idx, idx] += damp
    H = torch.linalg.cholesky(H)
    H = torch.cholesky_inverse(H)
    Hinv = torch.linalg.cholesky(H, upper=True)

    Q = torch.zeros_like(W)
    losses = torch.zeros_like(W)
    for i1 in range(0, columns, blocksize):
        i2 = min(i1 + blocksize, columns)
        W1 = W[:, i1:i2].clone()
        sc = torch.empty_like(W1)
        zc = torch.empty_like(W1)
        for i in range(i2 - i1):
            c = i1 + i
            if groupsize != -1 and c % groupsize == 0:
                # parameters of a new group come from W as it stands BEFORE this block's updates (:579-585)
                s, z = row_params(W[:, c:c + groupsize], maxq, sym)
                scales[:, c // groupsize] = s
                zeros[:, c // groupsize] = z
            sc[:, i] = s
            zc[:, i] = z
        Q1, E1, L1 = block_loop(W1, H_slice(Hinv, i1, i2), sc, zc, maxq)
        Q[:, i1:i2] = Q1
        losses[:, i1:i2] = L1 / 2
        W[:, i2:] -= E1.matmul(Hinv[i1:i2, i2:])
    if perm is not None:
        Q = Q[:, torch.argsort(perm)]
    return Q, scales, zeros, torch.sum(losses).item()


def H_slice(Hinv: torch.Tensor, i1: int, i2: int) -> torch.Tensor:
    return Hinv[i1:i2, i1:i2]

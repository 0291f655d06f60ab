"""GPTQ post-training weight quantisation on the MI355X: the producer of the int4 checkpoint format the decode path
consumes.

Drop-in for the reference's `GPTQQuantizer` (/root/reference lit_llama/quantization.py:426-616: same constructor,
`collect_input_stats` forward hook, `quantize() -> (ColBlockQuantizedLinear, error)`) and for
`llama_blockwise_quantization` (quantize/gptq.py:37-135).  The sequential inner loop — ~40 000 tiny PyTorch launches
per 4096-column linear in the reference — is ONE hand-written HIP launch per 128-column block
(csrc/gptq.hip, `mi355_gptq_block`); the dense linear algebra around it (Hessian accumulation, Cholesky, the
block-to-block update GEMM) is library work and stays with torch / rocBLAS / rocSOLVER.  GPU tensors only: the
block kernel has no CPU fallback.

Parity: oracle/gptq.py restates the reference bit for bit (pinned by tests/golden/gptq_*.npz).  The block kernel
reproduces that arithmetic exactly (no fused multiply-adds); the Cholesky factor comes from rocSOLVER instead of
LAPACK, so whole-quantiser results agree with the CPU reference up to those last-bit differences in Hinv: row
parameters exact, a fraction of a percent of the levels may move to the neighbouring level (tests/test_gptq_gpu.py).
"""
from __future__ import annotations

import gc
import math
from typing import Optional

import torch
from torch import nn

from . import ops
from .quantization import ColBlockQuantizedLinear


class GPTQQuantizer:
    """quantization.py:426-616.  `linear_module` must be an `nn.Linear` on the GPU."""

    def __init__(self, linear_module, *, bits, perchannel=True, sym=False, blocksize=128, percdamp=0.01, groupsize=-1,
                 actorder=False):
        assert isinstance(linear_module, nn.Linear)
        assert perchannel, "the ColBlock format stores per-row parameters (quantization.py:360-369)"
        assert 1 <= blocksize <= 128, "the block kernel keeps <= 128 columns in LDS"
        assert not (actorder and groupsize != -1), "The permutation trick does not work for grouped quantization"
        self.linear_module = linear_module
        self.dev = linear_module.weight.device
        if self.dev.type != "cuda":
            raise ValueError("GPTQQuantizer runs on the MI355X (module is on %s)" % self.dev)
        self.rows, self.columns = linear_module.weight.shape
        self.H = torch.zeros((self.columns, self.columns), device=self.dev)
        self.nsamples = 0
        self.bits, self.maxq = bits, 2**bits - 1
        self.sym, self.blocksize, self.percdamp = sym, blocksize, percdamp
        self.groupsize, self.actorder = groupsize, actorder
        self.tile_cols = self.columns if groupsize == -1 else groupsize
        n_groups = (self.columns + self.tile_cols - 1) // self.tile_cols
        self.scales = torch.zeros((self.rows, n_groups), dtype=linear_module.weight.dtype, device=self.dev)
        self.zeros = torch.zeros_like(self.scales)

    # ---- row parameters (quantization.py:472-513), native: torch's GPU `tensor / scalar` multiplies by the
    # reciprocal and would land one ulp off the reference
    def find_params_weight(self, x: torch.Tensor):
        x2 = x.flatten(1).float()
        scale, zero = ops.gptq_row_params(x2 if x2.stride(1) == 1 else x2.contiguous(), self.maxq, self.sym)
        shape = [-1] + [1] * (x.dim() - 1)
        return scale.reshape(shape), zero.reshape(shape)

    # ---- calibration statistics: running mean of 2 x x^T (quantization.py:515-529); usable as a forward hook
    def collect_input_stats(self, _module, inp, _out):
        x = inp[0].detach()
        self.last_inp = x
        batch = 1 if x.dim() == 2 else x.shape[0]
        x = x.reshape(-1, x.shape[-1])
        self.H *= self.nsamples / (self.nsamples + batch)
        self.nsamples += batch
        xt = math.sqrt(2 / self.nsamples) * x.t().float()
        self.H += xt.matmul(xt.t())

    # ---- the quantiser proper (quantization.py:531-616)
    @torch.no_grad()
    def quantize(self):
        W = self.linear_module.weight.detach().to(dtype=torch.float32, copy=True)
        scale, zero = self.find_params_weight(W)
        scale, zero = scale.reshape(-1), zero.reshape(-1)
        self.scales[:] = scale[:, None]
        self.zeros[:] = zero[:, None]

        H = self.H
        del self.H
        dead = torch.diag(H) == 0
        H[dead, dead] = 1
        W[:, dead] = 0
        perm: Optional[torch.Tensor] = None
        if self.actorder:
            perm = torch.argsort(torch.diag(H), descending=True)
            W = W[:, perm].contiguous()
            H = H[perm][:, perm]
        damp = self.percdamp * torch.mean(torch.diag(H))
        idx = torch.arange(self.columns, device=self.dev)
        H[idx, idx] += damp
        H = torch.linalg.cholesky(H)
        H = torch.cholesky_inverse(H)
        Hinv = torch.linalg.cholesky(H, upper=True).contiguous()

        Q = torch.zeros_like(W)
        total_loss = torch.zeros_like(W)
        for i1 in range(0, self.columns, self.blocksize):
            i2 = min(i1 + self.blocksize, self.columns)
            if self.groupsize == -1:
                sc, zc = scale, zero
            else:
                # a group's parameters come from W as it stands when its first column is reached, i.e. BEFORE this
                # block's own updates (quantization.py:579-585) — computed here for every column of the block
                cols_s, cols_z = [], []
                for c in range(i1, i2):
                    if c % self.groupsize == 0:
                        scale, zero = (t.reshape(-1) for t in self.find_params_weight(W[:, c:c + self.groupsize]))
                        self.scales[:, c // self.groupsize] = scale
                        self.zeros[:, c // self.groupsize] = zero
                    cols_s.append(scale)
                    cols_z.append(zero)
                sc, zc = torch.stack(cols_s, 1), torch.stack(cols_z, 1)
            Q1, E1, L1 = ops.gptq_block(W[:, i1:i2], Hinv[i1:i2, i1:i2], sc, zc, self.maxq)
            Q[:, i1:i2] = Q1
            total_loss[:, i1:i2] = L1 / 2
            W[:, i2:] -= E1.matmul(Hinv[i1:i2, i2:])
        if perm is not None:
            Q = Q[:, torch.argsort(perm)]
        weight = Q.reshape(self.linear_module.weight.shape).to(self.linear_module.weight.dtype)
        error = torch.sum(total_loss).item()

        q_module = ColBlockQuantizedLinear(self.linear_module.in_features, self.linear_module.out_features,
                                           self.linear_module.bias is not None, bits=self.bits,
                                           tile_cols=self.groupsize).to(self.dev)
        q_module.scales = self.scales
        q_module.zeros = self.zeros
        q_module.pack_weight(weight)
        q_module.bias = self.linear_module.bias
        return q_module, error


SUBMODULES = ("attn.c_attn", "attn.c_proj", "mlp.c_fc1", "mlp.c_fc2", "mlp.c_proj")  # quantize/gptq.py:63-69


@torch.no_grad()
def llama_blockwise_quantization(model, sample_inputs: torch.Tensor, working_device, *, bits=4, groupsize=-1,
                                 log=print):
    """quantize/gptq.py:37-135: quantise every linear of every block in order, each one calibrated on the outputs of
    the already-quantised layers before it; then lm_head on the normalised final hidden states.
    `sample_inputs` [n_samples, T] token ids; the model's blocks are moved to `working_device` one at a time."""
    from .model import build_rope_cache

    wd = torch.device(working_device)
    cfg = model.config
    model.transformer.wte.to(wd)
    sample_inputs = sample_inputs.to(wd)
    inps = model.transformer.wte(sample_inputs)
    T = sample_inputs.shape[1]
    # quantize/gptq.py:62 calls model.build_rope_cache(sample_inputs): the table is built from the INTEGER dtype of
    # the token ids, i.e. exactly in f32 (a bf16 arange rounds positions above 256 and theta to 8 mantissa bits:
    # wrong rotation angles in every calibration attention)
    rope = build_rope_cache(cfg.block_size, cfg.n_embd // cfg.n_head, sample_inputs.dtype, wd)[:T]
    mask = torch.tril(torch.ones((T, T), dtype=torch.bool, device=wd)).view(1, 1, T, T)
    outs = torch.zeros_like(inps)

    def run_block(block):
        for j in range(inps.size(0)):
            outs[j:j + 1], _ = block(inps[j:j + 1], rope, mask, cfg.block_size)

    errors = {}
    for i, block in enumerate(model.transformer.h):
        block.to(wd)
        for name in SUBMODULES:
            module = block.get_submodule(name)
            gptq = GPTQQuantizer(module, bits=bits, groupsize=groupsize, actorder=(groupsize == -1))
            handle = module.register_forward_hook(gptq.collect_input_stats)
            run_block(block)
            handle.remove()
            q_module, error = gptq.quantize()
            pname, dname = name.rsplit(".", 1)
            setattr(block.get_submodule(pname), dname, q_module)
            errors[f"transformer.h.{i}.{name}"] = error
            log(f"{i} {name} quantization error {error:.1f}")
            del gptq
            gc.collect()
        run_block(block)
        inps, outs = outs, inps
    model.transformer.ln_f.to(wd)
    for j in range(inps.size(0)):
        outs[j:j + 1] = model.transformer.ln_f(inps[j:j + 1])
    inps, outs = outs, inps
    model.lm_head.to(wd)
    gptq = GPTQQuantizer(model.lm_head, bits=bits, groupsize=groupsize, actorder=(groupsize == -1))
    handle = model.lm_head.register_forward_hook(gptq.collect_input_stats)
    for j in range(inps.size(0)):
        model.lm_head(inps[j:j + 1])
    handle.remove()
    model.lm_head, errors["lm_head"] = gptq.quantize()
    return errors

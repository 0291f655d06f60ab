"""Drop-in operator plug-ins: `ColBlockQuantizedLinear` (GPTQ int4 / int8) and `Linear8bitLt` (LLM.int8).

Same constructor signatures, registered buffers, state-dict keys, `pack_weight` / `get_weight` / `forward`
contract as /root/reference lit_llama/quantization.py:38-77 and :340-423, so `quantization(mode)` +
`LLaMA.from_name` + `load_state_dict` of a reference checkpoint work unchanged.
`forward` runs hand-written gfx950 kernels through the C ABI; there is neither Triton, nor bitsandbytes,
nor a CPU fallback behind it.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _native as nat
from . import ops


class ColBlockQuantizedLinear(torch.nn.Module):
    """k-bit (4 or 8) linear with per-row x per-column-tile scale / zero.

    Buffers (lit_llama/quantization.py:350-374):
      quant_weight uint8 [out, in * bits / 8] stored column-major (stride (1, out));
      byte j of row n = q[n, 2j] | q[n, 2j + 1] << 4 for bits = 4 (:387-390);
      scales, zeros [out, ceil(in / tile_cols)]; bias [out] or None.
    """

    def __init__(self, in_features, out_features, bias: bool, *, bits, tile_cols):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.tile_cols = tile_cols if tile_cols != -1 else self.in_features
        self.bits = bits
        self.entries_per_byte = 8 // bits
        assert self.entries_per_byte > 0 and self.entries_per_byte * self.bits == 8
        assert in_features % self.entries_per_byte == 0
        self.register_buffer(
            "quant_weight",
            torch.empty((self.out_features, self.in_features // self.entries_per_byte), dtype=torch.uint8)
            .t()
            .contiguous()
            .t(),
        )
        n_groups = (self.in_features + self.tile_cols - 1) // self.tile_cols
        self.register_buffer("scales", torch.empty((self.out_features, n_groups)))
        self.register_buffer("zeros", torch.empty_like(self.scales))
        assert isinstance(bias, bool)
        if bias:
            self.register_buffer("bias", torch.empty((self.out_features,)))
        else:
            self.register_buffer("bias", None)
        self._stream: Optional[torch.Tensor] = None  # repacked weight stream (fast path)
        self._stream_key = None
        # (stream, R, pair, which) while the reference-layout buffer is released: see release_reference_layout
        self._packed_src = None
        self.register_state_dict_pre_hook(lambda mod, prefix, keep_vars: mod._materialize())

    # ---- the reference-layout buffer can be given up while a native engine holds the same weights as a stream ----
    @property
    def quant_weight(self) -> torch.Tensor:
        """The `quant_weight` buffer (lit_llama/quantization.py:350-359).  Rebuilt from the engine's weight stream on
        first use after `release_reference_layout` (state_dict(), the module's own forward, a re-quantisation)."""
        try:
            if self._packed_src is not None:
                self._materialize()
            return self._buffers["quant_weight"]
        except KeyError:
            raise AttributeError("quant_weight") from None

    def release_reference_layout(self, stream: torch.Tensor, R: int, pair: bool, which: int = 0) -> None:
        """The engine has repacked this weight into `stream` (csrc/gemv.hip layout): drop the reference-layout copy
        (one more copy of the whole model otherwise: 3.3 GB for 7B, 32.5 GB for 65B).  The buffer stays registered
        (empty) and is rebuilt from the stream on demand."""
        if self.bits != 4 or self._packed_src is not None:
            return
        self._stream, self._stream_key = None, None
        self._packed_src = (stream, R, pair, which)
        self._buffers["quant_weight"] = torch.empty((self.out_features, 0), dtype=torch.uint8, device=stream.device)

    def _materialize(self) -> None:
        src, self._packed_src = self._packed_src, None
        if src is not None:
            stream, R, pair, which = src
            self._buffers["quant_weight"] = ops.unpack_q4_stream(stream, self.out_features, self.in_features, R, pair,
                                                                which)

    def _apply(self, fn, *args, **kwargs):  # .to() / .cuda(): the stream stays where the engine put it
        self._materialize()
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        if self._packed_src is not None:
            if prefix + "quant_weight" in state_dict:  # about to be overwritten: only the shape matters
                dev = self._packed_src[0].device
                self._packed_src = None
                self._buffers["quant_weight"] = torch.empty(
                    (self.out_features, self.in_features // self.entries_per_byte), dtype=torch.uint8, device=dev).t().contiguous().t()
            else:
                # a partial load (strict=False: adapter / LoRA / any subset of the keys) leaves this weight alone; the
                # stream is the only copy since release_reference_layout, and the model drops its engine on load
                self._materialize()
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    # ---- format utilities (checkpoint tooling: tensor reshuffling on whatever device the buffers live on; NOT compute entry
    # points — `forward` and every kernel-backed op refuse CPU tensors) ---------------------------------
    def _per_column(self, table: torch.Tensor) -> torch.Tensor:
        """[N, groups] scales / zeros -> what multiplies column k: [N, groups, 1] against the [N, groups, tile_cols] view of the
        weight when the groups tile K exactly, else [N, K] with group j covering columns j * tile_cols .. (j + 1) * tile_cols - 1
        and a SHORTER last group (the reference allocates ceil(in / tile_cols) groups and slices, lit_llama/quantization.py:360-369,
        :381-384, :404-410: GPTQ groupsize 512 against K = 11008 is such a shape; advisor r4)."""
        if self.in_features % self.tile_cols == 0:
            return table.unsqueeze(-1)
        return table.repeat_interleave(self.tile_cols, dim=1)[:, : self.in_features]

    def _grouped(self, t: torch.Tensor) -> torch.Tensor:
        """[N, K] -> the view `_per_column` broadcasts against: [N, groups, tile_cols], or [N, K] itself for a ragged last group."""
        if self.in_features % self.tile_cols == 0:
            return t.view(t.shape[0], self.scales.size(1), -1)
        return t

    def pack_weight(self, weight):
        """Quantise `weight` with the current scales / zeros and pack it: levels = clamp(w / scale + zero) with the reference's
        truncating float -> uint8 conversion, `entries_per_byte` consecutive columns per byte, low bits first
        (lit_llama/quantization.py:376-390).  All groups at once, in place in the weight's own dtype as the reference's per-slice
        `/=` and `+=` are."""
        dev = self.quant_weight.device
        w = weight.to(device=dev, copy=True).contiguous()
        g = self._grouped(w)
        g.div_(self._per_column(self.scales)).add_(self._per_column(self.zeros))
        levels = w.clamp_(min=0, max=2**self.bits - 1).to(dtype=torch.uint8)
        epb = self.entries_per_byte
        shifts = torch.arange(epb, device=dev, dtype=torch.int32) * self.bits
        packed = (levels.view(levels.shape[0], -1, epb).to(torch.int32) << shifts).sum(-1).to(torch.uint8)
        self.quant_weight.copy_(packed)  # (keeps the buffer's column-major strides)
        self._stream = None

    def get_weight(self, dtype=torch.float):
        """Dequantised [out, in] weight, (level - zero) * scale (lit_llama/quantization.py:392-411)."""
        if self.quant_weight.device.type == "cuda":
            return ops.colblock_dequant(self.quant_weight, self.scales, self.zeros, self.bits, self.tile_cols,
                                        self.in_features, dtype)
        # off the GPU: checkpoint inspection only; unpack all entries of a byte at once, then the reference's in-place - / *
        epb = self.entries_per_byte
        shifts = torch.arange(epb, dtype=torch.int32) * self.bits
        levels = (self.quant_weight.to(torch.int32).unsqueeze(-1) >> shifts) & ((1 << self.bits) - 1)
        weight = levels.reshape(self.out_features, self.in_features).to(dtype)
        self._grouped(weight).sub_(self._per_column(self.zeros)).mul_(self._per_column(self.scales))
        return weight

    # ---- hot path -----------------------------------------------------------------------------------
    def fast_eligible(self, dtype: torch.dtype) -> bool:
        """MFMA weight-streaming kernel: 4 bits, bf16 I/O, one scale/zero per row (gptq.int4's tile_cols = -1) or per
        row and group of 32 * 2^n columns (bf16 tables, at most 384 groups — groupsize 32 against K = 11008: `grouped_fast`)."""
        return (
            self.bits == 4
            and (self.scales.shape[1] == 1 or self.grouped_fast())
            and dtype == torch.bfloat16
            and self.scales.dtype in (torch.bfloat16, torch.float32)
            and self.in_features % 2 == 0
        )

    def grouped_fast(self) -> bool:
        g, ng = self.tile_cols, self.scales.shape[1]
        return (ng > 1 and g >= 32 and (g & (g - 1)) == 0 and ng <= 384 and self.scales.dtype == torch.bfloat16
                and ng == -(-self.in_features // g))

    def weight_stream(self, R: int = 1) -> torch.Tensor:
        if self._packed_src is not None and self._packed_src[1] == R and not self._packed_src[2]:
            return self._packed_src[0]  # the engine's own stream of this matrix
        key = (self.quant_weight.data_ptr(), self.quant_weight._version, R)
        if self._stream is None or self._stream_key != key:
            self._stream = ops.repack_q4(self.quant_weight, None, self.out_features, self.in_features, R)
            self._stream_key = key
        return self._stream

    def forward(self, inp):
        nat.require_gpu(inp, "ColBlockQuantizedLinear.forward")
        nat.require_gpu(self._buffers["quant_weight"], "ColBlockQuantizedLinear.forward (module buffers)")
        x2d = inp.reshape(-1, inp.shape[-1])
        if x2d.stride(-1) != 1:
            x2d = x2d.contiguous()
        grouped = self.scales.shape[1] > 1
        if (self.fast_eligible(inp.dtype) and x2d.shape[0] >= 32 and self.out_features % 4 == 0 and self.bias is None
                and (not grouped or self.in_features % 128 == 0)):
            # wide input (prompt / no-cache evaluation, evaluate/full.py:120-129): LDS-tiled MFMA GEMM over the stream
            y = ops.linear_gemm(x2d, self.weight_stream(1), 1, self.out_features, self.in_features,
                                scales=self.scales.reshape(-1), zeros=self.zeros.reshape(-1), out_dtype=inp.dtype,
                                group_cols=self.tile_cols if grouped else 0)
        elif self.fast_eligible(inp.dtype):
            R = 2 if self.out_features % 32 == 0 and self.out_features >= 16384 else 1
            y = ops.linear_fast(
                x2d, self.weight_stream(R), nat.W_Q4, R, self.out_features, self.in_features,
                scales=self.scales.reshape(-1), zeros=self.zeros.reshape(-1),
                bias=None if self.bias is None else self.bias.to(self.scales.dtype),
                out_dtype=inp.dtype, group_cols=self.tile_cols if grouped else 0,
            )
        else:
            y = ops.linear_colblock(x2d, self.quant_weight, self.scales, self.zeros, self.bits, self.tile_cols,
                                    self.bias, self.in_features)
        return y.view(*inp.shape[:-1], self.out_features)


class Linear8bitLt(torch.nn.Linear):
    """LLM.int8 linear for inference: int8 row-quantised weight (`weight.CB`, `weight.SCB`), quantised at
    construction and re-quantised when a float checkpoint is loaded (lit_llama/quantization.py:38-77).
    `threshold = 6.0`, `has_fp16_weights = False` as in the reference."""

    threshold = 6.0

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.weight.requires_grad_(False)
        if self.bias is not None:
            self.bias.requires_grad_(False)
        self._stream: Optional[torch.Tensor] = None
        self._stream_key = None
        self._pending_fp: Optional[torch.Tensor] = None
        self._quantize_weight(self.weight.data)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # Re-quantise the float `<prefix>weight` of the checkpoint on the GPU and let nn.Module load the bias
        # (lit_llama/quantization.py:52-67; the reference looks the key up by its `weight` suffix, here it is
        # addressed by the module prefix so the order of the checkpoint's keys does not matter).
        weight_key = prefix + "weight"
        if weight_key in state_dict:
            self._quantize_weight(state_dict.pop(weight_key))
        # else: an adapter-only checkpoint loaded with strict=False behind the pretrained one (generate/adapter_v2.py:94-101): the
        # int8 weight stays, the keys below still have to land (advisor r4: they used to be skipped with the weight)
        # the module's other parameters — the bias, and the adapter_scale / adapter_bias pair LLaMA-Adapter v2 attaches to every
        # nn.Linear, this class included (generate/adapter_v2.py with --quantize llm.int8) — go through nn.Module's loader, as the
        # reference's `if local_state_dict: super()._load_from_state_dict(...)` does (round 4: only the bias used to be loaded)
        if any(k.startswith(prefix) and "." not in k[len(prefix):] for k in state_dict):
            stash = self._parameters.pop("weight")
            try:
                super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
            finally:
                self._parameters["weight"] = stash

    def _quantize_weight(self, weight: torch.Tensor) -> None:
        """`bnb.functional.double_quant(weight.half().cuda())` -> CB, SCB (:69-77), on the module's GPU."""
        dev = self.weight.device
        if dev.type != "cuda":
            # the reference cannot even construct this module off-GPU; keep the float weight until `.to(cuda)`
            self._pending_fp = weight.detach()
            return
        w = weight.detach().to(device=dev)
        if w.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            w = w.float()
        CB, SCB = ops.int8_quant_rows(w.contiguous())
        self.weight.data = CB
        setattr(self.weight, "CB", CB)
        setattr(self.weight, "SCB", SCB)
        self._pending_fp = None
        self._stream = None

    def _apply(self, fn, recurse=True):
        pending = self._pending_fp
        out = super()._apply(fn, recurse)
        if pending is not None and self.weight.device.type == "cuda":
            self._quantize_weight(pending.to(self.weight.device))
        return out

    def weight_stream(self, R: int = 1) -> torch.Tensor:
        CB = self.weight.CB
        key = (CB.data_ptr(), CB._version, R)
        if self._stream is None or self._stream_key != key:
            self._stream = ops.repack_i8(CB, None, R)
            self._stream_key = key
        return self._stream

    def forward(self, x):
        nat.require_gpu(x, "Linear8bitLt.forward")
        if not hasattr(self.weight, "CB"):
            raise nat.NativeError("Linear8bitLt: weight is not quantised (module was never moved to the GPU)")
        x2d = x.reshape(-1, x.shape[-1])
        if x2d.stride(-1) != 1:
            x2d = x2d.contiguous()
        y = ops.linear_int8(
            x2d, self.weight_stream(1), self.weight.SCB, 1, self.out_features, self.in_features,
            threshold=self.threshold, bias=self.bias, out_dtype=x.dtype,
        )
        return y.view(*x.shape[:-1], self.out_features)


def qlinear_4bit_weight(inp: torch.Tensor, weight: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor) -> torch.Tensor:
    """Module-level entry of the reference's Triton kernel (lit_llama/quantization.py:284-333): inp [..., K] times the 4-bit
    ColBlock weight `weight` ([N, K / 2] uint8, the `quant_weight` buffer) with one (scale, zero) pair per output row
    (`scales`, `zeros` of shape [N, 1]).  Here: the exact generic kernel on the reference layout, no repack
    (`ColBlockQuantizedLinear.forward` streams the repacked weights through the MFMA kernels instead)."""
    nat.require_gpu(inp, "qlinear_4bit_weight")
    N, K = weight.shape[0], inp.shape[-1]
    assert weight.shape[1] * 2 == K, "incompatible dimensions"
    assert scales.shape == (N, 1) and zeros.shape == (N, 1)
    y = ops.linear_colblock(inp.reshape(-1, K), weight, scales, zeros, 4, K, None, K)
    return y.view(*inp.shape[:-1], N)


def __getattr__(name: str):
    # `from lit_llama.quantization import GPTQQuantizer` (quantize/gptq.py:17): the quantiser lives in lit_llama_amd/gptq.py,
    # which imports this module — resolved on first use
    if name == "GPTQQuantizer":
        from .gptq import GPTQQuantizer

        return GPTQQuantizer
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")

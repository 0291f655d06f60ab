"""Tensor-level wrappers over the C ABI (one call = one or two HIP kernel launches on the current stream).

PyTorch is used for device memory and streams only; nothing here computes on the host.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Tuple

import torch

from . import _native as nat
from ._native import (BF16, EPI_ACCUM, EPI_STORE, EPI_SWIGLU, F32, W_BF16, W_I8, W_Q4, AttnArgs, Int8Args, LinearArgs,
                      check, dtype_code, lib, ptr, require_gpu, stream_ptr)

MAX_M = 16


def fast_linear_max_m(K: int, R: int, fmt: int = W_Q4, waves: int = 8) -> int:
    """Largest M whose staged activations fit the 160 KiB LDS of one workgroup (mi355_linear_max_rows: the same
    arithmetic as the launchers in gemv.hip / int8.hip)."""
    return max(0, min(MAX_M, int(lib().mi355_linear_max_rows(fmt, K, R, waves))))


# ------------------------------------------------------------------------------------------ repack
def packed_bytes(fmt: int, N: int, K: int, R: int, pair: bool) -> int:
    n = int(lib().mi355_packed_bytes(fmt, N, K, R, 1 if pair else 0))
    if n <= 0:
        raise nat.NativeError(f"packed_bytes: bad shape fmt={fmt} N={N} K={K} R={R}")
    return n


def repack_q4(q0: torch.Tensor, q1: Optional[torch.Tensor], N: int, K: int, R: int,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """quant_weight [N, K/2] uint8 (any strides) -> Q4 stream (uint8 1-D; `out`: a slice of a caller's arena)."""
    require_gpu(q0, "repack_q4")
    assert q0.dtype == torch.uint8 and q0.shape == (N, K // 2)
    if q1 is not None:
        assert q1.dtype == torch.uint8 and q1.shape == q0.shape and q1.stride() == q0.stride()
    nbytes = packed_bytes(W_Q4, N, K, R, q1 is not None)
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=q0.device)
    elif out.dtype != torch.uint8 or out.numel() != nbytes or not out.is_contiguous() or out.device != q0.device:
        raise nat.NativeError(f"repack_q4: `out` must be a contiguous uint8 tensor of {nbytes} bytes on {q0.device}")
    check(lib().mi355_q4_repack(ptr(q0), ptr(q1), q0.stride(0), q0.stride(1), N, K, R, ptr(out), stream_ptr()),
          "mi355_q4_repack")
    return out


def unpack_q4_stream(stream: torch.Tensor, N: int, K: int, R: int, pair: bool, which: int = 0) -> torch.Tensor:
    """Inverse of `repack_q4` for ONE matrix of a Q4 stream: the reference layout `quant_weight` [N, K / 2] uint8 with
    stride (1, N), byte j of row n = q[n, 2j] | q[n, 2j + 1] << 4 (lit_llama/quantization.py:350-359, 387-390).
    Plain tensor ops on the stream's device (a state_dict() after the reference-layout buffers were released, not a hot
    path).  Stream: [tile][unit][r][lane = 16 g + row][dword d]; nibble p of dword d holds
    k = 128 u + 32 g + 8 d + j with j = 2 (p & 3) + (p >> 2); `pair`: r selects the matrix (`which`), else the row group."""
    assert stream.dtype == torch.uint8 and stream.dim() == 1 and K % 2 == 0
    units = (K + 127) // 128
    rows_per_tile = 16 if pair else 16 * R
    tiles = (N + rows_per_tile - 1) // rows_per_tile
    assert stream.numel() == tiles * units * R * 1024, "stream size does not match (N, K, R, pair)"
    b = stream.view(tiles, units, R, 64, 4, 4)
    if pair:
        b = b[:, :, which:which + 1]
    nib = torch.stack((b & 15, b >> 4), dim=-1).reshape(*b.shape[:4], 4, 8)  # nibble p = 2 * byte + half
    perm = torch.tensor([(j >> 1) + 4 * (j & 1) for j in range(8)], device=stream.device)
    q = nib.index_select(-1, perm)                                            # [tile, unit, r, lane, d, j]
    Rr = q.shape[2]
    q = q.view(tiles, units, Rr, 4, 16, 4, 8).permute(0, 2, 4, 1, 3, 5, 6)    # [tile, r, row, unit, g, d, j]
    q = q.reshape(tiles * Rr * 16, units * 128)[:N, :K]
    packed = q[:, 0::2] | (q[:, 1::2] << 4)
    return packed.t().contiguous().t()


def repack_bf16(w0: torch.Tensor, w1: Optional[torch.Tensor], R: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[N, K] f32 / bf16 weights -> BF16 stream (`out`: a slice of a caller's arena, as for `repack_q4`)."""
    require_gpu(w0, "repack_bf16")
    N, K = w0.shape
    w0 = w0.contiguous()
    if w1 is not None:
        w1 = w1.contiguous()
        assert w1.shape == w0.shape and w1.dtype == w0.dtype
    nbytes = packed_bytes(W_BF16, N, K, R, w1 is not None)
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=w0.device)
    elif out.dtype != torch.uint8 or out.numel() != nbytes or not out.is_contiguous() or out.device != w0.device:
        raise nat.NativeError(f"repack_bf16: `out` must be a contiguous uint8 tensor of {nbytes} bytes on {w0.device}")
    check(lib().mi355_bf16_repack(ptr(w0), ptr(w1), dtype_code(w0.dtype), N, K, R, ptr(out), stream_ptr()),
          "mi355_bf16_repack")
    return out


def repack_i8(c0: torch.Tensor, c1: Optional[torch.Tensor], R: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    require_gpu(c0, "repack_i8")
    N, K = c0.shape
    assert c0.dtype == torch.int8
    c0 = c0.contiguous()
    if c1 is not None:
        c1 = c1.contiguous()
    nbytes = packed_bytes(W_I8, N, K, R, c1 is not None)
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=c0.device)
    elif out.dtype != torch.uint8 or out.numel() != nbytes or not out.is_contiguous() or out.device != c0.device:
        raise nat.NativeError(f"repack_i8: `out` must be a contiguous uint8 tensor of {nbytes} bytes on {c0.device}")
    check(lib().mi355_i8_repack(ptr(c0), ptr(c1), N, K, R, ptr(out), stream_ptr()), "mi355_i8_repack")
    return out


def repack_u8(q0: torch.Tensor, q1: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """uint8 levels [N, K] of an 8-bit ColBlockQuantizedLinear (any strides: the reference keeps quant_weight column-major) -> the stream
    of mi355_fused_step's weight_fmt 6 (include/mi355_llama.h mi355_u8_repack); q1: the c_fc2 half of the c_fc1 / c_fc2 pair."""
    N, K = q0.shape
    if q0.dtype != torch.uint8 or q0.device.type != "cuda" or (q1 is not None and (q1.shape != q0.shape or q1.dtype != torch.uint8
                                                                                     or q1.stride() != q0.stride() or q1.device != q0.device)):
        raise nat.NativeError("repack_u8: uint8 CUDA matrices of one shape and one stride pattern")
    nbytes = N * K * (2 if q1 is not None else 1)
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=q0.device)
    elif out.dtype != torch.uint8 or out.numel() != nbytes or not out.is_contiguous() or out.device != q0.device:
        raise nat.NativeError(f"repack_u8: `out` must be a contiguous uint8 tensor of {nbytes} bytes on {q0.device}")
    check(lib().mi355_u8_repack(ptr(q0), ptr(q1), q0.stride(0), q0.stride(1), N, K, 2 if q1 is not None else 1, ptr(out), stream_ptr()),
          "mi355_u8_repack")
    return out


# ------------------------------------------------------------------------------------------ linears
def linear_fast(
    x2d: torch.Tensor,
    stream: torch.Tensor,
    fmt: int,
    R: int,
    N: int,
    K: int,
    *,
    scales: Optional[torch.Tensor] = None,
    zeros: Optional[torch.Tensor] = None,
    scales2: Optional[torch.Tensor] = None,
    zeros2: Optional[torch.Tensor] = None,
    norm_scale: Optional[torch.Tensor] = None,
    eps: float = 1e-5,
    bias: Optional[torch.Tensor] = None,
    epi: int = EPI_STORE,
    out: Optional[torch.Tensor] = None,
    out_dtype: Optional[torch.dtype] = None,
    waves: int = 0,
    grid: int = 0,
    prefetch: int = 0,
    flags: int = 0,
    attn_partials: Optional[torch.Tensor] = None,
    group_cols: int = 0,
) -> torch.Tensor:
    """y[M, N] = epi(x2d[M, K] . W^T) through the MFMA weight-streaming kernel; M is chunked to fit LDS.
    `group_cols` (Q4): input columns per (scale, zero) pair; scales / zeros are then [N, ceil(K / group_cols)] bf16.
    With `attn_partials` ([M, heads, splits, hs + 4] f32 records of a split attention) the activations are the
    combined attention output and x2d only provides M / dtype / device (it is not read)."""
    require_gpu(x2d, "linear_fast")
    assert x2d.dim() == 2 and x2d.shape[1] == K and x2d.stride(1) == 1
    M = x2d.shape[0]
    if out is None:
        assert epi != EPI_ACCUM, "accumulate epilogue needs `out`"
        out = torch.empty((M, N), dtype=out_dtype or x2d.dtype, device=x2d.device)
    assert out.shape == (M, N) and out.stride(1) == 1
    a = LinearArgs()
    a.fmt, a.R, a.w, a.N, a.K = fmt, R, ptr(stream), N, K
    a.x_dtype = dtype_code(x2d.dtype)
    a.ldx = x2d.stride(0)
    a.norm_scale = ptr(norm_scale)
    a.norm_dtype = dtype_code(norm_scale.dtype) if norm_scale is not None else F32
    a.eps = eps
    a.scales, a.zeros, a.scales2, a.zeros2 = ptr(scales), ptr(zeros), ptr(scales2), ptr(zeros2)
    a.sz_dtype = dtype_code(scales.dtype) if scales is not None else (dtype_code(bias.dtype) if bias is not None else F32)
    a.epi = epi
    a.bias = ptr(bias)
    a.y_dtype = dtype_code(out.dtype)
    a.ldy = out.stride(0)
    a.waves, a.grid, a.prefetch, a.flags = waves, grid, prefetch, flags
    a.group_cols = group_cols if 0 < group_cols < K else 0
    step = fast_linear_max_m(K, R, fmt, waves or 8)
    if a.group_cols:
        # the tile's (scale, zero) table sits behind the activation rows in LDS (2 x 16 R x groups dwords)
        assert scales is not None and scales.dtype == torch.bfloat16 and scales.is_contiguous() and zeros.is_contiguous()
        n_groups = -(-K // a.group_cols)
        row_bytes = (-(-K // 128) + 1) * 256 + 16
        step = max(1, min(step, step - -(-(2 * 16 * R * n_groups * 4 + 16) // row_bytes)))
    if step < 1:
        raise nat.NativeError(f"linear_fast: K={K} does not fit LDS even for M=1")
    s = stream_ptr()
    esz_x, esz_y = x2d.element_size(), out.element_size()
    if attn_partials is not None:
        assert attn_partials.dtype == torch.float32 and attn_partials.is_contiguous() and attn_partials.shape[0] == M
        _, a.attn_heads, a.attn_splits, rec = attn_partials.shape
        a.attn_hs = rec - 4
    for m0 in range(0, M, step):
        a.M = min(step, M - m0)
        a.x = x2d.data_ptr() + m0 * x2d.stride(0) * esz_x
        a.y = out.data_ptr() + m0 * out.stride(0) * esz_y
        if attn_partials is not None:
            a.attn_partials = attn_partials.data_ptr() + m0 * attn_partials.stride(0) * 4
        check(lib().mi355_linear_fast(C.byref(a), s), "mi355_linear_fast")
    return out


_GEMM_WS = {}  # device -> scratch tensor of the wide linear (staged bf16 operands + row statistics)


def linear_gemm(x2d: torch.Tensor, stream: torch.Tensor, R: int, N: int, K: int, *, scales: Optional[torch.Tensor] = None,
                zeros: Optional[torch.Tensor] = None, scales2: Optional[torch.Tensor] = None, zeros2: Optional[torch.Tensor] = None,
                norm_scale: Optional[torch.Tensor] = None, eps: float = 1e-5, epi: int = EPI_STORE,
                out: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None, fmt: int = W_Q4,
                group_cols: int = 0) -> torch.Tensor:
    """y[M, N] = epi(x2d[M, K] . W^T) for WIDE inputs through the LDS-tiled MFMA GEMM over the Q4 stream (scales / zeros
    per output row, or [N, ceil(K / group_cols)] bf16 tables with `group_cols`) or the BF16 stream (`fmt=W_BF16`,
    unquantised weights) — mi355_linear_gemm, csrc/gemm.hip: prompt prefill / no-cache evaluation
    (evaluate/full.py:120-129)."""
    assert fmt in (W_Q4, W_BF16) and (fmt == W_BF16 or (scales is not None and zeros is not None))
    require_gpu(x2d, "linear_gemm")
    assert x2d.dim() == 2 and x2d.shape[1] == K and x2d.stride(1) == 1
    M = x2d.shape[0]
    if out is None:
        assert epi != EPI_ACCUM, "accumulate epilogue needs `out`"
        out = torch.empty((M, N), dtype=out_dtype or x2d.dtype, device=x2d.device)
    assert out.shape == (M, N) and out.stride(1) == 1
    need = int(lib().mi355_linear_gemm_workspace_bytes(M, K))
    ws = _GEMM_WS.get(x2d.device)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 22), dtype=torch.uint8, device=x2d.device)
        _GEMM_WS[x2d.device] = ws
    a = LinearArgs()
    a.fmt, a.R, a.w, a.N, a.K, a.M = fmt, R, ptr(stream), N, K, M
    a.x, a.x_dtype, a.ldx = ptr(x2d), dtype_code(x2d.dtype), x2d.stride(0)
    a.norm_scale = ptr(norm_scale)
    a.norm_dtype = dtype_code(norm_scale.dtype) if norm_scale is not None else F32
    a.eps = eps
    a.scales, a.zeros, a.scales2, a.zeros2 = ptr(scales), ptr(zeros), ptr(scales2), ptr(zeros2)
    a.sz_dtype = dtype_code(scales.dtype) if scales is not None else BF16
    a.group_cols = group_cols if (fmt == W_Q4 and 0 < group_cols < K) else 0
    a.epi = epi
    a.y, a.y_dtype, a.ldy = ptr(out), dtype_code(out.dtype), out.stride(0)
    check(lib().mi355_linear_gemm(C.byref(a), ptr(ws), ws.numel(), stream_ptr()), "mi355_linear_gemm")
    return out


def linear_int8_gemm(x2d: torch.Tensor, stream: torch.Tensor, scb: torch.Tensor, R: int, N: int, K: int, *,
                     scb2: Optional[torch.Tensor] = None, norm_scale: Optional[torch.Tensor] = None, eps: float = 1e-5,
                     threshold: float = 6.0, epi: int = EPI_STORE, out: Optional[torch.Tensor] = None,
                     out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """LLM.int8 linear for WIDE inputs (mi355_linear_int8_gemm, csrc/int8_gemm.hip): outlier columns over all rows of the
    call (what MatMul8bitLt does), int8 product on the MFMA over 128-row blocks."""
    require_gpu(x2d, "linear_int8_gemm")
    assert x2d.dim() == 2 and x2d.shape[1] == K and x2d.stride(1) == 1 and scb.dtype == torch.float32
    M = x2d.shape[0]
    if out is None:
        assert epi != EPI_ACCUM
        out = torch.empty((M, N), dtype=out_dtype or x2d.dtype, device=x2d.device)
    need = int(lib().mi355_linear_int8_gemm_workspace_bytes(M, K))
    ws = _GEMM_WS.get(x2d.device)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 22), dtype=torch.uint8, device=x2d.device)
        _GEMM_WS[x2d.device] = ws
    a = Int8Args()
    a.w, a.scb, a.scb2, a.N, a.K, a.M, a.R = ptr(stream), ptr(scb), ptr(scb2), N, K, M, R
    a.x, a.x_dtype, a.ldx = ptr(x2d), dtype_code(x2d.dtype), x2d.stride(0)
    a.norm_scale = ptr(norm_scale)
    a.norm_dtype = dtype_code(norm_scale.dtype) if norm_scale is not None else F32
    a.eps, a.threshold, a.epi = eps, threshold, epi
    a.y, a.y_dtype, a.ldy = ptr(out), dtype_code(out.dtype), out.stride(0)
    check(lib().mi355_linear_int8_gemm(C.byref(a), ptr(ws), ws.numel(), stream_ptr()), "mi355_linear_int8_gemm")
    return out


def linear_int8(
    x2d: torch.Tensor,
    stream: torch.Tensor,
    scb: torch.Tensor,
    R: int,
    N: int,
    K: int,
    *,
    scb2: Optional[torch.Tensor] = None,
    norm_scale: Optional[torch.Tensor] = None,
    eps: float = 1e-5,
    threshold: float = 6.0,
    bias: Optional[torch.Tensor] = None,
    epi: int = EPI_STORE,
    out: Optional[torch.Tensor] = None,
    out_dtype: Optional[torch.dtype] = None,
    waves: int = 0,
    grid: int = 0,
    prefetch: int = 0,
    attn_partials: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """LLM.int8 linear (mi355_linear_int8; inputs of >= 32 rows without a bias go to linear_int8_gemm, which has no such
    deviation).  Known deviation from bitsandbytes for M > the LDS chunk (<= 16 rows):
    the rows are fed in chunks and the outlier COLUMN set (|x| >= threshold anywhere in the column) is determined
    per chunk, whereas MatMul8bitLt determines it over all B * T rows — a column that is an outlier in one chunk only
    takes the int8 path in the others.  Decode (M = 1) and prompts of up to one chunk are unaffected; prefill logits
    of longer prompts depend on the chunking at the int8 granularity (1/127 of a row's absmax)."""
    require_gpu(x2d, "linear_int8")
    assert x2d.dim() == 2 and x2d.shape[1] == K and x2d.stride(1) == 1
    assert scb.dtype == torch.float32
    M = x2d.shape[0]
    if out is None:
        assert epi != EPI_ACCUM
        out = torch.empty((M, N), dtype=out_dtype or x2d.dtype, device=x2d.device)
    a = Int8Args()
    a.w, a.scb, a.N, a.K = ptr(stream), ptr(scb), N, K
    a.x_dtype = dtype_code(x2d.dtype)
    a.ldx = x2d.stride(0)
    a.norm_scale = ptr(norm_scale)
    a.norm_dtype = dtype_code(norm_scale.dtype) if norm_scale is not None else F32
    a.eps, a.threshold, a.R = eps, threshold, R
    a.bias = ptr(bias)
    a.bias_dtype = dtype_code(bias.dtype) if bias is not None else F32
    a.epi = epi
    a.scb2 = ptr(scb2)
    a.y_dtype = dtype_code(out.dtype)
    a.ldy = out.stride(0)
    a.waves, a.grid, a.prefetch = waves, grid, prefetch
    if attn_partials is not None:
        # [1, n_head, n_split, hs + 4] f32 records of ops.attention(..., return_partials=True); x2d is only a shape
        assert attn_partials.dtype == torch.float32 and attn_partials.dim() == 4 and M == 1
        a.attn_partials = ptr(attn_partials)
        a.attn_heads, a.attn_splits, a.attn_hs = attn_partials.shape[1], attn_partials.shape[2], attn_partials.shape[3] - 4
    if M >= 32 and bias is None and attn_partials is None:
        return linear_int8_gemm(x2d, stream, scb, R, N, K, scb2=scb2, norm_scale=norm_scale, eps=eps, threshold=threshold,
                                epi=epi, out=out)
    step = fast_linear_max_m(K, R, W_I8, waves or 8)
    if step < 1:
        raise nat.NativeError(f"linear_int8: K={K} does not fit LDS even for M=1")
    s = stream_ptr()
    for m0 in range(0, M, step):
        a.M = min(step, M - m0)
        a.x = x2d.data_ptr() + m0 * x2d.stride(0) * x2d.element_size()
        a.y = out.data_ptr() + m0 * out.stride(0) * out.element_size()
        check(lib().mi355_linear_int8(C.byref(a), s), "mi355_linear_int8")
    return out


def int8_quant_rows(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """bnb.functional.double_quant(W)[0], [2]: (CB int8 [N,K], SCB f32 [N])."""
    require_gpu(w, "int8_quant_rows")
    w = w.contiguous()
    N, K = w.shape
    cb = torch.empty((N, K), dtype=torch.int8, device=w.device)
    scb = torch.empty((N,), dtype=torch.float32, device=w.device)
    check(lib().mi355_int8_quant_rows(ptr(w), dtype_code(w.dtype), N, K, ptr(cb), ptr(scb), stream_ptr()),
          "mi355_int8_quant_rows")
    return cb, scb


def linear_dense(x2d: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    require_gpu(x2d, "linear_dense")
    assert x2d.dtype == w.dtype and x2d.stride(1) == 1
    w = w.contiguous()
    M, K = x2d.shape
    N = w.shape[0]
    if bias is not None:
        bias = bias.to(x2d.dtype).contiguous()
    out = torch.empty((M, N), dtype=x2d.dtype, device=x2d.device)
    s = stream_ptr()
    for m0 in range(0, M, 32768):
        m = min(32768, M - m0)
        check(lib().mi355_linear_dense(x2d.data_ptr() + m0 * x2d.stride(0) * x2d.element_size(), x2d.stride(0), ptr(w),
                                       ptr(bias), out.data_ptr() + m0 * N * out.element_size(), N, m, N, K,
                                       dtype_code(x2d.dtype), s), "mi355_linear_dense")
    return out


def linear_colblock(x2d: torch.Tensor, qweight: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor, bits: int,
                    tile_cols: int, bias: Optional[torch.Tensor], K: int) -> torch.Tensor:
    require_gpu(x2d, "linear_colblock")
    M = x2d.shape[0]
    N = qweight.shape[0]
    scales = scales.contiguous()
    zeros = zeros.contiguous()
    assert scales.dtype == zeros.dtype
    if bias is not None:
        bias = bias.to(scales.dtype).contiguous()
    out = torch.empty((M, N), dtype=x2d.dtype, device=x2d.device)
    s = stream_ptr()
    for m0 in range(0, M, 32768):
        m = min(32768, M - m0)
        check(lib().mi355_linear_colblock(x2d.data_ptr() + m0 * x2d.stride(0) * x2d.element_size(), x2d.stride(0),
                                          ptr(qweight), qweight.stride(0), qweight.stride(1), ptr(scales), ptr(zeros),
                                          dtype_code(scales.dtype), scales.shape[1], tile_cols, bits, ptr(bias),
                                          out.data_ptr() + m0 * N * out.element_size(), N, m, N, K,
                                          dtype_code(x2d.dtype), s), "mi355_linear_colblock")
    return out


def colblock_dequant(qweight: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor, bits: int, tile_cols: int,
                     K: int, dtype: torch.dtype) -> torch.Tensor:
    require_gpu(qweight, "colblock_dequant")
    N = qweight.shape[0]
    scales = scales.contiguous()
    zeros = zeros.contiguous()
    out = torch.empty((N, K), dtype=dtype, device=qweight.device)
    check(lib().mi355_colblock_dequant(ptr(qweight), qweight.stride(0), qweight.stride(1), ptr(scales), ptr(zeros),
                                       dtype_code(scales.dtype), scales.shape[1], tile_cols, bits, ptr(out),
                                       dtype_code(dtype), N, K, stream_ptr()), "mi355_colblock_dequant")
    return out


# ------------------------------------------------------------------------------------------ small ops
def rmsnorm(x: torch.Tensor, scale: torch.Tensor, eps: float, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    require_gpu(x, "rmsnorm")
    C_ = x.shape[-1]
    x2 = x.reshape(-1, C_)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    out = torch.empty(x2.shape, dtype=out_dtype or x.dtype, device=x.device)
    check(lib().mi355_rmsnorm(ptr(x2), x2.stride(0), ptr(scale), dtype_code(scale.dtype), eps, ptr(out), C_,
                              x2.shape[0], C_, dtype_code(x2.dtype), dtype_code(out.dtype), stream_ptr()),
          "mi355_rmsnorm")
    return out.view(x.shape)


def apply_rope(x: torch.Tensor, rope: torch.Tensor) -> torch.Tensor:
    require_gpu(x, "apply_rope")
    B, T, nh, hs = x.shape
    xc = x.contiguous()
    rope = rope[:T].to(torch.float32).contiguous()
    out = torch.empty_like(xc)
    check(lib().mi355_apply_rope(ptr(xc), ptr(rope), ptr(out), B, T, nh, hs, dtype_code(x.dtype), stream_ptr()),
          "mi355_apply_rope")
    return out


def swiglu(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    require_gpu(a, "swiglu")
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a)
    check(lib().mi355_swiglu(ptr(a), ptr(b), ptr(out), a.numel(), dtype_code(a.dtype), stream_ptr()), "mi355_swiglu")
    return out


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    require_gpu(a, "add")
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a)
    check(lib().mi355_add(ptr(a), ptr(b), ptr(out), a.numel(), dtype_code(a.dtype), stream_ptr()), "mi355_add")
    return out


def embedding(idx: torch.Tensor, wte: torch.Tensor, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    require_gpu(wte, "embedding")
    assert idx.dtype in (torch.int32, torch.int64)
    flat = idx.reshape(-1).contiguous()
    wte = wte.contiguous()
    V, C_ = wte.shape
    out = torch.empty((flat.numel(), C_), dtype=out_dtype or wte.dtype, device=wte.device)
    check(lib().mi355_embedding(ptr(flat), 1 if idx.dtype == torch.int64 else 0, ptr(wte), dtype_code(wte.dtype),
                                ptr(out), dtype_code(out.dtype), flat.numel(), C_, V, stream_ptr()), "mi355_embedding")
    return out.view(*idx.shape, C_)


def argmax(logits: torch.Tensor) -> torch.Tensor:
    require_gpu(logits, "argmax")
    assert logits.dim() == 1 and logits.dtype == torch.float32
    logits = logits.contiguous()
    out = torch.empty((1,), dtype=torch.int32, device=logits.device)
    check(lib().mi355_argmax(ptr(logits), logits.numel(), ptr(out), None, None, stream_ptr()), "mi355_argmax")
    return out


def sample(logits: torch.Tensor, temperature: float, top_k: Optional[int], uniforms: torch.Tensor,
           pos: torch.Tensor, next_token: torch.Tensor, *, out_tokens: Optional[torch.Tensor] = None,
           tokens: Optional[torch.Tensor] = None, advance: bool = False, probs_out: Optional[torch.Tensor] = None) -> None:
    """Device-side tail of generate.py:68-85 (mi355_sample): temperature, exact top-k threshold, softmax and the
    inverse-CDF draw `min {i : cumsum(p)[i] > uniforms[pos[0]]}`; see csrc/sample.hip."""
    require_gpu(logits, "sample")
    assert logits.dtype == torch.float32 and logits.dim() == 1 and logits.is_contiguous()
    assert uniforms.dtype == torch.float32 and pos.dtype == torch.int32 and next_token.dtype == torch.int32
    check(lib().mi355_sample(ptr(logits), logits.numel(), float(temperature), int(top_k or 0), ptr(uniforms),
                             ptr(next_token), ptr(out_tokens), ptr(tokens), ptr(pos), 1 if advance else 0,
                             ptr(probs_out), stream_ptr()), "mi355_sample")


def attention(
    qkv: torch.Tensor,
    rope: torch.Tensor,
    n_head: int,
    *,
    pos: Optional[torch.Tensor] = None,
    kv_cache: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
    rope_gathered: bool = False,
    out_dtype: Optional[torch.dtype] = None,
    n_split: int = 1,
    return_partials: bool = False,
    adapter: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None,
) -> torch.Tensor:
    """qkv [B, T, 3C] -> y [B, T, C]; RoPE + (in-place) cache write + causal attention.  n_split > 1 spreads each
    head over n_split workgroups (flash-decoding) and combines the partial records with mi355_attn_combine, or
    returns them ([B*T, heads, n_split, hs + 4]) for a consumer that combines on the fly.  `adapter` = (ak, av, gate):
    the LLaMA-Adapter prefix term of lit_llama/adapter.py:134-151 is added (operands as `adapter_prefix` takes them)."""
    require_gpu(qkv, "attention")
    B, T, C3 = qkv.shape
    Cw = C3 // 3
    hs = Cw // n_head
    qkv = qkv.contiguous()
    assert rope.dtype == torch.float32 and rope.is_contiguous()
    y = torch.empty((B, T, Cw), dtype=out_dtype or qkv.dtype, device=qkv.device)
    a = AttnArgs()
    a.qkv, a.qkv_dtype, a.B, a.ld_qkv = ptr(qkv), dtype_code(qkv.dtype), B, C3
    a.rope = ptr(rope)
    a.T, a.n_head, a.hs = T, n_head, hs
    a.y, a.y_dtype, a.ldy = ptr(y), dtype_code(y.dtype), Cw
    a.rope_gathered = 1 if rope_gathered else 0
    keep = []
    if kv_cache is not None:
        k, v = kv_cache
        assert k.is_contiguous() and v.is_contiguous() and k.shape == v.shape and k.shape[0] == B
        assert k.shape[1] == n_head and k.shape[3] == hs and k.dtype == v.dtype
        assert pos is not None and pos.numel() == T
        p32 = pos.to(torch.int32).contiguous()
        keep.append(p32)
        a.pos, a.kcache, a.vcache, a.cache_dtype, a.S = ptr(p32), ptr(k), ptr(v), dtype_code(k.dtype), k.shape[2]
    else:
        cdt = qkv.dtype if qkv.dtype in (torch.float32, torch.bfloat16) else torch.float32
        tmp = torch.empty((2, B, n_head, T, hs), dtype=cdt, device=qkv.device)
        keep.append(tmp)
        a.kv_tmp, a.cache_dtype, a.S = ptr(tmp), dtype_code(cdt), T
    parts = None
    if n_split > 1:
        parts = torch.empty((B * T, n_head, n_split, hs + 4), dtype=torch.float32, device=qkv.device)
        a.n_split, a.partials = n_split, ptr(parts)
    if adapter is not None:
        ak, av, gate = adapter
        assert ak.dtype == av.dtype == gate.dtype == torch.float32 and ak.shape == av.shape == (n_head, ak.shape[1], hs)
        assert ak.is_contiguous() and av.is_contiguous() and gate.is_contiguous() and gate.numel() == n_head
        a.adapter_k, a.adapter_v, a.adapter_gate, a.adapter_len = ptr(ak), ptr(av), ptr(gate), ak.shape[1]
    check(lib().mi355_attention(C.byref(a), stream_ptr()), "mi355_attention")
    if parts is not None:
        if return_partials:
            return parts
        check(lib().mi355_attn_combine(ptr(parts), n_split, B * T, n_head, hs, ptr(y), dtype_code(y.dtype), Cw,
                                       stream_ptr()), "mi355_attn_combine")
    return y


def adapter_prefix(qkv: torch.Tensor, rope: torch.Tensor, n_head: int, ak: torch.Tensor, av: torch.Tensor,
                   gate: torch.Tensor, y: torch.Tensor, pos: Optional[torch.Tensor] = None,
                   rope_gathered: bool = True) -> torch.Tensor:
    """In place: y[b, t, h, :] += gate[h] * softmax(rope(q[b, t, h]) . ak[h]^T / sqrt(hs)) av[h] — the LLaMA-Adapter
    prefix term of lit_llama/adapter.py:134-151 (mi355_adapter_prefix, csrc/attention.hip).  qkv [B, T, 3 C] as
    `attention` takes it, ak / av f32 [n_head, aT, hs], gate f32 [n_head], y [B, T, C]."""
    require_gpu(qkv, "adapter_prefix")
    B, T, C3 = qkv.shape
    hs = C3 // 3 // n_head
    assert qkv.stride(-1) == 1 and qkv.stride(0) == T * qkv.stride(1) and y.shape == (B, T, n_head * hs) and y.is_contiguous()
    assert ak.dtype == av.dtype == gate.dtype == torch.float32 and ak.shape == av.shape == (n_head, ak.shape[1], hs)
    assert ak.is_contiguous() and av.is_contiguous() and gate.is_contiguous() and gate.numel() == n_head
    assert rope.dtype == torch.float32 and rope.is_contiguous()
    a = nat.AdapterArgs()
    a.qkv, a.qkv_dtype, a.B, a.ld_qkv = ptr(qkv), dtype_code(qkv.dtype), B, qkv.stride(1)
    a.rope, a.rope_gathered, a.T = ptr(rope), 1 if rope_gathered else 0, T
    if pos is not None:
        pos32 = pos.to(torch.int32).contiguous()
        a.pos = ptr(pos32)
    a.n_head, a.hs, a.aT = n_head, hs, ak.shape[1]
    a.ak, a.av, a.gate = ptr(ak), ptr(av), ptr(gate)
    a.y, a.y_dtype, a.ldy = ptr(y), dtype_code(y.dtype), y.stride(1)
    check(lib().mi355_adapter_prefix(C.byref(a), stream_ptr()), "mi355_adapter_prefix")
    return y


def kv_roll(k: torch.Tensor, v: torch.Tensor) -> None:
    require_gpu(k, "kv_roll")
    B, nh, S, hs = k.shape
    check(lib().mi355_kv_roll(ptr(k), ptr(v), dtype_code(k.dtype), B, nh, S, hs, stream_ptr()), "mi355_kv_roll")


# ------------------------------------------------------------------------------------------ GPTQ (offline)
def gptq_row_params(W: torch.Tensor, maxq: int, sym: bool = False):
    """(scale [N], zero [N]) of the rows of W [N, cols] f32 (column slices are fine): mi355_gptq_row_params."""
    require_gpu(W, "gptq_row_params")
    assert W.dtype == torch.float32 and W.dim() == 2 and W.stride(1) == 1
    N, cols = W.shape
    scale = torch.empty((N,), dtype=torch.float32, device=W.device)
    zero = torch.empty_like(scale)
    check(lib().mi355_gptq_row_params(ptr(W), W.stride(0), N, cols, int(maxq), 1 if sym else 0, ptr(scale), ptr(zero),
                                      stream_ptr()), "mi355_gptq_row_params")
    return scale, zero


def gptq_block(W1: torch.Tensor, Hinv1: torch.Tensor, scale: torch.Tensor, zero: torch.Tensor, maxq: int):
    """One block of GPTQ's inner loop (mi355_gptq_block): W1 [N, count] f32 (a column slice of W is fine),
    Hinv1 [count, count] f32, scale / zero [N] (per row) or [N, count] (per column).  Returns (Q1, Err1, Loss1)."""
    require_gpu(W1, "gptq_block")
    assert W1.dtype == torch.float32 and Hinv1.dtype == torch.float32 and W1.dim() == 2 and W1.stride(1) == 1
    N, count = W1.shape
    assert Hinv1.shape == (count, count)
    if Hinv1.stride(1) != 1:  # torch.linalg.cholesky(upper=True) returns a transposed view
        Hinv1 = Hinv1.contiguous()
    scale = scale.reshape(N, -1).float().contiguous()
    zero = zero.reshape(N, -1).float().contiguous()
    assert scale.shape == zero.shape and scale.shape[1] in (1, count)
    per_col = scale.shape[1] == count and count > 1
    Q1 = torch.empty((N, count), dtype=torch.float32, device=W1.device)
    E1 = torch.empty_like(Q1)
    L1 = torch.empty_like(Q1)
    check(lib().mi355_gptq_block(ptr(W1), W1.stride(0), N, count, ptr(Hinv1), Hinv1.stride(0), ptr(scale), ptr(zero),
                                 scale.stride(0), 1 if per_col else 0, int(maxq), ptr(Q1), ptr(E1), ptr(L1), count,
                                 stream_ptr()), "mi355_gptq_block")
    return Q1, E1, L1

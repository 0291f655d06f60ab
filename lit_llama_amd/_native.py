"""ctypes binding of libmi355llama.so (include/mi355_llama.h).

The product path has no CPU fallback: if the shared library is missing or a GPU is absent, the
operators raise.  Only the structs / prototypes live here; all arithmetic is in csrc/*.hip.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

import torch

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libmi355llama.so"

F32, BF16, F16 = 0, 1, 2
W_Q4, W_BF16, W_I8 = 0, 1, 2
EPI_STORE, EPI_ACCUM, EPI_SWIGLU = 0, 1, 2

_DTYPE_CODE = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}

c_void_p, c_int, c_int32, c_int64, c_float, c_size_t = C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_float, C.c_size_t


class NativeError(RuntimeError):
    pass


class LinearArgs(C.Structure):
    _fields_ = [
        ("fmt", c_int32), ("R", c_int32), ("w", c_void_p), ("N", c_int32), ("K", c_int32),
        ("x", c_void_p), ("x_dtype", c_int32), ("M", c_int32), ("ldx", c_int64),
        ("norm_scale", c_void_p), ("norm_dtype", c_int32), ("eps", c_float),
        ("scales", c_void_p), ("zeros", c_void_p), ("scales2", c_void_p), ("zeros2", c_void_p),
        ("sz_dtype", c_int32), ("epi", c_int32), ("bias", c_void_p),
        ("y", c_void_p), ("y_dtype", c_int32), ("group_cols", c_int32), ("ldy", c_int64),
        ("waves", c_int32), ("grid", c_int32), ("prefetch", c_int32), ("flags", c_int32),
        ("attn_partials", c_void_p), ("attn_splits", c_int32), ("attn_heads", c_int32), ("attn_hs", c_int32),
        ("reserved1", c_int32), ("debug_stamps", c_void_p),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("qkv", c_void_p), ("qkv_dtype", c_int32), ("B", c_int32), ("ld_qkv", c_int64),
        ("rope", c_void_p), ("pos", c_void_p), ("kcache", c_void_p), ("vcache", c_void_p),
        ("cache_dtype", c_int32), ("T", c_int32), ("n_head", c_int32), ("hs", c_int32),
        ("S", c_int32), ("y_dtype", c_int32), ("y", c_void_p), ("ldy", c_int64),
        ("kv_tmp", c_void_p), ("rope_gathered", c_int32), ("n_split", c_int32), ("partials", c_void_p),
        ("debug_stamps", c_void_p),
        ("adapter_k", c_void_p), ("adapter_v", c_void_p), ("adapter_gate", c_void_p),
        ("adapter_len", c_int32), ("reserved0", c_int32),
    ]


class AdapterArgs(C.Structure):
    _fields_ = [
        ("qkv", c_void_p), ("qkv_dtype", c_int32), ("B", c_int32), ("ld_qkv", c_int64),
        ("rope", c_void_p), ("pos", c_void_p), ("rope_gathered", c_int32), ("T", c_int32),
        ("n_head", c_int32), ("hs", c_int32), ("aT", c_int32), ("y_dtype", c_int32),
        ("ak", c_void_p), ("av", c_void_p), ("gate", c_void_p), ("y", c_void_p), ("ldy", c_int64),
    ]


class Int8Args(C.Structure):
    _fields_ = [
        ("w", c_void_p), ("scb", c_void_p), ("N", c_int32), ("K", c_int32),
        ("x", c_void_p), ("x_dtype", c_int32), ("M", c_int32), ("ldx", c_int64),
        ("norm_scale", c_void_p), ("norm_dtype", c_int32), ("eps", c_float),
        ("threshold", c_float), ("R", c_int32), ("bias", c_void_p), ("bias_dtype", c_int32),
        ("epi", c_int32), ("scb2", c_void_p), ("y", c_void_p), ("y_dtype", c_int32),
        ("waves", c_int32), ("ldy", c_int64), ("grid", c_int32), ("prefetch", c_int32),
        ("debug_stamps", c_void_p),
        ("attn_partials", c_void_p), ("attn_splits", c_int32), ("attn_heads", c_int32), ("attn_hs", c_int32),
        ("reserved0", c_int32),
    ]


class Weight(C.Structure):
    _fields_ = [
        ("fmt", c_int32), ("R", c_int32), ("w", c_void_p), ("N", c_int32), ("K", c_int32),
        ("scales", c_void_p), ("zeros", c_void_p), ("scales2", c_void_p), ("zeros2", c_void_p),
        ("scb", c_void_p), ("scb2", c_void_p),
        ("sz_dtype", c_int32), ("waves", c_int32), ("grid", c_int32), ("prefetch", c_int32),
        ("flags", c_int32), ("group_cols", c_int32),
    ]


class Layer(C.Structure):
    _fields_ = [
        ("rms1", c_void_p), ("rms2", c_void_p),
        ("attn", Weight), ("proj", Weight), ("fc", Weight), ("mproj", Weight),
        ("kcache", c_void_p), ("vcache", c_void_p),
        ("adapter_k", c_void_p), ("adapter_v", c_void_p), ("adapter_gate", c_void_p),
        ("adapter_len", c_int32), ("reserved0", c_int32),
    ]


class Model(C.Structure):
    _fields_ = [
        ("n_layer", c_int32), ("n_head", c_int32), ("n_embd", c_int32), ("hs", c_int32),
        ("n_hidden", c_int32), ("vocab", c_int32), ("S", c_int32), ("block_size", c_int32),
        ("param_dtype", c_int32), ("cache_dtype", c_int32), ("tp_world", c_int32), ("max_T", c_int32),
        ("eps", c_float), ("int8_threshold", c_float),
        ("wte", c_void_p), ("ln_f", c_void_p), ("lm_head", Weight), ("rope", c_void_p),
        ("layers", C.POINTER(Layer)),
        ("x", c_void_p), ("qkv", c_void_p), ("att", c_void_p), ("hbuf", c_void_p),
        ("partial", c_void_p), ("logits", c_void_p),
        ("tokens", c_void_p), ("pos", c_void_p), ("next_token", c_void_p), ("out_tokens", c_void_p),
        ("attn_part", c_void_p), ("attn_splits", c_int32), ("reserved0", c_int32),
        ("gemm_ws", c_void_p), ("gemm_ws_bytes", C.c_uint64),
    ]


class FusedStepArgs(C.Structure):
    _fields_ = [
        ("w", c_void_p), ("layer_stride", C.c_uint64),
        ("off_attn", C.c_uint32), ("off_proj", C.c_uint32), ("off_fc", C.c_uint32), ("off_mproj", C.c_uint32),
        ("layer_bytes", C.c_uint32), ("head_bytes", C.c_uint32),
        ("w_head", c_void_p), ("sz", c_void_p), ("sz_head", c_void_p), ("norms", c_void_p), ("wte", c_void_p),
        ("rope", c_void_p), ("kv", c_void_p),
        ("tokens", c_void_p), ("pos", c_void_p), ("next_token", c_void_p), ("out_tokens", c_void_p),
        ("logits", c_void_p), ("workspace", c_void_p), ("debug_stamps", c_void_p),
        ("n_layer", c_int32), ("n_head", c_int32), ("n_embd", c_int32), ("hs", c_int32),
        ("n_hidden", c_int32), ("vocab", c_int32), ("S", c_int32), ("mode", c_int32),
        ("eps", c_float), ("reserved0", c_int32),
        ("group_cols", c_int32), ("weight_fmt", c_int32), ("gt", c_void_p), ("gt_head", c_void_p),
        ("gt_layer_stride", C.c_uint64),
    ]


class TpComm(C.Structure):
    _fields_ = [
        ("world", c_int32), ("rank", c_int32), ("slot_floats", c_int32), ("reserved0", c_int32),
        ("peer_buf", c_void_p * 8), ("state", c_void_p),
    ]


# name -> (restype, argtypes); must list every function declared in include/mi355_llama.h
PROTOTYPES = {
    "mi355_version": (c_int, []),
    "mi355_last_error": (C.c_char_p, []),
    "mi355_num_cus": (c_int, []),
    "mi355_packed_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "mi355_q4_repack": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mi355_bf16_repack": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mi355_i8_repack": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mi355_u8_repack": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mi355_linear_fast": (c_int, [C.POINTER(LinearArgs), c_void_p]),
    "mi355_linear_gemm_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mi355_linear_gemm": (c_int, [C.POINTER(LinearArgs), c_void_p, c_size_t, c_void_p]),
    "mi355_linear_fast_batch": (c_int, [C.POINTER(LinearArgs), c_int, c_void_p]),
    "mi355_linear_dense": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                   c_int, c_void_p]),
    "mi355_linear_colblock": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int, c_int,
                                      c_int, c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                                      c_void_p]),
    "mi355_colblock_dequant": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                       c_void_p, c_int, c_int, c_int, c_void_p]),
    "mi355_rmsnorm": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_float, c_void_p, c_int64, c_int, c_int, c_int,
                              c_int, c_void_p]),
    "mi355_apply_rope": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mi355_swiglu": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mi355_add": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mi355_embedding": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mi355_argmax": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mi355_attention": (c_int, [C.POINTER(AttnArgs), c_void_p]),
    "mi355_adapter_prefix": (c_int, [C.POINTER(AdapterArgs), c_void_p]),
    "mi355_attn_combine": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int64, c_void_p]),
    "mi355_kv_roll": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mi355_int8_quant_rows": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "mi355_linear_int8": (c_int, [C.POINTER(Int8Args), c_void_p]),
    "mi355_linear_int8_gemm_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mi355_linear_int8_gemm": (c_int, [C.POINTER(Int8Args), c_void_p, c_size_t, c_void_p]),
    "mi355_set_step": (c_int, [C.POINTER(Model), c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mi355_forward": (c_int, [C.POINTER(Model), c_int, c_int, c_int, c_void_p]),
    "mi355_forward_embed": (c_int, [C.POINTER(Model), c_int, c_void_p]),
    "mi355_forward_segment": (c_int, [C.POINTER(Model), c_int, c_int, c_int, c_int, c_void_p]),
    "mi355_residual_add": (c_int, [C.POINTER(Model), c_int, c_void_p]),
    "mi355_forward_head": (c_int, [C.POINTER(Model), c_int, c_int, c_int, c_void_p]),
    "mi355_graph_capture": (c_int, [C.POINTER(Model), c_int, c_void_p, C.POINTER(c_void_p)]),
    "mi355_graph_launch": (c_int, [c_void_p, c_void_p]),
    "mi355_graph_destroy": (c_int, [c_void_p]),
    "mi355_fused_step_workspace_bytes": (c_size_t, [c_int]),
    "mi355_fused_step_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "mi355_fused_step": (c_int, [C.POINTER(FusedStepArgs), c_void_p]),
    "mi355_tp_comm_bytes": (c_size_t, [c_int, c_int]),
    "mi355_tp_buffer_alloc": (c_int, [c_size_t, C.POINTER(c_void_p)]),
    "mi355_tp_buffer_free": (c_int, [c_void_p]),
    "mi355_ipc_export": (c_int, [c_void_p, c_void_p]),
    "mi355_ipc_open": (c_int, [c_void_p, C.POINTER(c_void_p)]),
    "mi355_ipc_close": (c_int, [c_void_p]),
    "mi355_tp_step_begin": (c_int, [C.POINTER(TpComm), c_void_p]),
    "mi355_tp_allreduce": (c_int, [C.POINTER(TpComm), c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mi355_tp_argmax": (c_int, [C.POINTER(TpComm), c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                c_void_p]),
    "mi355_graph_begin": (c_int, [c_void_p]),
    "mi355_graph_end": (c_int, [c_void_p, C.POINTER(c_void_p)]),
    "mi355_sample": (c_int, [c_void_p, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                             c_void_p, c_void_p]),
    "mi355_sizeof": (c_int, [c_int]),
    "mi355_linear_max_rows": (c_int, [c_int, c_int, c_int, c_int]),
    "mi355_debug_time_next_launch": (c_int, [c_void_p, c_void_p]),
    "mi355_gptq_row_params": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "mi355_gptq_block": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                 c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
}

ABI_STRUCTS = [LinearArgs, AttnArgs, Int8Args, Weight, Layer, Model, FusedStepArgs, TpComm]

_lib: Optional[C.CDLL] = None


def _preload_hip_runtime() -> None:
    """Bind to the HIP runtime instance PyTorch ships (same soname libamdhip64.so.7 as /opt/rocm's)."""
    cand = Path(torch.__file__).resolve().parent / "lib" / "libamdhip64.so"
    if cand.exists():
        C.CDLL(str(cand), mode=C.RTLD_GLOBAL)


def lib() -> C.CDLL:
    """The loaded library; raises NativeError (never falls back) if it is not built."""
    global _lib
    if _lib is None:
        path = Path(os.environ.get("MI355_LLAMA_LIB", LIB_PATH))
        if not path.exists():
            raise NativeError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the hot path."
            )
        _preload_hip_runtime()
        try:
            handle = C.CDLL(str(path))
        except OSError as e:  # pragma: no cover
            raise NativeError(f"cannot load {path}: {e}") from e
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        if handle.mi355_version() != 1:
            raise NativeError("libmi355llama ABI version mismatch; rebuild")
        for i, st in enumerate(ABI_STRUCTS):
            if handle.mi355_sizeof(i) != C.sizeof(st):
                raise NativeError(f"ABI struct {st.__name__}: C sizeof {handle.mi355_sizeof(i)} != ctypes {C.sizeof(st)}")
        _lib = handle
    return _lib


E_ARG, E_SHAPE, E_DTYPE, E_STATE = -1, -2, -3, -4  # include/mi355_llama.h MI355_E_*


def last_error() -> str:
    return lib().mi355_last_error().decode(errors="replace")


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise NativeError(f"{what or 'libmi355llama'} failed (rc={rc}): {last_error()}")


def require_gpu(t: torch.Tensor, what: str) -> None:
    if t.device.type != "cuda":
        raise NativeError(
            f"{what}: tensor is on {t.device}; the MI355X hot path runs on the GPU only "
            "(the CPU restatement of the reference lives in oracle/ and is test infrastructure)."
        )


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DTYPE_CODE[dt]
    except KeyError:
        raise NativeError(f"unsupported dtype {dt}") from None


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream_ptr(stream: Optional["torch.cuda.Stream"] = None) -> Optional[int]:
    s = stream if stream is not None else torch.cuda.current_stream()
    return s.cuda_stream or None


def num_cus() -> int:
    return int(lib().mi355_num_cus())

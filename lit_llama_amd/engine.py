"""Native whole-forward engine behind `LLaMA.forward` (B = 1, KV cache).

Builds the `mi355_model` descriptor (include/mi355_llama.h) from the module's parameters / buffers:
repacked weight streams, scratch, RoPE table, in-place KV caches; `forward` then costs one tiny
`mi355_set_step` launch plus either a hipGraph replay (T == 1) or one `mi355_forward` call per chunk of
<= max_T prompt tokens.  Replaces the per-token Python walk of generate.py:63-91 -> model.py:76-122.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import torch
import torch.nn as nn

from . import _native as nat
from . import ops
from ._native import BF16, F32, W_BF16, W_I8, W_Q4, Layer, Model, Weight, check, dtype_code, lib, ptr



class _UncachedBytes:
    """Device bytes from `mi355_tp_buffer_alloc` (hipDeviceMallocUncached), seen by torch through the CUDA array interface."""

    def __init__(self, nbytes: int):
        buf = C.c_void_p()
        check(lib().mi355_tp_buffer_alloc(nbytes, C.byref(buf)), "mi355_tp_buffer_alloc")
        self.ptr, self.nbytes = buf.value, nbytes
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2}

    def __del__(self):
        try:
            lib().mi355_tp_buffer_free(C.c_void_p(self.ptr))
        except Exception:
            pass


def _handoff_workspace(nbytes: int, dev) -> torch.Tensor:
    """The persistent step's hand-off workspace, zeroed, in UNCACHED device memory: the all-gathers of the step run over it 3-7 %
    faster than over a plain allocation in the microbenchmark (profiles/r06_allgather_scalar_publish_microbench.txt) and the 7B
    int4 step 870-874 -> 860-863 us (profiles/r06_ab4_uncached_workspace.txt).  MI355_FUSED_WS_UNCACHED=0: a torch allocation."""
    if os.environ.get("MI355_FUSED_WS_UNCACHED", "1") != "0":
        try:
            with torch.cuda.device(dev):
                owner = _UncachedBytes(nbytes)
                t = torch.as_tensor(owner, device=dev)
            if t.data_ptr() != owner.ptr:
                raise RuntimeError("the uncached hand-off workspace was copied instead of wrapped")
            t._mi355_owner = owner  # keeps the allocation alive as long as the tensor
            return t
        except Exception as e:  # (a slower workspace, not a different result: say so and go on)
            import warnings

            warnings.warn(f"uncached hand-off workspace unavailable ({e}); using a plain allocation")
    return torch.zeros(nbytes, dtype=torch.uint8, device=dev)


class EngineUnavailable(RuntimeError):
    pass


def _env_int(name: str, default: int) -> int:
    try:
        return int(os.environ.get(name, default))
    except ValueError:
        return default


class _PackedWeight:
    """One entry of the model descriptor + the tensors that must stay alive for it."""

    def __init__(self):
        self.desc = Weight()
        self.keep: List[torch.Tensor] = []
        self.stream_bytes = 0
        self.side_bytes = 0
        self.q4_stream, self.q4_mods = None, []  # Q4: the stream and the module(s) whose weights it holds


def _kind(mod: nn.Module) -> str:
    from .quantization import ColBlockQuantizedLinear, Linear8bitLt

    # FIRST: LLaMA-Adapter v2 attaches adapter_scale / adapter_bias to every nn.Linear — Linear8bitLt subclasses nn.Linear, so a
    # quantised v2 model carries the pair too; the engine's streams have no such epilogue (advisor r3: the i8 check used to win)
    if getattr(mod, "adapter_scale", None) is not None:
        raise EngineUnavailable("LLaMA-Adapter v2 scale / bias on a linear: such models run op by op")
    if isinstance(mod, ColBlockQuantizedLinear):
        # 8-bit ColBlock (`--quantize gptq.int8`): the reference dequantises the whole matrix into the INPUT's dtype on every forward
        # call and runs a dense linear on it (lit_llama/quantization.py:376-423: get_weight(dtype=inp.dtype), F.linear).  Here that
        # bf16 matrix is built ONCE, at engine build, and streamed like any unquantised bf16 linear (`_dense_weight`): the same weight
        # values bit for bit (q - zero exact in bf16, one rounding of the product with the scale), 2 bytes per weight resident in HBM.
        if mod.bits == 4:
            return "q4"
        if mod.bits == 8:
            return "bf16"
        # (the reference constructor also takes bits 1 and 2; mi355_colblock_dequant handles 4 and 8: such models run op by op)
        raise EngineUnavailable(f"ColBlockQuantizedLinear with bits={mod.bits}: the engine streams 4-bit and (dequantised) 8-bit matrices")
    if isinstance(mod, Linear8bitLt):
        return "i8"
    if type(mod) is nn.Linear or getattr(mod, "_mi355_plain_weight", False):
        return "bf16"
    if hasattr(mod, "lora_A"):
        raise EngineUnavailable("LoRA update not merged into the weight yet: call model.eval()")
    raise EngineUnavailable(f"unsupported linear type {type(mod).__name__}")


def _dense_weight(mod: nn.Module) -> torch.Tensor:
    """The [N, K] matrix of a linear the engine streams as BF16: the parameter itself, or an 8-bit ColBlock's dequantised weight."""
    from .quantization import ColBlockQuantizedLinear

    if isinstance(mod, ColBlockQuantizedLinear):
        return mod.get_weight(torch.bfloat16)
    return mod.weight.detach()


def pack_linear(mod: nn.Module, R: int, pair: Optional[nn.Module] = None, tune: Optional[dict] = None,
                out: Optional[torch.Tensor] = None) -> _PackedWeight:
    """Repack one linear (or the c_fc1 / c_fc2 pair) into the weight stream of its format (`out`: the slice of a
    weight arena the Q4 stream is written to)."""
    from .quantization import ColBlockQuantizedLinear, Linear8bitLt

    kind = _kind(mod)
    if pair is not None and _kind(pair) != kind:
        raise EngineUnavailable("c_fc1 / c_fc2 use different linear types")
    pw = _PackedWeight()
    d = pw.desc
    d.R = 2 if pair is not None else R
    tune = tune or {}
    d.waves, d.grid, d.prefetch = tune.get("waves", 0), tune.get("grid", 0), tune.get("prefetch", 0)
    d.flags = tune.get("flags", 0)
    if kind == "q4":
        assert isinstance(mod, ColBlockQuantizedLinear)
        if not mod.fast_eligible(torch.bfloat16):
            raise EngineUnavailable("ColBlockQuantizedLinear layout not handled by the fast kernel "
                                    f"(bits={mod.bits}, groups={mod.scales.shape[1]}, scales {mod.scales.dtype})")
        if mod.bias is not None:
            raise EngineUnavailable("bias on a quantised linear")
        N, K = mod.out_features, mod.in_features
        stream = ops.repack_q4(mod.quant_weight, pair.quant_weight if pair is not None else None, N, K, d.R, out=out)
        s0, z0 = mod.scales.reshape(-1).contiguous(), mod.zeros.reshape(-1).contiguous()
        d.fmt, d.w, d.N, d.K = W_Q4, ptr(stream), N, K
        d.scales, d.zeros, d.sz_dtype = ptr(s0), ptr(z0), dtype_code(s0.dtype)
        d.group_cols = mod.tile_cols if mod.scales.shape[1] > 1 else 0  # grouped: [N, groups] tables, streamed per tile
        if pair is not None and (pair.tile_cols != mod.tile_cols or pair.scales.shape != mod.scales.shape):
            raise EngineUnavailable("c_fc1 / c_fc2 use different group sizes")
        pw.keep += [stream, s0, z0]
        pw.q4_stream, pw.q4_mods = stream, [mod] + ([pair] if pair is not None else [])
        pw.side_bytes = 2 * s0.numel() * s0.element_size()
        if pair is not None:
            s1, z1 = pair.scales.reshape(-1).contiguous(), pair.zeros.reshape(-1).contiguous()
            if s1.dtype != s0.dtype:
                raise EngineUnavailable("c_fc1 / c_fc2 scales differ in dtype")
            d.scales2, d.zeros2 = ptr(s1), ptr(z1)
            pw.keep += [s1, z1]
            pw.side_bytes *= 2
    elif kind == "i8":
        assert isinstance(mod, Linear8bitLt)
        if mod.bias is not None:
            raise EngineUnavailable("bias on a quantised linear")
        if not hasattr(mod.weight, "CB"):
            raise EngineUnavailable("Linear8bitLt weight not quantised yet")
        N, K = mod.out_features, mod.in_features
        stream = ops.repack_i8(mod.weight.CB, pair.weight.CB if pair is not None else None, d.R, out=out)
        d.fmt, d.w, d.N, d.K = W_I8, ptr(stream), N, K
        d.scb = ptr(mod.weight.SCB)
        pw.keep += [stream, mod.weight.SCB]
        pw.side_bytes = 4 * N
        if pair is not None:
            d.scb2 = ptr(pair.weight.SCB)
            pw.keep.append(pair.weight.SCB)
            pw.side_bytes *= 2
    else:
        if mod.bias is not None:
            raise EngineUnavailable("bias on a hot-path linear")
        w0 = _dense_weight(mod)
        if w0.dtype != torch.bfloat16:
            raise EngineUnavailable(f"dense weights are {w0.dtype}; the MFMA path needs bf16")
        N, K = w0.shape
        stream = ops.repack_bf16(w0, _dense_weight(pair) if pair is not None else None, d.R, out=out)
        del w0
        d.fmt, d.w, d.N, d.K = W_BF16, ptr(stream), N, K
        pw.keep.append(stream)
    pw.stream_bytes = pw.keep[0].numel()
    return pw


def model_fingerprint(model: "nn.Module") -> tuple:
    """(data_ptr, version, dtype) of every parameter and buffer the engine packed or points at.  The engine holds
    repacked copies and raw device pointers: `load_state_dict`, `.to()`, `.bfloat16()`, re-quantisation or an
    in-place edit make them stale, and `LLaMA.engine()` rebuilds the engine when this changes."""
    out = []
    for t in list(model.parameters()) + list(model.buffers()):
        out.append((t.data_ptr(), t._version, t.dtype, t.device))
    for mod in model.modules():
        w = getattr(mod, "weight", None)
        cb = getattr(w, "CB", None) if w is not None else None
        if cb is not None:  # Linear8bitLt keeps its int8 rows as attributes of the parameter
            out.append((cb.data_ptr(), cb._version, cb.dtype, cb.device))
    return tuple(out)


def _fused_plan(model, kinds, C_, nh, hs, H, V) -> Optional[dict]:
    """Byte layout of the weight arena of the fused decode step (csrc/fused_step.hip), or None when the model is
    not one the persistent launch handles (then every linear keeps its own stream tensor)."""
    if _env_int("MI355_FUSED", 1) == 0 or kinds not in ({"q4"}, {"bf16"}, {"i8"}):
        return None
    if any(hasattr(blk.attn, "adapter_wte") for blk in model.transformer.h):
        return None  # LLaMA-Adapter blocks: the prefix term lives in the launch-per-operator step
    sup = int(lib().mi355_fused_step_supported(C_, nh, hs, H, V, 1))  # 1: the 7B shape (ring kernel), 2: a wide shape (fused_step_wide.hip)
    if not sup:
        return None
    wide = sup == 2 or (_env_int("MI355_FUSED_WIDE", 0) != 0 and kinds == {"q4"})  # (the wide kernel on the 7B shape: a cross-check)
    if sup == 2 and kinds != {"q4"}:
        return None  # (the wide-shape kernel streams per-row int4 only)
    first = model.transformer.h[0]
    if kinds in ({"bf16"}, {"i8"}):
        # unquantised models (BASELINE configs[1]) and LLM.int8 models (configs[3]): the BF16 / int8 instantiations of the
        # register-ring kernel (round 4) over BF16 / I8 streams
        wfmt, fmt, env = (W_BF16, 1, "MI355_FUSED_BF16") if kinds == {"bf16"} else (W_I8, 2, "MI355_FUSED_INT8")
        if _env_int(env, 1) == 0:
            return None
        sizes = [ops.packed_bytes(wfmt, 3 * C_, C_, 1, False), ops.packed_bytes(wfmt, C_, C_, 1, False),
                 ops.packed_bytes(wfmt, H, C_, 2, True), ops.packed_bytes(wfmt, C_, H, 1, False)]
        offs = [0, sizes[0], sizes[0] + sizes[1], sizes[0] + sizes[1] + sizes[2]]
        layer_bytes = sum(sizes)
        if layer_bytes >= 1 << 31 or ops.packed_bytes(wfmt, V, C_, 1, False) >= 1 << 31:
            return None  # (a layer is addressed through one 32-bit buffer descriptor)
        plan = {"sizes": sizes, "offs": offs, "layer_bytes": layer_bytes, "group_cols": 0, "fmt": fmt}
        if fmt == 1 and _env_int("MI355_FUSED_U8", 1) != 0 and H // 128 <= 92:
            # round 6: `gptq.int8` (8-bit ColBlockQuantizedLinear everywhere, one (scale, zero) pair per row): the persistent step reads
            # the 8-bit levels themselves (weight_fmt 6, half the bytes of the dequantised bf16 matrices the launch path keeps streaming)
            from .quantization import ColBlockQuantizedLinear

            mods = [m_ for blk in model.transformer.h
                    for m_ in (blk.attn.c_attn, blk.attn.c_proj, blk.mlp.c_fc1, blk.mlp.c_fc2, blk.mlp.c_proj)] + [model.lm_head]
            if all(isinstance(m_, ColBlockQuantizedLinear) and m_.bits == 8 and m_.scales.shape[1] == 1 and m_.scales.dtype == torch.bfloat16
                   and m_.out_features % 16 == 0 and m_.in_features % 128 == 0 for m_ in mods):
                plan["u8"] = True
        return plan
    group_cols = 0
    mods = [m_ for blk in model.transformer.h
            for m_ in (blk.attn.c_attn, blk.attn.c_proj, blk.mlp.c_fc1, blk.mlp.c_fc2, blk.mlp.c_proj)] + [model.lm_head]
    if wide and any(m_.scales.shape[1] > 1 for m_ in mods):
        return None
    if all(m_.scales.shape[1] > 1 for m_ in mods):
        # grouped scales ("groupsize" checkpoints): the GRP instantiation of the register-ring kernel — one group size of
        # 128 * 2^n columns throughout, dividing n_embd and n_hidden, bf16 tables
        group_cols = first.attn.c_attn.tile_cols
        ok = (_env_int("MI355_FUSED_GROUPED", 1) != 0 and group_cols >= 128 and (group_cols & (group_cols - 1)) == 0
              and C_ % group_cols == 0 and H % group_cols == 0
              and all(m_.tile_cols == group_cols and m_.grouped_fast() for m_ in mods))
        if not ok or any(m_.out_features % 16 for m_ in mods):
            return None  # (group_table packs whole 16-row tiles: other row counts stay on the launch-per-operator engine)
    else:
        for mod in (first.attn.c_attn, first.attn.c_proj, first.mlp.c_fc1, first.mlp.c_fc2, first.mlp.c_proj, model.lm_head):
            if not mod.fast_eligible(torch.bfloat16) or mod.scales.dtype != torch.bfloat16 or mod.scales.shape[1] != 1:
                return None  # (mixed per-row / grouped layouts: launch-per-operator engine)
    sizes = [ops.packed_bytes(W_Q4, 3 * C_, C_, 1, False), ops.packed_bytes(W_Q4, C_, C_, 1, False),
             ops.packed_bytes(W_Q4, H, C_, 2, True), ops.packed_bytes(W_Q4, C_, H, 1, False)]
    offs = [0, sizes[0], sizes[0] + sizes[1], sizes[0] + sizes[1] + sizes[2]]
    layer_bytes = sum(sizes)
    if layer_bytes >= 1 << 32 or any(o % 16 for o in offs) or layer_bytes % 16:
        return None
    return {"sizes": sizes, "offs": offs, "layer_bytes": layer_bytes, "group_cols": group_cols, "fmt": 4 if wide else 0}


_U8_PERM = (0, 4, 1, 5, 2, 6, 3, 7)


def u8_stream(mats: List[torch.Tensor]) -> torch.Tensor:
    """uint8 [N, K] levels of an 8-bit ColBlockQuantizedLinear (one matrix: R = 1; the c_fc1 / c_fc2 pair: R = 2) -> the stream
    `mi355_fused_step` reads with weight_fmt 6: [tile of 16 rows][unit of 128 columns][r][piece e = 0, 1][lane = 16 g + row][16 B],
    byte b of lane (g, row) of piece e = column 128 u + 32 g + 16 e + 8 (b >> 3) + (0 4 1 5 2 6 3 7)[b & 7] — the octet order of
    the fp8 step's limb planes (tests/layouts.py f8_planes), so that `v & 0x0F0F0F0F` / `(v >> 4) & 0x0F0F0F0F` of the two pieces of
    a unit are the low- / high-nibble A operands against one B operand (csrc/fused_step_ring.hip FS_RUN_U)."""
    N, K = mats[0].shape
    assert N % 16 == 0 and K % 128 == 0 and all(m_.shape == (N, K) and m_.dtype == torch.uint8 for m_ in mats)
    dev = mats[0].device
    u, e, g, b = torch.meshgrid(torch.arange(K // 128, device=dev), torch.arange(2, device=dev), torch.arange(4, device=dev),
                                torch.arange(16, device=dev), indexing="ij")
    perm = torch.tensor(_U8_PERM, device=dev)
    col = (128 * u + 32 * g + 16 * e + 8 * (b >> 3) + perm[b & 7]).reshape(-1)
    outs = []
    for m_ in mats:
        t = m_.index_select(1, col).view(N // 16, 16, K // 128, 2, 4, 16)  # [tile, row, unit, e, g, b]
        outs.append(t.permute(0, 2, 3, 4, 1, 5))                               # [tile, unit, e, g, row, b]
    return torch.stack(outs, dim=2).contiguous().reshape(-1)                   # [tile, unit, r, e, lane, b]


def group_table(scales: torch.Tensor, zeros: torch.Tensor) -> torch.Tensor:
    """[N, groups] bf16 scales / zeros of a grouped ColBlockQuantizedLinear -> the table the GRP instantiation of the fused
    step reads (`mi355_fused_step_args.gt`): int32 [N / 16 tiles][group][16 rows] = scale bits | zero bits << 16, flattened —
    a streamer lane fetches rows 4 g .. 4 g + 3 of a tile for its group with one 16-B load."""
    N, G = scales.shape
    assert N % 16 == 0 and scales.dtype == torch.bfloat16 and zeros.dtype == torch.bfloat16 and zeros.shape == scales.shape
    word = (scales.contiguous().view(torch.int16).to(torch.int32) & 0xFFFF) | (zeros.contiguous().view(torch.int16).to(torch.int32) << 16)
    return word.view(N // 16, 16, G).permute(0, 2, 1).contiguous().reshape(-1)


class DecodeEngine:
    """Owns everything `mi355_forward` / `mi355_fused_step` need for one `LLaMA` instance."""

    def __init__(self, model: "nn.Module", *, tp_rank: int = 0, tp_world: int = 1, tune: Optional[dict] = None):
        cfg = model.config
        wte = model.transformer.wte.weight
        if wte.device.type != "cuda":
            raise EngineUnavailable(f"model is on {wte.device}")
        if wte.dtype != torch.bfloat16:
            raise EngineUnavailable(f"model dtype {wte.dtype}: the engine computes with bf16 MFMA operands; "
                                    "f32 models run op by op through the exact f32 kernels")
        for name, sc in [("ln_f", model.transformer.ln_f.scale)] + [
                (f"h.{i}.rms_{j}", getattr(blk, f"rms_{j}").scale) for i, blk in enumerate(model.transformer.h) for j in (1, 2)]:
            if sc.dtype != wte.dtype or sc.device != wte.device:
                raise EngineUnavailable(f"{name}.scale is {sc.dtype} on {sc.device}, wte is {wte.dtype} on {wte.device}: "
                                        "the engine reads every norm scale in the embedding's dtype")
        self.model = model
        self.device = wte.device
        self.cfg = cfg
        self.fingerprint = model_fingerprint(model)
        self.tp_world = tp_world
        self.tune = tune or {}
        self.stream = torch.cuda.Stream(device=self.device)  # capturable (the legacy default stream is not)
        self.use_graph = _env_int("MI355_GRAPH", 1) != 0
        self._graphs = {}  # argmax flag -> graph handle
        self._keep: List[object] = []

        C_, hs, nh = cfg.n_embd, cfg.n_embd // cfg.n_head, cfg.n_head
        with torch.cuda.device(self.device):
            first = model.transformer.h[0]
            self.n_hidden = first.mlp.c_fc1.out_features
            self.packed = []
            layers = (Layer * cfg.n_layer)()
            # Launch geometry: one persistent workgroup per CU looping over 16-row tiles.  Tile counts that are a
            # multiple of the CU count keep every CU equally busy (7B: c_attn 768 tiles = 3 per CU, c_proj /
            # mlp.c_proj 256 = 1 per CU, lm_head 2000 = 7.8); the c_fc1/c_fc2 pair is 688 tiles (2.7 per CU).
            cus = nat.num_cus() or 256
            # Measured (scripts/sweep_gemv.py, 7B shapes): one workgroup per CU is best for every per-layer linear
            # (c_attn 7.9 vs 8.1 us with two); only lm_head (7.8 tiles per CU) prefers two (15.5 vs 16.6 us).
            # The LLM.int8 launches (heavier quantising prologue) stay at one per CU throughout.
            i8 = _kind(first.attn.c_attn) == "i8"
            grids = {} if i8 else {"lm_head": 2 * cus}

            def dflt(key):
                # Round 5 (scripts/sweep_gemv.py over the 13B / 65B shapes, profiles/r05_gemv_geometry_13b_65b.txt): a single-matrix
                # Q4 launch runs TWO workgroups per CU (or one per tile when there are fewer tiles) whenever that does not raise the
                # largest number of tiles a CU ends up with — 65B c_attn 22.1 -> 20.5 us, attn.c_proj 9.3 -> 8.8, mlp.c_proj 20.2 -> 19.5;
                # 13B attn.c_proj 7.8 -> 7.3, mlp.c_proj 12.4 -> 11.6.  Every 7B shape keeps what it had (c_attn: 768 tiles = 3 per CU
                # against 2 x 2; 256-tile linears: one per CU either way; lm_head 2 x CUs).  The c_fc1 / c_fc2 pair stays at one per CU.
                g = grids.get(key, cus)
                mod = {"attn": first.attn.c_attn, "proj": first.attn.c_proj, "mproj": first.mlp.c_proj, "lm_head": model.lm_head}.get(key)
                if mod is not None and _kind(mod) == "q4":  # (per module: a bf16 lm_head in a q4 model keeps the generic rule)
                    n_tiles = -(-mod.out_features // 16)
                    g2 = min(n_tiles, 2 * cus)
                    # (rows of fewer than 32 units of 128 columns — the K-sharded c_proj / mlp.c_proj of a TP = 8 rank — keep one
                    # workgroup per CU: 5.2 / 5.7 us against 5.3 / 5.9 with two)
                    if mod.in_features // 128 >= 32 and (2 * -(-n_tiles // (2 * cus)) <= -(-n_tiles // cus) or n_tiles <= 2 * cus):
                        g = max(g2, 1)
                return {"grid": g, **self.tune.get(key, {})}
            kinds = {_kind(m_) for blk in model.transformer.h
                     for m_ in (blk.attn.c_attn, blk.attn.c_proj, blk.mlp.c_fc1, blk.mlp.c_fc2, blk.mlp.c_proj)}
            kinds.add(_kind(model.lm_head))
            # the fused decode step addresses a layer by a stride: all Q4 streams of the model live in ONE arena
            plan = None
            if tp_world == 1 and wte.dtype == torch.bfloat16:
                plan = _fused_plan(model, kinds, C_, nh, hs, self.n_hidden, model.lm_head.out_features)
            self.fused_plan = plan
            self.w_arena = None
            if plan is not None:
                self.w_arena = torch.empty(cfg.n_layer * plan["layer_bytes"], dtype=torch.uint8, device=self.device)

            def slot(i, j):
                if plan is None:
                    return None
                o = i * plan["layer_bytes"] + plan["offs"][j]
                return self.w_arena[o:o + plan["sizes"][j]]

            # LLaMA-Adapter: the k / v projections of the prefix rows through the block's own c_attn, once (the reference
            # keeps them as adapter_kv_cache, adapter.py:136-141) — through the module's forward, before its weights are repacked
            adapter_kv = {}
            for i, blk in enumerate(model.transformer.h):
                if hasattr(blk.attn, "adapter_wte"):
                    if tp_world != 1:
                        raise EngineUnavailable("LLaMA-Adapter blocks in a tensor-parallel shard")
                    adapter_kv[i] = blk.attn.adapter_prefix_kv(wte.dtype)
            for i, blk in enumerate(model.transformer.h):
                attn = pack_linear(blk.attn.c_attn, 1, tune=dflt("attn"), out=slot(i, 0))
                proj = pack_linear(blk.attn.c_proj, 1, tune=dflt("proj"), out=slot(i, 1))
                fc = pack_linear(blk.mlp.c_fc1, 2, pair=blk.mlp.c_fc2, tune=dflt("fc"), out=slot(i, 2))
                mproj = pack_linear(blk.mlp.c_proj, 1, tune=dflt("mproj"), out=slot(i, 3))
                self.packed += [attn, proj, fc, mproj]
                L = layers[i]
                L.rms1, L.rms2 = ptr(blk.rms_1.scale.detach()), ptr(blk.rms_2.scale.detach())
                L.attn, L.proj, L.fc, L.mproj = attn.desc, proj.desc, fc.desc, mproj.desc
                if i in adapter_kv:  # LLaMA-Adapter: prefix keys / values / gate of this block (adapter.py:134-151)
                    ak, av, gate = adapter_kv[i]
                    L.adapter_k, L.adapter_v, L.adapter_gate, L.adapter_len = ptr(ak), ptr(av), ptr(gate), ak.shape[1]
                    self._keep += [ak, av, gate]
            head = pack_linear(model.lm_head, 1, tune=dflt("lm_head"))
            self.packed.append(head)
            self.layers = layers

            # LDS bound on the rows one launch can stage (see ops.fast_linear_max_m)
            fmts = {p.desc.fmt for p in self.packed}
            worst_fmt = W_I8 if W_I8 in fmts else W_Q4
            # the prompt chunk is what the n_embd-wide linears can stage; the wider mlp.c_proj input is fed in
            # sub-chunks of rows by the native side (engine.hip run_linear)
            max_T = min(16, ops.fast_linear_max_m(C_, 2, worst_fmt))
            if ops.fast_linear_max_m(self.n_hidden, 1, worst_fmt) < 1:
                max_T = 0
            if max_T < 1:
                raise EngineUnavailable("hidden size does not fit LDS")
            # int4 and bf16 models: prompt chunks of >= 32 tokens go through the wide MFMA GEMM + flash attention (csrc/gemm.hip,
            # flash_prefill.hip), so the chunk is bounded by scratch memory only (0.5 GB at 2048 tokens of a 7B model,
            # most of it the f32 logits rows).  Large chunks matter: the [n_embd]-wide outputs (c_proj, mlp.c_proj) are
            # 16 row blocks, so 512 tokens fill only a quarter of the chip (9 % vs 27-36 % of the MFMA peak at 2048)
            self.gemm_ws = None
            if kinds <= {"q4", "bf16", "i8"} and _env_int("MI355_PREFILL_GEMM", 1):
                max_T = max(max_T, min(_env_int("MI355_PREFILL_T", 2048), cfg.block_size))
                need = max(int(lib().mi355_linear_gemm_workspace_bytes(max_T, max(C_, self.n_hidden))),
                           int(lib().mi355_linear_int8_gemm_workspace_bytes(max_T, max(C_, self.n_hidden))))
                self.gemm_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self.max_T = max_T

            local_heads = first.attn.c_attn.out_features // (3 * hs)
            self.local_heads = local_heads
            Cl = local_heads * hs
            V = model.lm_head.out_features
            dev = self.device
            self.x = torch.zeros((max_T, C_), dtype=torch.float32, device=dev)
            self.qkv = torch.zeros((max_T, 3 * Cl), dtype=torch.float32, device=dev)
            self.att = torch.zeros((max_T, Cl), dtype=torch.bfloat16, device=dev)
            self.hbuf = torch.zeros((max_T, self.n_hidden), dtype=torch.bfloat16, device=dev)
            self.partial = torch.zeros((max_T, C_), dtype=torch.float32, device=dev)
            self.logits = torch.zeros((max_T, V), dtype=torch.float32, device=dev)
            self.tokens = torch.zeros((max_T,), dtype=torch.int32, device=dev)
            self.pos = torch.zeros((max_T,), dtype=torch.int32, device=dev)
            self.next_token = torch.zeros((1,), dtype=torch.int32, device=dev)
            self.out_tokens = torch.zeros((cfg.block_size + 1,), dtype=torch.int32, device=dev)
            self.attn_splits = max(1, int(self.tune.get("attn_splits", _env_int("MI355_ATTN_SPLITS", 4))))
            self.attn_part = torch.zeros((local_heads, self.attn_splits, hs + 4), dtype=torch.float32, device=dev)
            if model.rope_cache is None or model.rope_cache.dtype != torch.float32:
                from .model import build_rope_cache

                rope = build_rope_cache(cfg.block_size, hs, torch.int64, dev)
            else:
                rope = model.rope_cache
            self.rope = rope.to(device=dev, dtype=torch.float32).contiguous()

        m = Model()
        m.n_layer, m.n_head, m.n_embd, m.hs = cfg.n_layer, local_heads, C_, hs
        m.n_hidden, m.vocab, m.S, m.block_size = self.n_hidden, wte.shape[0], 0, cfg.block_size
        m.param_dtype, m.cache_dtype = dtype_code(wte.dtype), BF16
        m.tp_world, m.max_T = tp_world, max_T
        m.eps = float(model.transformer.ln_f.eps)
        m.int8_threshold = 6.0
        m.wte, m.ln_f = ptr(wte.detach()), ptr(model.transformer.ln_f.scale.detach())
        m.lm_head = head.desc
        m.rope = ptr(self.rope)
        m.layers = C.cast(self.layers, C.POINTER(Layer))
        m.x, m.qkv, m.att, m.hbuf = ptr(self.x), ptr(self.qkv), ptr(self.att), ptr(self.hbuf)
        m.partial, m.logits = ptr(self.partial), ptr(self.logits)
        m.tokens, m.pos, m.next_token = ptr(self.tokens), ptr(self.pos), ptr(self.next_token)
        m.out_tokens = ptr(self.out_tokens)
        m.attn_part, m.attn_splits = ptr(self.attn_part), self.attn_splits
        if self.gemm_ws is not None:
            m.gemm_ws, m.gemm_ws_bytes = ptr(self.gemm_ws), self.gemm_ws.numel()
        self.m = m
        self.S = 0
        self._cache_pool = {}  # S -> list of (k, v): kept across reset_cache() so captured graphs stay valid
        self.fused = None
        self.fused_enabled = True  # tests / measurements switch between the persistent launch and the 162-launch step
        if self.fused_plan is not None:
            self._build_fused(model, head)
        # The int4 weights now exist twice: as the modules' reference-layout buffers and as this engine's streams.  Give
        # the first copy back (3.3 GB for 7B, 32.5 GB for 65B); ColBlockQuantizedLinear rebuilds it from the stream when
        # anything asks for it (state_dict(), the module's own forward, a new engine).  MI355_RELEASE_REFERENCE_LAYOUT=0
        # keeps both.
        self.released_bytes = 0
        if tp_world == 1 and _env_int("MI355_RELEASE_REFERENCE_LAYOUT", 1):
            for pw in self.packed:
                for which, mod in enumerate(pw.q4_mods):
                    self.released_bytes += mod._buffers["quant_weight"].numel()
                    mod.release_reference_layout(pw.q4_stream, pw.desc.R, len(pw.q4_mods) == 2, which)
            if self.released_bytes:
                self.fingerprint = model_fingerprint(model)

    # ---- fused decode step (csrc/fused_step.hip) -------------------------------------------------------
    def _build_fused(self, model, head) -> None:
        """Side arenas of the persistent decode launch: scales / zeros, norm scales, hand-off workspace."""
        cfg, plan, dev = self.cfg, self.fused_plan, self.device
        C_, H, V = cfg.n_embd, self.n_hidden, model.lm_head.out_features
        gc = plan.get("group_cols", 0)

        fmt = plan.get("fmt", 0)
        with torch.cuda.device(dev):
            sz = sz_head = gt = gt_head = None
            norms = torch.empty((2 * cfg.n_layer + 1, C_), dtype=torch.bfloat16, device=dev)
            if fmt == 1:
                pass  # BF16 streams: no scales / zeros
            elif fmt == 2:
                # LLM.int8: the f32 row scales SCB per layer, c_attn[3C] attn.c_proj[C] c_fc1[H] c_fc2[H] mlp.c_proj[C]; lm_head's [V]
                sz = torch.stack([torch.cat([mod.weight.SCB.reshape(-1).float() for mod in (blk.attn.c_attn, blk.attn.c_proj, blk.mlp.c_fc1,
                                                                                           blk.mlp.c_fc2, blk.mlp.c_proj)])
                                  for blk in model.transformer.h]).contiguous()
                assert sz.shape == (cfg.n_layer, 5 * C_ + 2 * H)
                sz_head = model.lm_head.weight.SCB.reshape(-1).float().contiguous()
            elif gc:
                rows = []
                for blk in model.transformer.h:
                    rows.append(torch.cat([group_table(mod.scales, mod.zeros) for mod in (blk.attn.c_attn, blk.attn.c_proj, blk.mlp.c_fc1,
                                                                        blk.mlp.c_fc2, blk.mlp.c_proj)]))
                gt = torch.stack(rows).contiguous()      # [n_layer, dwords per layer]
                gt_head = group_table(model.lm_head.scales, model.lm_head.zeros)
            else:
                sz = torch.empty((cfg.n_layer, 10 * C_ + 4 * H), dtype=torch.bfloat16, device=dev)
                sz_head = torch.cat([model.lm_head.scales.reshape(-1), model.lm_head.zeros.reshape(-1)]).contiguous()
            for i, blk in enumerate(model.transformer.h):
                if not gc and fmt in (0, 4):
                    parts = []
                    for mod in (blk.attn.c_attn, blk.attn.c_proj, blk.mlp.c_fc1, blk.mlp.c_fc2, blk.mlp.c_proj):
                        parts += [mod.scales.reshape(-1), mod.zeros.reshape(-1)]
                    sz[i].copy_(torch.cat(parts))
                norms[2 * i].copy_(blk.rms_1.scale.detach())
                norms[2 * i + 1].copy_(blk.rms_2.scale.detach())
            norms[2 * cfg.n_layer].copy_(model.transformer.ln_f.scale.detach())
            ws = _handoff_workspace(int(lib().mi355_fused_step_workspace_bytes(H)), dev)
        a = nat.FusedStepArgs()
        a.w, a.layer_stride = ptr(self.w_arena), plan["layer_bytes"]
        a.off_attn, a.off_proj, a.off_fc, a.off_mproj = plan["offs"]
        a.layer_bytes, a.head_bytes = plan["layer_bytes"], head.stream_bytes
        a.w_head = head.desc.w
        u8_arena = u8_head = None
        if plan.get("u8"):
            # 8-bit levels as their own arena (the bf16 arena stays what the launch-per-operator path and the prompt pass stream)
            with torch.cuda.device(dev):
                per_layer = []
                for blk in model.transformer.h:
                    per_layer.append(torch.cat([ops.repack_u8(blk.attn.c_attn.quant_weight), ops.repack_u8(blk.attn.c_proj.quant_weight),
                                                ops.repack_u8(blk.mlp.c_fc1.quant_weight, blk.mlp.c_fc2.quant_weight),
                                                ops.repack_u8(blk.mlp.c_proj.quant_weight)]))
                u8_arena = torch.stack(per_layer).contiguous()
                u8_head = ops.repack_u8(model.lm_head.quant_weight)  # (`u8_stream` above states the same layout in torch: tests/test_host_logic.py)
                sz = torch.empty((cfg.n_layer, 10 * C_ + 4 * H), dtype=torch.bfloat16, device=dev)
                for i, blk in enumerate(model.transformer.h):
                    parts = []
                    for mod in (blk.attn.c_attn, blk.attn.c_proj, blk.mlp.c_fc1, blk.mlp.c_fc2, blk.mlp.c_proj):
                        parts += [mod.scales.reshape(-1), mod.zeros.reshape(-1)]
                    sz[i].copy_(torch.cat(parts))
                sz_head = torch.cat([model.lm_head.scales.reshape(-1), model.lm_head.zeros.reshape(-1)]).contiguous()
            lb = u8_arena.shape[1]
            assert lb == 3 * C_ * C_ + C_ * C_ + 2 * H * C_ + C_ * H and lb < 1 << 31
            a.w, a.layer_stride, a.layer_bytes = ptr(u8_arena), lb, lb
            a.off_attn, a.off_proj, a.off_fc, a.off_mproj = 0, 3 * C_ * C_, 4 * C_ * C_, 4 * C_ * C_ + 2 * H * C_
            a.w_head, a.head_bytes = ptr(u8_head), u8_head.numel()
            fmt = 6
        a.sz, a.sz_head, a.norms = ptr(sz), ptr(sz_head), ptr(norms)
        a.weight_fmt = fmt
        # (round 6: group tables too — csrc/fused_step_ring.hip FS_RUN_FG: three MFMA columns per group, at most 15 groups per streamer wave)
        gc_f8_ok = not gc or ((H // 128 + 7) // 8 >> (gc // 128).bit_length() - 1) + 1 <= 15
        if fmt == 0 and gc_f8_ok and H // 128 <= 92 and _env_int("MI355_FUSED_F8", 1) != 0:
            # round 4: the same int4 streams through fp8 operands — one scaled K = 128 MFMA per 1-KiB piece instead of four f16 ones, the
            # publishers split the activations into three E4M3 limbs (csrc/fused_step_ring.hip FMT 3; DESIGN.md section 5: +2.4..2.8 % on
            # the headline, the same distance from the reference's run as the fp16 operands).  MI355_FUSED_F8=0 keeps weight_fmt 0;
            # tests/test_zz_fused_f8_gpu.py and scripts/ab_fused.py --f8 toggle `fused.weight_fmt` on a live engine.
            a.weight_fmt = 3
        if fmt == 4 and _env_int("MI355_FUSED_F8", 1) != 0:
            a.weight_fmt = 5  # round 6: the same operands in the wide-shape kernel (csrc/fused_step_wide.hip F8)
        if gc:
            a.group_cols, a.gt, a.gt_head, a.gt_layer_stride = gc, ptr(gt), ptr(gt_head), gt.shape[1] * 4
        a.wte, a.rope = self.m.wte, self.m.rope
        a.tokens, a.pos, a.next_token, a.out_tokens = self.m.tokens, self.m.pos, self.m.next_token, self.m.out_tokens
        a.logits, a.workspace = self.m.logits, ptr(ws)
        a.n_layer, a.n_head, a.n_embd, a.hs = cfg.n_layer, self.local_heads, C_, C_ // cfg.n_head
        a.n_hidden, a.vocab, a.eps = H, V, self.m.eps
        self.fused = a
        self._fused_top_fmt = int(a.weight_fmt)
        self._fused_keep = [sz, sz_head, norms, ws, gt, gt_head, u8_arena, u8_head]
        self._fused_ws = ws
        self._fused_warm = False
        self.fused_clipped = 0  # activation pairs clipped by the persistent step so far (check_status)
        self.fused_demotions: List[tuple] = []  # (why, to what, first bad position) every time the hand-off format proved too narrow
        self.fused_promotions = 0               # times the engine climbed back one rung after a clean run (check_status)
        # The ladder of hand-off formats, widest last (None = the launch-per-operator step: f32 residual, bf16 staging, no range to
        # leave).  Round 6: a clip moves the engine ONE rung down for `_hold` steps, not for good — a trained checkpoint's massive
        # activations fire on a few delimiter tokens, and a sticky ladder turned one of them into a permanent 2.7 % loss.
        # (the wide-shape kernel: weight_fmt 5 -> 4 -> the launch-per-operator step)
        self._full_rungs = {3: [3, 0, None], 5: [5, 4, None]}.get(int(a.weight_fmt), [int(a.weight_fmt), None])
        self._rungs = list(self._full_rungs)
        self._rung = 0
        self._demoted_before = False
        self._hold = self.HOLD0        # clean steps on a lower rung before the engine climbs back; doubles with every further clip
        self._steps_on_rung = 0        # decode steps issued since the last demotion / promotion

    HOLD0, HOLD_MAX = 16, 4096

    def fused_ready(self) -> bool:
        return (self.fused is not None and self.fused_enabled and self._rungs[self._rung] is not None and self.fused.kv is not None
                and self.fused.S == self.S and self.S > 0)

    def status_due(self) -> bool:
        """Is a periodic `check_status()` worth its device->host read: the persistent step is running (its hand-offs can clip), or the
        engine sits on a lower rung of the ladder and may climb back."""
        return self.fused is not None and self.fused_enabled and (self.fused_ready() or self._rung > 0)

    def _set_rung(self, i: int) -> str:
        """Select rung `i` of the ladder; the hand-off workspace is zeroed when the granule tag width changes (weight_fmt 3 carries
        16-bit tags: include/mi355_llama.h).  Everything here is enqueued on the engine's stream."""
        fmt = self._rungs[i]
        self._rung, self._steps_on_rung = i, 0
        if fmt is None:
            return "the launch-per-operator step"
        if int(self.fused.weight_fmt) != fmt:
            self.fused.weight_fmt = fmt
            with torch.cuda.stream(self.stream):
                self._fused_ws[256:].zero_()
        return {3: "fp8-limb operands (weight_fmt 3)", 0: "fp16 operands (weight_fmt 0)", 5: "fp8-limb operands (weight_fmt 5)",
                4: "fp16 operands (weight_fmt 4)", 6: "8-bit levels through fp8-limb operands (weight_fmt 6)"}.get(fmt, f"weight_fmt {fmt}")

    def _demote_fused(self, why: str, bad: Optional[int] = None) -> str:
        """The persistent step met a position its hand-off format is too narrow for — E4M3 limbs clip at +-448 x the edge's
        pre-scale, fp16 at +-65504, the LLM.int8 outlier list holds 1024 columns.  Move this engine ONE rung down the ladder
        fp8-limb operands (weight_fmt 3 / 5) -> fp16 operands (weight_fmt 0 / 4) -> launch-per-operator step (f32 residual, bf16 staging, no
        list limit) for the next `_hold` decode steps; `check_status` climbs back one rung after that many clean steps.  `_hold`
        starts at 16 and doubles with every further clip (cap 4096): a checkpoint that clips on every token settles on the lower
        rung by itself, one whose massive activations fire on a few tokens pays one replayed step each."""
        if self._demoted_before:
            self._hold = min(2 * self._hold, self.HOLD_MAX)
        self._demoted_before = True
        to = self._set_rung(min(self._rung + 1, len(self._rungs) - 1))
        self.fused_demotions.append((why, to, bad))
        return to

    def reset_fused_format(self) -> None:
        """Tests / measurements: back to the top rung of the ladder (what `_build_fused` chose), with a clean hand-off workspace."""
        if self.fused is None:
            return
        self._rungs = list(self._full_rungs)
        self._set_rung(0)
        self._hold, self._demoted_before = self.HOLD0, False
        self.fused_enabled = True

    def use_fused_format(self, fmt: int) -> None:
        """Tests / measurements: make `fmt` (a rung of this engine's ladder) the top rung — e.g. 0 on an engine whose default is 3
        times the fp16-operand kernel a clipping checkpoint runs on (bench.py `rungs`, scripts/ab_fused.py --f8)."""
        self._rungs = self._full_rungs[self._full_rungs.index(fmt):]
        self._set_rung(0)
        self._hold, self._demoted_before = self.HOLD0, False
        self.fused_enabled = True

    def clear_status(self) -> None:
        """Forget what the status words hold (clip count, lowest clipped position): a caller that starts a NEW sequence must not
        replay from a position of the previous one (advisor r5: an interrupted generate() or a direct run_step() caller leaves
        them set; they are sticky by design).  The abort word stays: a timed-out hand-off must still raise."""
        if self.fused is not None:
            with torch.cuda.stream(self.stream):
                self._fused_ws[8:16].zero_()

    def check_status(self) -> Optional[int]:
        """One device->host read of the persistent step's status words — call it where the host synchronises anyway (end of
        generate, after a timed loop, in tests).

        * a hand-off timed out / the step was entered with pos >= S (abort word): raises, the outputs are garbage;
        * activations exceeded the range of the step's hand-off format, or an LLM.int8 vector had more than 1024 outlier columns,
          since the last call: the steps from the returned position on did NOT compute what the reference computes.  The engine
          has then already moved one rung down (`_demote_fused`) and warned; the caller re-runs the generation from that position
          (`generate()` and `forward()` do) — a caller that drives `run_step` itself and ignores the return value keeps tokens
          that are wrong from there on, which is what the RuntimeWarning says;
        * otherwise returns None — and, when the engine has spent `_hold` clean steps on a lower rung, climbs back one rung."""
        if self.fused is None:
            return None
        words = self._fused_ws[:16].view(torch.int32).tolist()  # abort code, step counter, clipped pairs, 0x7FFFFFFF - first bad position
        code, clipped, first = words[0], words[2], words[3]
        overflow8 = 0x700 <= code < 0x800
        if code != 0 and not overflow8:  # (first: a time-out in the same window as a clip must not be cleared by the clip's branch)
            with torch.cuda.stream(self.stream):
                self._fused_ws[:4].zero_()
            raise nat.NativeError(f"fused decode step aborted (code 0x{code:x}): a workgroup hand-off timed out or the "
                                  "step was entered with pos >= S; the step's outputs are invalid")
        if clipped or overflow8:
            with torch.cuda.stream(self.stream):
                if overflow8:
                    self._fused_ws[:4].zero_()
                self._fused_ws[8:16].zero_()
            self.fused_clipped += clipped
            bad = 0x7FFFFFFF - first if first else 0
            was = self.fused.weight_fmt
            why = (f"{clipped} activation pairs past the range of the hand-off format (weight_fmt {was})" if clipped else
                   f"more than 1024 LLM.int8 outlier columns in one activation vector (abort code 0x{code:x})")
            to = self._demote_fused(why, bad)
            import warnings

            warnings.warn(f"fused decode step: {why} — clipped / invalid from position {bad} on; this engine decodes the next "
                          f"{self._hold} steps through {to} and the tokens from that position on must be recomputed (generate() and "
                          "LLaMA.forward do so themselves)", RuntimeWarning, stacklevel=2)
            return bad
        if self._rung > 0 and self.fused_enabled and self._steps_on_rung >= self._hold:
            self._set_rung(self._rung - 1)  # `_hold` clean steps behind the last clip: one rung up again
            self.fused_promotions += 1
        return None

    # ---- bookkeeping ---------------------------------------------------------------------------------
    def weight_stream_bytes(self) -> int:
        return sum(p.stream_bytes for p in self.packed)

    def side_bytes(self) -> int:
        return sum(p.side_bytes for p in self.packed)

    def _destroy_graphs(self) -> None:
        for g in self._graphs.values():
            lib().mi355_graph_destroy(g)
        self._graphs = {}

    def __del__(self):
        try:
            self._destroy_graphs()
        except Exception:
            pass

    def reset_cache(self) -> None:
        """`LLaMA.reset_cache()`: the next forward starts from zeroed caches (model.py:140-145)."""
        self.S = 0

    def _ensure_cache(self, S: int) -> None:
        if self.S == S and self.model.kv_caches:
            return
        cfg = self.cfg
        hs = cfg.n_embd // cfg.n_head
        pool = self._cache_pool.get(S)
        if pool is None:
            self._destroy_graphs()  # captured kernels hold the old cache pointers / S by value
            # one allocation [n_layer][2][1][n_head][S][hs]: the fused step addresses a layer's rows by a stride;
            # kv_caches keeps the reference's per-layer (k, v) tensors of shape [1, n_head, S, hs] as views
            self._cache_pool = {}
            arena = torch.zeros((cfg.n_layer, 2, 1, self.local_heads, S, hs), dtype=torch.bfloat16, device=self.device)
            pool = [(arena[i, 0], arena[i, 1]) for i in range(cfg.n_layer)]
            self._cache_pool = {S: pool}  # one cache geometry at a time
            self._kv_arena = arena
        else:
            for k, v in pool:
                k.zero_()
                v.zero_()
        for i, (k, v) in enumerate(pool):
            self.layers[i].kcache, self.layers[i].vcache = ptr(k), ptr(v)
        self.model.kv_caches = list(pool)
        self.m.S = S
        self.S = S
        if self.fused is not None:
            ok = bool(lib().mi355_fused_step_supported(cfg.n_embd, self.local_heads, hs, self.n_hidden,
                                                       self.fused.vocab, S))
            self.fused.kv = ptr(self._kv_arena) if ok else None
            self.fused.S = S if ok else 0

    # ---- execution -----------------------------------------------------------------------------------
    def _host_pos0(self, input_pos: torch.Tensor, T: int) -> Optional[int]:
        hint = getattr(input_pos, "_mi355_pos0", None)
        if hint is not None:
            return int(hint)
        ends = input_pos[[0, -1]].tolist()  # one device->host sync for callers that do not pass the hint
        if ends[1] - ends[0] != T - 1:
            return None
        return int(ends[0])

    def step_graph(self, argmax):
        key = int(argmax)  # 0 logits only, 1 + greedy argmax, 3 chained greedy step (mi355_forward)
        g = self._graphs.get(key)
        if g is None:
            handle = C.c_void_p()
            check(lib().mi355_graph_capture(C.byref(self.m), key, self.stream.cuda_stream, C.byref(handle)),
                  "mi355_graph_capture")
            g = handle
            self._graphs[key] = g
        return g

    def _warm_step(self, argmax) -> None:
        # one eager pass before the first capture: the launchers set per-kernel attributes
        # (hipFuncSetAttribute) lazily, which must not happen while the stream is capturing.  It recomputes
        # exactly what the captured step will compute again, so the state is unchanged.
        # (A chained step would advance the state, so its warm-up is the plain argmax step: same kernels plus
        # argmax_advance_kernel, which has no lazily-set attribute.)
        check(lib().mi355_forward(C.byref(self.m), 1, 1, int(argmax) & 1, self.stream.cuda_stream),
              "mi355_forward (warm-up)")
        if int(argmax) & 2:
            self.embed_step()  # the warm-up consumed the residual stream; restore the chained entry state

    def run_step(self, argmax, allow_fused: bool = True) -> None:
        """One T = 1 forward on self.stream (tokens / positions already placed by set_step).  argmax: False/0
        logits only, True/1 greedy argmax, 3 chained greedy step (needs `embed_step()` before the first one)."""
        s = self.stream.cuda_stream
        argmax = int(argmax)
        if self.fused is not None and self._rung > 0:
            self._steps_on_rung += 1
        while allow_fused and self.fused_ready():
            # one persistent launch per token; launches are asynchronous, so the host runs ahead without a graph
            self.fused.mode = argmax
            rc = lib().mi355_fused_step(C.byref(self.fused), s)
            if rc == 0:
                return
            if rc != nat.E_STATE:
                check(rc, "mi355_fused_step")
            # refused BEFORE anything was launched (the device does not admit a workgroup of this instantiation per CU — advisor r4:
            # the fp8-operand kernel used to fail every step here instead of falling back): next rung, same step
            self._demote_fused(f"mi355_fused_step refused weight_fmt {self.fused.weight_fmt}: {nat.last_error()}")
        if self.use_graph:
            try:
                if argmax not in self._graphs:
                    self._warm_step(argmax)
                g = self.step_graph(argmax)
            except nat.NativeError:
                if _env_int("MI355_GRAPH_STRICT", 0):
                    raise
                self.use_graph = False
                g = None
            if g is not None:
                check(lib().mi355_graph_launch(g, s), "mi355_graph_launch")
                return
        check(lib().mi355_forward(C.byref(self.m), 1, 1, argmax, s), "mi355_forward")

    def embed_step(self) -> None:
        """Embedding of tokens[0] into the residual stream: the entry state of a chained greedy step."""
        check(lib().mi355_forward_embed(C.byref(self.m), 1, self.stream.cuda_stream), "mi355_forward_embed")

    def set_step(self, idx: Optional[torch.Tensor], T: int, pos0: int, from_next: bool = False) -> None:
        is64 = 1 if (idx is not None and idx.dtype == torch.int64) else 0
        check(lib().mi355_set_step(C.byref(self.m), ptr(idx), is64, T, pos0, 1 if from_next else 0,
                                   self.stream.cuda_stream), "mi355_set_step")

    def prefill(self, idx: torch.Tensor, pos0: int, *, all_logits: bool, argmax: bool = False) -> Optional[torch.Tensor]:
        """Feed T prompt tokens starting at position pos0 (chunks of max_T); returns logits [T, V] if asked,
        otherwise leaves the last token's logits in self.logits[0]."""
        T = idx.numel()
        flat = idx.reshape(-1)
        out = torch.empty((T, self.logits.shape[1]), dtype=torch.float32, device=self.device) if all_logits else None
        s = self.stream.cuda_stream
        for t0 in range(0, T, self.max_T):
            n = min(self.max_T, T - t0)
            last = t0 + n == T
            self.set_step(flat[t0:t0 + n], n, pos0 + t0)
            mode = 2 if all_logits else (1 if last else 0)
            if n == 1 and mode == 1:
                self.run_step(argmax)
            else:
                check(lib().mi355_forward(C.byref(self.m), n, mode, 1 if (argmax and last) else 0, s), "mi355_forward")
            if all_logits:
                out[t0:t0 + n].copy_(self.logits[:n])
        return out

    def forward(self, idx: torch.Tensor, max_seq_length: int, input_pos: torch.Tensor) -> Optional[torch.Tensor]:
        """`LLaMA.forward` for B == 1 with a KV cache; returns logits [1, T, V] (f32) or None if the positions
        are not a contiguous run (then the caller takes the op-by-op path)."""
        T = idx.shape[1]
        pos0 = self._host_pos0(input_pos, T)
        if pos0 is None:
            return None
        roll = pos0 + T > max_seq_length
        if roll and (T != 1 or pos0 >= self.cfg.block_size):
            return None  # a multi-token call across the end of the cache: op-by-op path
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self._ensure_cache(max_seq_length)
            if T == 1:
                if roll:
                    # cache-roll regime (model.py:214-218: input_pos[-1] >= max_seq_length): every layer's cache moves
                    # up one row (mi355_kv_roll), the new row lands in the last slot (the attention kernel clamps the
                    # slot to S - 1), RoPE keeps the true position.  Stays on the engine: n_layer roll launches + the
                    # launch-per-operator step (the persistent step addresses the cache by position and refuses it).
                    for k, v in self.model.kv_caches:
                        ops.kv_roll(k, v)
                self.set_step(idx.reshape(-1), 1, pos0)
                self.run_step(False, allow_fused=not roll)  # logits land in row 0
                # raw `model(idx, S, input_pos)` callers (the reference's generate loop) never see the status words of the
                # persistent step otherwise: a timed-out hand-off would hand them garbage logits, silently and for good (the
                # word is sticky) -> raise; a step whose activations left the range of its hand-off format -> the engine has
                # moved to a wider one, compute THIS step again.  One 16-byte read; this path synchronises per token anyway.
                while self.fused is not None and self.check_status() is not None:
                    self.set_step(idx.reshape(-1), 1, pos0)
                    self.run_step(False, allow_fused=not roll)
                logits = self.logits[:1].clone()
            else:
                logits = self.prefill(idx, pos0, all_logits=True)
        cur.wait_stream(self.stream)
        return logits.view(1, T, -1)

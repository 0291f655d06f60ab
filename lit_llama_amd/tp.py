"""Tensor parallelism for the decode path (BASELINE.json configs[4]: LLaMA-65B gptq.int4, TP = 8 over xGMI).

The reference has no tensor parallelism; the partition is the one Meta's checkpoints use and the reference
records when it MERGES them — `shard_dims` in /root/reference scripts/convert_checkpoint.py:57-65:

    c_attn   dim 0 (per head, inside each of the stacked Q / K / V thirds, lit_llama/model.py:197)
    c_proj   dim 1        c_fc1 / c_fc2  dim 0        mlp.c_proj  dim 1        lm_head  dim 0
    wte      replicated here (524 MB in bf16 for 65B; a column split would add an all-gather per token)

Per layer two row-parallel linears produce partial sums -> one all-reduce(sum) of [T, n_embd] f32 each
(RCCL over xGMI through torch.distributed, backend "nccl"), plus one all-gather of the [V / world] logit shards
per step.  int4 per-row scale / zero shard with the rows of column-parallel linears and are replicated for
row-parallel ones (y = s (sum_k x_k q_k - z sum_k x_k) is additive over K shards).

`tp_forward` is the protocol (which collective where); it drives a list of `LocalShard`s: one `EngineShard` per
process under torch.distributed, or several on ONE GPU with `LoopbackComm` (how the protocol is tested on the
1-GPU box), or CPU stand-ins in the gloo tests.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

from .model import LLaMA, LLaMAConfig

SHARD_DIMS = {
    "attn.c_attn": 0, "attn.c_proj": 1, "mlp.c_fc1": 0, "mlp.c_fc2": 0, "mlp.c_proj": 1, "lm_head": 0,
}


def check_divisible(cfg: LLaMAConfig, world: int) -> None:
    if cfg.n_head % world or cfg.n_hidden % world or cfg.padded_vocab_size % world:
        raise ValueError(f"TP={world} does not divide n_head={cfg.n_head}, n_hidden={cfg.n_hidden} or "
                         f"vocab={cfg.padded_vocab_size}")
    if (cfg.n_hidden // world) % 2 or (cfg.n_embd // world) % 2:
        raise ValueError("int4 packing needs an even number of input columns per shard")


def _shard_rows(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    n = t.shape[0] // world
    return t[rank * n:(rank + 1) * n]


def _shard_qkv_rows(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """rows [Q; K; V] -> [Q_r; K_r; V_r] (heads are contiguous inside each third)."""
    third = t.shape[0] // 3
    n = third // world
    return torch.cat([t[i * third + rank * n: i * third + (rank + 1) * n] for i in range(3)], dim=0)


def _shard_cols(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    n = t.shape[1] // world
    return t[:, rank * n:(rank + 1) * n]


def shard_state_dict(sd: Dict[str, torch.Tensor], cfg: LLaMAConfig, rank: int, world: int) -> Dict[str, torch.Tensor]:
    """Rank-local view of a full (fp or gptq.int4) checkpoint with the reference's key names."""
    check_divisible(cfg, world)
    out: Dict[str, torch.Tensor] = {}
    for key, t in sd.items():
        name = next((n for n in SHARD_DIMS if (n + ".") in key), None)
        if name is None:  # wte, norm scales
            out[key] = t
            continue
        dim = SHARD_DIMS[name]
        leaf = key.rsplit(".", 1)[1]
        if dim == 0:
            if leaf == "bias":
                raise ValueError("hot-path linears have no bias")
            cut = _shard_qkv_rows if name == "attn.c_attn" else _shard_rows
            piece = cut(t, rank, world)
        else:
            # row-parallel: weights / packed bytes split along the input dim, per-row scale / zero replicated
            piece = t if leaf in ("scales", "zeros") else _shard_cols(t, rank, world)
        if leaf == "quant_weight":
            piece = piece.t().contiguous().t()  # keep the reference's column-major storage
        else:
            piece = piece.contiguous()
        out[key] = piece
    return out


def build_local_model(cfg: LLaMAConfig, world: int, *, device, dtype=torch.bfloat16, mode: Optional[str] = None) -> LLaMA:
    """An (empty) LLaMA whose linears have the rank-local shapes; fill it with `shard_state_dict` output."""
    from .utils import EmptyInitOnDevice

    check_divisible(cfg, world)
    local = LLaMAConfig(block_size=cfg.block_size, vocab_size=cfg.vocab_size, padded_vocab_size=cfg.padded_vocab_size,
                        n_layer=cfg.n_layer, n_head=cfg.n_head, n_embd=cfg.n_embd)
    local.tp_world = world
    with EmptyInitOnDevice(device=device, dtype=dtype, quantization_mode=mode):
        model = LLaMA(local)
    model.eval()
    return model


# ------------------------------------------------------------------------------------------------ communicators
class DistComm:
    """torch.distributed (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)

    def all_reduce_sum(self, tensors: Sequence[torch.Tensor]) -> None:
        assert len(tensors) == 1
        self.dist.all_reduce(tensors[0], op=self.dist.ReduceOp.SUM, group=self.group)

    def all_gather_cols(self, tensors: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        assert len(tensors) == 1
        t = tensors[0].contiguous()
        parts = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t, group=self.group)
        return [torch.cat(parts, dim=-1)]


class LoopbackComm:
    """All ranks live in this process (one GPU): the 'collective' is a local sum / concat, in rank order."""

    def __init__(self, world: int):
        self.world = world

    def all_reduce_sum(self, tensors: Sequence[torch.Tensor]) -> None:
        assert len(tensors) == self.world
        total = tensors[0].clone()
        for t in tensors[1:]:
            total += t
        for t in tensors:
            t.copy_(total)

    def all_gather_cols(self, tensors: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        assert len(tensors) == self.world
        full = torch.cat(list(tensors), dim=-1)
        return [full.clone() for _ in tensors]


class NativeComm:
    """Peer-write all-reduce of libmi355llama (csrc/tp_comm.hip, `mi355_tp_allreduce`): one launch per all-reduce,
    fused with the residual add, on the engine's stream — no host work between segments, capturable in a hipGraph.
    One process per GPU; the 64-byte IPC handles of the receive buffers travel through `exchange`, a callable that
    all-gathers a Python object across the ranks (default: torch.distributed.all_gather_object, which works over the
    gloo / nccl group the launcher created).  Two processes on ONE GPU work the same way (how the 1-GPU box tests it:
    RCCL refuses duplicate devices, HIP IPC does not)."""

    def __init__(self, rank: int, world: int, n_embd: int, device, exchange=None):
        from ._native import TpComm, check, lib

        self.rank, self.world, self.device = rank, world, torch.device(device)
        self.fused_residual = True
        L = lib()
        with torch.cuda.device(self.device):
            nbytes = int(L.mi355_tp_comm_bytes(world, n_embd))
            if nbytes <= 0:
                raise ValueError(f"tp all-reduce handles 1..8 ranks (got {world})")
            buf = C.c_void_p()
            check(L.mi355_tp_buffer_alloc(nbytes, C.byref(buf)), "mi355_tp_buffer_alloc")
            self._buf = buf
            handle = (C.c_ubyte * 64)()
            check(L.mi355_ipc_export(buf, handle), "mi355_ipc_export")
            if exchange is None:
                import torch.distributed as dist

                def exchange(obj):
                    out = [None] * world
                    dist.all_gather_object(out, obj)
                    return out
            handles = exchange(bytes(handle))
            self.state = torch.zeros(4, dtype=torch.int32, device=self.device)
            c = TpComm()
            c.world, c.rank, c.slot_floats = world, rank, n_embd
            self._opened = []
            for r in range(world):
                if r == rank:
                    c.peer_buf[r] = buf.value
                else:
                    peer = C.c_void_p()
                    hb = (C.c_ubyte * 64).from_buffer_copy(handles[r])
                    check(L.mi355_ipc_open(hb, C.byref(peer)), f"mi355_ipc_open (rank {r})")
                    c.peer_buf[r] = peer.value
                    self._opened.append(peer)
            c.state = self.state.data_ptr()
            self.c = c
            exchange("mapped")  # nobody frees / reuses before every peer has mapped every buffer
        self._call = 0

    def step_begin(self, stream) -> None:
        from ._native import check, lib

        check(lib().mi355_tp_step_begin(C.byref(self.c), stream.cuda_stream), "mi355_tp_step_begin")
        self._call = 0

    def reduce_add(self, shard, T: int, force: bool = False) -> None:
        """x[:T] += sum over ranks of partial[:T]  (all-reduce + residual add of model.py:166-167 in one launch).
        World 1: nothing to exchange — the engine's row-parallel segments accumulate into the residual stream themselves when
        tp_world == 1 (csrc/engine.hip run_segment), `partial` stays zero, and the 2 x n_layer launches per token would add that
        zero (round 5: 160 launches = 1.4 of 7.8 ms per 65B token on one GPU); `force` issues them anyway (bench.py times the
        collective alone with it)."""
        from ._native import check, lib

        if self.world == 1 and not force:
            return
        eng = shard.eng
        n = T * eng.cfg.n_embd
        if n > self.c.slot_floats:  # prompt chunks: row by row (decode, the case that matters, is one row)
            for t in range(T):
                self._one(eng, t * eng.cfg.n_embd, eng.cfg.n_embd)
            return
        self._one(eng, 0, n)

    MAX_CALLS = 1024  # csrc/tp_comm.hip kMaxCalls: collective calls per device step counter value

    def _one(self, eng, off: int, n: int) -> None:
        from ._native import check, lib

        if self._call >= self.MAX_CALLS - 2:
            # a prompt chunk reduced row by row can need more calls than one step's tag space holds (65B: 160 per row):
            # open the next step (every rank issues the same sequence, so the tags stay in lockstep)
            self.step_begin(eng.stream)
        check(lib().mi355_tp_allreduce(C.byref(self.c), eng.partial.data_ptr() + 4 * off, eng.x.data_ptr() + 4 * off, n,
                                       self._call, 1, eng.stream.cuda_stream), "mi355_tp_allreduce")
        self._call += 1

    def argmax(self, shard, advance: bool, row: int = 0) -> None:
        """Greedy sampling over the vocabulary shards, on the device (`mi355_tp_argmax`): next_token / out_tokens
        (indexed by the position of token `row` of the step + 1) and, with `advance`, the token / position slots of
        the next chained step."""
        from ._native import check, lib

        eng = shard.eng
        assert not (advance and row)
        check(lib().mi355_tp_argmax(C.byref(self.c), eng.logits.data_ptr(), int(eng.m.lm_head.N), self._call,
                                    eng.next_token.data_ptr(), eng.out_tokens.data_ptr(), eng.tokens.data_ptr(),
                                    eng.pos.data_ptr() + 4 * row, 1 if advance else 0, eng.stream.cuda_stream),
              "mi355_tp_argmax")
        self._call += 1

    def check_status(self) -> None:
        from ._native import NativeError

        code = int(self.state[1].item())
        if code:
            self.state[1] = 0
            raise NativeError(f"tensor-parallel all-reduce aborted (code 0x{code:x}): a peer did not deliver in time")

    def all_gather_cols(self, tensors: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        import torch.distributed as dist  # one gather of [V / world] logits per token: not on the per-layer path

        assert len(tensors) == 1
        t = tensors[0].contiguous()
        if dist.get_backend() == "gloo":  # test rigs (two processes on one GPU): gather on the host
            parts = [torch.empty_like(t, device="cpu") for _ in range(self.world)]
            dist.all_gather(parts, t.cpu())
            return [torch.cat(parts, dim=-1).to(t.device)]
        parts = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(parts, t)
        return [torch.cat(parts, dim=-1)]

    def close(self) -> None:
        from ._native import lib

        for p_ in self._opened:
            lib().mi355_ipc_close(p_)
        self._opened = []
        if self._buf is not None:
            lib().mi355_tp_buffer_free(self._buf)
            self._buf = None


# ------------------------------------------------------------------------------------------------ protocol
def tp_forward(shards: Sequence, comm, T: int, n_layer: int, want_logits: bool = True) -> Optional[List[torch.Tensor]]:
    """One forward over T tokens already placed in every shard.  Returns the full logits per shard
    ([rows, vocab], rows = whatever `head()` produced)."""
    fused = getattr(comm, "fused_residual", False)  # NativeComm: all-reduce + residual add in ONE launch on the stream
    if fused:
        assert len(shards) == 1
        comm.step_begin(shards[0].eng.stream)
    for s in shards:
        s.embed(T)
    for l in range(n_layer):
        for s in shards:
            s.attn_part(l, T)          # RMSNorm + c_attn shard + local heads + c_proj shard -> partial
        if fused:
            comm.reduce_add(shards[0], T)
        else:
            comm.all_reduce_sum([s.partial_view(T) for s in shards])
            for s in shards:
                s.residual_add(T)
        for s in shards:
            s.mlp_part(l, T)           # RMSNorm + fc shard + SwiGLU + mlp.c_proj shard -> partial
        if fused:
            comm.reduce_add(shards[0], T)
        else:
            comm.all_reduce_sum([s.partial_view(T) for s in shards])
            for s in shards:
                s.residual_add(T)
    if not want_logits:
        return None
    return comm.all_gather_cols([s.head(T) for s in shards])


class EngineShard:
    """`LocalShard` on the native engine: the rank's `mi355_model` driven segment by segment."""

    def __init__(self, model: LLaMA, world: int, tune: Optional[dict] = None):
        from .engine import DecodeEngine

        self.eng = DecodeEngine(model, tp_world=world, tune=tune)
        model._engine = self.eng
        self.model = model

    def _call(self, fn, *args):
        from ._native import check, lib

        check(getattr(lib(), fn)(C.byref(self.eng.m), *args, self.eng.stream.cuda_stream), fn)

    def embed(self, T):
        self._call("mi355_forward_embed", T)

    def attn_part(self, l, T):
        self._call("mi355_forward_segment", T, l, 0, 2)

    def mlp_part(self, l, T):
        self._call("mi355_forward_segment", T, l, 2, 4)

    def residual_add(self, T):
        self._call("mi355_residual_add", T)

    def partial_view(self, T):
        return self.eng.partial[:T]

    def head(self, T):
        self._call("mi355_forward_head", T, 1, 0)  # last token only
        return self.eng.logits[:1, : self.eng.m.lm_head.N]


class TPDecoder:
    """Greedy decode of ONE stream across `world` shards (list of EngineShard; length 1 per process under
    torch.distributed, length `world` with LoopbackComm)."""

    def __init__(self, shards: Sequence[EngineShard], comm, cfg: LLaMAConfig):
        self.shards, self.comm, self.cfg = list(shards), comm, cfg

    def _streams(self):
        return [s.eng.stream for s in self.shards]

    @torch.no_grad()
    def generate_chained(self, prompt: torch.Tensor, max_new_tokens: int, max_seq_length: Optional[int] = None,
                         use_graph: bool = True) -> torch.Tensor:
        """Greedy decode of one stream with `NativeComm`: the prompt goes through the segment protocol chunk by chunk;
        every decode step is then ONE hipGraph replay per rank — embedding, 2 x n_layer [segments + peer-write
        all-reduce fused with the residual add], lm_head shard, sharded arg-max that writes the next token and
        position — with no host work, no RCCL call and no device->host read inside the loop."""
        from ._native import check, lib

        assert len(self.shards) == 1 and getattr(self.comm, "fused_residual", False)
        shard, comm, cfg = self.shards[0], self.comm, self.cfg
        eng = shard.eng
        T = prompt.numel()
        S = max_seq_length or min(T + max_new_tokens, cfg.block_size)
        if S < T + max_new_tokens:
            raise ValueError(f"tensor-parallel decode needs max_seq_length >= prompt + new tokens ({S} < {T} + {max_new_tokens})")
        cur = torch.cuda.current_stream(prompt.device)
        eng.stream.wait_stream(cur)
        s = eng.stream.cuda_stream

        def step(advance: bool):
            comm.step_begin(eng.stream)
            shard.embed(1)
            for l in range(cfg.n_layer):
                shard.attn_part(l, 1)
                comm.reduce_add(shard, 1)
                shard.mlp_part(l, 1)
                comm.reduce_add(shard, 1)
            shard.head(1)
            comm.argmax(shard, advance)

        with torch.cuda.stream(eng.stream):
            eng._ensure_cache(S)
            eng.out_tokens[:T].copy_(prompt.to(torch.int32))
            pos = 0
            while pos < T:
                n = min(eng.max_T, T - pos)
                eng.set_step(prompt[pos:pos + n], n, pos)
                last = pos + n == T
                tp_forward([shard], comm, n, cfg.n_layer, want_logits=False)
                if last:
                    shard.head(n)
                    comm.argmax(shard, False, row=n - 1)  # out_tokens[T] = first generated token
                pos += n
            if max_new_tokens > 1 and comm.world == 1 and eng.fused_ready():
                # world 1 (BASELINE configs[4] on ONE GPU): nothing to exchange, so the whole decode step is the engine's
                # persistent launch (65B: csrc/fused_step_wide.hip, round 6) — the loop of lit_llama_amd.generate._generate_greedy
                from .generate import CHECK_EVERY, _replay_from

                eng.clear_status()
                eng.set_step(None, 1, T, from_next=True)
                eng.embed_step()
                done = 1
                while True:
                    bad = None
                    while done < max_new_tokens:
                        eng.run_step(3)
                        done += 1
                        if done % CHECK_EVERY == 0 and eng.status_due():
                            bad = eng.check_status()
                            if bad is not None:
                                break
                    if bad is None:
                        bad = eng.check_status()
                    if bad is None:
                        break
                    _replay_from(eng, bad, T, done)
                    done = bad + 1 - T
            elif max_new_tokens > 1:
                eng.set_step(None, 1, T, from_next=True)
                graph = None
                if use_graph:
                    step(False)  # eager warm-up (lazy kernel attributes); recomputes position T, state unchanged
                    check(lib().mi355_graph_begin(s), "mi355_graph_begin")
                    try:
                        step(True)
                    finally:
                        h = C.c_void_p()
                        rc = lib().mi355_graph_end(s, C.byref(h))
                    check(rc, "mi355_graph_end")
                    graph = h
                for _ in range(max_new_tokens - 1):
                    if graph is not None:
                        check(lib().mi355_graph_launch(graph, s), "mi355_graph_launch")
                    else:
                        step(True)
                if graph is not None:
                    eng.stream.synchronize()
                    lib().mi355_graph_destroy(graph)
            out = eng.out_tokens[:T + max_new_tokens].to(prompt.dtype).clone()
        cur.wait_stream(eng.stream)
        comm.check_status()
        return out

    @torch.no_grad()
    def generate(self, prompt: torch.Tensor, max_new_tokens: int, max_seq_length: Optional[int] = None) -> torch.Tensor:
        from .ops import argmax

        T = prompt.numel()
        S = max_seq_length or min(T + max_new_tokens, self.cfg.block_size)
        if S < T + max_new_tokens:
            # the single-GPU path rolls the cache (model.py:214-218, mi355_kv_roll); the shard engines do not yet, and
            # the attention kernel would clamp every later position onto the last slot: refuse instead of decoding
            # against stale rows
            raise ValueError(f"tensor-parallel decode needs max_seq_length >= prompt + new tokens "
                             f"({S} < {T} + {max_new_tokens}); the cache-roll regime is single-GPU only")
        dev = prompt.device
        out = torch.empty(T + max_new_tokens, dtype=prompt.dtype, device=dev)
        out[:T] = prompt
        eng0 = self.shards[0].eng
        cur = torch.cuda.current_stream(dev)
        for st in self._streams():
            st.wait_stream(cur)
        # the loopback shards share one GPU: run everything on shard 0's stream so the order is total
        run_stream = eng0.stream
        for s in self.shards:
            s.eng.stream = run_stream
        with torch.cuda.stream(run_stream):
            for s in self.shards:
                s.eng._ensure_cache(S)
            pos, logits = 0, None
            while pos < T:  # prompt in chunks every shard can stage
                n = min(min(s.eng.max_T for s in self.shards), T - pos)
                for s in self.shards:
                    s.eng.set_step(prompt[pos:pos + n], n, pos)
                logits = tp_forward(self.shards, self.comm, n, self.cfg.n_layer, want_logits=(pos + n == T))
                pos += n
            for i in range(max_new_tokens):
                nxt = argmax(logits[0].reshape(-1).float().contiguous())
                out[T + i: T + i + 1] = nxt.to(out.dtype)
                if i + 1 == max_new_tokens:
                    break
                for s in self.shards:
                    s.eng.set_step(nxt, 1, T + i)
                logits = tp_forward(self.shards, self.comm, 1, self.cfg.n_layer)
        cur.wait_stream(run_stream)
        return out

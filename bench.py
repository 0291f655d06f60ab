#!/usr/bin/env python
"""Benchmark of the hot path: LLaMA-7B `gptq.int4`, batch 1, greedy decode on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one decoded token: one pass of the whole forward + on-device greedy argmax over synthetic random-init
weights of the 7B architecture — ONE persistent launch per token for 7B gptq.int4 (csrc/fused_step.hip; the
fallback and the other configs: 32 layers x 5 launches + lm_head + argmax replayed from a hipGraph) — with the
prompt already prefilled and everything resident in HBM when the timed region starts.  W untimed steps, then
EXACTLY K timed steps between (barrier +) torch.cuda.synchronize() on both sides — three such blocks over the same
positions, the MEDIAN block is reported (`blocks_ms_per_step` holds all three); MAX over ranks; rank 0 prints
ONE JSON line.  With N > 1 every GPU decodes its own independent stream (the bs=1 7B path does not shard —
SURVEY.md §8e: "replicas only"), so scaling is "weak" and `value` is the sum over ranks.

Extra objects on the same line:
  roofline     — the dominant kernel: with the fused step that is `fused_step_ring_kernel`, one launch = one token
                 (algorithmic bytes = every weight byte once + scales / zeros + norm scales + the KV rows read at
                 that position); otherwise the c_fc1/c_fc2 + SwiGLU weight-streaming launch.  Algorithmic bytes per
                 launch / average launch duration — the dispatch's own begin / end timestamps, delivered into HIP
                 events by hipExtLaunchKernel on the launch stream (mi355_debug_time_next_launch; the clock
                 rocprofv3 reads) — vs the 8.0 TB/s HBM peak.  `traffic` is NOT measured in this run: it is the
                 FETCH_SIZE x 2 of the committed rocprofv3 --pmc pass over the same command (`traffic_source`);
  cpu_baseline — the reference's CPU path timed on the host cores over a bounded sample, extrapolated to 32 layers:
                 the real /root/reference code when that tree exists (kind "reference"; it does not on the GPU box),
                 else oracle/oracle.py, its line-by-line restatement (kind "port").  Both dequantise every weight on
                 every call, as the reference does off the Triton branch;
  generate     — tokens/s of lit_llama_amd.generate INCLUDING the prompt (what generate.py:146-153 prints).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK = 8.0e12  # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)
METRIC = "decode tokens/sec/GPU LLaMA-7B gptq.int4 bs=1; % HBM roofline"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--prompt-len", type=int, default=128)
    ap.add_argument("--model", default="7B")
    ap.add_argument("--quantize", default="gptq.int4", choices=["gptq.int4", "llm.int8", "none", "gptq.int8"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tp-timeout", type=float, default=180.0, help="seconds after which a hanging TP leg is abandoned")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--group-cols", type=int, default=0, help="gptq.int4 with one (scale, zero) pair per row and group of this "
                    "many input columns (GPTQ groupsize, e.g. 128) instead of one pair per row (not a BASELINE config)")
    ap.add_argument("--adapter", action="store_true", help="LLaMA-Adapter variant (generate/adapter.py): prefix attention "
                    "with random adaption prompts / gates in every block from adapter_start_layer on (not a BASELINE config)")
    ap.add_argument("--dry-run", action="store_true", help="check the launch contract only (ranks, world size, the tensor-parallel "
                    "part's output contract with stub legs); no GPU work")
    ap.add_argument("--dry-run-fail-native", type=int, default=-1, help="(dry run) the rank whose native TP leg fails")
    ap.add_argument("--no-tp", action="store_true", help="skip the tensor-parallel leg (65B gptq.int4, TP = --gpus)")
    ap.add_argument("--tp-model", default="65B")
    ap.add_argument("--tp-steps", type=int, default=48)
    ap.add_argument("--tp-comm", default="native", choices=["native", "rccl"])
    ap.add_argument("--tune", default=None, help="JSON dict of per-linear launch tuning (engine.DecodeEngine tune)")
    return ap.parse_args()


def bytes_per_token(cfg, mode: str, u8_stream: bool = False):
    """Algorithmic HBM bytes of one decode step: every linear weight once + its per-row
    scale/zero + norm scales + one embedding row; KV traffic is reported separately (position dependent)."""
    C_, H, V, L = cfg.n_embd, cfg.n_hidden, cfg.padded_vocab_size, cfg.n_layer
    params = L * (3 * C_ * C_ + C_ * C_ + 3 * C_ * H) + V * C_
    rows = L * (3 * C_ + C_ + 2 * H + C_) + V
    # (gptq.int8: the reference dequantises an 8-bit ColBlock matrix into a bf16 one on every forward call, quantization.py:413-423;
    # the engine builds it once and streams it — 2 bytes per weight is what a decode step reads)
    # (round 6: the persistent step reads the 8-bit levels themselves, weight_fmt 6 — 1 byte per weight + the per-row scale / zero pairs)
    wbytes = {"gptq.int4": params // 2, "llm.int8": params, "none": params * 2, "gptq.int8": params if u8_stream else params * 2}[mode]
    side = {"gptq.int4": rows * 4, "llm.int8": rows * 4, "none": 0, "gptq.int8": rows * 4 if u8_stream else 0}[mode]
    other = (2 * L + 1) * C_ * 2 + C_ * 2
    return dict(weights=wbytes, total=wbytes + side + other, kv_per_pos=2 * L * C_ * 2)


def build_model(args, dev):
    from lit_llama_amd import synth
    from lit_llama_amd.model import LLaMA, LLaMAConfig
    from lit_llama_amd.utils import EmptyInitOnDevice

    cfg = LLaMAConfig.from_name(args.model)
    mode = None if args.quantize == "none" else args.quantize
    if args.adapter:
        from lit_llama_amd import adapter as A

        cfg = A.LLaMAConfig.from_name(args.model)
        cfg.vocab_size = cfg.padded_vocab_size  # (the reference's adapter model sizes its head by vocab_size)
        LLaMA = A.LLaMA
    with EmptyInitOnDevice(device=dev, dtype=torch.bfloat16, quantization_mode=mode):
        model = LLaMA(cfg)
    model.eval()
    if args.group_cols:
        assert mode == "gptq.int4", "--group-cols is a gptq.int4 option"
        from lit_llama_amd.quantization import ColBlockQuantizedLinear

        for _, mod in list(model.named_modules()):
            for cname, child in list(mod.named_children()):
                if isinstance(child, ColBlockQuantizedLinear):
                    q = ColBlockQuantizedLinear(child.in_features, child.out_features, bias=False, bits=4, tile_cols=args.group_cols)
                    setattr(mod, cname, q.to(device=dev, dtype=torch.bfloat16))
    if args.adapter:
        gen_a = torch.Generator(device=dev).manual_seed(7)
        with torch.no_grad():
            for blk in model.transformer.h:
                if hasattr(blk.attn, "adapter_wte"):
                    w = blk.attn.adapter_wte.weight
                    w.copy_(torch.randn(w.shape, generator=gen_a, device=dev).to(w.dtype))
                    blk.attn.gating_factor.fill_(0.5)
    if mode in ("gptq.int4", "gptq.int8"):
        synth.fill_model_random_int4(model, seed=0)
    else:
        gen = torch.Generator(device=dev).manual_seed(0)
        with torch.no_grad():
            for name, p in model.named_parameters():
                if name.endswith("scale"):
                    p.copy_((1 + 0.1 * torch.randn(p.shape, generator=gen, device=dev)).to(p.dtype))
                elif name.endswith("wte.weight"):
                    p.copy_(torch.randn(p.shape, generator=gen, device=dev).to(p.dtype))
            for mod in model.modules():
                if isinstance(mod, torch.nn.Linear):
                    w = torch.randn(mod.weight.shape, generator=gen, device=dev) * mod.in_features**-0.5
                    if mode == "llm.int8":
                        mod._quantize_weight(w)
                    else:
                        mod.weight.data.copy_(w.to(mod.weight.dtype))
    if args.tune:
        model._engine = None
        from lit_llama_amd.engine import DecodeEngine

        model._engine = DecodeEngine(model, tune=json.loads(args.tune))
    return model, cfg


def measure_fused_step(eng, n: int = 48):
    """Average duration of the fused_step_ring_kernel launch (one token), dispatch timestamps via hipExtLaunchKernel; the
    steps are chained greedy steps continuing the timed loop (state on the device)."""
    from lit_llama_amd._native import check, lib

    evs = []
    with torch.cuda.stream(eng.stream):
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()  # creates the underlying hipEvent_t objects (lazily allocated by torch)
            e1.record()
            check(lib().mi355_debug_time_next_launch(C.c_void_p(e0.cuda_event), C.c_void_p(e1.cuda_event)), "hook")
            eng.run_step(3)
            evs.append((e0, e1))
    eng.stream.synchronize()
    times = sorted(a.elapsed_time(b) * 1e-3 for a, b in evs[8:])
    return sum(times) / len(times), times[len(times) // 2]


def measure_dominant_kernel(eng, reps: int = 3):
    """Average duration of the c_fc1/c_fc2 + SwiGLU launch (segment 2 of every layer), events on eng.stream."""
    from lit_llama_amd._native import check, lib

    n_layer = eng.cfg.n_layer
    s = eng.stream.cuda_stream
    evs = []
    # Q4 / bf16: the launcher hands the events to hipExtLaunchKernel, so they carry the dispatch's own begin / end
    # timestamps (what rocprofv3 --kernel-trace reports).  int8 (no hook): events recorded around the launch.
    hook = eng.m.layers[0].fc.fmt != 2
    with torch.cuda.stream(eng.stream):
        for _ in range(reps):
            for l in range(n_layer):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if hook:
                    e0.record()  # creates the underlying hipEvent_t objects (lazily allocated by torch)
                    e1.record()
                    check(lib().mi355_debug_time_next_launch(C.c_void_p(e0.cuda_event), C.c_void_p(e1.cuda_event)), "hook")
                else:
                    e0.record()
                check(lib().mi355_forward_segment(C.byref(eng.m), 1, l, 2, 3, s), "forward_segment")
                if not hook:
                    e1.record()
                evs.append((e0, e1))
    eng.stream.synchronize()
    times = sorted(a.elapsed_time(b) * 1e-3 for a, b in evs[n_layer:])  # first pass = warm-up
    return sum(times) / len(times), times[len(times) // 2]


def cpu_baseline_reference(cfg, mode: str):
    """The REAL reference (imported from /root/reference through oracle/_stubs) on the same bounded sample as the
    port: ONE 7B-width Block + ln_f + lm_head, 10 decode steps after a 1-token prompt, extrapolated to n_layer."""
    sys.path[:0] = [str(ROOT / "oracle" / "_stubs"), "/root/reference"]
    import lit_llama as ref
    from lit_llama.utils import quantization as ref_quantization

    from lit_llama_amd import synth
    from lit_llama_amd.model import LLaMAConfig

    qmode = mode if mode != "none" else None
    small = LLaMAConfig(n_layer=1, n_head=cfg.n_head, n_embd=cfg.n_embd)
    with ref_quantization(qmode):
        model = ref.LLaMA(ref.LLaMAConfig(n_layer=1, n_head=cfg.n_head, n_embd=cfg.n_embd))
    model.load_state_dict(synth.make_state_dict(small, seed=0, mode=qmode))
    model.eval()
    steps = 10
    with torch.no_grad():
        model(torch.tensor([[1]]), 16, torch.tensor([0]))
        blk = model.transformer.h[0]
        t_layer = t_head = 0.0
        for i in range(steps):
            x = torch.randn(1, 1, cfg.n_embd)
            pos = torch.tensor([1 + i])
            rope = model.rope_cache.index_select(0, pos)
            mask = model.mask_cache.index_select(2, pos)[:, :, :, :16]
            t0 = time.perf_counter()
            x, model.kv_caches[0] = blk(x, rope, mask, 16, pos, model.kv_caches[0])
            t1 = time.perf_counter()
            model.lm_head(model.transformer.ln_f(x))
            t2 = time.perf_counter()
            t_layer += (t1 - t0) / steps
            t_head += (t2 - t1) / steps
    per_token = cfg.n_layer * t_layer + t_head
    return dict(value=1.0 / per_token, unit="tokens/s", cores=torch.get_num_threads(), kind="reference",
                sample=f"/root/reference lit_llama (unmodified), 1 of {cfg.n_layer} layers at {args_model_name(cfg)} width "
                       f"+ lm_head, {steps} decode steps ({steps * (t_layer + t_head):.1f} s measured), extrapolated: "
                       f"{cfg.n_layer} x {t_layer:.2f} s + {t_head:.2f} s per token")


def cpu_baseline(cfg, mode: str):
    """The reference's CPU path on a bounded sample: ONE 7B-width layer + lm_head, 10 decode steps after a 1-token
    prompt (~10 s); per-token time extrapolated to n_layer layers.  The real reference when its tree is present,
    else the oracle (its restatement)."""
    if Path("/root/reference/lit_llama/model.py").exists():
        try:
            return cpu_baseline_reference(cfg, mode)
        except Exception:  # fall through to the port
            pass
    from lit_llama_amd import synth
    from lit_llama_amd.model import LLaMAConfig
    from oracle import oracle

    small = LLaMAConfig(n_layer=1, n_head=cfg.n_head, n_embd=cfg.n_embd)
    sd = synth.make_state_dict(small, seed=0, mode=mode if mode != "none" else None)
    om = oracle.Model(oracle.Config(n_layer=1, n_head=cfg.n_head, n_embd=cfg.n_embd), sd,
                      mode=mode if mode != "none" else None)
    idx = torch.tensor([[1]], dtype=torch.int64)
    with torch.no_grad():
        om(idx, 16, torch.tensor([0]))  # prefill (also builds caches)
        t_layer = t_head = 0.0
        steps = 10
        for i in range(steps):
            x = torch.randn(1, 1, cfg.n_embd)
            rope = om.rope_cache.index_select(0, torch.tensor([1 + i]))
            mask = om.mask_cache.index_select(2, torch.tensor([1 + i]))[:, :, :, :16]
            t0 = time.perf_counter()
            x, om.kv_caches[0] = om.block(0, x, rope, mask, 16, torch.tensor([1 + i]), om.kv_caches[0])
            t1 = time.perf_counter()
            oracle.linear(om.sd, "lm_head", oracle.rmsnorm(x, om.p("transformer.ln_f.scale")), om.mode)
            t2 = time.perf_counter()
            t_layer += (t1 - t0) / steps
            t_head += (t2 - t1) / steps
    per_token = cfg.n_layer * t_layer + t_head
    return dict(value=1.0 / per_token, unit="tokens/s", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle/oracle.py (reference CPU path restated), 1 of {cfg.n_layer} layers at {args_model_name(cfg)} "
                       f"width + lm_head, {steps} decode steps ({steps * (t_layer + t_head):.1f} s measured), "
                       f"extrapolated: {cfg.n_layer} x {t_layer:.2f} s + {t_head:.2f} s per token")


def tp_leg(args, dev, rank, world, dist):
    """BASELINE.json configs[4]: LLaMA-65B gptq.int4 decoded tensor-parallel over the `world` GPUs of this job (TP = 1
    on a single GPU: the same code path at world 1).  Every rank holds its shard (scripts/convert_checkpoint.py:57-65),
    a decode step is one hipGraph replay per rank: 2 x n_layer [segments + peer-write all-reduce] + lm_head shard +
    sharded arg-max (lit_llama_amd/tp.py, csrc/tp_comm.hip); `--tp-comm rccl` runs the host-driven protocol over
    torch.distributed instead (the comparison point)."""
    from lit_llama_amd import synth, tp
    from lit_llama_amd.model import LLaMAConfig

    cfg = LLaMAConfig.from_name(args.tp_model)
    tp.check_divisible(cfg, world)
    model = tp.build_local_model(cfg, world, device=dev, mode="gptq.int4")
    synth.fill_tp_shard_random_int4(model, seed=0, rank=rank)
    shard = tp.EngineShard(model, world)
    T, K = 32, args.tp_steps
    prompt = synth.make_prompt(T, vocab=cfg.vocab_size, seed=99).to(dev)
    if args.tp_comm == "native":
        if world > 1:
            exchange = None  # torch.distributed.all_gather_object over the job's process group
        else:
            exchange = lambda obj: [obj]  # noqa: E731
        comm = tp.NativeComm(rank, world, cfg.n_embd, dev, exchange=exchange)
    else:
        comm = tp.DistComm() if world > 1 else tp.LoopbackComm(1)
    dec = tp.TPDecoder([shard], comm, cfg)
    run = (lambda n: dec.generate_chained(prompt, n, max_seq_length=T + K + 8)) if args.tp_comm == "native" else \
          (lambda n: dec.generate(prompt, n, max_seq_length=T + K + 8))
    run(8)  # warm: cache geometry, kernel attributes
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    # prompt + 8 tokens vs prompt + 8 + K tokens: the difference is K decode steps
    t0 = time.perf_counter()
    run(8)
    torch.cuda.synchronize(dev)
    t_short = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    toks = run(8 + K)
    torch.cuda.synchronize(dev)
    t_long = time.perf_counter() - t0
    dt = torch.tensor([t_long - t_short], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    per_token = float(dt.item()) / K
    # the collective alone: 2 x n_layer all-reduces of [n_embd] f32 per token
    ar_us = None
    if args.tp_comm == "native":
        eng = shard.eng
        with torch.cuda.stream(eng.stream):
            comm.step_begin(eng.stream)
            for _ in range(16):
                comm.reduce_add(shard, 1, force=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            comm.step_begin(eng.stream)
            e0.record(eng.stream)
            for _ in range(2 * cfg.n_layer):
                comm.reduce_add(shard, 1, force=True)  # (at world 1 the decode step itself issues none: tp.NativeComm.reduce_add)
            e1.record(eng.stream)
        e1.synchronize()
        comm.check_status()
        ar_us = e0.elapsed_time(e1) * 1e3 / (2 * cfg.n_layer)
    ranks_seen = world
    if args.tp_comm != "native" and world > 1 and dist is not None:
        # RCCL alone: 2 x n_layer all-reduces of [n_embd] f32, back to back on the current stream; and how many ranks it reaches
        buf = torch.ones(cfg.n_embd, dtype=torch.float32, device=dev)
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        for _ in range(8):
            dist.all_reduce(buf)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2 * cfg.n_layer):
            dist.all_reduce(buf)
        e1.record()
        e1.synchronize()
        ar_us = e0.elapsed_time(e1) * 1e3 / (2 * cfg.n_layer)
    bpt = bytes_per_token(cfg, "gptq.int4")
    per_gpu = (bpt["weights"] - cfg.padded_vocab_size * cfg.n_embd // 2) // world + cfg.padded_vocab_size * cfg.n_embd // 2 // world
    assert int(toks.min()) >= 0 and int(toks.max()) < cfg.padded_vocab_size
    out = {
        "workload": f"LLaMA-{args.tp_model} gptq.int4 bs=1 greedy decode, TP={world} (configs[4]"
                    + ("" if world == 8 and args.tp_model == "65B" else f" at TP={world}") + "), random-init shards",
        "tokens_per_s": round(1.0 / per_token, 2),
        "ms_per_token": round(per_token * 1e3, 4),
        "steps": K,
        "comm": "peer-write all-reduce (mi355_tp_allreduce, HIP IPC over xGMI), hipGraph per step" if args.tp_comm == "native"
                else "RCCL all-reduce via torch.distributed, host-driven segments",
        "weight_bytes_per_gpu_per_token": int(per_gpu),
        "frac_of_int4_weight_roofline_per_gpu": round(per_gpu / per_token / HBM_PEAK, 4),
        "allreduce_us": None if ar_us is None else round(ar_us, 2),
        "collective_us_per_token": None if ar_us is None else (0.0 if world == 1 else round(ar_us * 2 * cfg.n_layer, 1)),
        "collective_launches_per_token": 0 if (world == 1 and args.tp_comm == "native") else 2 * cfg.n_layer + 1,
        "ranks_seen": ranks_seen,
        "note": "TP > 1 has not been run on hardware by the builder (1-GPU boxes only): the curve is whatever the driver's "
                "multi-GPU run prints here",
    }
    if hasattr(comm, "close"):
        comm.close()
    return out


def _agree(bad: bool, dev, dist) -> bool:
    """MAX over ranks of a failure flag (one all-reduce): every rank takes the same decision about a retry."""
    t = torch.tensor([1.0 if bad else 0.0], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item()) > 0


def run_tp_legs(args, world, dist, leg, agree):
    """The tensor-parallel part of the line (BASELINE.json configs[4]), watchdogged: returns (`tp` object, hung).

    world == 1: ONE leg (`--tp-comm`, default the native collective, which issues nothing at world 1).
    world > 1 : BOTH collectives in one run, so that the first lease of a multi-GPU node yields a comparison and not a stack trace
                (north_star says "RCCL all-reduce over xGMI", the default is the builder's own peer-write
                collective): `tp` = the native leg — or, LOUDLY (`fallback_from_native` + a line on stderr), the RCCL leg when the
                native one failed on any rank — and `tp["rccl"]` = the same decode over RCCL (tokens/s, ms per token,
                collective_us_per_token, ranks RCCL saw).  `leg(args)` runs one leg on this rank and returns its dict; `agree(bad)` is
                the MAX of a failure flag over the ranks.  A leg that HANGS (a rank that failed alone leaves its peers in a
                collective / a peer-write that never arrives) is abandoned after --tp-timeout seconds: the headline line is printed
                and the process exits without joining it."""
    import copy
    import threading

    box = {}

    def one(comm):
        a2 = copy.copy(args)
        a2.tp_comm = comm
        try:
            return leg(a2), None
        except BaseException as e:  # noqa: BLE001 (the TP leg must never take the headline down with it)
            return None, repr(e)

    def _run():
        if world == 1:
            res, err = one(args.tp_comm)
            box["res"] = res if err is None else {"error": err}
            return
        try:
            native, err = one("native")
            failed = agree(err is not None)           # the ranks agree: a native leg that failed ANYWHERE is not a result
            rccl, err2 = one("rccl")
            failed2 = agree(err2 is not None)
            if failed2:
                rccl = {"error": err2 or "the RCCL leg failed on another rank"}
            if failed:
                why = err or "the native leg failed on another rank"
                print(f"bench.py: the native peer-write collective FAILED at world {world} ({why}); `tp` reports the RCCL leg",
                      file=sys.stderr, flush=True)
                res = dict(rccl)
                res["fallback_from_native"] = why
            else:
                res = native
                res["rccl"] = rccl
            box["res"] = res
        except BaseException as e:  # noqa: BLE001
            box["res"] = {"error": f"tensor-parallel legs: {e!r}"}

    th = threading.Thread(target=_run, daemon=True)
    th.start()
    th.join(args.tp_timeout * (1 if world == 1 else 2))
    hung = th.is_alive()
    res = {"error": f"timeout: the tensor-parallel legs did not finish within {args.tp_timeout * (1 if world == 1 else 2)} s"} if hung \
        else box.get("res")
    return res, hung


def args_model_name(cfg):
    return {4096: "7B", 5120: "13B", 6656: "30B", 8192: "65B"}.get(cfg.n_embd, f"n_embd={cfg.n_embd}")


def spawn_command(n: int, argv):
    """`python bench.py --gpus N` without a launcher -> the launch line the driver uses for N > 1 (one rank per GPU,
    rendezvous on 127.0.0.1: the container hostname may not resolve)."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), *argv]


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a plain `python bench.py --gpus N` (the form used for N = 1): start the N ranks ourselves, one per GPU
        import subprocess

        cmd = spawn_command(args.gpus, sys.argv[1:])
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the job must have one rank per requested GPU")
    if args.dry_run:
        # launch-contract check without a GPU (tests/test_host_logic.py): every rank joins a gloo group, rank 0 reports
        seen = [(rank, local_rank)]
        tp_res = None
        if world > 1:
            import torch.distributed as dist_

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist_.init_process_group("gloo")
            seen = [None] * world
            dist_.all_gather_object(seen, (rank, local_rank))
        if not args.no_tp:
            # the output contract of the tensor-parallel part, with stub legs over gloo: which legs run, what the line carries, and
            # that a native leg failing on ONE rank (--dry-run-fail-native R) makes EVERY rank report the RCCL leg, loudly
            def stub(a2):
                saw = world
                if world > 1:
                    t = torch.ones(1)
                    dist_.all_reduce(t)
                    saw = int(t.item())
                if a2.tp_comm == "native" and args.dry_run_fail_native == rank:
                    raise RuntimeError("dry run: native collective failed on this rank")
                return {"workload": f"dry run, TP={world}", "tokens_per_s": 0.0, "ms_per_token": 0.0, "steps": 0, "comm": a2.tp_comm,
                        "collective_us_per_token": 0.0, "ranks_seen": saw}

            def agree(bad):
                if world == 1:
                    return bad
                t = torch.tensor([1.0 if bad else 0.0])
                dist_.all_reduce(t, op=dist_.ReduceOp.MAX)
                return float(t.item()) > 0

            tp_res, _ = run_tp_legs(args, world, dist_ if world > 1 else None, stub, agree)
        if world > 1:
            dist_.destroy_process_group()
        if rank == 0:
            out = {"dry_run": True, "n_gpus": world, "ranks": sorted(r for r, _ in seen)}
            if tp_res is not None:
                out["tp"] = tp_res
            print(json.dumps(out))
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible); there is no CPU fallback for the hot path")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    if args.no_graph:
        os.environ["MI355_GRAPH"] = "0"

    from lit_llama_amd import synth

    model, cfg = build_model(args, dev)
    eng = model.engine()
    if eng is None:
        raise SystemExit(f"native engine unavailable: {model._engine_failed}")
    T, W, K = args.prompt_len, args.warmup, args.steps
    S = T + W + K + 1 + 64  # + the launches the roofline measurement appends to the chained loop
    BLOCKS = 3  # timed blocks of EXACTLY K steps over the SAME positions; the median block is reported
    if S > cfg.block_size:
        raise SystemExit(f"prompt + warmup + steps + 1 = {S} exceeds block_size {cfg.block_size}")
    prompt = synth.make_prompt(T, vocab=cfg.vocab_size, seed=1234 + rank).to(dev)

    with torch.cuda.stream(eng.stream):
        eng._ensure_cache(S)
        eng.out_tokens[:T].copy_(prompt)
        t_pf0 = time.perf_counter()
        eng.prefill(prompt, 0, all_logits=False, argmax=True)
        eng.stream.synchronize()
        t_prefill = time.perf_counter() - t_pf0
        pos = T
        # same loop as lit_llama_amd.generate._generate_greedy: chained greedy steps, one graph replay per token
        eng.set_step(None, 1, pos, from_next=True)
        eng.embed_step()
        for _ in range(W):
            eng.run_step(3)
            pos += 1
        # Three timed blocks of EXACTLY K steps, each between (barrier +) torch.cuda.synchronize() on both sides.  Blocks 2 and 3
        # rewind the chain to the first timed position — the same tokens are decoded again over the same cache rows — so that every
        # block is the workload `config.workload` names; the MEDIAN block (of the per-block MAX over ranks) is what `value` and
        # `ms_per_step` report, all three are on the line (`blocks_ms_per_step`).  One block of 20 steps is 18 ms on boxes whose
        # clocks wander by 2-5 %.
        p0 = pos
        blocks = []
        for b in range(BLOCKS):
            if b:
                eng.set_step(eng.out_tokens[p0:p0 + 1], 1, p0)
                eng.embed_step()
                pos = p0
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(K):
                eng.run_step(3)
                pos += 1
            torch.cuda.synchronize(dev)
            if dist is not None:
                dist.barrier()
            blocks.append(time.perf_counter() - t0)
    if dist is not None:
        t = torch.tensor(blocks, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        blocks = [float(v) for v in t.tolist()]
    elapsed = sorted(blocks)[len(blocks) // 2]
    # a timed-out hand-off of the fused step must never be reported as a rate (raises); neither must steps whose activations left
    # the range of the step's hand-off format (the engine would have demoted itself: the blocks then mix two kernels)
    if eng.check_status() is not None:
        raise SystemExit(f"bench: the persistent step clipped activations on the synthetic bench model ({eng.fused_demotions}): "
                         "the timed rate is not the rate of one kernel")
    tokens = eng.out_tokens[: pos + 1].tolist()
    assert all(0 <= t_ < cfg.padded_vocab_size for t_ in tokens), "decode produced invalid ids"
    fused = eng.fused_ready()
    eng_fmt = int(eng.fused.weight_fmt) if fused else -1
    f8_operands = fused and int(eng.fused.weight_fmt) in (3, 5, 6)  # (the fp8-limb operand path of the int4 / 8-bit steps)
    hipgraph_used = bool(eng.use_graph and eng._graphs) and not fused
    if rank == 0:
        bpt = bytes_per_token(cfg, args.quantize, u8_stream=eng_fmt == 6)
        tok_s_gpu = K / elapsed
        mean_pos = T + W + K / 2
        # ---- dominant kernel roofline (c_fc1/c_fc2 + SwiGLU)
        C_, H = cfg.n_embd, cfg.n_hidden
        wb = {"gptq.int4": C_ * H, "llm.int8": 2 * C_ * H, "none": 4 * C_ * H, "gptq.int8": 4 * C_ * H}[args.quantize]
        side = {"gptq.int4": 2 * 2 * H * 2, "llm.int8": 2 * H * 4, "none": 0, "gptq.int8": 0}[args.quantize]
        algo = wb + side + C_ * 4 + C_ * 2 + H * 2
        traffic, traffic_source = None, None
        pmc = ROOT / "profiles" / "pmc_traffic.json"
        if fused:
            # one launch = one token: every weight byte once + side operands + the KV rows read at the step's position
            pos_mid = pos + 24  # the 48 measured launches continue the chained loop
            algo = bpt["total"] + int(bpt["kv_per_pos"] * (pos_mid + 1))
            avg_s, med_s = measure_fused_step(eng)
            assert eng.check_status() is None
            key = "fused_step_bytes_per_launch"
        else:
            avg_s, med_s = measure_dominant_kernel(eng)
            key = "fc_swiglu_bytes_per_launch"
        if pmc.exists() and args.model == "7B" and args.quantize == "gptq.int4" and not args.group_cols and not args.adapter:
            try:  # (the committed PMC pass is of the headline configuration)
                doc = json.loads(pmc.read_text())
                traffic = doc.get(key)
                kname = None
                if fused:
                    # the entry of the kernel that was TIMED: fused_step_ring_kernel<GRP, FMT> with FMT = the engine's weight_fmt
                    kname = f"fused_step_ring_kernel<false, {eng_fmt}>"
                    ent = doc.get("kernels", {}).get(kname)
                    traffic = round(ent["corrected_bytes_per_launch"]) if ent else None
                traffic_source = (f"profiles/pmc_traffic.json: committed rocprofv3 --pmc FETCH_SIZE pass over this command"
                                  + (f", kernel {kname}" if kname else "") +
                                  " (x2 gfx950 wide-read correction); static, not measured in this run") if traffic else None
            except Exception:
                traffic = None
        # what a checkpoint that leaves the fp8 hand-off's range gets: the SAME K steps over the same positions on
        # the next rung of the engine's ladder — fp16 operands, `fused_step_ring_kernel<false, 0>` (wide shapes: weight_fmt 4) — one block,
        # same engine, same box
        rungs = None
        if f8_operands and eng_fmt in (3, 5):
            with torch.cuda.stream(eng.stream):
                eng.use_fused_format(0 if eng_fmt == 3 else 4)
                eng.set_step(eng.out_tokens[p0:p0 + 1], 1, p0)
                eng.embed_step()
                for _ in range(4):
                    eng.run_step(3)
                eng.set_step(eng.out_tokens[p0:p0 + 1], 1, p0)
                eng.embed_step()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(K):
                    eng.run_step(3)
                torch.cuda.synchronize(dev)
                t_r = time.perf_counter() - t0
            clipped0 = eng.check_status() is not None
            eng.reset_fused_format()
            rungs = {"fp8x3": round(tok_s_gpu, 2), "fp16": None if clipped0 else round(K / t_r, 2),
                     "what": "tokens/s of the same K steps on each rung of the persistent step's hand-off ladder (fp8-limb operands -> fp16 "
                             "operands -> launch-per-operator step); a position whose activations leave a rung's range is replayed one "
                             "rung down and the engine climbs back after 16 clean steps (engine.check_status)"}
        # tokens/s of generate() including the prompt (generate.py:146-153 prints this figure)
        gen_new = 64
        import lit_llama_amd

        model.reset_cache()
        lit_llama_amd.generate(model, prompt, 4, top_k=1, max_seq_length=T + gen_new)  # warm (cache geometry, graphs)
        torch.cuda.synchronize(dev)
        t_g0 = time.perf_counter()
        lit_llama_amd.generate(model, prompt, gen_new, top_k=1, max_seq_length=T + gen_new)
        torch.cuda.synchronize(dev)
        t_gen = time.perf_counter() - t_g0
        # the same run with sampling (temperature 0.8, top_k 200: generate.py's defaults) on the device
        model.reset_cache()
        lit_llama_amd.generate(model, prompt, 4, temperature=0.8, top_k=200, max_seq_length=T + gen_new)
        torch.cuda.synchronize(dev)
        t_s0 = time.perf_counter()
        lit_llama_amd.generate(model, prompt, gen_new, temperature=0.8, top_k=200, max_seq_length=T + gen_new)
        torch.cuda.synchronize(dev)
        t_samp = time.perf_counter() - t_s0
        demotions = [f"{why} -> {to} (from position {bad})" for why, to, bad in getattr(eng, "fused_demotions", [])]
        # prompt prefill at the reference's evaluation length (evaluate/full.py:120-129: T = 2048): wide int4 GEMM +
        # flash attention, MFMA-bound
        prefill = None
        if args.quantize in ("gptq.int4", "none", "gptq.int8") and cfg.block_size >= 2048:
            T2 = 2048
            long_prompt = synth.make_prompt(T2, vocab=cfg.vocab_size, seed=4321).to(dev)
            pf_ms = []
            with torch.cuda.stream(eng.stream):
                eng._ensure_cache(T2 + 8)
                eng.prefill(long_prompt, 0, all_logits=False, argmax=True)  # warm
                # three timed passes over the same prompt, the MEDIAN is reported like the decode leg's blocks (VERDICT r5 weak 3:
                # the driver saw 23.99 and then 25.37 ms from one unchanged kernel, one shot each)
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(eng.stream)
                    eng.prefill(long_prompt, 0, all_logits=False, argmax=True)
                    e1.record(eng.stream)
                    e1.synchronize()
                    pf_ms.append(e0.elapsed_time(e1))
            ms = sorted(pf_ms)[1]
            L = cfg.n_layer
            flops = 2.0 * T2 * L * (4 * C_ * C_ + 3 * C_ * H) + 2.0 * 2.0 * L * C_ * T2 * (T2 + 1) / 2 + 2.0 * C_ * cfg.padded_vocab_size
            prefill = {"tokens": T2, "ms": round(ms, 2), "runs_ms": [round(v, 2) for v in pf_ms], "timing": "median of 3 passes",
                       "tokens_per_s": round(T2 / ms * 1e3, 1),
                       "tflops": round(flops / ms / 1e9, 1), "peak_tflops": 2500.0, "bound": "mfma",
                       "frac": round(flops / ms / 1e9 / 2500.0, 4), "chunk": eng.max_T,
                       "chain": "staged (MI355_GEMM_FUSE=0)" if os.environ.get("MI355_GEMM_FUSE", "1")[:1] == "0" else
                       "fused: producers' epilogues write the next operand / the K, V cache rows (csrc/gemm_fuse.h)",
                       "kernels": ("gemm_q4_kernel (int4 stream -> bf16 MFMA 16x16x32)" if args.quantize == "gptq.int4" else
                                   "gemm_q4_kernel<BF16> (bf16 stream, MFMA 16x16x32)") + " + flash_prefill_kernel"}
    tp_res = None
    if not args.no_tp and args.quantize == "gptq.int4":
        # free the 7B replica first: the 65B shard of a small world is tens of GB
        del eng
        model._engine = None
        del model
        torch.cuda.empty_cache()
        torch.cuda.set_device(dev)
        tp_res, tp_hung = run_tp_legs(args, world, dist, lambda a2: tp_leg(a2, dev, rank, world, dist),
                                      agree=lambda bad: _agree(bad, dev, dist))
    else:
        tp_hung = False
    if rank != 0:
        if tp_hung:
            os._exit(0)
        if dist is not None:
            dist.destroy_process_group()
        return
    variant = args.adapter or args.group_cols > 0
    default_cfg = args.model == "7B" and args.quantize == "gptq.int4" and not variant
    cfg_idx = {"none": 1, "gptq.int4": 2, "llm.int8": 3}.get(args.quantize) if args.model == "7B" and not variant else None
    # (gptq.int8: the persistent step streams the 8-bit levels, weight_fmt 6; with MI355_FUSED_U8=0 the dequantised bf16 matrices: engine._dense_weight)
    wname = {"gptq.int4": "int4", "llm.int8": "int8", "none": "bf16", "gptq.int8": "int8" if eng_fmt == 6 else "bf16"}[args.quantize]
    out = {
        "metric": METRIC if default_cfg else f"decode tokens/sec/GPU LLaMA-{args.model}{' + LLaMA-Adapter' if args.adapter else ''} "
                                             f"{args.quantize}{' groupsize %d' % args.group_cols if args.group_cols else ''} bs=1; % HBM roofline",
        "value": round(tok_s_gpu * world, 2),
        "unit": "tokens/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": round(1e3 * elapsed / K, 4),
        "blocks_ms_per_step": [round(1e3 * b / K, 4) for b in blocks],
        "timing": f"median of {len(blocks)} blocks of {K} steps each over the same positions (every block between synchronize() on both sides)",
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        # what the timed kernels compute with (not a precision claim): the persistent int4 step feeds the MX-scaled fp8 MFMA (int4
        # levels are exact E4M3 codes; every activation travels as three E4M3 limbs = 12 significant bits, one more than the fp16
        # granules of MI355_FUSED_F8=0; distance from the reference's f32 run at full depth 0.016 logit-std either way, the
        # reference's own bf16 run 0.086), the launch-per-operator paths bf16 operands, LLM.int8 int8 x int8 -> int32
        # plus f16 outlier columns; f32 accumulation and a bf16 KV cache throughout (>= the reference's bf16-true run)
        "dtype": ("fp8x3" if f8_operands else
                  "fp16" if fused and args.quantize == "gptq.int4" else "int8" if args.quantize == "llm.int8" else "bf16"),
        "dtype_detail": ("int4 weights as exact E4M3 bytes x activations as THREE E4M3 limbs (12 significant bits) on the MX-scaled fp8 MFMA "
                         f"(weight_fmt {eng_fmt}, the default; MI355_FUSED_F8=0: fp16 operands), f32 accumulate, bf16 KV cache" if f8_operands else
                         "int4 weights -> fp16 MFMA operands x fp16 activations, f32 accumulate, bf16 KV cache"
                         if fused and args.quantize == "gptq.int4" else
                         "int8 x int8 -> int32 MFMA + f16 outlier columns, bf16 KV cache" if args.quantize == "llm.int8" else
                         "bf16 MFMA operands, f32 accumulate, bf16 KV cache"),
        "data": "synthetic",
        "config": {
            "workload": f"LLaMA-{args.model} {args.quantize} bs=1 greedy decode "
                        + (f"(configs[{cfg_idx}]), " if cfg_idx is not None else "(single GPU, not a BASELINE config), ")
                        + "random-init weights, "
                        f"prompt {T} tokens, positions {T + W}..{T + W + K - 1}",
            "prompt_len": T,
            # (the synthetic checkpoint's statistics: zero-mean int4 levels of unit gain since round 5 — rounds 1-4 timed zero point 8 /
            # gain 2.2; the rate does not see the values, the parity bars do: tests/test_model_gpu.py runs both)
            "model_stats": {"int4_zero_point": 7.5, "gain": 1.0} if args.quantize in ("gptq.int4", "gptq.int8") else None,
            "max_seq_length": S,
            "parallelism": "single GPU" if world == 1 else f"{world} independent replicas (bs=1 path does not shard)",
            "hipgraph": hipgraph_used,
            "fused_step": fused,
            "launches_per_token": 1 if fused else cfg.n_layer * 5 + 2,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": (("fused_step_wide_kernel" if int(eng_fmt) in (4, 5) else "fused_step_ring_kernel") +
                       " (the whole decode step, one launch per token)") if fused else
                      {"gptq.int4": "gemv_kernel<Q4,R=2,SwiGLU>", "none": "gemv_kernel<BF16,R=2,SwiGLU>", "gptq.int8": "gemv_kernel<BF16,R=2,SwiGLU>",
                       "llm.int8": "int8_gemv_kernel<R=2>"}[args.quantize] + " (c_fc1/c_fc2 pair)",
            "achieved": round(algo / avg_s / 1e9, 1),
            "peak": HBM_PEAK / 1e9,
            "unit": "GB/s",
            "frac": round(algo / avg_s / HBM_PEAK, 4),
            "traffic": traffic if default_cfg else None,
            "traffic_source": traffic_source if default_cfg else None,
            "algorithmic_bytes_per_launch": algo,
            "avg_launch_us": round(avg_s * 1e6, 2),
            "median_launch_us": round(med_s * 1e6, 2),
        },
        "decode_roofline": {
            "bytes_per_token_weights": bpt["weights"],
            "bytes_per_token_total": bpt["total"] + int(bpt["kv_per_pos"] * (mean_pos + 2)),
            "tokens_per_s_at_100pct": round(HBM_PEAK / bpt["weights"], 1),
            f"frac_of_{wname}_weight_roofline": round(tok_s_gpu * bpt["weights"] / HBM_PEAK, 4),
        },
        "rungs": rungs,
        "prefill_s": round(t_prefill, 4),
        "prefill": prefill,
        "generate": {"tokens_per_s_incl_prompt": round(gen_new / t_gen, 1), "sampled_tokens_per_s_incl_prompt": round(gen_new / t_samp, 1),
                     "prompt_len": T, "new_tokens": gen_new,
                     # (the random bench model leaves the fp8 hand-off's range on some SAMPLED tokens: the engine then moves to fp16 operands
                     # and generate() recomputes from the clipped position — the sampled figure includes that replay when it happened)
                     "hand_off_demotions": demotions,
                     "what": "lit_llama_amd.generate(top_k=1) wall time incl. prefill, as generate.py:146-153 reports"},
    }
    if tp_res is not None:
        out["tp"] = tp_res
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(cfg, args.quantize)
        except Exception as e:  # the baseline must never take the headline down with it
            out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": f"failed: {e!r}"}
    print(json.dumps(out), flush=True)
    if tp_hung:
        sys.stdout.flush()
        os._exit(0)  # a hung collective cannot be joined
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

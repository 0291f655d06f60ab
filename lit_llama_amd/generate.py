"""Token sampling loop with the signature of /root/reference generate.py:20-91.

`generate(model, idx, max_new_tokens, *, max_seq_length, temperature, top_k, eos_id)` returns the prompt
followed by the generated ids, like the reference.  Two ways through it:

  * greedy (`top_k == 1`, which is how the reference spells greedy: generate.py:70-76 leaves a single finite
    logit) on an engine-backed model: prefill once, then per token one `set_step` launch + one hipGraph replay
    whose last node is the on-device argmax that feeds the next step.  The host never reads a device value
    inside the loop (the reference syncs once per layer per token, model.py:214), so launches run ahead of
    the GPU;
  * everything else: the reference's loop, calling `model(x, max_seq_length, input_pos)` and sampling with
    torch ops on the returned logits.
"""
from __future__ import annotations

from typing import Optional

import torch


def _host_pos(t: torch.Tensor, pos0: int) -> torch.Tensor:
    """Attach the host-known first position so the engine needs no device->host sync to learn it."""
    t._mi355_pos0 = pos0
    return t


@torch.no_grad()
def generate(
    model,
    idx: torch.Tensor,
    max_new_tokens: int,
    *,
    max_seq_length: Optional[int] = None,
    temperature: float = 1.0,
    top_k: Optional[int] = None,
    eos_id: Optional[int] = None,
    sample_on_device: bool = True,
) -> torch.Tensor:
    T = idx.size(0)
    T_new = T + max_new_tokens
    if max_seq_length is None:
        max_seq_length = min(T_new, model.config.block_size)
    device, dtype = idx.device, idx.dtype

    eng = model.engine() if (getattr(model, "use_engine", False) and device.type == "cuda") else None
    if eng is not None and top_k == 1 and T_new <= max_seq_length and max_new_tokens > 0:
        return _generate_greedy(model, eng, idx, max_new_tokens, max_seq_length, eos_id)
    if eng is not None and temperature > 0 and T_new <= max_seq_length and max_new_tokens > 0 and sample_on_device:
        return _generate_sampled(model, eng, idx, max_new_tokens, max_seq_length, temperature, top_k, eos_id)

    # ---- reference loop (generate.py:45-91)
    empty = torch.empty(T_new, dtype=dtype, device=device)
    empty[:T] = idx
    idx = empty
    pos_host = 0
    input_pos = _host_pos(torch.arange(0, T, device=device), 0)
    for _ in range(max_new_tokens):
        x = idx.index_select(0, input_pos).view(1, -1)
        logits = model(x, max_seq_length, input_pos)
        logits = logits[0, -1] / temperature
        if top_k is not None:
            v, _ = torch.topk(logits, min(top_k, logits.size(-1)))
            logits = torch.where(logits < v[[-1]], -float("Inf"), logits)
        probs = torch.nn.functional.softmax(logits, dim=-1)
        idx_next = torch.multinomial(probs, num_samples=1).to(dtype=dtype)
        pos_host = pos_host + input_pos.numel() if input_pos.numel() > 1 else pos_host + 1
        input_pos = _host_pos(input_pos[-1:] + 1, pos_host)
        idx = idx.index_copy(0, input_pos, idx_next)
        if eos_id is not None and idx_next == eos_id:
            # generate.py:88-89 returns `idx[:input_pos]`: despite its comment the slice stops BEFORE the
            # EOS position; kept as is so callers see the same length as with the reference
            return idx[:input_pos]
    return idx


# The persistent step records a hand-off that left the range of its format; the host learns of it at its next status read.  The loops
# below read every CHECK_EVERY tokens (one 16-byte device->host read: ~30 us against ~14 ms of decoding) and at their end, so a clip
# costs at most that many recomputed steps — not the rest of a long generation.
CHECK_EVERY = 16


def _replay_from(eng, bad: int, T: int, done: int):
    """The persistent step's activations left the range of its hand-off format at position `bad` (DecodeEngine.check_status: the
    engine has moved to a wider format / the launch-per-operator step by now).  out_tokens[: bad + 1] and the cache rows below
    `bad` come from unclipped steps: make out_tokens[bad] at position `bad` the step to run next.  The caller recomputes from there.
    `bad` must be a position THIS call decoded (the prompt's last token .. the last step issued): the status words are sticky, and a
    position left behind by another sequence (an interrupted call, a caller that drove run_step itself) would make the replay
    overwrite prompt tokens or skip ahead — generate() clears the words when it starts, and refuses anything out of range here."""
    if not (T - 1 <= bad <= T + done - 1):
        from ._native import NativeError

        raise NativeError(f"fused decode step: clipped position {bad} is outside this call's decoded range [{T - 1}, {T + done - 1}] "
                          "(stale status words: was the engine driven by another caller meanwhile?)")
    eng.set_step(eng.out_tokens[bad:bad + 1], 1, bad)
    eng.embed_step()


def _generate_sampled(model, eng, idx, max_new_tokens, max_seq_length, temperature, top_k, eos_id):
    """generate.py:63-91 with sampling, entirely on the device: per token one decode step (logits) + one `mi355_sample`
    launch (temperature, exact top-k threshold, softmax, inverse-CDF draw from a uniform of THIS call's torch generator
    state), which also writes the next step's token / position.  No device->host read inside the loop.  The draws are
    inverse-CDF, not torch.multinomial's: the same distribution, a different stream of samples for a given seed
    (`sample_on_device=False` runs the reference's torch ops instead)."""
    from . import ops

    device, dtype = idx.device, idx.dtype
    T = idx.size(0)
    cur = torch.cuda.current_stream(device)
    eng.stream.wait_stream(cur)
    with torch.cuda.stream(eng.stream):
        uniforms = torch.rand(max_seq_length + 1, device=device, dtype=torch.float32)  # indexed by position
        eng._ensure_cache(max_seq_length)
        eng.clear_status()  # (a clip position of an earlier sequence must not steer this one's replay)
        eng.out_tokens[:T].copy_(idx.to(torch.int32))
        eng.prefill(idx, 0, all_logits=False, argmax=False)  # logits of the last prompt token in row 0
        eng.set_step(idx[-1:], 1, T - 1)                      # position slot = T - 1: the draw lands at out_tokens[T]
        row = eng.logits[0, : eng.m.lm_head.N]
        kw = dict(out_tokens=eng.out_tokens, tokens=eng.tokens, advance=True)
        ops.sample(row, temperature, top_k, uniforms, eng.pos, eng.next_token, **kw)
        done = 1
        stop = False
        while True:
            bad = None
            while done < max_new_tokens and not stop:
                eng.run_step(0)
                ops.sample(row, temperature, top_k, uniforms, eng.pos, eng.next_token, **kw)
                done += 1
                if eos_id is not None and (done % 16 == 0 or done == max_new_tokens):
                    toks = eng.out_tokens[T:T + done].tolist()
                    stop = eos_id in toks
                if done % CHECK_EVERY == 0 and eng.status_due():
                    bad = eng.check_status()
                    if bad is not None:
                        break
            if bad is None:
                bad = eng.check_status()
            if bad is None:
                break
            # a step left the range of the persistent step's hand-off format: the draws are per position, so the replay through the
            # wider format continues the very same sample path
            _replay_from(eng, bad, T, done)
            done, stop = bad + 1 - T, False  # (0 when the clipped step was the prompt's last token, run as a T = 1 chunk)
        out = eng.out_tokens[:T + done].to(dtype).clone()
    cur.wait_stream(eng.stream)
    if eos_id is not None:
        gen = out[T:].tolist()
        if eos_id in gen:
            out = out[: T + gen.index(eos_id)]
    return out


def _generate_greedy(model, eng, idx, max_new_tokens, max_seq_length, eos_id):
    """Greedy decode entirely on the device: out[pos + 1] = argmax(logits(pos))."""
    device, dtype = idx.device, idx.dtype
    T = idx.size(0)
    cur = torch.cuda.current_stream(device)
    eng.stream.wait_stream(cur)
    with torch.cuda.stream(eng.stream):
        eng._ensure_cache(max_seq_length)
        eng.clear_status()  # (a clip position of an earlier sequence must not steer this one's replay)
        eng.out_tokens[:T].copy_(idx.to(torch.int32))
        # prompt: all but the last token without logits, then the last one with logits + argmax
        eng.prefill(idx, 0, all_logits=False, argmax=True)
        done = 1
        if max_new_tokens > 1:
            # entry state of the chained steps: token = the prompt's argmax, position T, its embedding in x.
            # From here every step's last node writes the next token / position / embedding itself.
            eng.set_step(None, 1, T, from_next=True)
            eng.embed_step()
        stop = False
        while True:
            bad = None
            while done < max_new_tokens and not stop:
                eng.run_step(3)
                done += 1
                if eos_id is not None and (done % 16 == 0 or done == max_new_tokens):
                    # bounded-lag EOS check: the reference tests every token (generate.py:88-89) and pays a
                    # device->host sync for it; here the sync is amortised and the tail is cut off afterwards
                    toks = eng.out_tokens[T:T + done].tolist()
                    stop = eos_id in toks
                if done % CHECK_EVERY == 0 and eng.status_due():
                    bad = eng.check_status()
                    if bad is not None:
                        break
            # a read after the loop as well: a hand-off of the fused step timed out -> raise, never garbage; activations past the range
            # of its hand-off format at some position -> the engine has moved to a wider format, recompute from that position
            if bad is None:
                bad = eng.check_status()
            if bad is None:
                break
            _replay_from(eng, bad, T, done)
            done, stop = bad + 1 - T, False  # out_tokens[T .. bad] stand; the step at `bad` produces out_tokens[bad + 1]
        out = eng.out_tokens[:T + done].to(dtype).clone()
    cur.wait_stream(eng.stream)
    if eos_id is not None:
        gen = out[T:].tolist()
        if eos_id in gen:
            out = out[: T + gen.index(eos_id)]  # same cut as the reference's `idx[:input_pos]`
    return out

"""GPU parity tests of the individual kernels, through the C ABI, against the oracle on the same seeded inputs.

Tolerances (written next to each check):
  * integer / byte work (repack layouts, int8 accumulation, argmax, embedding gather): bit-exact;
  * f32 generic kernels: |err| <= 2e-5 * scale (different summation order than torch's CPU matmul);
  * MFMA bf16-operand kernels with f32 output: |err| <= 1e-3 * rms(y) against an f64 product of the SAME
    bf16-rounded activations (what remains is f32 accumulation order and the +128 offset cancellation);
  * bf16 outputs: additionally one bf16 rounding, |err| <= 2^-8 |y|.
"""
import numpy as np
import pytest
import torch

from lit_llama_amd import _native as nat
from lit_llama_amd import ops, synth
from oracle import oracle

import layouts  # tests/layouts.py

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _rms(t):
    return float(t.double().pow(2).mean().sqrt())


# ---------------------------------------------------------------------------------------------- stream layouts
@pytest.mark.parametrize("N,K,R", [(64, 256, 1), (96, 384, 2), (40, 200, 1)])
def test_q4_repack_layout_is_the_documented_one(dev, N, K, R):
    gen = torch.Generator().manual_seed(N + K)
    q = torch.randint(0, 16, (N, K), generator=gen, dtype=torch.uint8)
    packed = synth.pack_colblock(q).to(dev)
    stream = ops.repack_q4(packed, None, N, K, R).cpu().numpy()
    assert np.array_equal(stream, layouts.q4_levels_to_stream(q.numpy(), None, R))  # bit-exact
    assert np.array_equal(layouts.q4_stream_to_levels(stream, N, K, R, False)[0], q.numpy())
    # row-major packed input (different strides) must give the same stream
    stream2 = ops.repack_q4(packed.contiguous(), None, N, K, R).cpu().numpy()
    assert np.array_equal(stream, stream2)


def test_q4_repack_pair_interleaves_two_matrices(dev):
    N, K = 48, 256
    gen = torch.Generator().manual_seed(5)
    q0 = torch.randint(0, 16, (N, K), generator=gen, dtype=torch.uint8)
    q1 = torch.randint(0, 16, (N, K), generator=gen, dtype=torch.uint8)
    s = ops.repack_q4(synth.pack_colblock(q0).to(dev), synth.pack_colblock(q1).to(dev), N, K, 2).cpu().numpy()
    assert np.array_equal(s, layouts.q4_levels_to_stream(q0.numpy(), q1.numpy(), 2))


def test_bf16_and_i8_repack_layouts(dev):
    gen = torch.Generator().manual_seed(6)
    w = torch.randn((40, 200), generator=gen).to(torch.bfloat16)
    for R in (1, 2):
        s = ops.repack_bf16(w.to(dev), None, R).cpu().numpy()
        assert np.array_equal(s, layouts.bf16_bits_to_stream(w.view(torch.int16).numpy().view(np.uint16), R))
    cb = torch.randint(-127, 128, (48, 256), generator=gen, dtype=torch.int8)
    for R in (1, 2):
        s = ops.repack_i8(cb.to(dev), None, R).cpu().numpy()
        assert np.array_equal(s, layouts.i8_to_stream(cb.numpy(), R))


def test_u8_repack_layout_of_the_8_bit_colblock_stream(dev):
    """mi355_u8_repack (weight_fmt 6 of mi355_fused_step) == tests/layouts.py u8_to_stream == lit_llama_amd.engine.u8_stream, from the
    reference's column-major quant_weight (lit_llama/quantization.py:350-359: `.t().contiguous().t()`) and from a row-major matrix, single and
    as the c_fc1 / c_fc2 pair; shapes the stream cannot hold are refused."""
    from lit_llama_amd import _native as nat
    from lit_llama_amd.engine import u8_stream

    gen = torch.Generator().manual_seed(9)
    q = torch.randint(0, 256, (2, 48, 384), generator=gen, dtype=torch.uint8)
    want1, want2 = layouts.u8_to_stream(q[0].numpy(), 1), layouts.u8_to_stream(q.numpy(), 2)
    for colmajor in (True, False):
        a, b = (t.to(dev).t().contiguous().t() if colmajor else t.to(dev).contiguous() for t in (q[0], q[1]))
        assert a.stride() == ((1, 48) if colmajor else (384, 1))
        assert np.array_equal(ops.repack_u8(a).cpu().numpy(), want1)
        assert np.array_equal(ops.repack_u8(a, b).cpu().numpy(), want2)
    assert np.array_equal(u8_stream([q[0].to(dev), q[1].to(dev)]).cpu().numpy(), want2)
    with pytest.raises(nat.NativeError):
        ops.repack_u8(torch.zeros((40, 384), dtype=torch.uint8, device=dev))
    with pytest.raises(nat.NativeError):
        ops.repack_u8(torch.zeros((48, 200), dtype=torch.uint8, device=dev))


# ---------------------------------------------------------------------------------------------- int4 fast linear
def _q4_problem(N, K, M, seed, dev, x_scale=1.0):
    gen = torch.Generator().manual_seed(seed)
    w = torch.randn((N, K), generator=gen) * K**-0.5
    q, scale, zero = synth.rtn_quantize_rows(w)
    scale = scale.to(torch.bfloat16).float()
    x = torch.randn((M, K), generator=gen) * x_scale
    wdq = (q.float() - zero[:, None]) * scale[:, None]
    return dict(q=q, packed=synth.pack_colblock(q).to(dev), scale=scale, zero=zero, x=x, wdq=wdq)


Q4_SHAPES = [
    # N, K, M, R, waves, grid, prefetch
    (64, 128, 1, 1, 0, 0, 0),       # one unit: 7 of 8 waves have no work (combine-only path)
    (96, 256, 3, 2, 4, 0, 0),
    (48, 384, 2, 2, 8, 0, 8),       # N not a multiple of 32: padded tile
    (40, 200, 1, 1, 0, 0, 0),       # K and N both padded
    (512, 1024, 16, 1, 8, 3, 4),    # 3 workgroups loop over 32 tiles (persistent path), M = 16
    (4096, 4096, 1, 1, 0, 0, 0),    # 7B attn.c_proj
    (4096, 4096, 1, 1, 4, 128, 8),
    (12288, 4096, 2, 2, 0, 0, 0),   # 7B c_attn
    (4096, 11008, 1, 1, 0, 0, 0),   # 7B mlp.c_proj: 86 units over 8 waves (uneven split)
    (4096, 11008, 5, 1, 8, 200, 8),
    (32000, 4096, 1, 2, 0, 0, 0),   # lm_head
    (8192, 2752, 1, 1, 0, 0, 0),    # 65B TP=8 mlp.c_proj shard: K = 21.5 units
    (256, 22016, 1, 1, 0, 0, 0),    # 65B mlp.c_proj width on one GPU: six 16-B vectors per thread
    (256, 13824, 2, 1, 0, 0, 0),    # 13B mlp.c_proj width
]


@pytest.mark.parametrize("N,K,M,R,waves,grid,prefetch", Q4_SHAPES)
def test_q4_linear_matches_oracle(dev, N, K, M, R, waves, grid, prefetch):
    p = _q4_problem(N, K, M, seed=N * 7 + K + M, dev=dev)
    xb = p["x"].to(torch.bfloat16)
    stream = ops.repack_q4(p["packed"], None, N, K, R)
    sc, ze = p["scale"].to(torch.bfloat16).to(dev), p["zero"].to(torch.bfloat16).to(dev)
    y = ops.linear_fast(xb.to(dev), stream, nat.W_Q4, R, N, K, scales=sc, zeros=ze, out_dtype=torch.float32,
                        waves=waves, grid=grid, prefetch=prefetch).cpu()
    ref64 = xb.double() @ p["wdq"].double().t()
    err = (y.double() - ref64).abs().max().item()
    assert err <= 1e-3 * _rms(ref64), f"max err {err:.3e} vs rms {_rms(ref64):.3e}"
    # and the oracle's own forward (reference CPU path: full dequant + F.linear) on the same inputs
    yo = oracle.colblock_linear(xb.float(), synth.pack_colblock(p["q"]).contiguous(), p["scale"][:, None],
                                p["zero"][:, None], 4, K)
    assert (y - yo).abs().max().item() <= 1e-3 * _rms(ref64)
    # bf16 output = one extra rounding
    yb = ops.linear_fast(xb.to(dev), stream, nat.W_Q4, R, N, K, scales=sc, zeros=ze, out_dtype=torch.bfloat16,
                         waves=waves, grid=grid, prefetch=prefetch).cpu().double()
    assert bool(((yb - ref64).abs() <= 2.0**-8 * ref64.abs() + 1e-3 * _rms(ref64)).all())


def test_q4_linear_f32_scales_and_determinism(dev):
    N, K, M = 1024, 2048, 4
    p = _q4_problem(N, K, M, seed=99, dev=dev)
    xb = p["x"].to(torch.bfloat16).to(dev)
    stream = ops.repack_q4(p["packed"], None, N, K, 1)
    kw = dict(scales=p["scale"].to(dev), zeros=p["zero"].to(dev), out_dtype=torch.float32)
    y0 = ops.linear_fast(xb, stream, nat.W_Q4, 1, N, K, **kw)
    y1 = ops.linear_fast(xb, stream, nat.W_Q4, 1, N, K, grid=7, **kw)   # other tile -> workgroup mapping
    y2 = ops.linear_fast(xb, stream, nat.W_Q4, 1, N, K, **kw)
    assert torch.equal(y0, y1) and torch.equal(y0, y2)  # bit-reproducible, independent of the launch geometry
    ref64 = xb.cpu().double() @ p["wdq"].double().t()
    assert (y0.cpu().double() - ref64).abs().max().item() <= 1e-3 * _rms(ref64)
    # an f32 activation is rounded to bf16 exactly once (a different staging path: only the order of the f32
    # row sum may differ)
    y3 = ops.linear_fast(xb.float(), stream, nat.W_Q4, 1, N, K, **kw)
    assert (y0 - y3).abs().max().item() <= 1e-4 * _rms(ref64)


@pytest.mark.parametrize("N,K,M", [(256, 512, 3), (128, 8192, 1), (64, 5120, 2), (64, 16384, 1)])
def test_q4_linear_fused_rmsnorm_accumulate_and_bias(dev, N, K, M):
    # K = 5120 / 8192: the four-vector staging mode of the 13B .. 65B widths; 16384: past it (element loop)
    p = _q4_problem(N, K, M, seed=3, dev=dev, x_scale=3.0)
    gen = torch.Generator().manual_seed(8)
    nscale = (1 + 0.1 * torch.randn(K, generator=gen)).to(torch.bfloat16)
    bias = torch.randn(N, generator=gen).to(torch.bfloat16)
    resid = torch.randn((M, N), generator=gen)
    stream = ops.repack_q4(p["packed"], None, N, K, 1)
    sc, ze = p["scale"].to(torch.bfloat16).to(dev), p["zero"].to(torch.bfloat16).to(dev)
    out = resid.clone().to(dev)
    ops.linear_fast(p["x"].to(dev), stream, nat.W_Q4, 1, N, K, scales=sc, zeros=ze, norm_scale=nscale.to(dev), eps=1e-5,
                    bias=bias.to(dev), epi=nat.EPI_ACCUM, out=out)
    # the kernel stages bf16(scale_k * x_k) and applies 1/rms in the epilogue (same map, the one bf16 rounding sits
    # before instead of after the scalar): exact reference in that order, oracle order at bf16-rounding tolerance
    xs = (nscale.float() * p["x"]).to(torch.bfloat16)
    rinv = torch.rsqrt((p["x"] * p["x"]).mean(-1, keepdim=True) + 1e-5).double()
    ref_k = (xs.double() @ p["wdq"].double().t()) * rinv + bias.double() + resid.double()
    err = (out.cpu().double() - ref_k).abs().max().item()
    assert err <= 1e-3 * _rms(ref_k), err
    xn = oracle.rmsnorm(p["x"], nscale.float(), 1e-5).to(torch.bfloat16)
    ref_o = xn.double() @ p["wdq"].double().t() + bias.double() + resid.double()
    assert (out.cpu().double() - ref_o).abs().max().item() <= 1e-2 * _rms(ref_o)


def test_q4_linear_swiglu_pair(dev):
    N, K, M = 11008, 4096, 2   # 7B c_fc1 / c_fc2
    a = _q4_problem(N, K, M, seed=41, dev=dev)
    b = _q4_problem(N, K, M, seed=42, dev=dev)
    x = a["x"].to(torch.bfloat16)
    stream = ops.repack_q4(a["packed"], b["packed"], N, K, 2)
    bf = lambda t: t.to(torch.bfloat16).to(dev)  # noqa: E731
    y = ops.linear_fast(x.to(dev), stream, nat.W_Q4, 2, N, K, scales=bf(a["scale"]), zeros=bf(a["zero"]),
                        scales2=bf(b["scale"]), zeros2=bf(b["zero"]), epi=nat.EPI_SWIGLU, out_dtype=torch.float32).cpu()
    h1 = x.double() @ a["wdq"].double().t()
    h2 = x.double() @ b["wdq"].double().t()
    ref = torch.nn.functional.silu(h1) * h2
    assert (y.double() - ref).abs().max().item() <= 2e-3 * _rms(ref)


# ---- grouped scales ("groupsize": one (scale, zero) pair per output row and group of columns, quantization.py:284-333
# with tile_cols > 0) in the MFMA streaming kernel
def _q4_grouped_problem(N, K, M, g, seed, dev):
    gen = torch.Generator().manual_seed(seed)
    w = torch.randn((N, K), generator=gen) * K**-0.5 * (1 + torch.arange(K) % 7)[None, :]  # group-dependent ranges
    ng = -(-K // g)
    q = torch.empty((N, K), dtype=torch.uint8)
    scale, zero = torch.empty((N, ng)), torch.empty((N, ng))
    for j in range(ng):
        qj, sj, zj = synth.rtn_quantize_rows(w[:, j * g:(j + 1) * g])
        q[:, j * g:(j + 1) * g], scale[:, j], zero[:, j] = qj, sj, zj
    scale = scale.to(torch.bfloat16).float()
    x = torch.randn((M, K), generator=gen)
    wdq = (q.float() - zero.repeat_interleave(g, 1)[:, :K]) * scale.repeat_interleave(g, 1)[:, :K]
    return dict(q=q, packed=synth.pack_colblock(q).to(dev), scale=scale, zero=zero, x=x, wdq=wdq)


def test_q4_grouped_fast_kernel_matches_reference_golden(dev, golden):
    """The reference module's own output for a 64-column-group layer (tests/golden/colblock.npz, generated from
    /root/reference lit_llama/quantization.py by oracle/gen_golden.py).  The fast kernel rounds x and the scale table to
    bf16: checked tightly against the same arithmetic on the rounded operands, loosely against the f32 golden."""
    g = golden("colblock")
    N, K, bits, tc = (int(v) for v in g["b4_g64_meta"])
    assert (bits, tc) == (4, 64)
    q = _t(g["b4_g64_q"]).t().contiguous().t().to(dev)
    scales, zeros, x = _t(g["b4_g64_scales"]), _t(g["b4_g64_zeros"]), _t(g["b4_g64_x"])
    sb, zb, xb = scales.to(torch.bfloat16), zeros.to(torch.bfloat16), x.to(torch.bfloat16)
    for R in (1, 2):
        stream = ops.repack_q4(q, None, N, K, R)
        y = ops.linear_fast(x.to(dev), stream, nat.W_Q4, R, N, K, scales=sb.to(dev).reshape(-1).contiguous(),
                            zeros=zb.to(dev).reshape(-1).contiguous(), out_dtype=torch.float32, group_cols=tc).cpu()
        wdq = _t(g["b4_g64_wdq"])  # (q - z) * s with the f32 tables
        ref = _t(g["b4_g64_y"])
        assert (y - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
        # exact arithmetic on the rounded operands: levels recovered from the golden dequantised weight
        lev = torch.round(wdq / scales.repeat_interleave(tc, 1) + zeros.repeat_interleave(tc, 1))
        w2 = (lev - zb.float().repeat_interleave(tc, 1)) * sb.float().repeat_interleave(tc, 1)
        ref2 = xb.double() @ w2.double().t()
        assert (y.double() - ref2).abs().max().item() <= 1e-3 * _rms(ref2)


GROUPED_SHAPES = [
    # N, K, M, R, group, waves
    (4096, 4096, 1, 1, 128, 0),     # 7B attn.c_proj with groupsize 128
    (4096, 4096, 7, 1, 128, 0),
    (12288, 4096, 1, 2, 128, 0),    # 7B c_attn, two row groups per tile
    (4096, 11008, 1, 1, 128, 0),    # 7B mlp.c_proj: 86 groups, uneven unit split (groups end inside a wave's range)
    (4096, 11008, 3, 1, 256, 8),    # a group spans two units; 43 groups
    (512, 4096, 2, 1, 32, 8),       # a group per MFMA k-block: 128 groups
    (96, 1024, 1, 2, 64, 4),        # two groups per unit
    (40, 200, 1, 1, 64, 0),         # K and N padded: the last group is short
    (64, 512, 16, 1, 512, 0),       # group size == K: one group (per-row path)
]


@pytest.mark.parametrize("N,K,M,R,g,waves", GROUPED_SHAPES)
def test_q4_grouped_linear_matches_oracle(dev, N, K, M, R, g, waves):
    p = _q4_grouped_problem(N, K, M, g, seed=N + K + M + g, dev=dev)
    xb = p["x"].to(torch.bfloat16)
    stream = ops.repack_q4(p["packed"], None, N, K, R)
    sc = p["scale"].to(torch.bfloat16).to(dev).reshape(-1).contiguous()
    ze = p["zero"].to(torch.bfloat16).to(dev).reshape(-1).contiguous()
    y = ops.linear_fast(xb.to(dev), stream, nat.W_Q4, R, N, K, scales=sc, zeros=ze, out_dtype=torch.float32,
                        waves=waves, group_cols=g).cpu()
    ref64 = xb.double() @ p["wdq"].double().t()
    err = (y.double() - ref64).abs().max().item()
    assert err <= 1e-3 * _rms(ref64), f"max err {err:.3e} vs rms {_rms(ref64):.3e}"
    yo = oracle.colblock_linear(xb.float(), synth.pack_colblock(p["q"]).contiguous(), p["scale"], p["zero"], 4, g)
    assert (y - yo).abs().max().item() <= 1e-3 * _rms(ref64)
    # the generic (scalar) kernel on the reference layout agrees: two independent HIP implementations
    yg = ops.linear_colblock(xb.to(dev).float(), p["packed"], p["scale"].to(dev), p["zero"].to(dev), 4, g, None, K).cpu()
    assert (y - yg).abs().max().item() <= 1e-3 * _rms(ref64)
    y2 = ops.linear_fast(xb.to(dev), stream, nat.W_Q4, R, N, K, scales=sc, zeros=ze, out_dtype=torch.float32,
                         waves=waves, grid=5, group_cols=g).cpu()
    assert torch.equal(y, y2)  # bit-reproducible, independent of the tile -> workgroup mapping


def test_q4_grouped_swiglu_pair_norm_and_accumulate(dev):
    N, K, M, g = 11008, 4096, 2, 128   # 7B c_fc1 / c_fc2 with groupsize 128, fused RMSNorm prologue
    a = _q4_grouped_problem(N, K, M, g, seed=51, dev=dev)
    b = _q4_grouped_problem(N, K, M, g, seed=52, dev=dev)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn((M, K), generator=gen) * 3
    nscale = (1 + 0.1 * torch.randn(K, generator=gen)).to(torch.bfloat16)
    # the kernel stages bf16(x * norm_scale) and applies 1 / rms to the finished dot products
    xnb = (x * nscale.float()).to(torch.bfloat16)
    rinv = torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-5).double()
    stream = ops.repack_q4(a["packed"], b["packed"], N, K, 2)
    bf = lambda t: t.to(torch.bfloat16).to(dev).reshape(-1).contiguous()  # noqa: E731
    y = ops.linear_fast(x.to(dev), stream, nat.W_Q4, 2, N, K, scales=bf(a["scale"]), zeros=bf(a["zero"]),
                        scales2=bf(b["scale"]), zeros2=bf(b["zero"]), norm_scale=nscale.to(dev), eps=1e-5,
                        epi=nat.EPI_SWIGLU, out_dtype=torch.float32, group_cols=g).cpu()
    h1 = (xnb.double() @ a["wdq"].double().t()) * rinv
    h2 = (xnb.double() @ b["wdq"].double().t()) * rinv
    ref = torch.nn.functional.silu(h1) * h2
    assert (y.double() - ref).abs().max().item() <= 2e-3 * _rms(ref)
    # accumulate epilogue (residual add) on a c_proj-shaped grouped layer
    c = _q4_grouped_problem(512, 4096, 3, g, seed=53, dev=dev)
    xb = c["x"].to(torch.bfloat16)
    out = torch.randn((3, 512), generator=gen)
    acc = out.clone().to(dev)
    ops.linear_fast(xb.to(dev), ops.repack_q4(c["packed"], None, 512, 4096, 1), nat.W_Q4, 1, 512, 4096,
                    scales=bf(c["scale"]), zeros=bf(c["zero"]), epi=nat.EPI_ACCUM, out=acc, group_cols=g)
    ref2 = out.double() + xb.double() @ c["wdq"].double().t()
    assert (acc.cpu().double() - ref2).abs().max().item() <= 1e-3 * _rms(ref2)


def test_q4_fast_kernel_agrees_with_generic_kernel_at_full_size(dev):
    """Two independent HIP implementations (MFMA stream kernel on the repacked layout vs the scalar kernel on
    the reference layout) on a 7B-sized matrix, plus linearity — size-independent checks at BASELINE sizes."""
    N, K = 11008, 4096
    gen = torch.Generator(device=dev).manual_seed(1)
    packed = torch.randint(0, 256, (K // 2, N), generator=gen, device=dev, dtype=torch.uint8).t()
    scales = (0.005 + 0.005 * torch.rand((N, 1), generator=gen, device=dev)).to(torch.bfloat16)
    zeros = torch.randint(0, 16, (N, 1), generator=gen, device=dev).to(torch.bfloat16)
    x = torch.randn((2, K), generator=gen, device=dev).to(torch.bfloat16)
    stream = ops.repack_q4(packed, None, N, K, 2)
    fast = ops.linear_fast(x, stream, nat.W_Q4, 2, N, K, scales=scales.reshape(-1), zeros=zeros.reshape(-1),
                           out_dtype=torch.float32)
    slow = ops.linear_colblock(x.float(), packed, scales.float(), zeros.float(), 4, K, None, K)
    assert (fast - slow).abs().max().item() <= 1e-3 * _rms(slow)
    # linearity in x (exact inputs: powers of two keep bf16 products exact)
    y2 = ops.linear_fast((x.float() * 2).to(torch.bfloat16), stream, nat.W_Q4, 2, N, K, scales=scales.reshape(-1),
                         zeros=zeros.reshape(-1), out_dtype=torch.float32)
    assert (y2 - 2 * fast).abs().max().item() <= 1e-3 * _rms(slow)


# ---------------------------------------------------------------------------------------------- bf16 fast linear
@pytest.mark.parametrize("N,K,M,R,grid", [(64, 128, 1, 1, 0), (96, 384, 4, 2, 0), (4096, 4096, 1, 1, 0),
                                          (4096, 11008, 2, 1, 100), (32000, 4096, 1, 2, 0), (40, 200, 2, 1, 0)])
def test_bf16_linear_matches_oracle(dev, N, K, M, R, grid):
    gen = torch.Generator().manual_seed(N + K + M)
    w = (torch.randn((N, K), generator=gen) * K**-0.5).to(torch.bfloat16)
    x = torch.randn((M, K), generator=gen).to(torch.bfloat16)
    stream = ops.repack_bf16(w.to(dev), None, R)
    y = ops.linear_fast(x.to(dev), stream, nat.W_BF16, R, N, K, out_dtype=torch.float32, grid=grid).cpu()
    ref64 = x.double() @ w.double().t()
    assert (y.double() - ref64).abs().max().item() <= 2e-5 * max(1.0, _rms(ref64)) * 10  # plain f32 accumulation
    # repack from f32 weights rounds them once to bf16
    stream32 = ops.repack_bf16(w.float().to(dev), None, R)
    assert torch.equal(stream32, stream)


# ---------------------------------------------------------------------------------------------- generic kernels
def test_colblock_generic_kernels_match_reference_golden(dev, golden):
    g = golden("colblock")
    for tag in ("b4_row", "b4_g64", "b8_row"):
        N, K, bits, tc = (int(v) for v in g[f"{tag}_meta"])
        q = _t(g[f"{tag}_q"]).t().contiguous().t().to(dev)  # reference (column-major) storage
        scales, zeros = _t(g[f"{tag}_scales"]).to(dev), _t(g[f"{tag}_zeros"]).to(dev)
        wdq = ops.colblock_dequant(q, scales, zeros, bits, tc, K, torch.float32).cpu()
        assert torch.equal(wdq, _t(g[f"{tag}_wdq"]))  # (q - z) * s in f32: bit-exact
        y = ops.linear_colblock(_t(g[f"{tag}_x"]).to(dev), q, scales, zeros, bits, tc, None, K).cpu()
        ref = _t(g[f"{tag}_y"])
        assert (y - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


def test_qlinear_4bit_weight_entry_matches_reference_golden(dev, golden):
    """`lit_llama_amd.quantization.qlinear_4bit_weight` — the module-level name of the reference's Triton wrapper
    (lit_llama/quantization.py:284-333) — on the reference module's own data and output (tests/golden/colblock.npz, b4_row)."""
    from lit_llama_amd.quantization import qlinear_4bit_weight

    g = golden("colblock")
    N, K, bits, tc = (int(v) for v in g["b4_row_meta"])
    assert bits == 4
    q = _t(g["b4_row_q"]).t().contiguous().t().to(dev)
    scales, zeros, x = _t(g["b4_row_scales"]).to(dev), _t(g["b4_row_zeros"]).to(dev), _t(g["b4_row_x"]).to(dev)
    assert scales.shape == (N, 1)
    y = qlinear_4bit_weight(x.view(1, *x.shape), q, scales, zeros).cpu()
    ref = _t(g["b4_row_y"])
    assert y.shape == (1, *ref.shape)
    assert (y[0] - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


def test_dense_rmsnorm_rope_swiglu_add_embedding_argmax(dev, golden):
    g = golden("blocks")
    # RMSNorm / RoPE against the reference's own outputs
    y = ops.rmsnorm(_t(g["rms_x"]).to(dev), _t(g["rms_scale"]).to(dev), 1e-6).cpu()
    assert (y - _t(g["rms_y"])).abs().max().item() <= 2e-6
    yr = ops.apply_rope(_t(g["rope_x"]).to(dev), _t(g["rope_cache"]).to(dev)).cpu()
    assert (yr - _t(g["rope_y"])).abs().max().item() <= 1e-6
    gen = torch.Generator().manual_seed(2)
    x, w, b = torch.randn((5, 200), generator=gen), torch.randn((33, 200), generator=gen), torch.randn(33, generator=gen)
    yd = ops.linear_dense(x.to(dev), w.to(dev), b.to(dev)).cpu()
    ref = torch.nn.functional.linear(x, w, b)
    assert (yd - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    a, c = torch.randn((3, 77), generator=gen), torch.randn((3, 77), generator=gen)
    assert (ops.swiglu(a.to(dev), c.to(dev)).cpu() - torch.nn.functional.silu(a) * c).abs().max().item() <= 1e-6
    assert torch.equal(ops.add(a.to(dev), c.to(dev)).cpu(), a + c)
    ab, cb = a.to(torch.bfloat16), c.to(torch.bfloat16)
    assert torch.equal(ops.add(ab.to(dev), cb.to(dev)).cpu(), ab + cb)
    wte = torch.randn((50, 16), generator=gen)
    idx = torch.tensor([[3, 49, 0], [7, 7, 12]])
    assert torch.equal(ops.embedding(idx.to(dev), wte.to(dev)).cpu(), wte[idx])
    assert torch.equal(ops.embedding(idx.int().to(dev), wte.to(dev)).cpu(), wte[idx])
    logits = torch.randn(32000, generator=gen)
    logits[[123, 31999]] = logits.max() + 1.0  # tie: lowest index wins
    assert int(ops.argmax(logits.to(dev)).item()) == 123


# ---------------------------------------------------------------------------------------------- attention
def _oracle_attention_steps(qkv_steps, n_head, rope, S, prompt):
    """Run the oracle's CausalSelfAttention core (rope, cache, SDPA) over a prompt + single-token steps."""
    C = qkv_steps.shape[-1] // 3
    hs = C // n_head
    cache = (torch.zeros(1, n_head, S, hs), torch.zeros(1, n_head, S, hs))
    mask_cache = torch.tril(torch.ones(rope.shape[0], rope.shape[0], dtype=torch.bool))[None, None]
    outs, pos = [], 0
    chunks = [qkv_steps[:prompt]] + [qkv_steps[i:i + 1] for i in range(prompt, qkv_steps.shape[0])]
    for ch in chunks:
        T = ch.shape[0]
        input_pos = torch.arange(pos, pos + T)
        q, k, v = ch.view(1, T, 3 * C).split(C, dim=2)
        q = oracle.apply_rope(q.view(1, T, n_head, hs), rope.index_select(0, input_pos)).transpose(1, 2)
        k = oracle.apply_rope(k.view(1, T, n_head, hs), rope.index_select(0, input_pos)).transpose(1, 2)
        v = v.view(1, T, n_head, hs).transpose(1, 2)
        ck, cv = cache
        ip = input_pos
        if input_pos[-1] >= S:
            ip = torch.tensor(S - 1)
            ck, cv = torch.roll(ck, -1, dims=2), torch.roll(cv, -1, dims=2)
        ck, cv = ck.index_copy(2, ip, k), cv.index_copy(2, ip, v)
        cache = (ck, cv)
        mask = mask_cache.index_select(2, input_pos)[:, :, :, :S]
        y = torch.nn.functional.scaled_dot_product_attention(q, ck, cv, attn_mask=mask)
        outs.append(y.transpose(1, 2).reshape(T, C))
        pos += T
    return torch.cat(outs), cache


@pytest.mark.parametrize("n_head,hs,cache_dtype,tol", [(4, 64, torch.float32, 2e-5), (2, 128, torch.float32, 2e-5),
                                                       (4, 2, torch.float32, 2e-5), (2, 128, torch.bfloat16, 2e-2)])
def test_attention_with_cache_prefill_decode_and_roll(dev, n_head, hs, cache_dtype, tol):
    S, prompt, steps = 12, 5, 11   # positions 0..15: the last 4 steps run in the cache-roll regime
    C = n_head * hs
    gen = torch.Generator().manual_seed(hs)
    qkv = torch.randn((prompt + steps, 3 * C), generator=gen)
    rope = oracle.build_rope_cache(64, hs, dtype=torch.int64)
    ref, (rk, rv) = _oracle_attention_steps(qkv, n_head, rope, S, prompt)
    k = torch.zeros((1, n_head, S, hs), dtype=cache_dtype, device=dev)
    v = torch.zeros_like(k)
    rope_d = rope.to(dev)
    outs, pos = [], 0
    for T in [prompt] + [1] * steps:
        ip = torch.arange(pos, pos + T, device=dev)
        if pos + T - 1 >= S:
            ops.kv_roll(k, v)
        y = ops.attention(qkv[pos:pos + T].view(1, T, 3 * C).to(dev), rope_d, n_head, pos=ip, kv_cache=(k, v))
        outs.append(y.view(T, C).float().cpu())
        pos += T
    got = torch.cat(outs)
    assert (got - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    assert (k.float().cpu() - rk).abs().max().item() <= tol * max(1.0, rk.abs().max().item())
    assert (v.float().cpu() - rv).abs().max().item() <= tol * max(1.0, rv.abs().max().item())


@pytest.mark.parametrize("n_split", [2, 4, 8])
def test_split_attention_equals_single_workgroup_and_feeds_the_projection(dev, n_split):
    """Flash-decoding: n_split workgroups per head + combine == one workgroup per head; and the c_proj linear that
    combines the partial records in its prologue == the same linear on the combined attention output."""
    n_head, hs, S = 32, 128, 512
    C = n_head * hs
    gen = torch.Generator(device=dev).manual_seed(n_split)
    rope = oracle.build_rope_cache(2048, hs, dtype=torch.int64).to(dev)
    for pos in (0, 1, 5, 130, 511):
        k = torch.randn((1, n_head, S, hs), generator=gen, device=dev).to(torch.bfloat16)
        v = torch.randn((1, n_head, S, hs), generator=gen, device=dev).to(torch.bfloat16)
        qkv = torch.randn((1, 1, 3 * C), generator=gen, device=dev)
        p_t = torch.tensor([pos], device=dev)
        k1, v1, k2, v2 = k.clone(), v.clone(), k.clone(), v.clone()
        y1 = ops.attention(qkv, rope, n_head, pos=p_t, kv_cache=(k1, v1), out_dtype=torch.float32)
        y2 = ops.attention(qkv, rope, n_head, pos=p_t, kv_cache=(k2, v2), out_dtype=torch.float32, n_split=n_split)
        assert torch.equal(k1, k2) and torch.equal(v1, v2)
        assert (y1 - y2).abs().max().item() <= 2e-5 * max(1.0, y1.abs().max().item())
    if n_split > 4:
        return  # the fused c_proj prologue takes at most 4 splits
    # projection fed by the partial records
    parts = ops.attention(qkv, rope, n_head, pos=p_t, kv_cache=(k2, v2), n_split=n_split, return_partials=True)
    # ... the LLM.int8 linear: same rows (bf16-rounded combine), so the same int8 levels up to that rounding
    gen_c = torch.Generator().manual_seed(17)
    cb, scb = ops.int8_quant_rows((torch.randn((256, C), generator=gen_c) * C**-0.5).to(dev))
    s8 = ops.repack_i8(cb, None, 1)
    ref8 = ops.linear_int8(y2.to(torch.bfloat16).view(1, C), s8, scb, 1, 256, C, out_dtype=torch.float32)
    got8 = ops.linear_int8(y2.to(torch.bfloat16).view(1, C), s8, scb, 1, 256, C, out_dtype=torch.float32,
                           attn_partials=parts)
    assert (got8 - ref8).abs().max().item() <= 2e-2 * _rms(ref8)
    p = _q4_problem(4096, C, 1, seed=5, dev=dev)
    stream = ops.repack_q4(p["packed"], None, 4096, C, 1)
    sc, ze = p["scale"].to(torch.bfloat16).to(dev), p["zero"].to(torch.bfloat16).to(dev)
    y_bf = y2.to(torch.bfloat16).view(1, C)
    ref = ops.linear_fast(y_bf, stream, nat.W_Q4, 1, 4096, C, scales=sc, zeros=ze, out_dtype=torch.float32)
    got = ops.linear_fast(y_bf, stream, nat.W_Q4, 1, 4096, C, scales=sc, zeros=ze, out_dtype=torch.float32,
                          attn_partials=parts)
    # same bf16 activations up to one rounding boundary of the combine
    assert (got - ref).abs().max().item() <= 5e-3 * _rms(ref)


def test_attention_without_cache_batched(dev):
    B, T, n_head, hs = 3, 9, 4, 8
    C = n_head * hs
    gen = torch.Generator().manual_seed(12)
    qkv = torch.randn((B, T, 3 * C), generator=gen)
    rope = oracle.build_rope_cache(T, hs, dtype=torch.int64)
    q, k, v = qkv.split(C, dim=2)
    q = oracle.apply_rope(q.view(B, T, n_head, hs), rope).transpose(1, 2)
    k = oracle.apply_rope(k.view(B, T, n_head, hs), rope).transpose(1, 2)
    v = v.view(B, T, n_head, hs).transpose(1, 2)
    mask = torch.tril(torch.ones(T, T, dtype=torch.bool))[None, None]
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=mask).transpose(1, 2).reshape(B, T, C)
    got = ops.attention(qkv.to(dev), rope.to(dev), n_head, rope_gathered=True).cpu()
    assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


def test_attention_long_context_decode_bf16(dev):
    """7B head geometry at a long position: 32 heads x 128, cache bf16, position 1500."""
    n_head, hs, S, pos = 32, 128, 2048, 1500
    C = n_head * hs
    gen = torch.Generator(device=dev).manual_seed(4)
    k = (torch.randn((1, n_head, S, hs), generator=gen, device=dev)).to(torch.bfloat16)
    v = (torch.randn((1, n_head, S, hs), generator=gen, device=dev)).to(torch.bfloat16)
    qkv = torch.randn((1, 1, 3 * C), generator=gen, device=dev)
    rope = oracle.build_rope_cache(2048, hs, dtype=torch.int64)
    y = ops.attention(qkv, rope.to(dev), n_head, pos=torch.tensor([pos], device=dev), kv_cache=(k, v),
                      out_dtype=torch.float32).cpu()
    kc, vc = k.float().cpu(), v.float().cpu()  # includes the row the kernel wrote at `pos`
    q = oracle.apply_rope(qkv.cpu()[..., :C].view(1, 1, n_head, hs), rope[pos:pos + 1]).transpose(1, 2)
    att = torch.softmax((q @ kc[:, :, :pos + 1].transpose(-1, -2)) / hs**0.5, dim=-1) @ vc[:, :, :pos + 1]
    ref = att.transpose(1, 2).reshape(1, 1, C)
    assert (y - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())
    knew = oracle.apply_rope(qkv.cpu()[..., C:2 * C].view(1, 1, n_head, hs), rope[pos:pos + 1])
    assert torch.equal(kc[0, :, pos], knew[0, 0].to(torch.bfloat16).float())


# ---------------------------------------------------------------------------------------------- LLM.int8
def test_int8_weight_quantisation_is_bit_exact(dev):
    gen = torch.Generator().manual_seed(21)
    w = torch.randn((96, 512), generator=gen) * 0.05
    cb, scb = ops.int8_quant_rows(w.to(dev))
    ocb, oscb = oracle.int8_quant_rows(w)
    assert torch.equal(scb.cpu(), oscb)
    assert torch.equal(cb.cpu(), ocb)
    cb16, scb16 = ops.int8_quant_rows(w.to(torch.bfloat16).to(dev))
    ocb16, oscb16 = oracle.int8_quant_rows(w.to(torch.bfloat16))
    assert torch.equal(cb16.cpu(), ocb16) and torch.equal(scb16.cpu(), oscb16)


@pytest.mark.parametrize("N,K,M,R,outliers", [(64, 128, 1, 1, 0), (96, 512, 3, 2, 2), (4096, 4096, 1, 1, 5),
                                              (4096, 11008, 2, 1, 7), (40, 200, 2, 1, 1),
                                              (256, 4096, 1, 1, 12), (128, 11008, 1, 2, 3)])
def test_int8_linear_matches_oracle(dev, N, K, M, R, outliers):
    gen = torch.Generator().manual_seed(N + K)
    w = torch.randn((N, K), generator=gen) * K**-0.5
    x = torch.randn((M, K), generator=gen)
    for i in range(outliers):
        x[i % M, (37 * i + 5) % K] = 6.0 + i  # |x| >= 6: outlier columns
    x = x.to(torch.bfloat16)
    ocb, oscb = oracle.int8_quant_rows(w)
    ref = oracle.llm_int8_linear(x, ocb, oscb).float()
    cb, scb = ops.int8_quant_rows(w.to(dev))
    stream = ops.repack_i8(cb, None, R)
    y = ops.linear_int8(x.to(dev), stream, scb, R, N, K, out_dtype=torch.bfloat16).float().cpu()
    # integer accumulation and every f16 rounding are restated exactly; the f16 outlier side product is summed
    # in a different order than torch's matmul -> allow one f16 ulp on those rows
    close = (y - ref).abs() <= 2.0**-9 * ref.abs() + 1e-6
    assert bool(close.all()), f"{int((~close).sum())} of {close.numel()} outputs differ"
    if outliers == 0:
        assert torch.equal(y, ref)
    # sanity: the quantised product tracks the fp product
    fp = x.float() @ w.t()
    assert (y - fp).abs().max().item() <= 5e-2 * fp.abs().max().item()


# ---------------------------------------------------------------------------------------------- wide int4 GEMM (prefill)
@pytest.mark.parametrize("M,N,K,epi", [(32, 64, 128, "store"), (128, 4096, 4096, "store"), (200, 4096, 4096, "accum"),
                                        (257, 11008, 4096, "swiglu"), (96, 4096, 11008, "accum"), (40, 72, 200, "store"),
                                        (512, 12288, 4096, "store"),
                                        # short prompts against N = 4096 / 12288: few blocks -> deterministic split-K (8 / 4 / 2 slices)
                                        (128, 4096, 4096, "accum"), (128, 4096, 11008, "store"), (200, 4096, 4096, "store"),
                                        (128, 12288, 4096, "store"), (128, 11008, 4096, "swiglu")])
def test_linear_gemm_matches_the_skinny_kernel_and_oracle(dev, M, N, K, epi):
    """mi355_linear_gemm (LDS-tiled MFMA GEMM over the Q4 stream) against (a) the CPU oracle's dequantise-then-F.linear
    (lit_llama/quantization.py:422-423) within the bf16-operand tolerance and (b) the skinny weight-streaming kernel,
    which computes the same bf16 products with f32 accumulation in another order."""
    gen = torch.Generator().manual_seed(M * 7 + N + K)
    def qw():
        q = torch.randint(0, 16, (N, K), generator=gen, dtype=torch.uint8)
        scales = (torch.rand((N, 1), generator=gen) * 0.02 + 0.005).to(torch.bfloat16)
        zeros = torch.randint(5, 11, (N, 1), generator=gen).to(torch.bfloat16)
        return q, scales, zeros
    q0, s0, z0 = qw()
    x = (torch.randn((M, K), generator=gen)).to(torch.bfloat16)
    pk = lambda q: synth.pack_colblock(q, 4).to(dev)  # noqa: E731
    w0 = (q0.float() - z0.float()) * s0.float()
    xd = x.to(dev)
    if epi == "swiglu":
        q1, s1, z1 = qw()
        w1 = (q1.float() - z1.float()) * s1.float()
        stream = ops.repack_q4(pk(q0), pk(q1), N, K, 2)
        kw = dict(scales=s0.reshape(-1).to(dev), zeros=z0.reshape(-1).to(dev), scales2=s1.reshape(-1).to(dev),
                  zeros2=z1.reshape(-1).to(dev), epi=nat.EPI_SWIGLU, out_dtype=torch.bfloat16)
        got = ops.linear_gemm(xd, stream, 2, N, K, **kw).float().cpu()
        ref = torch.nn.functional.silu(x.float() @ w0.t()) * (x.float() @ w1.t())
        skinny = ops.linear_fast(xd, stream, nat.W_Q4, 2, N, K, **kw).float().cpu()
    else:
        stream = ops.repack_q4(pk(q0), None, N, K, 1)
        kw = dict(scales=s0.reshape(-1).to(dev), zeros=z0.reshape(-1).to(dev))
        base = torch.randn((M, N), generator=gen)
        if epi == "accum":
            o1, o2 = base.clone().to(dev), base.clone().to(dev)
            got = ops.linear_gemm(xd, stream, 1, N, K, epi=nat.EPI_ACCUM, out=o1, **kw).float().cpu()
            skinny = ops.linear_fast(xd, stream, nat.W_Q4, 1, N, K, epi=nat.EPI_ACCUM, out=o2, **kw).float().cpu()
            ref = base + x.float() @ w0.t()
        else:
            got = ops.linear_gemm(xd, stream, 1, N, K, out_dtype=torch.float32, **kw).float().cpu()
            skinny = ops.linear_fast(xd, stream, nat.W_Q4, 1, N, K, out_dtype=torch.float32, **kw).float().cpu()
            ref = x.float() @ w0.t()
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 2e-2 * scale + 1e-3, f"vs oracle: {(got - ref).abs().max().item():.4e} (scale {scale:.3f})"
    assert (got - skinny).abs().max().item() <= (8e-3 if epi == "swiglu" else 1e-4) * scale + 1e-5, \
        f"vs skinny kernel: {(got - skinny).abs().max().item():.4e} (scale {scale:.3f})"


def test_linear_gemm_with_fused_rmsnorm_matches_the_skinny_kernel(dev):
    gen = torch.Generator().manual_seed(5)
    M, N, K = 160, 512, 4096
    q = torch.randint(0, 16, (N, K), generator=gen, dtype=torch.uint8)
    scales = (torch.rand((N,), generator=gen) * 0.02 + 0.005).to(torch.bfloat16).to(dev)
    zeros = torch.randint(5, 11, (N,), generator=gen).to(torch.bfloat16).to(dev)
    x = (torch.randn((M, K), generator=gen) * 3).to(dev)  # f32 residual stream
    g = (1 + 0.1 * torch.randn((K,), generator=gen)).to(torch.bfloat16).to(dev)
    stream = ops.repack_q4(synth.pack_colblock(q, 4).to(dev), None, N, K, 1)
    a = ops.linear_gemm(x, stream, 1, N, K, scales=scales, zeros=zeros, norm_scale=g, eps=1e-5, out_dtype=torch.float32)
    b = ops.linear_fast(x, stream, nat.W_Q4, 1, N, K, scales=scales, zeros=zeros, norm_scale=g, eps=1e-5,
                        out_dtype=torch.float32)
    assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item()


GROUPED_GEMM_SHAPES = [
    # M, N, K, group, epi
    (512, 4096, 4096, 128, "store"),     # 8-tile blocks of 4 waves; one group per unit
    (128, 4096, 4096, 128, "accum"),     # short prompt: 4 K-slices of 8 groups
    (40, 96, 1024, 64, "store"),         # 1-wave blocks; two groups per unit (masked passes)
    (200, 512, 4096, 32, "accum"),       # a group per MFMA k-block
    (96, 4096, 11008, 256, "store"),     # a group spans two units, 43 groups
    (257, 4096, 11008, 512, "store"),    # last group short (21.5 groups of 4 units)
    (257, 11008, 4096, 128, "swiglu"),   # c_fc1 / c_fc2 pair with two tables, fused RMSNorm
    (130, 1000, 512, 128, "store"),      # N not a multiple of 16
    (128, 11008, 4096, 128, "swiglu"),   # short prompt: K-slices of whole groups (deterministic split-K) on the pair stream
    (64, 12288, 4096, 64, "store"),      # split-K over units with sub-unit groups
    (2048, 4096, 4096, 128, "accum"),    # long prompt: blocks of 128 tokens x 8 waves (kGrpWideM of csrc/gemm.hip)
    (2049, 4096, 11008, 256, "accum"),   # the same tiling, last block of one token, a group over two units
    (2050, 2064, 4096, 128, "swiglu"),   # the same tiling on the pair stream, last tile column partial
]


@pytest.mark.parametrize("M,N,K,g,epi", GROUPED_GEMM_SHAPES)
def test_linear_gemm_with_grouped_scales_matches_oracle(dev, M, N, K, g, epi):
    """Wide GEMM over the Q4 stream with one (scale, zero) pair per row and group of `g` input columns (GPTQ groupsize,
    /root/reference lit_llama/quantization.py:284-333, :404-410 with tile_cols > 0): against exact arithmetic on the
    bf16-rounded operands, the CPU oracle's colblock_linear, and the skinny streaming kernel."""
    a = _q4_grouped_problem(N, K, M, g, seed=M + N + K + g, dev=dev)
    bf = lambda t: t.to(torch.bfloat16).to(dev).reshape(-1).contiguous()  # noqa: E731
    gen = torch.Generator().manual_seed(g + M)
    if epi == "swiglu":
        b = _q4_grouped_problem(N, K, M, g, seed=M + N + K + g + 1, dev=dev)
        x = torch.randn((M, K), generator=gen) * 3
        nscale = (1 + 0.1 * torch.randn(K, generator=gen)).to(torch.bfloat16)
        xnb = (x * nscale.float()).to(torch.bfloat16)
        rinv = torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-5).double()
        stream = ops.repack_q4(a["packed"], b["packed"], N, K, 2)
        kw = dict(scales=bf(a["scale"]), zeros=bf(a["zero"]), scales2=bf(b["scale"]), zeros2=bf(b["zero"]),
                  norm_scale=nscale.to(dev), eps=1e-5, epi=nat.EPI_SWIGLU, out_dtype=torch.bfloat16, group_cols=g)
        y = ops.linear_gemm(x.to(dev), stream, 2, N, K, **kw).float().cpu()
        h1 = (xnb.double() @ a["wdq"].double().t()) * rinv
        h2 = (xnb.double() @ b["wdq"].double().t()) * rinv
        ref = torch.nn.functional.silu(h1) * h2
        assert (y.double() - ref).abs().max().item() <= 8e-3 * ref.abs().max().item()  # bf16 output
        ys = torch.cat([ops.linear_fast(x[i:i + 4].to(dev), stream, nat.W_Q4, 2, N, K, **kw).float().cpu()
                        for i in range(0, 16, 4)])
        assert (y[:16] - ys).abs().max().item() <= 8e-3 * ref.abs().max().item()
        return
    xb = a["x"].to(torch.bfloat16)
    stream = ops.repack_q4(a["packed"], None, N, K, 1)
    kw = dict(scales=bf(a["scale"]), zeros=bf(a["zero"]), group_cols=g)
    ref64 = xb.double() @ a["wdq"].double().t()
    if epi == "accum":
        base = torch.randn((M, N), generator=gen)
        out = base.clone().to(dev)
        y = ops.linear_gemm(xb.to(dev), stream, 1, N, K, epi=nat.EPI_ACCUM, out=out, **kw).cpu()
        ref64 = ref64 + base.double()
    else:
        y = ops.linear_gemm(xb.to(dev), stream, 1, N, K, out_dtype=torch.float32, **kw).cpu()
        yo = oracle.colblock_linear(xb.float(), synth.pack_colblock(a["q"]).contiguous(), a["scale"], a["zero"], 4, g)
        assert (y - yo).abs().max().item() <= 1e-3 * _rms(ref64)
        rows = slice(0, 8)   # the skinny kernel (exact per-group sums) on the first rows
        ysk = ops.linear_fast(xb[rows].to(dev), stream, nat.W_Q4, 1, N, K, out_dtype=torch.float32, **kw).cpu()
        assert (y[rows] - ysk).abs().max().item() <= 1e-3 * _rms(ref64)
    err = (y.double() - ref64).abs().max().item()
    assert err <= 1e-3 * _rms(ref64), f"max err {err:.3e} vs rms {_rms(ref64):.3e}"


@pytest.mark.parametrize("M,N,K,epi", [(32, 64, 128, "store"), (200, 4096, 4096, "accum"), (257, 11008, 4096, "swiglu"),
                                        (96, 4096, 11008, "store"), (40, 72, 200, "store"), (128, 4096, 4096, "accum")])
def test_linear_gemm_over_the_bf16_stream(dev, M, N, K, epi):
    """The same GEMM over UNQUANTISED weights (BASELINE configs[1]; FMT = BF16 of csrc/gemm.hip: a stream piece is an MFMA
    A fragment as it lies) against x @ W^T in f32 on the CPU (nn.Linear of lit_llama/model.py:57,177-179,247-249) within
    the bf16-operand tolerance, and against the skinny streaming kernel over the same stream (same products, other order)."""
    gen = torch.Generator().manual_seed(M + N + K)
    w0 = (torch.randn((N, K), generator=gen) * K**-0.5).to(torch.bfloat16)
    x = torch.randn((M, K), generator=gen).to(torch.bfloat16)
    xd = x.to(dev)
    if epi == "swiglu":
        w1 = (torch.randn((N, K), generator=gen) * K**-0.5).to(torch.bfloat16)
        stream = ops.repack_bf16(w0.to(dev), w1.to(dev), 2)
        got = ops.linear_gemm(xd, stream, 2, N, K, epi=nat.EPI_SWIGLU, out_dtype=torch.bfloat16, fmt=nat.W_BF16).float().cpu()
        skinny = ops.linear_fast(xd, stream, nat.W_BF16, 2, N, K, epi=nat.EPI_SWIGLU, out_dtype=torch.bfloat16).float().cpu()
        ref = torch.nn.functional.silu(x.float() @ w0.float().t()) * (x.float() @ w1.float().t())
    else:
        stream = ops.repack_bf16(w0.to(dev), None, 1)
        ref = x.float() @ w0.float().t()
        if epi == "accum":
            base = torch.randn((M, N), generator=gen)
            o1, o2 = base.clone().to(dev), base.clone().to(dev)
            got = ops.linear_gemm(xd, stream, 1, N, K, epi=nat.EPI_ACCUM, out=o1, fmt=nat.W_BF16).float().cpu()
            skinny = ops.linear_fast(xd, stream, nat.W_BF16, 1, N, K, epi=nat.EPI_ACCUM, out=o2).float().cpu()
            ref = base + ref
        else:
            got = ops.linear_gemm(xd, stream, 1, N, K, out_dtype=torch.float32, fmt=nat.W_BF16).float().cpu()
            skinny = ops.linear_fast(xd, stream, nat.W_BF16, 1, N, K, out_dtype=torch.float32).float().cpu()
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 2e-2 * scale + 1e-3, f"vs f32 reference: {(got - ref).abs().max().item():.4e} (scale {scale:.3f})"
    assert (got - skinny).abs().max().item() <= (8e-3 if epi == "swiglu" else 1e-4) * scale + 1e-5, \
        f"vs skinny kernel: {(got - skinny).abs().max().item():.4e} (scale {scale:.3f})"


# ---------------------------------------------------------------------------------------------- wide LLM.int8 GEMM (prefill)
@pytest.mark.parametrize("N,K,M,epi,outliers", [(64, 128, 32, "store", 0), (512, 1024, 100, "store", 5), (4096, 4096, 160, "accum", 9),
                                                 (11008, 4096, 130, "swiglu", 7), (4096, 11008, 96, "store", 70),
                                                 (72, 200, 40, "store", 3),
                                                 # one token block against N = 4096 / 11008: split-K (8 / 2 slices of int32 partials)
                                                 (4096, 4096, 128, "accum", 6), (11008, 4096, 128, "swiglu", 4), (4096, 11008, 64, "store", 0)])
def test_int8_gemm_matches_oracle(dev, N, K, M, epi, outliers):
    """mi355_linear_int8_gemm against oracle.llm_int8_linear (the restated MatMul8bitLt forward: outlier columns over ALL rows,
    row absmax over sub-threshold entries, int32 accumulation, f16 roundings) — integer work exact, one f16 ulp on rows with
    an outlier side product (summed by the f16 MFMA here, in torch's matmul order there); 70 outlier columns exercise the
    second pass of the epilogue's LDS staging."""
    gen = torch.Generator().manual_seed(N + K + M)
    w = torch.randn((N, K), generator=gen) * K**-0.5
    x = torch.randn((M, K), generator=gen)
    for i in range(outliers):
        x[(7 * i) % M, (37 * i + 5) % K] = (6.0 + (i % 5)) * (1 if i % 2 else -1)  # |x| >= 6: outlier columns
    x = x.to(torch.bfloat16)
    ocb, oscb = oracle.int8_quant_rows(w)
    cb, scb = ops.int8_quant_rows(w.to(dev))
    xd = x.to(dev)
    # (the oracle returns x.dtype = bf16: its f16 result rounded once more; STORE into a bf16 output reproduces that exactly,
    # the f32 outputs of ACCUM / SwiGLU keep the f16 value: half a bf16 ulp = 2^-9 of slack on top)
    if epi == "swiglu":
        w2 = torch.randn((N, K), generator=gen) * K**-0.5
        ocb2, oscb2 = oracle.int8_quant_rows(w2)
        cb2, scb2 = ops.int8_quant_rows(w2.to(dev))
        stream = ops.repack_i8(cb, cb2, 2)
        y = ops.linear_int8_gemm(xd, stream, scb, 2, N, K, scb2=scb2, epi=nat.EPI_SWIGLU, out_dtype=torch.float32).float().cpu()
        a, b = oracle.llm_int8_linear(x, ocb, oscb).float(), oracle.llm_int8_linear(x, ocb2, oscb2).float()
        ref = torch.nn.functional.silu(a) * b
        tol = 2.0**-7 * (a.abs() * b.abs() + b.abs()) + 1e-5  # an ulp of either factor through the product
    else:
        stream = ops.repack_i8(cb, None, 1)
        ref = oracle.llm_int8_linear(x, ocb, oscb).float()
        if epi == "accum":
            base = torch.randn((M, N), generator=gen)
            out = base.clone().to(dev)
            y = ops.linear_int8_gemm(xd, stream, scb, 1, N, K, epi=nat.EPI_ACCUM, out=out).float().cpu() - base
            tol = 2.0**-8 * ref.abs() + 1e-5 + 2.0**-22 * base.abs()
        else:
            y = ops.linear_int8_gemm(xd, stream, scb, 1, N, K, out_dtype=torch.bfloat16).float().cpu()
            tol = 2.0**-9 * ref.abs() + 1e-6
    close = (y - ref).abs() <= tol
    if outliers and epi == "store":
        # (round 6: the f16 side product runs on v_mfma_f32_16x16x32_f16 — exact products, f32 sums in the hardware's order instead of
        # ascending k; where that moves an f16 rounding AND the value sits on a bf16 rounding boundary the stored bf16 moves one ulp:
        # a handful of outputs in 4 x 10^5, never more than that ulp)
        assert float((~close).float().mean()) <= 1e-4, f"{int((~close).sum())} of {close.numel()} outputs differ"
        # (... of the SIDE PRODUCT's magnitude: where it cancels against the int8 part the output is small and the ulp is not)
        loose = 4 * tol + 2.0**-10 * float(ref.abs().max())
        assert bool(((y - ref).abs() <= loose).all()), f"worst {float(((y - ref).abs() - loose).max()):.3e} past one f16 ulp of the side product"
    else:
        assert bool(close.all()), f"{int((~close).sum())} of {close.numel()} outputs differ; worst {float(((y - ref).abs() - tol).max()):.3e}"
    if outliers == 0 and epi == "store":
        assert torch.equal(y, ref)
    # linear_int8 routes wide inputs here: same result through the public entry point
    if epi == "store":
        y2 = ops.linear_int8(xd, stream, scb, 1, N, K, out_dtype=torch.bfloat16).float().cpu()
        assert torch.equal(y2, y)

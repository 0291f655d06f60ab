// Skinny (M <= 16) weight-streaming linear for gfx950: the decode hot kernel.
//
// Replaces, for decode shapes, the reference's Triton `linear_kernel_4bit_weight`
// (/root/reference lit_llama/quantization.py:187-333, reached from
// ColBlockQuantizedLinear.forward :413-421) and the dense F.linear of lit_llama/model.py.
//
// Design:
//  * HBM-bound: each weight byte is read exactly once, as fully coalesced 1-KiB wave loads
//    (64 lanes x 16 B, non-temporal) from a load-time repacked stream
//    [tile][unit][r][lane][16 B] — straight into VGPRs, no LDS round trip for weights
//    (they are not shared between waves), a P-deep register ring keeps P KiB per wave in flight
//    across tile boundaries and across the block-level reduction.
//  * The multiply runs on the matrix pipe, which is otherwise idle in a GEMV:
//    v_mfma_f32_16x16x32_bf16 with A = 16 weight rows x 32 k, B = the (<= 16) activation rows.
//    int4 weights become bf16 MFMA operands with 7 VALU ops per 8 weights:
//    (w >> 4i) & 0x000F000F | 0x43004300 is the bf16 pair (128 + q_a, 128 + q_b) exactly;
//    the +128 and the GPTQ zero point are removed in the epilogue:
//        y[n] = scale[n] * (acc[n] - (128 + zero[n]) * sum_k x[k]).
//  * The activation vector (optionally RMSNorm'ed on the fly) is staged once per workgroup
//    into LDS as bf16; B fragments are 16-B LDS broadcasts.
//  * K is split over the waves of a workgroup; partial 16x16 tiles are combined through LDS in
//    a fixed order (deterministic), then the epilogue (dequant, bias, residual accumulate or
//    SwiGLU of an interleaved c_fc1/c_fc2 pair) writes the 16*R outputs of the tile.
#include <mutex>

#include <hip/hip_ext.h>

#include "common.h"

#ifndef MI355_GEMV_AND_OR
#define MI355_GEMV_AND_OR 1
#endif


namespace {

constexpr int kUnitK = 128;        // input columns per stream unit
constexpr int kMaxM = 16;
constexpr int kLdsHeader = 1024;   // wss[16][16] floats
constexpr int kMaxLds = 160 * 1024;

struct GemvParams {
    const uint8_t* w;
    const void* x;
    const void* norm_scale;
    const void* scales;
    const void* zeros;
    const void* scales2;
    const void* zeros2;
    const void* bias;
    void* y;
    int64_t ldx, ldy;
    int N, K, M, n_tiles, units;
    int u_q, u_r;  // units / waves, units % waves: wave w takes u_q (+1 if w < u_r) consecutive units of every tile
    int nu_pad;    // ring turns are whole: every wave steps through nu_pad = ceil(max units per wave / P) * P units per
                   // tile; the steps past its own units load nothing (out-of-range offsets) and multiply zeros
    int t_q, t_r;  // n_tiles / grid, n_tiles % grid: workgroup b takes t_q (+1 if b < t_r) tiles, b, b + grid, ...
    int x_dtype, norm_dtype, sz_dtype, y_dtype, epi;
    int xs_stride;  // bytes per LDS activation row
    int vec_mode;   // 0 scalar staging, 1 f32 + bf16-scale RMSNorm vectorised, 2 bf16 copy vectorised
    unsigned w_bytes;  // size of the weight stream (buffer descriptor bound)
    const float* attn_part;  // vec_mode 3: activations = combine of split-attention partial records
    int attn_splits, attn_heads, attn_hs;
    unsigned long long* dbg;  // optional wall-clock stamps [grid][8]
    float eps;
    // grouped scales (GRP kernels): scales / zeros are [N, n_groups] bf16, one pair per 32 << gq_shift input columns
    int n_groups, gq_shift;
    float inv_ng;  // 1 / n_groups (element index -> row without an integer division)
};

template <int FMT>
struct Fmt;
template <>
struct Fmt<MI355_W_Q4> {
    static constexpr int kPieces = 1;  // 16-B pieces per (unit, r) per lane
};
template <>
struct Fmt<MI355_W_BF16> {
    static constexpr int kPieces = 4;
};

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// f32 / bf16 only (keeps the unrolled hot loop small; f16 tensors go through the generic operators)
__device__ __forceinline__ float ld2(const void* p, int64_t i, int dtype) {
    return dtype == MI355_F32 ? ((const float*)p)[i] : bf16_to_f32(((const bf16_t*)p)[i]);
}
__device__ __forceinline__ void st2(void* p, int64_t i, int dtype, float v) {
    if (dtype == MI355_F32)
        ((float*)p)[i] = v;
    else
        ((bf16_t*)p)[i] = f32_to_bf16(v);
}

template <bool NT>
__device__ __forceinline__ u32x4 ldw(const uint8_t* p) {
    if constexpr (NT)
        return __builtin_nontemporal_load((const u32x4*)p);
    else
        return *(const u32x4*)p;
}

// ------------------------------------------------------------------------------------ activation staging
// The activation rows go to LDS as bf16 (the MFMA B operand).  What the timeline stamps showed for the first
// version (load -> block reduce -> normalise -> block reduce -> barrier, issued AFTER the weight ring): ~4 us of a
// 12-14 us launch before the first MFMA.  So:
//   * RMSNorm's 1/rms is a per-row scalar and commutes with the linear map: rows are staged as bf16(scale_k x_k)
//     and the epilogue multiplies by rsqrt(mean(x^2) + eps) (lit_llama/model.py:274-277, same arithmetic up to
//     where the one bf16 rounding sits).  The sum of squares needs no barrier of its own: per-wave partials are
//     published together with the staged row.
//   * the row sum needed to undo the +128 / zero-point offset comes out of the matrix pipe: one extra MFMA per
//     k-step with an all-ones A fragment yields sum_k x_k of exactly the rounded operands.
//   * the row's loads are issued BEFORE the weight ring: VMEM returns in order, so loads queued behind 4-8 KiB
//     of HBM weight traffic per wave would only be usable when that traffic has landed.
// Vectorised modes (p.vec_mode, chosen on the host; all need K % 8 == 0 and 16-B aligned rows):
//   1: x f32 with a bf16 norm scale, <= 2 x 8 elements per thread      (decode: c_attn, c_fc1/c_fc2, lm_head)
//   4: the same with <= 4 x 8 elements per thread                      (n_embd 5120 .. 8192: 13B .. 65B)
//   2: x bf16, no norm, <= 6 x 8 elements per thread                   (decode: mlp.c_proj; attn.c_proj unsplit)
//   3: x = combine of split-attention partial records, <= 8 elements per thread, <= 4 splits   (attn.c_proj)
//   0: any dtype / shape, element loop.
template <int VMODE>
struct Stager;

template <>
struct Stager<0> {  // element loop, any dtype
    __device__ __forceinline__ void load(const GemvParams&, int) {}
    __device__ __forceinline__ float store(const GemvParams& p, int m, char* xs) {
        const int tid = threadIdx.x, nt = blockDim.x;
        const bool norm = p.norm_scale != nullptr;
        const int64_t base = (int64_t)m * p.ldx;
        bf16_t* row = (bf16_t*)(xs + (size_t)m * p.xs_stride);
        float ss = 0.f;
        for (int k = tid; k < p.K; k += nt) {
            float v = ld2(p.x, base + k, p.x_dtype);
            if (norm) {
                ss += v * v;
                v = ld2(p.norm_scale, k, p.norm_dtype) * v;
            }
            row[k] = f32_to_bf16(v);
        }
        return ss;
    }
};

template <int NC>
struct StagerF32Norm {  // f32 row, bf16 norm scale, <= NC x 8 elements per thread
    f32x4 xa[NC], xb[NC];
    u32x4 ns[NC];
    __device__ __forceinline__ void load(const GemvParams& p, int m) {
        const int tid = threadIdx.x, nt = blockDim.x, nvec = p.K >> 3;
        const float* xrow = (const float*)p.x + (int64_t)m * p.ldx;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int v = tid + c * nt;
            if (v < nvec) {
                xa[c] = *(const f32x4*)(xrow + v * 8);
                xb[c] = *(const f32x4*)(xrow + v * 8 + 4);
                ns[c] = *(const u32x4*)((const bf16_t*)p.norm_scale + v * 8);
            }
        }
    }
    __device__ __forceinline__ float store(const GemvParams& p, int m, char* xs) {
        const int tid = threadIdx.x, nt = blockDim.x, nvec = p.K >> 3;
        char* row = xs + (size_t)m * p.xs_stride;
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int v = tid + c * nt;
            if (v < nvec) {
                u32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float x0 = i < 2 ? xa[c][2 * i] : xb[c][2 * i - 4];
                    const float x1 = i < 2 ? xa[c][2 * i + 1] : xb[c][2 * i - 3];
                    ss += x0 * x0 + x1 * x1;
                    const bf16_t a = f32_to_bf16(__uint_as_float(ns[c][i] << 16) * x0);
                    const bf16_t b = f32_to_bf16(__uint_as_float(ns[c][i] & 0xffff0000u) * x1);
                    o[i] = (uint32_t)a | ((uint32_t)b << 16);
                }
                *(u32x4*)(row + v * 16) = o;
            }
        }
        return ss;
    }
};
template <>
struct Stager<1> : StagerF32Norm<2> {};  // n_embd <= 4096 at 512 threads (7B)
template <>
struct Stager<4> : StagerF32Norm<4> {};  // n_embd <= 8192 (13B .. 65B)

template <>
struct Stager<2> {  // bf16 row, no norm, <= 6 x 8 elements per thread (65B mlp.c_proj: K = 22016)
    u32x4 r[6];
    __device__ __forceinline__ void load(const GemvParams& p, int m) {
        const int tid = threadIdx.x, nt = blockDim.x, nvec = p.K >> 3;
        const bf16_t* xrow = (const bf16_t*)p.x + (int64_t)m * p.ldx;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int v = tid + c * nt;
            if (v < nvec) r[c] = *(const u32x4*)(xrow + v * 8);
        }
    }
    __device__ __forceinline__ float store(const GemvParams& p, int m, char* xs) {
        const int tid = threadIdx.x, nt = blockDim.x, nvec = p.K >> 3;
        char* row = xs + (size_t)m * p.xs_stride;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int v = tid + c * nt;
            if (v < nvec) *(u32x4*)(row + v * 16) = r[c];
        }
        return 0.f;
    }
};

template <>
struct Stager<3> {  // split-attention partial records, <= 8 elements per thread, <= 4 splits
    float mj[4], lj[4];
    f32x4 oa[4], ob[4];
    __device__ __forceinline__ void load(const GemvParams& p, int m) {
        const int tid = threadIdx.x, nvec = p.K >> 3;
        if (tid < nvec) {
            const int hs = p.attn_hs, rs = hs + 4;
            const int k0 = tid * 8, h = k0 / hs, d0 = k0 - h * hs;
            const float* rec = p.attn_part + ((int64_t)m * p.attn_heads + h) * p.attn_splits * rs;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < p.attn_splits) {
                    const float* q = rec + j * rs;
                    mj[j] = q[0];
                    lj[j] = q[1];
                    oa[j] = *(const f32x4*)(q + 4 + d0);
                    ob[j] = *(const f32x4*)(q + 8 + d0);
                }
        }
    }
    __device__ __forceinline__ float store(const GemvParams& p, int m, char* xs) {
        const int tid = threadIdx.x, nvec = p.K >> 3;
        if (tid < nvec) {
            // x[h*hs + d] = sum_j e^{m_j - M} o_j[d] / sum_j e^{m_j - M} l_j  (flash-decoding combine)
            float M_ = -1.0e30f, L = 0.f;
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < p.attn_splits) {
                    const float Mn = fmaxf(M_, mj[j]);
                    const float c_old = expf(M_ - Mn), c_new = expf(mj[j] - Mn);
                    L = L * c_old + lj[j] * c_new;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[i] = acc[i] * c_old + oa[j][i] * c_new;
                        acc[4 + i] = acc[4 + i] * c_old + ob[j][i] * c_new;
                    }
                    M_ = Mn;
                }
            const float inv = 1.0f / L;
            u32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                o[i] = (uint32_t)f32_to_bf16(acc[2 * i] * inv) | ((uint32_t)f32_to_bf16(acc[2 * i + 1] * inv) << 16);
            *(u32x4*)(xs + (size_t)m * p.xs_stride + tid * 16) = o;
        }
        return 0.f;
    }
};

// Per-output operands of one tile's epilogue, owned by thread (e_row, e_col) of the first 256 threads and
// fetched ONE TILE AHEAD: an epilogue that issued its own loads would have to wait for them with vmcnt(0),
// draining the weight ring at every tile.
// The loads are kept as RAW bits (converted at use): converting right after the load would make hipcc wait
// for it on the spot.
template <int R>
struct EpiOps {
    uint32_t s[R], z[R];   // dword holding the Q4 scale / zero of the output row
    uint32_t bias[R];
    uint32_t old[R];       // dword holding the value to accumulate into
    uint32_t sh_sz[R], sh_y[R];  // bit offset of a 16-bit element inside its dword (0 for f32)
};

// Buffer descriptors of the epilogue operands.  Every operand load is ONE unconditional buffer_load_dword: rows
// past N, non-owner threads and absent operands (descriptor of 0 bytes) fall out of range, which returns zero
// without a memory request and — unlike a guarded load — keeps hipcc's vmcnt bookkeeping exact, so that waiting
// for these operands at a tile's end does not also drain the weight ring.  A 16-bit element is fetched as the
// aligned dword that contains it.
struct EpiRsrc {
    __amdgpu_buffer_rsrc_t s0, z0, s1, z1, bias, y;
    int sz_shift, y_shift;  // log2(element bytes)
};
__device__ __forceinline__ EpiRsrc make_epi_rsrc(const GemvParams& p, int M) {
    EpiRsrc r;
    r.sz_shift = p.sz_dtype == MI355_F32 ? 2 : 1;
    r.y_shift = p.y_dtype == MI355_F32 ? 2 : 1;
    const int nb = ((p.N << r.sz_shift) + 3) & ~3;
    const int nbg = (((p.N * (p.n_groups > 0 ? p.n_groups : 1)) << r.sz_shift) + 3) & ~3;  // [N, n_groups] when grouped
    auto mk = [](const void* ptr, int bytes) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)ptr, 0, ptr != nullptr ? bytes : 0, 0x00020000);
    };
    r.s0 = mk(p.scales, nbg);
    r.z0 = mk(p.zeros, nbg);
    r.s1 = mk(p.scales2, nbg);
    r.z1 = mk(p.zeros2, nbg);
    r.bias = mk(p.bias, nb);
    r.y = mk(p.y, (int)(((((int64_t)(M - 1) * p.ldy + p.N) << r.y_shift) + 3) & ~(int64_t)3));
    return r;
}
__device__ __forceinline__ uint32_t ld_epi_dword(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, bool ok) {
    return __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (byte_off & ~3u) : 0xFFFFFFF0u, 0, 0);
}
__device__ __forceinline__ float cvt2(uint32_t raw, int dtype) {
    return dtype == MI355_F32 ? __uint_as_float(raw) : __uint_as_float(raw << 16);
}

template <int FMT, int R, int EPI, int GRP = 0>
__device__ __forceinline__ void load_epi(const GemvParams& p, const EpiRsrc& er, int tile, int e_row, int e_col,
                                         bool e_owner, EpiOps<R>& o) {
    constexpr bool sw = EPI == MI355_EPI_SWIGLU;
    const bool live = e_owner && tile < p.n_tiles;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int n = sw ? tile * 16 + e_row : (tile * R + r) * 16 + e_row;
        const bool ok = live && n < p.N;
        const unsigned off = (unsigned)n << er.sz_shift;
        o.sh_sz[r] = er.sz_shift == 2 ? 0u : (off & 2u) * 8u;
        if constexpr (FMT == MI355_W_Q4 && !GRP) {
            o.s[r] = ld_epi_dword((sw && r == 1) ? er.s1 : er.s0, off, ok);
            o.z[r] = ld_epi_dword((sw && r == 1) ? er.z1 : er.z0, off, ok);
        }
        if constexpr (!sw) {
            o.bias[r] = ld_epi_dword(er.bias, off, ok);
            if constexpr (EPI == MI355_EPI_ACCUM) {
                const unsigned yo = (unsigned)(((int64_t)e_col * p.ldy + n) << er.y_shift);
                o.sh_y[r] = er.y_shift == 2 ? 0u : (yo & 2u) * 8u;
                o.old[r] = ld_epi_dword(er.y, yo, ok);
            }
        }
    }
}

// Combine the W partial 16x16 tiles of `buf` in wave order and write the tile's outputs (stores only).
// RS = partial tiles per wave: R weight tiles (+ the all-ones tile carrying sum_k x_k for Q4).
// GRP: the partial tiles are already dequantised (grouped scales are applied inside the k loop).
template <int FMT, int R, int EPI, bool MULTI, int GRP = 0>
__device__ __forceinline__ void tile_epilogue(const GemvParams& p, const char* part, int buf, int W, int tile,
                                              int e_row, int e_col, const EpiOps<R>& o, float rinv) {
    constexpr bool kDeq = FMT == MI355_W_Q4 && !GRP;
    constexpr int RS = R + (kDeq ? 1 : 0);
    // D layout of mfma_f32_16x16x32: lane (row >> 2) * 16 + col holds D[row][col] in register (row & 3)
    float sx = 0.f;
    float v[R];
    if (!MULTI) {
        // decode: the 16 lanes of a row's group each fetch ONE wave's partial and the group is summed with DPP
        // (fixed butterfly order -> reproducible); a serial loop over W x (R + 1) LDS reads per owner cost ~0.6 us
        // per tile on the critical path.  Here e_col doubles as the wave index.
        const int src = (e_row >> 2) << 4;  // column 0
        const float* base = (const float*)(part + (size_t)(buf * W * RS) * 1024) + src * 4 + (e_row & 3);
        float t[RS];
#pragma unroll
        for (int r = 0; r < RS; ++r) t[r] = e_col < W ? base[(e_col * RS + r) * 256] : 0.f;
#pragma unroll
        for (int r = 0; r < RS; ++r) t[r] = group_sum(t[r], 16);
        if (e_col != 0) return;
        if constexpr (kDeq) sx = t[RS - 1];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = t[r];
    } else {
        const int src = ((e_row >> 2) << 4) | e_col;
        const float* base = (const float*)(part + (size_t)(buf * W * RS) * 1024) + src * 4 + (e_row & 3);
        if constexpr (kDeq) {
            for (int w = 0; w < W; ++w) sx += base[(w * RS + R) * 256];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float s = 0.f;
            for (int w = 0; w < W; ++w) s += base[(w * RS + r) * 256];
            v[r] = s;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float s = v[r];
        if constexpr (kDeq)
            s = cvt2(o.s[r] >> o.sh_sz[r], p.sz_dtype) * (s - (128.f + cvt2(o.z[r] >> o.sh_sz[r], p.sz_dtype)) * sx);
        v[r] = s * rinv;
    }
    if constexpr (EPI == MI355_EPI_SWIGLU) {
        if constexpr (R == 2) {
            const int n = tile * 16 + e_row;
            if (n < p.N) st2(p.y, (int64_t)e_col * p.ldy + n, p.y_dtype, swiglu_f32(v[0], v[1]));
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = (tile * R + r) * 16 + e_row;
            if (n < p.N) {
                float out = v[r];
                if (p.bias != nullptr) out += cvt2(o.bias[r] >> o.sh_sz[r], p.sz_dtype);
                if (EPI == MI355_EPI_ACCUM) out += cvt2(o.old[r] >> o.sh_y[r], p.y_dtype);
                st2(p.y, (int64_t)e_col * p.ldy + n, p.y_dtype, out);
            }
        }
    }
}

// EPI is a template parameter so that the SwiGLU pair kernel (c_fc1/c_fc2: the largest launch of a decode
// step) is its own symbol in profiles, and the epilogue carries no runtime switch.
// Up to 16 waves (1024 threads) per workgroup for the lean Q4 P=4 variants (<= 128 VGPRs); the register-hungrier
// ones (deep ring, bf16 weights: 4 pieces per unit) stay at 8 waves.
template <int FMT, int P, int GRP = 0>
constexpr int kMaxThreads = (FMT == MI355_W_Q4 && P <= 4 && !GRP) ? 1024 : 512;

// MULTI = false is the decode step (M == 1): no row loop, no per-row branches — the loop around the row-staging
// loads alone cost 0.4-1 us per launch through hipcc's conservative vmcnt waits (12.9 -> 11.9 us for the fc pair).
// GRP (Q4 only): one (scale, zero) pair per output row AND group of 32 << gq_shift input columns (GPTQ "groupsize",
// quantization.py:284-333 with tile_cols > 0).  A tile's pairs ([16 R rows][n_groups], bf16) are fetched ONE TILE AHEAD
// with the epilogue-operand trick (fixed number of unconditional buffer loads per thread, kept as raw bits) and parked
// in LDS at the tile switch; inside the k loop every wave applies  accf += s (acc - (128 + z) sum_x)  at each group
// boundary of its unit range (sum_x from the all-ones MFMA of the same columns) and restarts acc / sum_x.  The partial
// tiles that reach the epilogue are then plain sums.
constexpr int kGrpLoadsMax = 12; // GRP = (scale, zero) pairs per thread, matrix and tile: 16 n_groups <= GRP x threads (1, 4 or 12;
                                 // every pair is an issued buffer load whether it fetches or not: 32 groups over 512 threads need one).
                                 // 12 (round 6) admits 384 groups per row: groupsize 32 against K = 11008 (344), which the cap of 4 x 512 / 16
                                 // = 128 groups had kept — with groupsize 64 (172) — off the engine altogether
constexpr int kGrpLoadsMid = 4;

template <int FMT, int R, int P, int EPI, int VMODE, bool MULTI, int GRP = 0>
__global__ __launch_bounds__((kMaxThreads<FMT, P, GRP>)) void gemv_kernel(const GemvParams p) {
    static_assert(!GRP || FMT == MI355_W_Q4, "grouped scales are a Q4 feature");
    const int M = MULTI ? p.M : 1;
    constexpr bool NT = true;  // weights are read once: non-temporal
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wss = (float*)smem;  // [W][16] per-wave partial sums of x^2 (RMSNorm)
    char* part = smem + kLdsHeader;

    constexpr int kPieces = Fmt<FMT>::kPieces;
    constexpr int kSlot = R * kPieces;  // 16-B pieces per lane per unit
    constexpr int RS = R + ((FMT == MI355_W_Q4 && !GRP) ? 1 : 0);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = blockDim.x >> 6;
    // int4 -> bf16: nibble mask and exponent pattern as OPAQUE register values — hipcc then selects ONE v_and_or_b32 per
    // field (7 VALU per 8 weights; from literals it emits v_and + v_or: 11) and pads the VALU -> MFMA hazard itself, which an
    // inline-asm v_and_or_b32 does not get (csrc/fused_step_ring.hip nib2f16, NOTES.md)
    [[maybe_unused]] uint32_t cmask = 0x000F000Fu, cmagic = 0x43004300u;
#if MI355_GEMV_AND_OR
    asm volatile("" : "+s"(cmask));
    asm volatile("" : "+v"(cmagic));
#endif
    char* xs = part + 2 * W * RS * 1024;
    // GRP: [2][R * 16 rows][n_groups] dwords (scale bf16 | zero bf16 << 16) behind the activation rows
    uint32_t* szl = (uint32_t*)(xs + (((size_t)M * p.xs_stride + 15) & ~(size_t)15));
#define MI355_STAMP(i)                                                                          \
    do {                                                                                        \
        if (p.dbg != nullptr && threadIdx.x == 0) p.dbg[blockIdx.x * 8 + (i)] = wall_clock64(); \
    } while (0)
    MI355_STAMP(0);

    const int units = p.units;
    // K split over the waves / tiles over the workgroups, from host-computed quotients (integer divisions by
    // runtime values cost ~60 instructions of every launch's prologue)
    const int nu = p.u_q + (wave < p.u_r ? 1 : 0);
    const int u0 = wave * p.u_q + (wave < p.u_r ? wave : p.u_r);
    const int bid = blockIdx.x, nb = gridDim.x;
    const int my_tiles = p.t_q + (bid < p.t_r ? 1 : 0);
    const int nu_pad = p.nu_pad;
    const int total = my_tiles * nu_pad;

    // ---- row 0 of the activations: loads first (in-order VMEM return, see Stager)
    Stager<VMODE> stager;
    stager.load(p, 0);
    // ---- weight prefetch ring: P units in flight per wave.
    // Every refill is an UNCONDITIONAL load: a conditional refill makes the ring registers phi nodes, and hipcc
    // then drains the whole ring with s_waitcnt vmcnt(0) at the loop back-edge to copy them.  Past the end of the
    // wave's work the offset is pushed beyond the buffer descriptor's bound: the hardware returns zeros and
    // issues NO memory request.  (Re-reading a fixed dummy address instead funnels every wave of the chip into
    // one L2 channel: measured ~8 us of a 16 us launch.)
    u32x4 ring[P][kSlot];
    int pf_tile = bid, pf_u = 0, pf_n = 0;  // pf_u: step inside the tile, 0 .. nu_pad - 1 (real units: pf_u < nu)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    const unsigned lane_off = lane * 16;
    const unsigned unit_bytes32 = (unsigned)kSlot * 1024u;
#define MI355_ISSUE(slot)                                                                                       \
    do {                                                                                                        \
        const bool ok__ = pf_n < total && pf_u < nu;                                                            \
        const unsigned off__ =                                                                                  \
            ok__ ? ((unsigned)pf_tile * (unsigned)units + (unsigned)(u0 + pf_u)) * unit_bytes32 + lane_off      \
                 : 0xFFFFF000u;                                                                                 \
        _Pragma("unroll") for (int s__ = 0; s__ < kSlot; ++s__) ring[slot][s__] = __builtin_bit_cast(           \
            u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off__ + s__ * 1024, 0, NT ? 2 : 0));             \
        ++pf_n;                                                                                                 \
        if (++pf_u == nu_pad) {                                                                                 \
            pf_u = 0;                                                                                           \
            pf_tile += nb;                                                                                      \
        }                                                                                                       \
    } while (0)

    // ---- per-tile epilogue operands, fetched one tile ahead (threads < 256 own output (row, col)).  The first
    // tile's are requested BEFORE the ring: they sit in a divergent branch, so hipcc cannot count them, and a wait
    // for the activation loads then also waits for as many of the OLDEST later loads — which must not be weights.
    // (The owners are always waves 0-3: the CU's memory pipeline serves the waves in issue order, so these get
    // their weights first and have slack; taking turns with waves 4-7 was measured 4 % slower end to end.)
    const int e_row = (threadIdx.x >> 4) & 15, e_col = threadIdx.x & 15;
    const bool e_owner = threadIdx.x < 256 && e_col < M;
    EpiOps<R> eo;
#pragma unroll
    for (int r = 0; r < R; ++r) eo.s[r] = eo.z[r] = eo.bias[r] = eo.old[r] = eo.sh_sz[r] = eo.sh_y[r] = 0u;
    const EpiRsrc er = make_epi_rsrc(p, M);
    load_epi<FMT, R, EPI, GRP>(p, er, bid, e_row, e_col, e_owner, eo);
    // GRP: the (scale, zero) pairs of a tile, element e = thread + k * threads of the [16][n_groups] block of matrix r
    constexpr int kGrpLoads = GRP ? GRP : 1;
    uint32_t gs[GRP ? R : 1][kGrpLoads], gz[GRP ? R : 1][kGrpLoads];
    auto grp_load = [&](int t) {
        if constexpr (GRP) {
            constexpr bool sw = EPI == MI355_EPI_SWIGLU;
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int k = 0; k < kGrpLoads; ++k) {
                    const int e = (int)threadIdx.x + k * (int)blockDim.x;
                    const int row = (int)(((float)e + 0.5f) * p.inv_ng);
                    const int grp = e - row * p.n_groups;
                    const int n = sw ? t * 16 + row : (t * R + r) * 16 + row;
                    const bool ok = row < 16 && t < p.n_tiles && n < p.N;
                    const unsigned off = (unsigned)(n * p.n_groups + grp) * 2u;
                    gs[r][k] = ld_epi_dword((sw && r == 1) ? er.s1 : er.s0, off, ok);
                    gz[r][k] = ld_epi_dword((sw && r == 1) ? er.z1 : er.z0, off, ok);
                }
            }
        }
    };
    auto grp_park = [&](int t, int b) {  // raw bits -> LDS block b; rows past N carry (0, 0): they multiply zeros
        if constexpr (GRP) {
            constexpr bool sw = EPI == MI355_EPI_SWIGLU;
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int k = 0; k < kGrpLoads; ++k) {
                    const int e = (int)threadIdx.x + k * (int)blockDim.x;
                    const int row = (int)(((float)e + 0.5f) * p.inv_ng);
                    const int grp = e - row * p.n_groups;
                    const int n = sw ? t * 16 + row : (t * R + r) * 16 + row;
                    const unsigned sh = ((unsigned)(n * p.n_groups + grp) & 1u) * 16u;
                    if (row < 16)
                        szl[((b * R + r) * 16 + row) * p.n_groups + grp] =
                            ((gs[r][k] >> sh) & 0xFFFFu) | ((gz[r][k] >> sh) << 16);
                }
            }
        }
    };
    grp_load(bid);

#pragma unroll
    for (int j = 0; j < P; ++j) MI355_ISSUE(j);
    MI355_STAMP(1);

    // ---- stage the activation rows; per-wave partial sums of squares ride along (no extra barrier).
    // Row 0 (the only row of a decode step) is staged by straight-line code: inside the loop over rows the loads
    // of row m + 1 make hipcc wait with vmcnt(0), i.e. for the whole ring prefill (measured: "x staged" moved
    // from 2.0 to 3.4 us after ring issue and grew with the ring depth).
    auto stage_row = [&](int m) {
        float ss = stager.store(p, m, xs);
        for (int k = p.K + (int)threadIdx.x; k < (p.units + 1) * kUnitK; k += blockDim.x)
            ((bf16_t*)(xs + (size_t)m * p.xs_stride))[k] = 0;  // stream padding + the all-zero unit of the idle steps
        if constexpr (VMODE == 0 || VMODE == 1 || VMODE == 4) {
            ss = group_sum(ss, 64);  // 4 DPP steps + 2 ds_bpermute (wave_sum: 6 ds_bpermute, ~0.2 us more)
            if (lane == 0) wss[wave * 16 + m] = ss;
        }
    };
    stage_row(0);
    for (int m = 1; m < M; ++m) {
        stager.load(p, m);
        stage_row(m);
    }
    grp_park(bid, 0);
    grp_load(bid + nb);
    __syncthreads();
    MI355_STAMP(2);

    float rinv = 1.f;
    if (p.norm_scale != nullptr && e_owner) {
        float ss = 0.f;
        for (int w = 0; w < W; ++w) ss += wss[w * 16 + e_col];
        rinv = rsqrtf(ss / (float)p.K + p.eps);
    }

    f32x4 acc[R], acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accf[GRP ? R : 1];  // GRP: the dequantised sums of the groups finished so far
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < (GRP ? R : 1); ++r) accf[r] = f32x4{0.f, 0.f, 0.f, 0.f};

    int tile = bid, buf = 0;

    // ---- main loop
    const int g = lane >> 4, c = lane & 15;
    const int xrow = c < M ? c : M - 1;
    const char* xl = xs + (size_t)xrow * p.xs_stride + g * 64;
    const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};  // 8 x bf16 1.0
    int uu = 0;

    // GRP decode step (see consume): first group of the 16-group column window of this wave, and the end of a window
    int gbase = GRP ? (u0 >> (p.gq_shift >= 2 ? p.gq_shift - 2 : 0)) : 0;
    auto apply_window = [&]() {
        if constexpr (GRP && !MULTI) {
            int grp = gbase + c;
            grp = grp < p.n_groups ? grp : p.n_groups - 1;  // (columns past the wave's range hold zeros)
            const uint32_t* sz = szl + ((buf * R) * 16 + 4 * g) * p.n_groups + grp;
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const uint32_t w = sz[(r * 16 + rr) * p.n_groups];
                    const float sc = __uint_as_float(w << 16), zp = __uint_as_float(w & 0xffff0000u);
                    accf[r][rr] += sc * (acc[r][rr] - (128.f + zp) * acc1[rr]);
                }
                acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // one k-unit: operands from LDS, int4 -> bf16, MFMAs
    auto consume = [&](int j) {
        const char* xb = xl + (uu < nu ? u0 + uu : units) * (kUnitK * 2);  // idle step: the all-zero unit
        bf16x8 b[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) b[d] = *(const bf16x8*)(xb + 16 * d);

        if constexpr (GRP && !MULTI) {
            if (p.gq_shift >= 2) {
                // Decode step, groups of whole units: the MFMA's 16 token columns are idle (M = 1), so the activations
                // of group g go to COLUMN (g - gbase) & 15 (every other column reads the all-zero unit) and ONE running
                // accumulator holds the sums of 16 groups side by side — no per-group work in the k loop at all; the
                // scales are applied once per tile (or every 16 groups of a wave's range) in apply_window().
                const int grp = (u0 + uu) >> (p.gq_shift - 2);
                if (uu < nu && grp - gbase >= 16) {  // (wave-uniform) the window is full
                    apply_window();
                    gbase += 16;
                }
                const bool mine = uu < nu && c == ((grp - gbase) & 15);
                const char* xc = xs + g * 64 + (mine ? u0 + uu : units) * (kUnitK * 2);
                bf16x8 bc[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) bc[d] = *(const bf16x8*)(xc + 16 * d);
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ones), bc[d], acc1, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const u32x4 q = ring[j][r];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const uint32_t v = q[d];
                        u32x4 a;
                        a[0] = (v & cmask) | cmagic;
                        a[1] = ((v >> 4) & cmask) | cmagic;
                        a[2] = ((v >> 8) & cmask) | cmagic;
                        a[3] = ((v >> 12) & cmask) | cmagic;
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a), bc[d], acc[r], 0, 0, 0);
                    }
                }
                return;
            }
        }
        if constexpr (GRP) {
            // The k dimension of ONE MFMA is spread over the unit (lane group g holds columns 32 g + 8 d .. + 7 of
            // MFMA d), so the four MFMAs of a unit always add up whole 128-column units.  Groups of >= 128 columns
            // end at unit boundaries; groups of 32 / 64 columns are lane groups {g} / {g >> 1} of the B operand:
            // one pass per sub-group with the other lane groups' activations zeroed (the matrix pipe is idle
            // otherwise; the conversions are redone per pass — this layout is the rare one).
            const bool real = uu < nu;
            const int unit = u0 + uu;
            const int sub_shift = p.gq_shift < 2 ? p.gq_shift : 2;      // lane groups per sub-group = 1 << sub_shift
            const int nsub = 4 >> sub_shift;                            // passes over the unit: 4, 2 or 1
#pragma unroll 1
            for (int sub = 0; sub < nsub; ++sub) {
                const bool mine = (g >> sub_shift) == sub;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const u32x4 raw = __builtin_bit_cast(u32x4, b[d]);
                    const bf16x8 bm = as_bf16x8(mine ? raw : u32x4{0u, 0u, 0u, 0u});
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ones), bm, acc1, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t v = ring[j][r][d];
                        u32x4 a;
                        a[0] = (v & cmask) | cmagic;
                        a[1] = ((v >> 4) & cmask) | cmagic;
                        a[2] = ((v >> 8) & cmask) | cmagic;
                        a[3] = ((v >> 12) & cmask) | cmagic;
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a), bm, acc[r], 0, 0, 0);
                    }
                }
                // end of a group (sub-unit groups: every pass; wider ones: the group's last unit), or the last unit
                // of this wave's range (wave-uniform)
                const int q_end = unit * 4 + ((sub + 1) << sub_shift);  // 32-column blocks up to here
                if (real && (((q_end & ((1 << p.gq_shift) - 1)) == 0) || (uu == nu - 1 && sub == nsub - 1))) {
                    int grp = (q_end - 1) >> p.gq_shift;
                    grp = grp < p.n_groups ? grp : p.n_groups - 1;  // stream padding past K: x is zero there
                    const uint32_t* sz = szl + ((buf * R) * 16 + 4 * g) * p.n_groups + grp;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const uint32_t w = sz[(r * 16 + rr) * p.n_groups];
                            const float sc = __uint_as_float(w << 16), zp = __uint_as_float(w & 0xffff0000u);
                            accf[r][rr] += sc * (acc[r][rr] - (128.f + zp) * acc1[rr]);
                        }
                        acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            return;
        }
        if constexpr (FMT == MI355_W_Q4) {
            // sum_k x_k of the rounded operands, from the otherwise idle matrix pipe
#pragma unroll
            for (int d = 0; d < 4; ++d)
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ones), b[d], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if constexpr (FMT == MI355_W_Q4) {
                const u32x4 q = ring[j][r];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const uint32_t v = q[d];
                    u32x4 a;
                    a[0] = (v & cmask) | cmagic;
                    a[1] = ((v >> 4) & cmask) | cmagic;
                    a[2] = ((v >> 8) & cmask) | cmagic;
                    a[3] = ((v >> 12) & cmask) | cmagic;
                    acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a), b[d], acc[r], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ring[j][r * kPieces + d]), b[d], acc[r], 0,
                                                                     0, 0);
            }
        }
    };
    // tile done for this wave: publish the partial 16x16 tiles, combine, epilogue
    auto flush = [&]() {
        if constexpr (GRP && !MULTI) {
            if (p.gq_shift >= 2) {  // the window's groups lie side by side in the 16 columns: scale them, add them up
                apply_window();
                gbase = u0 >> (p.gq_shift - 2);
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) accf[r][rr] = group_sum(accf[r][rr], 16);
            }
        }
        f32x4* pp = (f32x4*)(part + (size_t)((buf * W + wave) * RS) * 1024) + lane;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if constexpr (GRP) {
                pp[r * 64] = accf[r];
                accf[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                pp[r * 64] = acc[r];
            }
            acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (FMT == MI355_W_Q4 && !GRP) pp[R * 64] = acc1;
        acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
        // GRP: the next tile's pairs (requested a tile ago) go to the other LDS block, which no wave reads any more:
        // its tile ended at the previous flush's barrier
        grp_park(tile + nb, buf ^ 1);
        if (tile == bid) MI355_STAMP(3);
        __syncthreads();
        if (e_owner || (!MULTI && threadIdx.x < 256))
            tile_epilogue<FMT, R, EPI, MULTI, GRP>(p, part, buf, W, tile, e_row, e_col, eo, rinv);
        if (tile == bid) MI355_STAMP(4);
        tile += nb;
        buf ^= 1;
        load_epi<FMT, R, EPI, GRP>(p, er, tile, e_row, e_col, e_owner, eo);
        grp_load(tile + nb);
    };

    // Every wave steps through nu_pad (a multiple of P) units per tile, so a tile can only end after slot P - 1:
    // ONE copy of the tile-end code instead of P, and the wait for the epilogue operands leaves the whole ring in
    // flight (with a tile end possible after any slot hipcc must assume the previous one was a single unit ago and
    // waits for all but one slot).  K = 4096 over 8 waves is exactly one ring turn per tile; where the split is
    // uneven (K = 11008: 10 or 11 units per wave, nu_pad = 12) the idle steps cost a few MFMAs on zeros.
    for (int t = 0; t < total; t += P) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            consume(j);
            ++uu;
            MI355_ISSUE(j);
        }
        if (uu == nu_pad) {
            uu = 0;
            flush();
        }
    }
    MI355_STAMP(5);
#undef MI355_ISSUE
#undef MI355_STAMP
}

// ------------------------------------------------------------------------------------ repack kernels
// Rows >= N and columns >= K of the padded stream are written as zero (nibble 0 / bf16 0 / int8 0): the
// staged activations are zero there too, so padding contributes exactly 0 to every accumulator.

__global__ void q4_repack_kernel(const uint8_t* q0, const uint8_t* q1, int64_t stride_n, int64_t stride_kb, int N,
                                 int K, int R, uint32_t* out, int64_t n_dwords) {
    const int units = (K + kUnitK - 1) / kUnitK;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_dwords;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i & 3);
        const int lane = (int)((i >> 2) & 63);
        int64_t rest = i >> 8;
        const int r = (int)(rest % R);
        rest /= R;
        const int u = (int)(rest % units);
        const int tile = (int)(rest / units);
        const int g = lane >> 4, row = lane & 15;
        const uint8_t* src;
        int n;
        if (q1 != nullptr) {
            src = r == 0 ? q0 : q1;
            n = tile * 16 + row;
        } else {
            src = q0;
            n = (tile * R + r) * 16 + row;
        }
        uint32_t o = 0;
        if (n < N) {
#pragma unroll
            for (int pnib = 0; pnib < 8; ++pnib) {
                const int j = 2 * (pnib & 3) + (pnib >> 2);
                const int k = kUnitK * u + 32 * g + 8 * d + j;
                if (k < K) {
                    const uint8_t byte = src[(int64_t)n * stride_n + (int64_t)(k >> 1) * stride_kb];
                    const uint32_t nib = (k & 1) ? (byte >> 4) : (byte & 0xF);
                    o |= nib << (4 * pnib);
                }
            }
        }
        out[i] = o;
    }
}

// one thread per 16-B piece of 8 bf16
__global__ void bf16_repack_kernel(const void* w0, const void* w1, int dtype, int N, int K, int R, u32x4* out,
                                   int64_t n_pieces) {
    const int units = (K + kUnitK - 1) / kUnitK;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pieces;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const int d = (int)((i >> 6) & 3);
        int64_t rest = i >> 8;
        const int r = (int)(rest % R);
        rest /= R;
        const int u = (int)(rest % units);
        const int tile = (int)(rest / units);
        const int g = lane >> 4, row = lane & 15;
        const void* src;
        int n;
        if (w1 != nullptr) {
            src = r == 0 ? w0 : w1;
            n = tile * 16 + row;
        } else {
            src = w0;
            n = (tile * R + r) * 16 + row;
        }
        const int k0 = kUnitK * u + 32 * g + 8 * d;
        u32x4 o = {0u, 0u, 0u, 0u};
        if (n < N) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ka = k0 + 2 * q, kb = ka + 1;
                const bf16_t lo = ka < K ? f32_to_bf16(ld_as_f32(src, (int64_t)n * K + ka, dtype)) : (bf16_t)0;
                const bf16_t hi = kb < K ? f32_to_bf16(ld_as_f32(src, (int64_t)n * K + kb, dtype)) : (bf16_t)0;
                o[q] = (uint32_t)lo | ((uint32_t)hi << 16);
            }
        }
        out[i] = o;
    }
}

// 8-bit ColBlock levels -> the stream of mi355_fused_step's weight_fmt 6: one thread per 16 B of lane (g, row) of piece e; byte b =
// column 128 u + 32 g + 16 e + 8 (b >> 3) + (0 4 1 5 2 6 3 7)[b & 7] (the octet order of the fp8 step's limb planes)
__global__ void u8_repack_kernel(const uint8_t* q0, const uint8_t* q1, int64_t stride_n, int64_t stride_k, int N, int K, int R, u32x4* out,
                                 int64_t n_pieces) {
    const int units = K / kUnitK;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pieces; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const int e = (int)((i >> 6) & 1);
        int64_t rest = i >> 7;
        const int r = (int)(rest % R);
        rest /= R;
        const int u = (int)(rest % units);
        const int tile = (int)(rest / units);
        const int g = lane >> 4, row = lane & 15;
        const uint8_t* src = (q1 != nullptr && r == 1) ? q1 : q0;
        const int n = q1 != nullptr ? tile * 16 + row : (tile * R + r) * 16 + row;
        const int k0 = kUnitK * u + 32 * g + 16 * e;
        u32x4 o = {0u, 0u, 0u, 0u};
        if (n < N) {
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const int j = b & 7;
                const int k = k0 + 8 * (b >> 3) + (j >> 1) + 4 * (j & 1);  // (0 4 1 5 2 6 3 7)[j]
                o[b >> 2] |= (uint32_t)src[(int64_t)n * stride_n + (int64_t)k * stride_k] << (8 * (b & 3));
            }
        }
        out[i] = o;
    }
}

// one thread per 16-B piece of 16 int8
__global__ void i8_repack_kernel(const int8_t* c0, const int8_t* c1, int N, int K, int R, u32x4* out,
                                 int64_t n_pieces) {
    const int units = (K + kUnitK - 1) / kUnitK;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pieces;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const int e = (int)((i >> 6) & 1);
        int64_t rest = i >> 7;
        const int r = (int)(rest % R);
        rest /= R;
        const int u = (int)(rest % units);
        const int tile = (int)(rest / units);
        const int g = lane >> 4, row = lane & 15;
        const int8_t* src;
        int n;
        if (c1 != nullptr) {
            src = r == 0 ? c0 : c1;
            n = tile * 16 + row;
        } else {
            src = c0;
            n = (tile * R + r) * 16 + row;
        }
        const int k0 = kUnitK * u + 64 * e + 16 * g;
        u32x4 o = {0u, 0u, 0u, 0u};
        if (n < N) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t v = 0;
#pragma unroll
                for (int bidx = 0; bidx < 4; ++bidx) {
                    const int k = k0 + 4 * q + bidx;
                    if (k < K) v |= (uint32_t)(uint8_t)src[(int64_t)n * K + k] << (8 * bidx);
                }
                o[q] = v;
            }
        }
        out[i] = o;
    }
}


template <int FMT, int R, int P, int EPI, int VMODE, bool MULTI, int GRP>
int launch_gemv_m(const GemvParams& p, int grid, int waves, size_t lds, hipStream_t stream) {
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute((const void*)gemv_kernel<FMT, R, P, EPI, VMODE, MULTI, GRP>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    });
    if (attr_err != hipSuccess) {
        mi355_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(attr_err));
        return (int)attr_err;
    }
    if (waves * 64 > kMaxThreads<FMT, P, GRP>) {  // the host split (u_q / u_r) was computed for `waves`
        mi355_set_error("internal: %d waves exceed the variant's workgroup size", waves);
        return MI355_E_ARG;
    }
    if (t_time_start != nullptr) {
        // measurement hook (mi355_debug_time_next_launch): the events receive the dispatch's own begin / end
        // timestamps, i.e. the duration rocprofv3 reports for this launch
        hipEvent_t e0 = t_time_start, e1 = t_time_stop;
        t_time_start = t_time_stop = nullptr;
        hipExtLaunchKernelGGL((gemv_kernel<FMT, R, P, EPI, VMODE, MULTI, GRP>), dim3(grid), dim3(waves * 64), (uint32_t)lds,
                              stream, e0, e1, 0, p);
    } else {
        hipLaunchKernelGGL((gemv_kernel<FMT, R, P, EPI, VMODE, MULTI, GRP>), dim3(grid), dim3(waves * 64), lds, stream, p);
    }
    MI355_LAUNCH_CHECK();
    return 0;
}

template <int FMT, int R, int P, int EPI, int VMODE, int GRP>
int launch_gemv_v(const GemvParams& p, int grid, int waves, size_t lds, hipStream_t stream) {
    if constexpr (VMODE == 3) {  // split-attention input exists for M = 1 only (host check)
        return launch_gemv_m<FMT, R, P, EPI, VMODE, false, GRP>(p, grid, waves, lds, stream);
    } else {
        return p.M > 1 ? launch_gemv_m<FMT, R, P, EPI, VMODE, true, GRP>(p, grid, waves, lds, stream)
                       : launch_gemv_m<FMT, R, P, EPI, VMODE, false, GRP>(p, grid, waves, lds, stream);
    }
}

template <int FMT, int R, int P, int EPI, int GRP>
int launch_gemv_epi(const GemvParams& p, int grid, int waves, size_t lds, hipStream_t stream) {
    switch (p.vec_mode) {
        case 1:
            if constexpr (EPI != MI355_EPI_ACCUM) return launch_gemv_v<FMT, R, P, EPI, 1, GRP>(p, grid, waves, lds, stream);
            break;
        case 4:
            if constexpr (EPI != MI355_EPI_ACCUM) return launch_gemv_v<FMT, R, P, EPI, 4, GRP>(p, grid, waves, lds, stream);
            break;
        case 2:
            if constexpr (EPI != MI355_EPI_SWIGLU) return launch_gemv_v<FMT, R, P, EPI, 2, GRP>(p, grid, waves, lds, stream);
            break;
        case 3:
            if constexpr (EPI != MI355_EPI_SWIGLU) return launch_gemv_v<FMT, R, P, EPI, 3, GRP>(p, grid, waves, lds, stream);
            break;
        default: break;
    }
    if (p.vec_mode == 3) {
        mi355_set_error("split-attention prologue is not available for this epilogue");
        return MI355_E_ARG;
    }
    return launch_gemv_v<FMT, R, P, EPI, 0, GRP>(p, grid, waves, lds, stream);
}

template <int FMT, int R, int P, int GRP>
int launch_gemv(const GemvParams& p, int grid, int waves, size_t lds, hipStream_t stream) {
    switch (p.epi) {
        case MI355_EPI_STORE: return launch_gemv_epi<FMT, R, P, MI355_EPI_STORE, GRP>(p, grid, waves, lds, stream);
        case MI355_EPI_ACCUM: return launch_gemv_epi<FMT, R, P, MI355_EPI_ACCUM, GRP>(p, grid, waves, lds, stream);
        default:
            if constexpr (R == 2) return launch_gemv_epi<FMT, R, P, MI355_EPI_SWIGLU, GRP>(p, grid, waves, lds, stream);
            mi355_set_error("SwiGLU epilogue needs R == 2");
            return MI355_E_ARG;
    }
}

template <int FMT, int R>
int dispatch_p(const GemvParams& p, int /*prefetch*/, int grid, int waves, size_t lds, hipStream_t s) {
    // One ring depth per format: 4 units (Q4) / 2 units (bf16) in flight per wave.  Rings twice as deep were
    // slower on every 7B shape (the CU's memory pipeline accepts requests in order, so a longer prefill only
    // delays the activation loads; scripts/sweep_gemv.py) and doubled the code size; `prefetch` is accepted and
    // ignored.
    constexpr int PA = FMT == MI355_W_Q4 ? 4 : 2;
    if constexpr (FMT == MI355_W_Q4) {
        if (p.n_groups > 0)
            return 16 * p.n_groups <= waves * 64                  ? launch_gemv<FMT, R, PA, 1>(p, grid, waves, lds, s)
                   : 16 * p.n_groups <= kGrpLoadsMid * waves * 64 ? launch_gemv<FMT, R, PA, kGrpLoadsMid>(p, grid, waves, lds, s)
                                                                   : launch_gemv<FMT, R, PA, kGrpLoadsMax>(p, grid, waves, lds, s);
    }
    return launch_gemv<FMT, R, PA, false>(p, grid, waves, lds, s);
}

int g_num_cus = 0;

}  // namespace

// measurement hook shared by the launchers of this library (mi355_debug_time_next_launch; declared in common.h)
thread_local hipEvent_t t_time_start = nullptr, t_time_stop = nullptr;

extern "C" int mi355_num_cus(void) {
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        g_num_cus = prop.multiProcessorCount;
    }
    return g_num_cus;
}

static inline int64_t padded_rows(int N, int R, bool pair) {
    const int rows_per_tile = pair ? 16 : 16 * R;
    const int64_t tiles = ((int64_t)N + rows_per_tile - 1) / rows_per_tile;
    return tiles * (pair ? 32 : rows_per_tile);  // a pair stream carries 2 x 16 rows per tile
}
static inline int64_t padded_cols(int K) { return ((int64_t)K + kUnitK - 1) / kUnitK * kUnitK; }

extern "C" size_t mi355_packed_bytes(int fmt, int N, int K, int R, int pair) {
    if (N <= 0 || K <= 0 || (R != 1 && R != 2) || (pair && R != 2)) return 0;
    const int64_t elems = padded_rows(N, R, pair != 0) * padded_cols(K);
    switch (fmt) {
        case MI355_W_Q4: return (size_t)(elems / 2);
        case MI355_W_BF16: return (size_t)(elems * 2);
        case MI355_W_I8: return (size_t)elems;
        default: return 0;
    }
}

static int check_repack(int N, int K, int R, bool pair) {
    MI355_CHECK_ARG(R == 1 || R == 2, MI355_E_ARG, "repack: R must be 1 or 2 (got %d)", R);
    MI355_CHECK_ARG(!pair || R == 2, MI355_E_ARG, "repack: interleaving two matrices needs R == 2");
    MI355_CHECK_ARG(N > 0 && K > 0, MI355_E_SHAPE, "repack: N=%d K=%d must be positive", N, K);
    return 0;
}

static inline int repack_grid(int64_t n) {
    const int64_t b = (n + 255) / 256;
    return (int)(b > 262144 ? 262144 : b);
}

extern "C" int mi355_q4_repack(const uint8_t* q0, const uint8_t* q1, int64_t stride_n, int64_t stride_kb, int N, int K,
                               int R, uint8_t* out, mi355_stream_t stream) {
    MI355_CHECK_ARG(q0 && out, MI355_E_ARG, "q4_repack: null pointer");
    MI355_CHECK_ARG(K % 2 == 0, MI355_E_SHAPE, "q4_repack: K=%d must be even (two entries per byte)", K);
    if (int rc = check_repack(N, K, R, q1 != nullptr)) return rc;
    const int64_t n_dwords = padded_rows(N, R, q1 != nullptr) * padded_cols(K) / 8;
    hipLaunchKernelGGL(q4_repack_kernel, dim3(repack_grid(n_dwords)), dim3(256), 0, (hipStream_t)stream, q0, q1,
                       stride_n, stride_kb, N, K, R, (uint32_t*)out, n_dwords);
    MI355_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi355_bf16_repack(const void* w0, const void* w1, int dtype, int N, int K, int R, void* out,
                                 mi355_stream_t stream) {
    MI355_CHECK_ARG(w0 && out, MI355_E_ARG, "bf16_repack: null pointer");
    MI355_CHECK_ARG(dtype == MI355_F32 || dtype == MI355_BF16, MI355_E_DTYPE, "bf16_repack: dtype must be f32 or bf16");
    if (int rc = check_repack(N, K, R, w1 != nullptr)) return rc;
    const int64_t n_pieces = padded_rows(N, R, w1 != nullptr) * padded_cols(K) / 8;
    hipLaunchKernelGGL(bf16_repack_kernel, dim3(repack_grid(n_pieces)), dim3(256), 0, (hipStream_t)stream, w0, w1,
                       dtype, N, K, R, (u32x4*)out, n_pieces);
    MI355_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi355_i8_repack(const int8_t* cb0, const int8_t* cb1, int N, int K, int R, int8_t* out,
                               mi355_stream_t stream) {
    MI355_CHECK_ARG(cb0 && out, MI355_E_ARG, "i8_repack: null pointer");
    if (int rc = check_repack(N, K, R, cb1 != nullptr)) return rc;
    const int64_t n_pieces = padded_rows(N, R, cb1 != nullptr) * padded_cols(K) / 16;
    hipLaunchKernelGGL(i8_repack_kernel, dim3(repack_grid(n_pieces)), dim3(256), 0, (hipStream_t)stream, cb0, cb1, N,
                       K, R, (u32x4*)out, n_pieces);
    MI355_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi355_u8_repack(const uint8_t* q0, const uint8_t* q1, int64_t stride_n, int64_t stride_k, int N, int K, int R, uint8_t* out,
                               mi355_stream_t stream) {
    MI355_CHECK_ARG(q0 && out, MI355_E_ARG, "u8_repack: null pointer");
    MI355_CHECK_ARG(N > 0 && K > 0 && N % 16 == 0 && K % kUnitK == 0 && (R == 1 || (R == 2 && q1 != nullptr)) && (q1 == nullptr || R == 2),
                    MI355_E_SHAPE, "u8_repack: N %d must be a multiple of 16, K %d of 128, R %d in {1, 2} (2: a pair q0 / q1)", N, K, R);
    const int64_t n_pieces = (int64_t)N * (q1 != nullptr ? 2 : 1) * K / 16;
    hipLaunchKernelGGL(u8_repack_kernel, dim3(repack_grid(n_pieces)), dim3(256), 0, (hipStream_t)stream, q0, q1, stride_n, stride_k, N, K, R,
                       (u32x4*)out, n_pieces);
    MI355_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi355_linear_fast(const mi355_linear_args* a, mi355_stream_t stream) {
    MI355_CHECK_ARG(a != nullptr, MI355_E_ARG, "linear_fast: null args");
    MI355_CHECK_ARG(a->fmt == MI355_W_Q4 || a->fmt == MI355_W_BF16, MI355_E_ARG,
                    "linear_fast: fmt %d not handled here (int8 goes through mi355_linear_int8)", a->fmt);
    MI355_CHECK_ARG(a->R == 1 || a->R == 2, MI355_E_ARG, "linear_fast: R must be 1 or 2");
    MI355_CHECK_ARG(a->w && (a->x || a->attn_partials) && a->y, MI355_E_ARG, "linear_fast: null w/x/y");
    if (a->attn_partials != nullptr) {
        MI355_CHECK_ARG(a->M == 1 && a->attn_splits >= 1 && a->attn_heads >= 1 && a->attn_hs % 8 == 0 && a->norm_scale == nullptr &&
                            a->K == a->attn_heads * a->attn_hs && a->epi != MI355_EPI_SWIGLU,
                        MI355_E_ARG, "linear_fast: bad split-attention activation spec");
    }
    MI355_CHECK_ARG(a->M >= 1 && a->M <= kMaxM, MI355_E_SHAPE, "linear_fast: M=%d outside 1..%d", a->M, kMaxM);
    MI355_CHECK_ARG(a->K > 0 && a->N > 0, MI355_E_SHAPE, "linear_fast: N=%d K=%d must be positive", a->N, a->K);
    const bool swiglu = a->epi == MI355_EPI_SWIGLU;
    MI355_CHECK_ARG(a->epi >= MI355_EPI_STORE && a->epi <= MI355_EPI_SWIGLU, MI355_E_ARG, "linear_fast: bad epi");
    MI355_CHECK_ARG(!swiglu || a->R == 2, MI355_E_ARG, "linear_fast: SwiGLU epilogue needs the interleaved R=2 stream");
    const int rows_per_tile = swiglu ? 16 : 16 * a->R;
    if (a->fmt == MI355_W_Q4) {
        MI355_CHECK_ARG(a->scales && a->zeros, MI355_E_ARG, "linear_fast: Q4 needs scales and zeros");
        MI355_CHECK_ARG(!swiglu || (a->scales2 && a->zeros2), MI355_E_ARG, "linear_fast: SwiGLU Q4 needs scales2/zeros2");
    }
    auto two = [](int d) { return d == MI355_F32 || d == MI355_BF16; };
    MI355_CHECK_ARG(two(a->x_dtype) && two(a->y_dtype), MI355_E_DTYPE, "linear_fast: x/y dtype must be f32 or bf16");
    MI355_CHECK_ARG(a->norm_scale == nullptr || two(a->norm_dtype), MI355_E_DTYPE,
                    "linear_fast: norm scale dtype must be f32 or bf16");
    MI355_CHECK_ARG(a->fmt != MI355_W_Q4 || two(a->sz_dtype), MI355_E_DTYPE,
                    "linear_fast: scales/zeros dtype must be f32 or bf16");
    MI355_CHECK_ARG(a->ldx >= a->K || a->M == 1 || a->attn_partials, MI355_E_SHAPE, "linear_fast: ldx < K");

    GemvParams p;
    p.w = (const uint8_t*)a->w;
    p.x = a->x;
    p.norm_scale = a->norm_scale;
    p.scales = a->scales;
    p.zeros = a->zeros;
    p.scales2 = a->scales2;
    p.zeros2 = a->zeros2;
    p.bias = swiglu ? nullptr : a->bias;
    p.y = a->y;
    p.ldx = a->ldx;
    p.ldy = a->ldy;
    p.N = a->N;
    p.K = a->K;
    p.M = a->M;
    p.n_tiles = (a->N + rows_per_tile - 1) / rows_per_tile;
    p.units = (a->K + kUnitK - 1) / kUnitK;
    p.x_dtype = a->x_dtype;
    p.norm_dtype = a->norm_dtype;
    p.sz_dtype = a->sz_dtype;
    p.y_dtype = a->y_dtype;
    p.epi = a->epi;
    p.xs_stride = (p.units + 1) * kUnitK * 2 + 16;  // + one all-zero unit (idle ring steps read it)
    p.eps = a->eps;
    int waves = a->waves > 0 ? a->waves : 8;
    if (waves > 16) waves = 16;
    if ((a->fmt != MI355_W_Q4 || (a->group_cols > 0 && a->group_cols < a->K)) && waves > 8) waves = 8;  // see kMaxThreads
    if (waves < 4) waves = 4;  // the combine step needs 256 owner threads
    {
        const int esz = a->x_dtype == MI355_F32 ? 4 : 2;
        const bool aligned = a->x != nullptr && ((uintptr_t)a->x % 16 == 0) && ((a->ldx * esz) % 16 == 0 || a->M == 1) && a->K % 8 == 0;
        p.vec_mode = 0;
        const int nvec = a->K / 8, nthr = waves * 64;
        if (aligned && a->norm_scale != nullptr && a->x_dtype == MI355_F32 && a->norm_dtype == MI355_BF16 &&
            (uintptr_t)a->norm_scale % 16 == 0 && nvec <= 4 * nthr)
            p.vec_mode = nvec <= 2 * nthr ? 1 : 4;
        else if (aligned && a->norm_scale == nullptr && a->x_dtype == MI355_BF16 && nvec <= 6 * nthr)
            p.vec_mode = 2;
        if (a->attn_partials != nullptr) {
            MI355_CHECK_ARG(a->attn_splits <= 4 && nvec <= nthr && ((uintptr_t)a->attn_partials % 16) == 0,
                            MI355_E_SHAPE, "linear_fast: split-attention prologue handles <= 4 splits and K <= %d",
                            8 * nthr);
            p.vec_mode = 3;
        }
        p.attn_part = a->attn_partials;
        p.attn_splits = a->attn_splits;
        p.attn_heads = a->attn_heads;
        p.attn_hs = a->attn_hs;
        p.dbg = (unsigned long long*)a->debug_stamps;
    }
    {
        const size_t wb = mi355_packed_bytes(a->fmt, a->N, a->K, a->R, swiglu ? 1 : 0);
        MI355_CHECK_ARG(wb > 0 && wb < 0xFFFFFFF0ull, MI355_E_SHAPE, "linear_fast: weight stream of %zu B exceeds 4 GiB", wb);
        p.w_bytes = (unsigned)wb;
    }

    // grouped scales: group_cols = input columns per (scale, zero) pair; 0 or >= K: one pair per output row
    p.n_groups = 0;
    p.gq_shift = 0;
    p.inv_ng = 0.f;
    const int group_cols = a->group_cols;
    if (a->fmt == MI355_W_Q4 && group_cols > 0 && group_cols < a->K) {
        int sh = 0;
        while ((32 << sh) < group_cols) ++sh;
        MI355_CHECK_ARG((32 << sh) == group_cols, MI355_E_SHAPE,
                        "linear_fast: group size %d is not 32 * 2^n (use the generic kernel)", group_cols);
        MI355_CHECK_ARG(a->sz_dtype == MI355_BF16, MI355_E_DTYPE, "linear_fast: grouped scales / zeros must be bf16");
        p.n_groups = (a->K + group_cols - 1) / group_cols;
        p.gq_shift = sh;
        p.inv_ng = 1.0f / (float)p.n_groups;
        MI355_CHECK_ARG(16 * p.n_groups <= kGrpLoadsMax * waves * 64, MI355_E_SHAPE,
                        "linear_fast: %d groups per row exceed what %d waves prefetch per tile (%d)", p.n_groups, waves,
                        kGrpLoadsMax * waves * 4);
        MI355_CHECK_ARG((int64_t)a->N * p.n_groups * 2 < 0x7FFFFFF0ll, MI355_E_SHAPE, "linear_fast: scale table too large");
    }
    const int RS = a->R + ((a->fmt == MI355_W_Q4 && p.n_groups == 0) ? 1 : 0);
    const size_t lds = kLdsHeader + (size_t)2 * waves * RS * 1024 + (((size_t)a->M * p.xs_stride + 15) & ~(size_t)15) +
                       (size_t)2 * a->R * 16 * p.n_groups * 4;
    MI355_CHECK_ARG(lds <= (size_t)kMaxLds, MI355_E_SHAPE,
                    "linear_fast: M=%d x K=%d activations do not fit LDS (%zu B > %d B); chunk M", a->M, a->K, lds,
                    kMaxLds);
    int grid = a->grid;
    if (grid <= 0) {
        const int cus = mi355_num_cus() > 0 ? mi355_num_cus() : 256;
        grid = cus * 2;
    }
    if (grid > p.n_tiles) grid = p.n_tiles;
    p.u_q = p.units / waves;
    p.u_r = p.units % waves;
    {
        const int P = a->fmt == MI355_W_Q4 ? 4 : 2, nu_max = p.u_q + (p.u_r ? 1 : 0);
        p.nu_pad = (nu_max + P - 1) / P * P;
    }
    p.t_q = p.n_tiles / grid;
    p.t_r = p.n_tiles % grid;
    hipStream_t s = (hipStream_t)stream;
    if (a->fmt == MI355_W_Q4) {
        return a->R == 1 ? dispatch_p<MI355_W_Q4, 1>(p, a->prefetch, grid, waves, lds, s)
                         : dispatch_p<MI355_W_Q4, 2>(p, a->prefetch, grid, waves, lds, s);
    }
    return a->R == 1 ? dispatch_p<MI355_W_BF16, 1>(p, a->prefetch, grid, waves, lds, s)
                     : dispatch_p<MI355_W_BF16, 2>(p, a->prefetch, grid, waves, lds, s);
}

extern "C" int mi355_debug_time_next_launch(void* start_event, void* stop_event) {
    MI355_CHECK_ARG((start_event == nullptr) == (stop_event == nullptr), MI355_E_ARG,
                    "debug_time_next_launch: pass both events or neither");
    t_time_start = (hipEvent_t)start_event;
    t_time_stop = (hipEvent_t)stop_event;
    return 0;
}

extern "C" int mi355_linear_max_rows(int fmt, int K, int R, int waves) {
    if (K <= 0 || R < 1 || R > 2) return 0;
    int w = waves > 0 ? waves : 8;
    if (w > 16) w = 16;
    if (w < 4) w = 4;
    const int64_t kp = (int64_t)((K + kUnitK - 1) / kUnitK) * kUnitK;
    int64_t fixed, per_row;
    if (fmt == MI355_W_I8) {
        if (w > 8) w = 8;
        fixed = 512 + (int64_t)2 * w * R * 1024 + kp * 2 + kp / 8 + 16;  // int8.hip
        per_row = (kp + 16) + kp * 2;
    } else if (fmt == MI355_W_Q4 || fmt == MI355_W_BF16) {
        const int RS = R + (fmt == MI355_W_Q4 ? 1 : 0);
        fixed = kLdsHeader + (int64_t)2 * w * RS * 1024;
        per_row = (kp + kUnitK) * 2 + 16;  // + the all-zero unit
    } else {
        return 0;
    }
    const int64_t rows = (kMaxLds - fixed) / per_row;
    return rows < 0 ? 0 : (rows > kMaxM ? kMaxM : (int)rows);
}

extern "C" int mi355_linear_fast_batch(const mi355_linear_args* a, int count, mi355_stream_t stream) {
    MI355_CHECK_ARG(a != nullptr && count >= 0, MI355_E_ARG, "linear_fast_batch: bad argument");
    for (int i = 0; i < count; ++i)
        if (int rc = mi355_linear_fast(a + i, stream)) return rc;
    return 0;
}

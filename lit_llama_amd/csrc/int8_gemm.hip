// Wide (M >= 32 tokens) LLM.int8 linear for gfx950: prompt prefill / no-cache evaluation of a `--quantize llm.int8` model.
//
// Replaces, for B * T >= 32 rows, bitsandbytes' MatMul8bitLt forward as Linear8bitLt reaches it
// (/root/reference lit_llama/quantization.py:38-77; the algorithm is restated in oracle/oracle.py::llm_int8_linear — PARITY
// UNPINNED: bitsandbytes is not vendored).  The skinny kernel (int8.hip) feeds such inputs in chunks
// of <= 16 rows and determines the outlier COLUMN set per chunk; this path determines it over all rows of the call, as the
// algorithm does, and runs the int8 product on v_mfma_i32_16x16x64_i8 over 128-token blocks:
//   1. i8_stage_kernel   (one workgroup per row) — xh = f16(norm_scale * x * 1/rms) (or f16(x)), the row's absmax over its
//                        sub-threshold entries, and the column mask |xh| >= threshold OR-ed over all rows (atomicOr: order
//                        does not matter);
//   2. i8_quant_kernel   — CA = rint(xh * 127 / absmax) with the masked columns zeroed, int8; workgroup 0 also writes the
//                        ascending list of outlier columns;
//   3. i8_gemm_kernel    — the int8 weight stream of int8.hip ([tile][unit][r][piece e][lane][16 x int8], one piece = one
//                        MFMA A operand) against CA staged through LDS (128 tokens x 128 k, XOR-swizzled), two row tiles per
//                        wave, 8 waves; epilogue with the arithmetic of int8.hip, rounding for rounding:
//                            d = f16((acc * 6.200012e-05 * absmax[m]) * SCB[n]);   o = sum over outlier columns k (ascending) of
//                            xh[m, k] * f16(CB[n, k] * SCB[n] / 127);   y = f16(d + f16(o))   (the last step only with outliers)
//                        the outlier operands of the block (128 tokens x n_out, 256 rows x n_out) are staged in LDS first.
#include "common.h"

namespace {

constexpr int kBM = 128;      // tokens per block
constexpr int kTPW = 2;       // 16-row tile slots per wave
constexpr int kWaves = 8;
constexpr int kSlots = kWaves * kTPW;
constexpr int kThreads = 64 * kWaves;
constexpr int kXTile = kBM * 128;        // bytes of one activation tile (int8)
constexpr int kOutChunk = 64;            // outlier columns staged per pass of the epilogue
constexpr int kLdsMain = 2 * kXTile;
constexpr int kSoLd = kOutChunk + 8;     // row pitch of the staged outlier weights (halves): 144 B, so that 16 rows written side by side fall into 16 banks
constexpr int kLdsEpi = (kBM * kOutChunk + kSlots * 16 * kSoLd) * 2;  // f16 [128 tokens][kOutChunk] + [256 rows][kSoLd]
constexpr int kLds = kLdsMain > kLdsEpi ? kLdsMain : kLdsEpi;

__device__ __forceinline__ float f16r(float v) { return f16_to_f32(f32_to_f16(v)); }
// the same for a value that must be an f32 PRODUCT rounded to f32 first (MatMul8bitLt's dequantisation multiplies in f32 and converts): hipcc
// otherwise may contract the last multiply and the conversion into v_fma_mixlo_f16 — one rounding instead of two, a bf16 ulp in ~1e-5 of the
// outputs against the oracle (seen when the outlier side product moved to MFMA and the register allocation around it changed)
__device__ __forceinline__ float f16r_of_f32(float v) {
    asm volatile("" : "+v"(v));
    return f16r(v);
}
// 16-B chunk swizzle of a 128-B activation row in LDS: rows alternate bank halves (128-B pitch), pairs of rows rotate the
// 8 chunks — the 16 lanes a ds_read_b128 serves together (tokens {0-3, 12-15} with one k-group, {4-11} with the next)
// land in 16 different bank groups
__device__ __forceinline__ int swz8(int tok) { return (tok >> 1) & 7; }

struct I8GemmParams {
    const uint8_t* w;
    unsigned w_bytes;
    const int8_t* ca;      // [M, Kp]
    const f16_t* xh;       // [M, Kp]
    const float* sca;      // [M] row absmax
    const int* olist;      // [0] = n_out, [1 ..] ascending outlier columns
    const float* scb;
    const float* scb2;
    void* y;
    int64_t ldy;
    int M, N, K, Kp, units, n_tiles, n_blocks, per_xcd, total_blocks;
    int y_dtype;
    // split-K for short prompts (a 128-token prompt is ONE token block: 16 workgroups for N = 4096): MODE 1 launches
    // total_blocks x ksplit workgroups that store their int32 partial tiles to part[ks][M][ldp]; MODE 2 (total_blocks
    // workgroups) sums them — integer sums: exact in any order — and runs the epilogue
    int ksplit;
    int64_t ldp;
    int* part;
};
constexpr size_t kSplitBudget = (size_t)32 << 20;
constexpr int kMaxSplit = 8;

// ---- 1. rows: f16 operand, absmax over the sub-threshold entries, outlier column mask
__global__ __launch_bounds__(256) void i8_stage_kernel(const void* x, int x_dtype, int64_t ldx, const void* norm_scale,
                                                       int norm_dtype, float eps, float threshold, int K, int Kp, f16_t* xh,
                                                       float* sca, unsigned* mask) {
    __shared__ float red[32];
    const int m = blockIdx.x;
    float rinv = 1.f;
    if (norm_scale != nullptr) {
        float ss = 0.f;
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            const float v = ld_as_f32(x, (int64_t)m * ldx + k, x_dtype);
            ss += v * v;
        }
        rinv = rsqrtf(block_sum(ss, red) / (float)K + eps);
    }
    float amax = 0.f;
    for (int k = threadIdx.x; k < Kp; k += blockDim.x) {
        f16_t h = 0;
        if (k < K) {
            float v = ld_as_f32(x, (int64_t)m * ldx + k, x_dtype);
            if (norm_scale != nullptr) v = ld_as_f32(norm_scale, k, norm_dtype) * (v * rinv);
            h = f32_to_f16(v);
            const float a = fabsf(f16_to_f32(h));
            if (threshold > 0.f && a >= threshold)
                atomicOr(mask + (k >> 5), 1u << (k & 31));
            else
                amax = fmaxf(amax, a);
        }
        xh[(int64_t)m * Kp + k] = h;
    }
    amax = block_max(amax, red);
    if (threadIdx.x == 0) sca[m] = amax;
}

// ---- 2. rows: int8 operand; workgroup 0: the outlier list
__global__ __launch_bounds__(256) void i8_quant_kernel(const f16_t* xh, const float* sca, const unsigned* mask, int Kp, int8_t* ca,
                                                       int* olist) {
    const int m = blockIdx.x;
    const float a = sca[m];
    const float inv = a > 0.f ? __fdiv_rn(127.0f, a) : 0.f;  // IEEE division: parity with the oracle
    for (int k4 = threadIdx.x; k4 < Kp / 4; k4 += blockDim.x) {
        const int k = 4 * k4;
        const unsigned mw = mask[k >> 5] >> (k & 31);
        const u32x2 hv = *(const u32x2*)(xh + (int64_t)m * Kp + k);
        uint32_t q4 = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f16_t h = (f16_t)((j & 1) ? (hv[j >> 1] >> 16) : (hv[j >> 1] & 0xffffu));
            const float q = ((mw >> j) & 1u) ? 0.f : rintf(f16_to_f32(h) * inv);
            q4 |= ((uint32_t)(int)q & 0xffu) << (8 * j);
        }
        *(uint32_t*)(ca + (int64_t)m * Kp + k) = q4;
    }
    if (m == 0 && threadIdx.x == 0) {
        int n = 0;
        for (int w = 0; w < Kp / 32; ++w) {
            unsigned bits = mask[w];
            while (bits != 0u) {
                const int b = __builtin_ctz(bits);
                olist[1 + n++] = w * 32 + b;
                bits &= bits - 1u;
            }
        }
        olist[0] = n;
    }
}

// ---- 3. the product
// MODE 0: product + epilogue; 1: product of one K-slice -> partial tiles; 2: sum of the partial tiles + epilogue
template <int EPI, bool PAIR, int MODE>
__global__ __launch_bounds__(kThreads) void i8_gemm_kernel(const I8GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3;
    const int L2 = xcd * p.per_xcd + j0;
    // MODE 1: a workgroup per (block, K-slice); MODE 2: a workgroup per (block, 16-token tile) — 8 x the workgroups of the
    // product launch, so that summing the slices and the epilogue are not left to a dozen workgroups
    const int nsplit = MODE == 1 ? p.ksplit : MODE == 2 ? 8 : 1;
    if (j0 >= p.per_xcd || L2 >= p.total_blocks * nsplit) return;
    const int L = L2 / nsplit, ks = L2 - L * nsplit;
    // MODE 2: workgroup ks of a block finishes the rows of slot group ks (the rows wave ks owns in the product launch) for all 128
    // tokens, wave w its token tile w — so the outlier columns' weights (the expensive side of the staging: byte gathers and an IEEE
    // division each) are staged once per block, not once per token tile
    const int tt_lo = MODE == 2 ? wave : 0, tt_hi = MODE == 2 ? wave + 1 : 8;  // token tiles this wave finishes
    const int wslot = MODE == 2 ? ks : wave;                                   // slot group of this wave's row tiles
    const int mb = L / p.n_blocks, nb = L - mb * p.n_blocks;
    const int m0 = mb * kBM;
    const int u_lo = MODE == 1 ? (int)((int64_t)ks * p.units / p.ksplit) : 0;
    const int u_hi = MODE == 1 ? (int)((int64_t)(ks + 1) * p.units / p.ksplit) : p.units;
    constexpr int R = PAIR ? 2 : 1;

    int tile[kTPW], rr[kTPW];
#pragma unroll
    for (int t = 0; t < kTPW; ++t) {
        if (PAIR) {
            tile[t] = nb * (kSlots / 2) + (kTPW / 2) * wslot + (t >> 1);  // pair tile: 16 rows of c_fc1 (r = 0) and c_fc2 (r = 1)
            rr[t] = t & 1;
        } else {
            tile[t] = nb * kSlots + kTPW * wslot + t;
            rr[t] = 0;
        }
    }
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.ca, 0, (int)((int64_t)p.M * p.Kp), 0x00020000);
    const unsigned lane_off = lane * 16;
    auto wload = [&](int t, int u, u32x4 (&dst)[2]) {
        const bool ok = tile[t] < p.n_tiles && u < u_hi;
        const unsigned off = (unsigned)((tile[t] * p.units + u) * R + rr[t]) * 2048u;
#pragma unroll
        for (int e = 0; e < 2; ++e)
            dst[e] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ok ? rw : rw0, lane_off, ok ? off + e * 1024u : 0u, 0));
    };
    // activation tile of unit u: 1024 chunks of 16 B, two per thread; chunk = (token, 16-B column)
    u32x4 stage[2];
    auto xload = [&](int u) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = i * kThreads + threadIdx.x;
            const int tok = ch >> 3, col = ch & 7;
            const unsigned off = (unsigned)((int64_t)(m0 + tok) * p.Kp) + (unsigned)u * 128u + (unsigned)col * 16u;
            stage[i] = __builtin_bit_cast(
                u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ((m0 + tok) < p.M && u < u_hi) ? off : 0xFFFFFFF0u, 0, 0));
        }
    };
    auto xstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = i * kThreads + threadIdx.x;
            const int tok = ch >> 3, col = ch & 7;
            *(u32x4*)(smem + buf * kXTile + tok * 128 + ((col ^ swz8(tok)) << 4)) = stage[i];
        }
    };

    i32x4 acc[kTPW][8];
#pragma unroll
    for (int t = 0; t < kTPW; ++t)
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) acc[t][tt] = i32x4{0, 0, 0, 0};

    if constexpr (MODE != 2) {
        u32x4 wcur[kTPW][2], wnext[kTPW][2];
#pragma unroll
        for (int t = 0; t < kTPW; ++t) wload(t, u_lo, wcur[t]);
        xload(u_lo);
        xstore(u_lo & 1);
        __syncthreads();
        for (int u = u_lo; u < u_hi; ++u) {
            const int buf = u & 1;
            xload(u + 1);  // unconditional: past the last unit the offsets are out of the descriptor (zeros, no traffic)
#pragma unroll
            for (int t = 0; t < kTPW; ++t) wload(t, u + 1, wnext[t]);
            const char* xs = smem + buf * kXTile;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
#pragma unroll
                for (int tt = 0; tt < 8; ++tt) {
                    const int tok = tt * 16 + c;
                    const i32x4 b = *(const i32x4*)(xs + tok * 128 + (((4 * e + g) ^ swz8(tok)) << 4));
#pragma unroll
                    for (int t = 0; t < kTPW; ++t)
                        acc[t][tt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, wcur[t][e]), b, acc[t][tt], 0, 0, 0);
                }
            }
            xstore(buf ^ 1);
#pragma unroll
            for (int t = 0; t < kTPW; ++t) {
                wcur[t][0] = wnext[t][0];
                wcur[t][1] = wnext[t][1];
            }
            __syncthreads();
        }
    }
    // partial tiles of a K-slice: column n of c_fc2 sits at N + n
    if constexpr (MODE == 1) {
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
            const int m = m0 + tt * 16 + c;
            if (m >= p.M) continue;
#pragma unroll
            for (int t = 0; t < kTPW; ++t) {
                const int n = tile[t] * 16 + 4 * g;
                if (tile[t] < p.n_tiles && n < p.N)
                    *(i32x4*)(p.part + ((int64_t)ks * p.M + m) * p.ldp + (PAIR && rr[t] == 1 ? p.N : 0) + n) = acc[t][tt];
            }
        }
        return;
    }
    if constexpr (MODE == 2) {
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
            const int m = m0 + tt * 16 + c;
            if (m >= p.M || tt < tt_lo || tt >= tt_hi) continue;
#pragma unroll
            for (int t = 0; t < kTPW; ++t) {
                const int n = tile[t] * 16 + 4 * g;
                if (tile[t] < p.n_tiles && n < p.N)
                    for (int s2 = 0; s2 < p.ksplit; ++s2)
                        acc[t][tt] += *(const i32x4*)(p.part + ((int64_t)s2 * p.M + m) * p.ldp + (PAIR && rr[t] == 1 ? p.N : 0) + n);
            }
        }
    }

    // ---- epilogue.  lane (g, c) holds rows 4 g .. 4 g + 3 of its tiles for token tt * 16 + c
    float scb[kTPW][4];
#pragma unroll
    for (int t = 0; t < kTPW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = tile[t] * 16 + 4 * g + r;
            scb[t][r] = n < p.N ? ((PAIR && rr[t] == 1) ? p.scb2 : p.scb)[n] : 0.f;
        }
    float d[kTPW][8][4];
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) {
        const int m = m0 + tt * 16 + c;
        const float sa = m < p.M ? p.sca[m] : 0.f;
#pragma unroll
        for (int t = 0; t < kTPW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) d[t][tt][r] = f16r_of_f32((((float)acc[t][tt][r] * 6.200012e-05f) * sa) * scb[t][r]);
    }
    const int n_out = p.olist[0];
    if (n_out > 0) {
        // mixed-precision decomposition: the outlier columns in f16, ascending k, kOutChunk columns per pass through LDS
        f16_t* xo = (f16_t*)smem;                     // [128 tokens][kOutChunk]
        f16_t* so = xo + kBM * kOutChunk;             // [kSlots x 16 rows][kOutChunk]: f16(CB[n, k] * SCB[n] / 127)
        // (round 6: the side product runs on v_mfma_f32_16x16x32_f16 — A = 16 rows x 32 outlier columns of `so`, B = 16 tokens x the same
        // columns of `xo`, result in the accumulators' own layout: lane (g, c) holds rows 4 g .. 4 g + 3 for token c.  The products of two f16
        // values are exact in f32 and the sum is f32 as before; its ORDER inside a block of 32 columns is the hardware's instead of
        // ascending k — bitsandbytes' own f16 GEMM promises no order either, and the tests allow the one f16 ulp this can move.  The scalar
        // loop it replaces (64 FMAs per lane and outlier column) took 1.4 ms per attn.c_proj / mlp.c_proj launch at T = 2048, where ~10 % of
        // the columns of a 2048-token call hold an outlier: profiles/r06_prefill_int8_outliers.txt.)
        typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
        f32x4 o[kTPW][8];
#pragma unroll
        for (int t = 0; t < kTPW; ++t)
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) o[t][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int base = 0; base < n_out; base += kOutChunk) {
            const int nc = n_out - base < kOutChunk ? n_out - base : kOutChunk;
            __syncthreads();  // (the main loop / the previous pass is done with the LDS)
            // a thread's outlier column is the same in every iteration (kThreads is a multiple of kOutChunk): its list entry and
            // the column's place inside a unit are read once per pass, which leaves ONE gather per element in loops that can be unrolled
            // (with the list read inside, each of the 32 iterations waited for two dependent loads: ~40 us per pass of 64 columns)
            static_assert(kThreads % kOutChunk == 0 && (kOutChunk & (kOutChunk - 1)) == 0, "column index per thread is loop-invariant");
            const int oi = threadIdx.x & (kOutChunk - 1);
            const bool col_ok = oi < nc;
            const int k = col_ok ? p.olist[1 + base + oi] : 0;
#pragma unroll 8
            for (int idx = threadIdx.x; idx < kBM * kOutChunk; idx += kThreads) {
                const int tok = idx / kOutChunk;
                const int m = m0 + tok;
                xo[idx] = (m < p.M && col_ok) ? p.xh[(int64_t)m * p.Kp + k] : (f16_t)0;
            }
            // weights: ROW fastest — 16 lanes = the 16 rows of a tile at one outlier column (16 B apart in the stream: two cache lines),
            // four columns per wave (the column-fastest order touched 64 lines per wave load); a thread sees two columns, oc and oc + 32
            static_assert(kThreads == 512 && kOutChunk == 64, "element e = j * 512 + thread: row e & 15, column (e >> 4) & 63, slot e >> 10 = j >> 1");
            const int oc = threadIdx.x >> 4, row = threadIdx.x & 15;
            const bool c_ok[2] = {oc < nc, oc + 32 < nc};
            const int k2[2] = {c_ok[0] ? p.olist[1 + base + oc] : 0, c_ok[1] ? p.olist[1 + base + oc + 32] : 0};
            const int j_lo = MODE == 2 ? 2 * ks * kTPW : 0, j_hi = MODE == 2 ? j_lo + 2 * kTPW : 2 * kSlots;  // MODE 2: the rows of slot group ks only
#pragma unroll 8
            for (int j = j_lo; j < j_hi; ++j) {
                const int slot = j >> 1, h = j & 1;
                const int wv = slot / kTPW, t = slot - wv * kTPW;
                int tl, r1;
                if (PAIR) {
                    tl = nb * (kSlots / 2) + (kTPW / 2) * wv + (t >> 1);
                    r1 = t & 1;
                } else {
                    tl = nb * kSlots + kTPW * wv + t;
                    r1 = 0;
                }
                const int n = tl * 16 + row;
                const int kk = h ? k2[1] : k2[0];
                f16_t sub = 0;
                if ((h ? c_ok[1] : c_ok[0]) && tl < p.n_tiles && n < p.N) {
                    const int ku = kk >> 7, ke = (kk >> 6) & 1, kgg = (kk >> 4) & 3, kjj = kk & 15;
                    const int64_t off = ((((int64_t)tl * p.units + ku) * R + r1) * 2 + ke) * 1024 + (kgg * 16 + row) * 16 + kjj;
                    const float cb = (float)(int8_t)p.w[off];
                    const float s = ((PAIR && r1 == 1) ? p.scb2 : p.scb)[n];
                    sub = f32_to_f16(__fdiv_rn(cb * s, 127.0f));
                }
                so[(slot * 16 + row) * kSoLd + oc + 32 * h] = sub;
            }
            __syncthreads();
#pragma unroll
            for (int kb = 0; kb < kOutChunk / 32; ++kb) {
                if (kb * 32 >= nc) break;
                f16x8_t a[kTPW];
#pragma unroll
                for (int t = 0; t < kTPW; ++t) a[t] = *(const f16x8_t*)(so + ((wslot * kTPW + t) * 16 + c) * kSoLd + kb * 32 + 8 * g);
#pragma unroll
                for (int tt = 0; tt < 8; ++tt) {
                    if (MODE == 2 && (tt < tt_lo || tt >= tt_hi)) continue;
                    const f16x8_t b = *(const f16x8_t*)(xo + (tt * 16 + c) * kOutChunk + kb * 32 + 8 * g);
#pragma unroll
                    for (int t = 0; t < kTPW; ++t) o[t][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t], b, o[t][tt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < kTPW; ++t)
#pragma unroll
            for (int tt = 0; tt < 8; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) d[t][tt][r] = f16r(d[t][tt][r] + f16r(o[t][tt][r]));
    }
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) {
        const int m = m0 + tt * 16 + c;
        if (m >= p.M || tt < tt_lo || tt >= tt_hi) continue;
        // four consecutive output columns per lane and tile: one 16-B (f32) / 8-B (bf16, f16) access when the row allows it
        const bool vec = (p.N & 3) == 0 && (p.ldy & 3) == 0 && ((uintptr_t)p.y & 15) == 0;
        if constexpr (EPI == MI355_EPI_SWIGLU) {
#pragma unroll
            for (int t = 0; t < kTPW; t += 2) {
                const int n = tile[t] * 16 + 4 * g;
                if (tile[t] >= p.n_tiles || n >= p.N) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = swiglu_f32(d[t][tt][r], d[t + 1][tt][r]);
                if (vec && p.y_dtype == MI355_BF16) {
                    u32x2 pk;
                    pk[0] = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
                    pk[1] = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
                    *(u32x2*)((bf16_t*)p.y + (int64_t)m * p.ldy + n) = pk;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < p.N) st_from_f32(p.y, (int64_t)m * p.ldy + n + r, p.y_dtype, v[r]);
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < kTPW; ++t) {
                const int n = tile[t] * 16 + 4 * g;
                if (tile[t] >= p.n_tiles || n >= p.N) continue;
                const int64_t yi = (int64_t)m * p.ldy + n;
                if (vec && p.y_dtype == MI355_F32) {
                    f32x4 o = {d[t][tt][0], d[t][tt][1], d[t][tt][2], d[t][tt][3]};
                    if constexpr (EPI == MI355_EPI_ACCUM) o += *(const f32x4*)((const float*)p.y + yi);
                    *(f32x4*)((float*)p.y + yi) = o;
                } else if (vec && p.y_dtype == MI355_BF16) {
                    float o[4] = {d[t][tt][0], d[t][tt][1], d[t][tt][2], d[t][tt][3]};
                    if constexpr (EPI == MI355_EPI_ACCUM) {
                        const u32x2 old = *(const u32x2*)((const bf16_t*)p.y + yi);
                        o[0] += __uint_as_float(old[0] << 16);
                        o[1] += __uint_as_float(old[0] & 0xffff0000u);
                        o[2] += __uint_as_float(old[1] << 16);
                        o[3] += __uint_as_float(old[1] & 0xffff0000u);
                    }
                    u32x2 pk;
                    pk[0] = (uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16);
                    pk[1] = (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16);
                    *(u32x2*)((bf16_t*)p.y + yi) = pk;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (n + r >= p.N) continue;
                        float v = d[t][tt][r];
                        if constexpr (EPI == MI355_EPI_ACCUM) v += ld_as_f32(p.y, yi + r, p.y_dtype);
                        st_from_f32(p.y, yi + r, p.y_dtype, v);
                    }
                }
            }
        }
    }
}

template <int EPI, bool PAIR, int MODE>
int launch_i8_mode(const I8GemmParams& q, hipStream_t s) {
    static hipError_t attr_err =
        hipFuncSetAttribute((const void*)i8_gemm_kernel<EPI, PAIR, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (attr_err != hipSuccess) {
        mi355_set_error("hipFuncSetAttribute(i8_gemm) failed: %s", hipGetErrorString(attr_err));
        return (int)attr_err;
    }
    I8GemmParams r = q;
    r.per_xcd = (q.total_blocks * (MODE == 1 ? q.ksplit : MODE == 2 ? 8 : 1) + 7) / 8;
    hipLaunchKernelGGL((i8_gemm_kernel<EPI, PAIR, MODE>), dim3(8 * r.per_xcd), dim3(kThreads), kLds, s, r);
    MI355_LAUNCH_CHECK();
    return 0;
}
template <int EPI, bool PAIR>
int launch_i8_gemm(const I8GemmParams& p, hipStream_t s) {
    const int per_block = PAIR ? kSlots / 2 : kSlots;
    I8GemmParams q = p;
    q.n_blocks = (p.n_tiles + per_block - 1) / per_block;
    q.total_blocks = q.n_blocks * ((p.M + kBM - 1) / kBM);
    if (q.ksplit <= 1) return launch_i8_mode<EPI, PAIR, 0>(q, s);
    if (int rc = launch_i8_mode<EPI, PAIR, 1>(q, s)) return rc;
    return launch_i8_mode<EPI, PAIR, 2>(q, s);
}

size_t pad16(size_t v) { return (v + 15) & ~(size_t)15; }

}  // namespace

extern "C" size_t mi355_linear_int8_gemm_workspace_bytes(int M, int K) {
    if (M <= 0 || K <= 0) return 0;
    const size_t kp = ((size_t)K + 127) / 128 * 128;
    // xh f16 [M, Kp], CA int8 [M, Kp], absmax [M], column mask [Kp / 32], outlier list [1 + Kp]
    return pad16((size_t)M * kp * 2) + pad16((size_t)M * kp) + pad16((size_t)M * 4) + pad16(kp / 8) + pad16((1 + kp) * 4) + 256 +
           kSplitBudget;  // int32 partial tiles of the split-K launches (short prompts)
}

extern "C" int mi355_linear_int8_gemm(const mi355_int8_args* a, void* workspace, size_t workspace_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(a != nullptr && workspace != nullptr && a->w && a->scb && a->x && a->y, MI355_E_ARG, "linear_int8_gemm: null argument");
    MI355_CHECK_ARG(a->R == 1 || a->R == 2, MI355_E_ARG, "linear_int8_gemm: R must be 1 or 2");
    MI355_CHECK_ARG(a->M >= 1 && a->N > 0 && a->K > 0, MI355_E_SHAPE, "linear_int8_gemm: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
    MI355_CHECK_ARG(a->epi >= MI355_EPI_STORE && a->epi <= MI355_EPI_SWIGLU, MI355_E_ARG, "linear_int8_gemm: bad epi");
    const bool swiglu = a->epi == MI355_EPI_SWIGLU;
    MI355_CHECK_ARG(swiglu ? (a->R == 2 && a->scb2) : a->R == 1, MI355_E_ARG,
                    "linear_int8_gemm: STORE / ACCUM take the R = 1 stream, SWIGLU the interleaved R = 2 stream");
    MI355_CHECK_ARG(a->bias == nullptr && a->attn_partials == nullptr, MI355_E_ARG, "linear_int8_gemm: no bias / attention prologue");
    auto three = [](int d) { return d == MI355_F32 || d == MI355_BF16 || d == MI355_F16; };
    MI355_CHECK_ARG(three(a->x_dtype) && three(a->y_dtype), MI355_E_DTYPE, "linear_int8_gemm: x / y dtype");
    MI355_CHECK_ARG(a->norm_scale == nullptr || three(a->norm_dtype), MI355_E_DTYPE, "linear_int8_gemm: norm scale dtype");
    MI355_CHECK_ARG(workspace_bytes >= mi355_linear_int8_gemm_workspace_bytes(a->M, a->K) && (uintptr_t)workspace % 16 == 0,
                    MI355_E_SHAPE, "linear_int8_gemm: workspace of %zu bytes is too small (need %zu)", workspace_bytes,
                    mi355_linear_int8_gemm_workspace_bytes(a->M, a->K));
    hipStream_t s = (hipStream_t)stream;
    const int units = (a->K + 127) / 128, kp = units * 128;
    MI355_CHECK_ARG((size_t)a->M * kp < 0xFFFFFFF0ull, MI355_E_SHAPE, "linear_int8_gemm: M x K too large for one launch");
    char* ws = (char*)workspace;
    f16_t* xh = (f16_t*)ws;
    ws += pad16((size_t)a->M * kp * 2);
    int8_t* ca = (int8_t*)ws;
    ws += pad16((size_t)a->M * kp);
    float* sca = (float*)ws;
    ws += pad16((size_t)a->M * 4);
    unsigned* mask = (unsigned*)ws;
    ws += pad16((size_t)kp / 8);
    int* olist = (int*)ws;
    ws += pad16((size_t)(1 + kp) * 4);
    int* part = (int*)ws;
    // split-K: launches of fewer than 96 blocks are cut into up to 8 K-slices (as many as fit the partial buffer)
    int ksplit = 1;
    const int64_t ldp = (int64_t)a->N * (swiglu ? 2 : 1);
    {
        const int rows_per_block = swiglu ? 16 * kSlots / 2 : 16 * kSlots;
        const int blocks = ((a->N + rows_per_block - 1) / rows_per_block) * ((a->M + kBM - 1) / kBM);
        while (ksplit < kMaxSplit && blocks * ksplit < 96 && units >= 8 * ksplit &&
               (size_t)2 * ksplit * a->M * ldp * 4 <= kSplitBudget)
            ksplit *= 2;
    }
    MI355_HIP(hipMemsetAsync(mask, 0, (size_t)kp / 8, s));
    hipLaunchKernelGGL(i8_stage_kernel, dim3(a->M), dim3(256), 0, s, a->x, a->x_dtype, a->ldx, a->norm_scale, a->norm_dtype, a->eps,
                       a->threshold, a->K, kp, xh, sca, mask);
    MI355_LAUNCH_CHECK();
    hipLaunchKernelGGL(i8_quant_kernel, dim3(a->M), dim3(256), 0, s, xh, sca, mask, kp, ca, olist);
    MI355_LAUNCH_CHECK();
    I8GemmParams p;
    memset(&p, 0, sizeof(p));
    p.w = (const uint8_t*)a->w;
    {
        const size_t wb = mi355_packed_bytes(MI355_W_I8, a->N, a->K, a->R, swiglu ? 1 : 0);
        MI355_CHECK_ARG(wb > 0 && wb < 0xFFFFFFF0ull, MI355_E_SHAPE, "linear_int8_gemm: weight stream of %zu B", wb);
        p.w_bytes = (unsigned)wb;
    }
    p.ca = ca;
    p.xh = xh;
    p.sca = sca;
    p.olist = olist;
    p.scb = a->scb;
    p.scb2 = a->scb2;
    p.y = a->y;
    p.ldy = a->ldy;
    p.M = a->M;
    p.N = a->N;
    p.K = a->K;
    p.Kp = kp;
    p.units = units;
    p.n_tiles = (a->N + 15) / 16;
    p.y_dtype = a->y_dtype;
    p.ksplit = ksplit;
    p.ldp = ldp;
    p.part = part;
    if (swiglu) return launch_i8_gemm<MI355_EPI_SWIGLU, true>(p, s);
    if (a->epi == MI355_EPI_ACCUM) return launch_i8_gemm<MI355_EPI_ACCUM, false>(p, s);
    return launch_i8_gemm<MI355_EPI_STORE, false>(p, s);
}

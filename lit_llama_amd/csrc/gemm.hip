// Wide (M >= 32 tokens) linear over the int4 or the bf16 weight stream for gfx950: prompt prefill and no-cache evaluation.
//
// Replaces the reference's Triton `linear_kernel_4bit_weight` (/root/reference lit_llama/quantization.py:187-333,
// reached from ColBlockQuantizedLinear.forward :413-421) at the shapes of evaluate/full.py:120-129 (T = 2048, no cache)
// and of generate.py's prompt pass — where the skinny weight-streaming kernel (gemv.hip) re-reads the whole weight
// stream once per 13 tokens.
//
// Design:
//  * the weights stay in the decode path's stream layout [tile of 16 rows][unit of 128 k][r][lane][16 B]: a wave reads
//    a (tile, unit) piece with ONE coalesced 1-KiB load straight into registers (no LDS for weights: a piece is
//    private to its wave) and turns it into four MFMA A fragments with 7 VALU ops per 8 weights — amortised over the
//    8 token tiles of the block, so the conversion that bounds the M = 1 path costs 1/8 MFMA slot here;
//  * the activations of a block (128 tokens x 128 k, bf16) go through LDS, double buffered, the 16-B chunks of a row
//    XOR-swizzled by swz(token & 15) — a GF(2)-linear map chosen by exhaustive search so that every ds_read_b128 lane
//    group of a B-fragment read (the hardware serves {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... together) touches
//    16 different 16-B slots: the obvious (token & 7) was 2-way conflicted and made the kernel LDS-bound;
//  * workgroup = 8 waves, block tile = 128 tokens x 16 row tiles (2 per wave; for the c_fc1 / c_fc2 pair stream the two
//    tiles of a wave are the fc1 / fc2 halves of the same 16 rows, so SwiGLU stays in the epilogue): every B fragment
//    read from LDS feeds two MFMAs, 64 x v_mfma_f32_16x16x32_bf16 per wave and unit (4 tiles per wave measured
//    slower: 184 VGPRs halve the waves per SIMD);
//  * XCD-aware block order: workgroup b runs on XCD b % 8 (each XCD has its own 4 MiB L2).  The (token block, row
//    block) pairs are cut, token-block major, into 8 contiguous ranges, one per XCD: the workgroups an XCD runs side by
//    side share one or two token blocks (1 MiB of activations each), fetched into that L2 once;
//  * operands are staged once per linear by stage_rows_kernel: bf16(norm_scale * x) with RMSNorm's 1/rms kept as a
//    per-row factor for the epilogue, and the per-row operand sum that undoes the +128 / zero-point offset:
//        y[m, n] = scale[n] * (acc[m, n] - (128 + zero[n]) * sum_k xb[m, k]) * rinv[m]      (same arithmetic as gemv.hip).
#include <stdlib.h>

#include "common.h"
#include "gemm_fuse.h"

// The epilogues' conversions on the hardware paths (round 6: with IEEE division + expf in SiLU and the six-instruction software rounding
// the c_fc1 / c_fc2 epilogue was VALU-bound — 33 us of a 269-us launch at T = 2048 against 9 us for its 45 MB of stores,
// profiles/r06_prefill_epilogue_cost.txt): v_cvt_pk_bf16_f32 rounds to nearest even exactly as f32_to_bf16 does (finite inputs),
// v_exp_f32 / v_rcp_f32 are within 1 ulp of f32 — 2^-15 of the bf16 step the result is rounded to.
__device__ __forceinline__ bf16_t bf16_hw(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float swiglu_fast(float a, float b) {
    // silu(a) * b = a / (1 + 2^(-a log2 e)) * b; a -> -inf: 2^(+inf) = inf, rcp = 0; a -> +inf: rcp(1) = 1
    return (a * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a * -1.44269504088896340736f))) * b;
}

namespace {

constexpr int kBM = 128;     // tokens per block
#ifndef MI355_GEMM_TPW
#define MI355_GEMM_TPW 2
#endif
constexpr int kTPW = MI355_GEMM_TPW;
#ifndef MI355_GEMM_XDMA
#define MI355_GEMM_XDMA 1
#endif
#ifndef MI355_GEMM_BPIPE
#define MI355_GEMM_BPIPE 4   // 128-token blocks: B fragments requested this many LDS reads ahead of the MFMAs that take them (0: hipcc's
                             // order, which waits for every pair of reads right behind their issue); 2 / 3 / 4: -1.4 / -1.9 / -2.0 % per 2048-token
                             // prompt (profiles/r06_prefill_gemm_ab.txt); 124-127 VGPRs at 4, still two workgroups per CU
#endif
#ifndef MI355_GEMM_PIN_LOADS
#define MI355_GEMM_PIN_LOADS 1
#endif
#ifndef MI355_GEMM_AND_OR
#define MI355_GEMM_AND_OR 1
#endif
#ifndef MI355_GEMM_PIN_W
#define MI355_GEMM_PIN_W 0
#endif
#ifndef MI355_GEMM_BF16_REFILL
#define MI355_GEMM_BF16_REFILL 1
#endif
#ifndef MI355_GEMM_BF16_ROWMAJOR
// workgroup order of the BF16 streams: row block major (an XCD runs every token block of a few row blocks side by side).  Token block major
// (the int4 order) carried every weight piece across the fabric once per token block: 16 x 13.5 GB = 216 GB per 2048-token pass of the 7B
// model, 6.3 TB/s of its 34 ms.  2048 tokens 34.0 -> 28.5 ms, 512 tokens 16.5 -> 10.6 ms (profiles/r06_ab_bf16_block_order.txt)
#define MI355_GEMM_BF16_ROWMAJOR 1
#endif
#ifndef MI355_GEMM_ONE_PER_CU
#define MI355_GEMM_ONE_PER_CU 1
#endif
#ifndef MI355_GEMM_Q4_ROWMAJOR
#define MI355_GEMM_Q4_ROWMAJOR 0
#endif
#ifndef MI355_GEMM_BF16_OCC
// 1 held the BF16-stream kernels at 128 VGPRs (two workgroups per CU) since round 3 — with 100-112 bytes of scratch per lane in the 128-token
// blocks, inside the unit loop.  Round 6 measured it: a 2048-token prompt of the bf16 7B model 69.3 ms with it, 37.5 ms without (130-136 VGPRs,
// one workgroup per CU, no scratch); 64-token blocks (118 VGPRs either way) and prompts up to 512 tokens are unchanged
// (profiles/r06_prefill_bf16_occupancy.txt).
#define MI355_GEMM_BF16_OCC 0
#endif  // 16-row tile slots per wave: one B fragment read from LDS feeds kTPW MFMAs
// Waves per workgroup: 8 (a block = 16 row tiles x 128 tokens) for prompts that fill the chip; 2 (4 row tiles) when the
// launch would otherwise be a few dozen workgroups — a 128-token prompt of a 7B model is ONE token block, i.e. 16 workgroups
// for attn.c_proj / mlp.c_proj (N = 4096) on 256 CUs, each streaming its 0.5-1.4 MB of weights alone: ~96 us per launch.

struct GemmParams {
    const uint8_t* w;
    unsigned w_bytes;
    const bf16_t* xb;      // [M, ldxb] staged operands
    int64_t ldxb;
    const float* rinv;     // [M]
    const float* sx;       // [M]
    const void* scales;
    const void* zeros;
    const void* scales2;
    const void* zeros2;
    void* y;
    int64_t ldy;
    int M, N, K, units, n_tiles, n_blocks, per_xcd, total_blocks;
    int sz_dtype, y_dtype;
    // split-K (few blocks: short prompts against N = 4096): slice ks of the units -> part[ks][M][N] f32, summed in order
    // by splitk_reduce_kernel; sx then holds [M][ksplit] per-slice operand sums
    int ksplit;
    float* part;
    // grouped scales (GRP kernels): gtab[group 0..n_groups][r][tab_ld rows] = bf16 scale | bf16 zero << 16 (row n_groups
    // is all zero), sxt[group][M] = operand sums per group; upg = 128-column units per group (groups of >= 128 columns)
    const uint32_t* gtab;
    const float* sxt;
    int n_groups, gq_shift, upg, tab_ld;
    // producer / consumer fusion (gemm_fuse.h; FUSE kernels only).  f_in: xb is the caller's operand, 1/rms and the operand
    // sums of the block's rows are rebuilt from the partial sums in_ss / in_sx ([partial][M])
    int f_in, in_sx_n, in_ss_n, in_ppu;  // in_ppu: partial operand sums per 128-column unit (a K-slice sums its own units' shares)
    const float* in_sx;
    const float* in_ss;
    float eps;
    bf16_t* out_xb;       // ACCUM: bf16(next_norm * y_new)
    int64_t out_ld;
    const void* next_norm;
    int next_norm_dtype;
    float* out_ss;        // [n_blocks][M]
    float* out_sx;
    const float* rope;    // STORE of c_attn: k rotated + K / V cache rows written here
    const int32_t* pos;
    bf16_t* kcache;
    bf16_t* vcache;
    int S, C, rope_gathered;
    float q_scale;        // != 0: q leaves rotated, scaled by q_scale and rounded to bf16 (gemm_fuse.h: q_scale), rows of 2 * ldy bf16
};

// c_attn's q rows as the prompt attention multiplies them (flash_prefill.hip's own prologue, moved to the producer: the same f32
// arithmetic and the one rounding, so the same bits): four outputs = two interleaved pairs at row m, column n < C
__device__ __forceinline__ void store_q_ready(const GemmParams& p, int m, int n, int posm, float o0, float o1, float o2, float o3) {
    const f32x4 cs = *(const f32x4*)(p.rope + (int64_t)(p.rope_gathered ? m : posm) * 128 + (n & 127));
    const float qs = p.q_scale;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    const bf2 a = {(__bf16)(qs * (o0 * cs[0] - o1 * cs[1])), (__bf16)(qs * (o1 * cs[0] + o0 * cs[1]))};
    const bf2 b = {(__bf16)(qs * (o2 * cs[2] - o3 * cs[3])), (__bf16)(qs * (o3 * cs[2] + o2 * cs[3]))};
    u32x2 pk;
    pk[0] = __builtin_bit_cast(uint32_t, a);
    pk[1] = __builtin_bit_cast(uint32_t, b);
    *(u32x2*)((bf16_t*)p.y + (int64_t)m * (2 * p.ldy) + n) = pk;
}

constexpr size_t kSplitBudget = (size_t)32 << 20;  // bytes of split-K partials a workspace holds
constexpr int kMaxSplit = 8;
constexpr size_t kFuseScratch = (size_t)8 << 20;  // partial sums of the fused prompt chain: <= 2048 tokens x ~500 shares x 4 B
// first unit of K-slice s of `ksplit` (slice s = units [lo(s), lo(s + 1)))
__host__ __device__ __forceinline__ int slice_lo(int s, int units, int ksplit) { return (int)((int64_t)s * units / ksplit); }

// 16-B chunk swizzle of an activation row in LDS (see the header): bits 0, 1 of the token stay, bit 2 -> 8, bit 3 -> 12
#ifdef MI355_NO_SWZ
__device__ __forceinline__ int swz(int tok) { return tok & 7; }
#else
__device__ __forceinline__ int swz(int tok) { return (tok & 3) ^ (((tok >> 2) & 1) << 3) ^ (((tok >> 3) & 1) * 12); }
#endif

__device__ __forceinline__ float ldsz(const void* p, int i, int dtype) {
    return dtype == MI355_F32 ? ((const float*)p)[i] : bf16_to_f32(((const bf16_t*)p)[i]);
}

// one row per workgroup: xb = bf16(norm_scale * x), rinv = rsqrt(mean(x^2) + eps) (1 without norm), sx[m][s] = sum of xb over
// K-slice s (ksplit = 1: the whole row)
__global__ __launch_bounds__(256) void stage_rows_kernel(const void* x, int x_dtype, int64_t ldx, const void* norm_scale,
                                                         int norm_dtype, float eps, int K, int Kp, bf16_t* xb, int64_t ldxb,
                                                         float* rinv, float* sx, int ksplit) {
    __shared__ float red[32];
    const int m = blockIdx.x;
    const int units = Kp >> 7;
    float ss = 0.f;
    for (int s = 0; s < ksplit; ++s) {
        const int k_lo = slice_lo(s, units, ksplit) << 7, k_hi = slice_lo(s + 1, units, ksplit) << 7;
        float sum = 0.f;
        for (int k = k_lo + threadIdx.x; k < k_hi; k += blockDim.x) {
            bf16_t o = 0;
            if (k < K) {
                float v = ld_as_f32(x, (int64_t)m * ldx + k, x_dtype);
                if (norm_scale != nullptr) {
                    ss += v * v;
                    v *= ld_as_f32(norm_scale, k, norm_dtype);
                }
                o = bf16_hw(v);
                sum += bf16_to_f32(o);
            }
            xb[(int64_t)m * ldxb + k] = o;
        }
        const float tot = block_sum(sum, red);
        if (threadIdx.x == 0) sx[(int64_t)m * ksplit + s] = tot;
    }
    const float tss = block_sum(ss, red);
    if (threadIdx.x == 0) rinv[m] = norm_scale != nullptr ? rsqrtf(tss / (float)K + eps) : 1.0f;
}

// grouped scales: as stage_rows_kernel, but sxt[grp][m] = sum of xb over the group's input columns.  A wave per group
// (wave reductions only); one row per workgroup
__global__ __launch_bounds__(256) void stage_rows_grouped_kernel(const void* x, int x_dtype, int64_t ldx, const void* norm_scale,
                                                                 int norm_dtype, float eps, int K, int Kp, bf16_t* xb,
                                                                 int64_t ldxb, float* rinv, float* sxt, int M, int group_cols,
                                                                 int n_groups) {
    __shared__ float red[32];
    const int m = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float ss = 0.f;
    for (int grp = wave; grp < n_groups; grp += 4) {
        const int k_lo = grp * group_cols;
        const int k_hi = k_lo + group_cols < Kp ? k_lo + group_cols : Kp;
        float sum = 0.f;
        for (int k = k_lo + lane; k < k_hi; k += 64) {
            bf16_t o = 0;
            if (k < K) {
                float v = ld_as_f32(x, (int64_t)m * ldx + k, x_dtype);
                if (norm_scale != nullptr) {
                    ss += v * v;
                    v *= ld_as_f32(norm_scale, k, norm_dtype);
                }
                o = bf16_hw(v);
                sum += bf16_to_f32(o);
            }
            xb[(int64_t)m * ldxb + k] = o;
        }
        sum = wave_sum(sum);
        if (lane == 0) sxt[(int64_t)grp * M + m] = sum;
    }
    for (int k = n_groups * group_cols + (int)threadIdx.x; k < Kp; k += 256) xb[(int64_t)m * ldxb + k] = 0;
    const float tss = block_sum(ss, red);
    if (threadIdx.x == 0) rinv[m] = norm_scale != nullptr ? rsqrtf(tss / (float)K + eps) : 1.0f;
}

// the reference's [N, n_groups] bf16 scale / zero tables -> gtab[grp][r][n] dwords (n fastest: the four rows a lane holds
// are one 16-B load); row n_groups and the rows past N are zero
__global__ __launch_bounds__(256) void group_table_kernel(const bf16_t* s0, const bf16_t* z0, const bf16_t* s1, const bf16_t* z1,
                                                          int N, int tab_ld, int n_groups, int R, uint32_t* gtab) {
    const int64_t total = (int64_t)(n_groups + 1) * R * tab_ld;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % tab_ld);
        const int64_t gr = i / tab_ld;
        const int r = (int)(gr % R), grp = (int)(gr / R);
        uint32_t v = 0;
        if (n < N && grp < n_groups) {
            const int64_t src = (int64_t)n * n_groups + grp;
            v = (uint32_t)(r == 1 ? s1 : s0)[src] | ((uint32_t)(r == 1 ? z1 : z0)[src] << 16);
        }
        gtab[i] = v;
    }
}

// y[m][n] (+)= sum over the K-slices, in slice order (deterministic)
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* part, int ksplit, int M, int N, void* y, int y_dtype,
                                                            int64_t ldy) {
    const int64_t n4 = N >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)M * n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4), n = (int)(i - (int64_t)m * n4) * 4;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if constexpr (EPI == MI355_EPI_SWIGLU) {
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < ksplit; ++s) {
                const float* row = part + ((int64_t)s * M + m) * (2 * (int64_t)N);
                o += *(const f32x4*)(row + n);
                b += *(const f32x4*)(row + N + n);
            }
            bf16_t* dst = (bf16_t*)y + (int64_t)m * ldy + n;
            u32x2 pk;
            pk[0] = (uint32_t)bf16_hw(swiglu_fast(o[0], b[0])) | ((uint32_t)bf16_hw(swiglu_fast(o[1], b[1])) << 16);
            pk[1] = (uint32_t)bf16_hw(swiglu_fast(o[2], b[2])) | ((uint32_t)bf16_hw(swiglu_fast(o[3], b[3])) << 16);
            *(u32x2*)dst = pk;
        } else if (y_dtype == MI355_F32) {
            float* dst = (float*)y + (int64_t)m * ldy + n;
            if constexpr (EPI == MI355_EPI_ACCUM) o = *(const f32x4*)dst;
            for (int s = 0; s < ksplit; ++s) o += *(const f32x4*)(part + ((int64_t)s * M + m) * N + n);
            *(f32x4*)dst = o;
        } else {
            bf16_t* dst = (bf16_t*)y + (int64_t)m * ldy + n;
            if constexpr (EPI == MI355_EPI_ACCUM) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = bf16_to_f32(dst[r]);
            }
            for (int s = 0; s < ksplit; ++s) o += *(const f32x4*)(part + ((int64_t)s * M + m) * N + n);
            u32x2 pk;
            pk[0] = (uint32_t)bf16_hw(o[0]) | ((uint32_t)bf16_hw(o[1]) << 16);
            pk[1] = (uint32_t)bf16_hw(o[2]) | ((uint32_t)bf16_hw(o[3]) << 16);
            *(u32x2*)dst = pk;
        }
    }
}

// The split-K reduction of a launch of the fused chain (gemm_fuse.h): sums the K-slices in slice order as splitk_reduce_kernel
// does, and is the producer epilogue the whole-K kernel has — the residual rows also leave as the next linear's bf16 operand
// with their partial sums, the SwiGLU output with its partial operand sums, c_attn's k rotated and K / V written to the cache.
// One half wave (32 lanes x 4 columns) per (row, 128-column unit): a unit is what a consumer's K-slices are cut at.
template <int EPI>
__global__ __launch_bounds__(256) void splitk_fused_reduce_kernel(const GemmParams p) {
    const int units_n = p.N >> 7;  // N % 128 == 0 (host check)
    const int64_t groups = (int64_t)p.M * units_n;
    const int l32 = threadIdx.x & 31;
    for (int64_t grp = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5; grp < groups; grp += (int64_t)gridDim.x * 8) {
        const int m = (int)(grp / units_n), un = (int)(grp - (int64_t)m * units_n);
        const int n = un * 128 + l32 * 4;
        if constexpr (EPI == MI355_EPI_SWIGLU) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
            for (int sl = 0; sl < p.ksplit; ++sl) {
                const float* row = p.part + ((int64_t)sl * p.M + m) * (2 * (int64_t)p.N);
                a += *(const f32x4*)(row + n);
                b += *(const f32x4*)(row + p.N + n);
            }
            bf16_t ob[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[r] = bf16_hw(swiglu_fast(a[r], b[r]));
            u32x2 pk;
            pk[0] = (uint32_t)ob[0] | ((uint32_t)ob[1] << 16);
            pk[1] = (uint32_t)ob[2] | ((uint32_t)ob[3] << 16);
            *(u32x2*)((bf16_t*)p.y + (int64_t)m * p.ldy + n) = pk;
            if (p.out_sx != nullptr) {
                const float s1 = group_sum((bf16_to_f32(ob[0]) + bf16_to_f32(ob[1])) + (bf16_to_f32(ob[2]) + bf16_to_f32(ob[3])), 32);
                if (l32 == 0) p.out_sx[(int64_t)un * p.M + m] = s1;
            }
        } else {
            float* dst = (float*)p.y + (int64_t)m * p.ldy + n;  // (f32 outputs only: host check)
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            if constexpr (EPI == MI355_EPI_ACCUM) o = *(const f32x4*)dst;
            for (int sl = 0; sl < p.ksplit; ++sl) o += *(const f32x4*)(p.part + ((int64_t)sl * p.M + m) * p.N + n);
            if constexpr (EPI == MI355_EPI_STORE) {
                const int sec = p.rope != nullptr ? (n >= p.C) + (n >= 2 * p.C) : 0;
                if (sec != 0) {  // as the whole-K epilogue: k rotated in f32, K / V rows to the bf16 cache
                    const int cn = n - sec * p.C, h = cn >> 7, d = cn & 127;
                    const int posm = p.pos[m];
                    if (sec == 1) {
                        const f32x4 cs = *(const f32x4*)(p.rope + (int64_t)(p.rope_gathered ? m : posm) * 128 + d);
                        const float k0 = o[0] * cs[0] - o[1] * cs[1], k1 = o[1] * cs[0] + o[0] * cs[1];
                        const float k2 = o[2] * cs[2] - o[3] * cs[3], k3 = o[3] * cs[2] + o[2] * cs[3];
                        o = f32x4{k0, k1, k2, k3};
                    }
                    const int slot = posm < p.S - 1 ? posm : p.S - 1;
                    u32x2 pk;
                    pk[0] = (uint32_t)bf16_hw(o[0]) | ((uint32_t)bf16_hw(o[1]) << 16);
                    pk[1] = (uint32_t)bf16_hw(o[2]) | ((uint32_t)bf16_hw(o[3]) << 16);
                    *(u32x2*)((sec == 1 ? p.kcache : p.vcache) + ((int64_t)h * p.S + slot) * 128 + d) = pk;
                    continue;
                }
                if (p.rope != nullptr && p.q_scale != 0.f) {
                    store_q_ready(p, m, n, p.pos[m], o[0], o[1], o[2], o[3]);
                    continue;
                }
            }
            *(f32x4*)dst = o;
            if constexpr (EPI == MI355_EPI_ACCUM) {
                if (p.out_xb != nullptr) {
                    bf16_t ob[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) ob[r] = bf16_hw(o[r] * ldsz(p.next_norm, n + r, p.next_norm_dtype));
                    u32x2 pk;
                    pk[0] = (uint32_t)ob[0] | ((uint32_t)ob[1] << 16);
                    pk[1] = (uint32_t)ob[2] | ((uint32_t)ob[3] << 16);
                    *(u32x2*)(p.out_xb + (int64_t)m * p.out_ld + n) = pk;
                    const float s1 = group_sum((bf16_to_f32(ob[0]) + bf16_to_f32(ob[1])) + (bf16_to_f32(ob[2]) + bf16_to_f32(ob[3])), 32);
                    const float s2 = group_sum((o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]), 32);
                    if (l32 == 0) {
                        p.out_sx[(int64_t)un * p.M + m] = s1;
                        p.out_ss[(int64_t)un * p.M + m] = s2;
                    }
                }
            }
        }
    }
}

// FMT = MI355_W_Q4: int4 stream, one 1-KiB piece per (tile, unit), converted below; MI355_W_BF16: unquantised weights
// (BASELINE configs[1]), four 1-KiB pieces per (tile, unit) whose piece d IS the A fragment of k-quarter d — same k
// order as the int4 conversion produces, no conversion, scale 1 / zero-point 0 in the epilogue
// GRP (Q4): 0 one (scale, zero) pair per output row; 1 / 2: one pair per row and group of input columns (GPTQ "groupsize",
// /root/reference lit_llama/quantization.py:284-333 with tile_cols > 0), groups of whole 128-column units (1) or of 32 / 64
// columns (2).  The MFMA accumulators keep running over K (cum_g = columns up to the end of group g) and at every group
// end     accf += (s_g - s_{g+1}) cum_g - s_g (128 + z_g) X_g      (Abel summation of sum_g s_g (A_g - (128 + z_g) X_g),
// s past the last group = 0; X_g = the group's operand sum from stage_rows_grouped_kernel): two FMAs per output element
// and group, no accumulator reset.  The k columns of ONE MFMA are spread over the unit (lane group g holds columns
// 32 g + 8 d ..), so sub-unit groups take one pass per group with the other lane groups' activations zeroed, as in gemv.hip.
// FUSE: the producer / consumer fusion of gemm_fuse.h (ungrouped launches without a K split): operand sums and 1/rms from
// partial sums instead of the staging pass; the residual epilogue emits the next linear's bf16 operand and its partial
// sums, the SwiGLU epilogue the partial operand sums of its output, the c_attn epilogue rotates k and writes the K / V cache
// (BF16 streams, 8 waves: four pieces per (tile, unit) in flight took the kernel to 130-136 VGPRs, i.e. ONE workgroup per CU;
// the second launch-bound argument — waves per SIMD — holds it at 128)
template <int EPI, bool PAIR, int FMT, int kWaves, int BM, int GRP = 0, bool FUSE = false>
__global__ __launch_bounds__(64 * kWaves, (FMT == MI355_W_BF16 && kWaves == 8 && MI355_GEMM_BF16_OCC) ? 4 : 1) void gemm_q4_kernel(const GemmParams p) {
    static_assert(GRP == 0 || FMT == MI355_W_Q4, "grouped scales are a Q4 feature");
    static_assert(!FUSE || GRP == 0, "the fused chain runs over per-row scales");
    constexpr int kTT = BM / 16;                   // 16-token tiles per block
    constexpr int kSlots = kWaves * kTPW;          // tile slots per block
    constexpr int kThreads = 64 * kWaves;
    constexpr int kXChunks = BM * 16 / kThreads;   // 16-B activation chunks per thread and unit
    constexpr int kWP = FMT == MI355_W_BF16 ? 4 : 1;  // pieces per (tile, unit)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    // b -> (token block, row block): the (token block, row block) pairs, token block major, are cut into 8 contiguous
    // ranges, one per XCD (b % 8): the workgroups an XCD runs side by side share one or two token blocks
    const int n_blocks = p.n_blocks;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int L2 = xcd * p.per_xcd + j;
    if (j >= p.per_xcd || L2 >= p.total_blocks * p.ksplit) return;
    const int L = L2 / p.ksplit, ks = L2 - L * p.ksplit;  // (ksplit = 1: ks = 0, all units)
    int mb, nb;
    if constexpr (FMT == MI355_W_BF16 ? MI355_GEMM_BF16_ROWMAJOR != 0 : MI355_GEMM_Q4_ROWMAJOR != 0) {
        // row block major — an XCD runs every token block of a few row blocks side by side, so a weight piece (BF16: 4 x the
        // bytes of an int4 one) crosses the fabric once per XCD instead of once per token block
        const int m_blocks = p.total_blocks / n_blocks;
        nb = L / m_blocks;
        mb = L - nb * m_blocks;
    } else {
        mb = L / n_blocks;
        nb = L - mb * n_blocks;
    }
    const int m0 = mb * BM;
    [[maybe_unused]] float* fr = (float*)(smem + 2 * BM * 256);  // FUSE: [BM] 1/rms, [BM] operand sums of the block's rows
    // K-slice: units [u_lo, u_hi); GRP 1 cuts at group boundaries (groups [g_lo, g_hi) of upg units each)
    const int g_lo = GRP == 1 ? slice_lo(ks, p.n_groups, p.ksplit) : 0, g_hi = GRP == 1 ? slice_lo(ks + 1, p.n_groups, p.ksplit) : 0;
    const int u_lo = GRP == 1 ? g_lo * p.upg : slice_lo(ks, p.units, p.ksplit);
    const int u_hi = GRP == 1 ? (g_hi * p.upg < p.units ? g_hi * p.upg : p.units) : slice_lo(ks + 1, p.units, p.ksplit);

    // this wave's row tiles
    int tile[kTPW], rr[kTPW];
#pragma unroll
    for (int t = 0; t < kTPW; ++t) {
        if (PAIR) {
            tile[t] = nb * (kSlots / 2) + (kTPW / 2) * wave + (t >> 1);  // pair tile: 16 rows of c_fc1 (r = 0) and c_fc2 (r = 1)
            rr[t] = t & 1;
        } else {
            tile[t] = nb * kSlots + kTPW * wave + t;
            rr[t] = 0;
        }
    }
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0, 0x00020000);
    const unsigned xbytes = (unsigned)((int64_t)p.M * p.ldxb * 2);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.xb, 0, (int)xbytes, 0x00020000);
    const unsigned lane_off = lane * 16;
    auto wload = [&](int t, int u, u32x4 (&dst)[kWP]) {
        // past the last unit / tile the load goes through a zero-sized descriptor: the scalar offset operand of a raw
        // buffer load is NOT range-checked (an unconditional prefetch of unit `units` of the last tile faulted)
        const bool ok = tile[t] < p.n_tiles && u < u_hi;
        const unsigned off = (unsigned)((tile[t] * p.units + u) * (PAIR ? 2 : 1) + rr[t]) * (1024u * kWP);
#pragma unroll
        for (int d = 0; d < kWP; ++d)
            dst[d] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ok ? rw : rw0, lane_off, ok ? off + d * 1024u : 0u, 0));
    };
    // BF16 streams (round 6): piece d of the NEXT unit is requested into the registers piece d of this unit just left (behind its MFMAs):
    // no second register set — 32 registers less, two workgroups per CU without the scratch the occupancy hint had bought them with
    constexpr bool kRefill = FMT == MI355_W_BF16 && GRP == 0 && MI355_GEMM_BF16_REFILL;
    [[maybe_unused]] auto wload1 = [&](int t, int u, int d, u32x4& dst) {
        const bool ok = tile[t] < p.n_tiles && u < u_hi;
        const unsigned off = (unsigned)((tile[t] * p.units + u) * (PAIR ? 2 : 1) + rr[t]) * (1024u * kWP);
        dst = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ok ? rw : rw0, lane_off, ok ? off + d * 1024u : 0u, 0));
    };
    // activation block of unit u: 2048 chunks of 16 B, kXChunks per thread; chunk = (token, 16-B column)
    // Round 6: the activation block of the NEXT unit goes global -> LDS by LDS-DMA (buffer_load ... lds, 16 B per lane: one instruction
    // fills 1 KiB = four token rows of the buffer; the XOR swizzle is applied on the SOURCE side — LDS slot c of token t takes column
    // c ^ swz(t)), requested at the unit's start.  The staged version held the block in 32 registers per thread, which hipcc could only
    // afford by sinking the requests to the unit's END, right in front of the LDS stores that wait for them: every unit paid the memory
    // latency in front of its barrier (rocprofv3 --pmc: SQ_WAIT_ANY 0.53 of the wave cycles, 0.28 issue stalls, 0.19 active;
    // profiles/r06_prefill_gemm_ab.txt).
    constexpr bool kXDma = MI355_GEMM_XDMA != 0 && GRP == 0;
    [[maybe_unused]] auto xdma = [&](int u, int buf) {
#pragma unroll
        for (int i = 0; i < kXChunks; ++i) {
            const int ch = i * kThreads + threadIdx.x;   // = LDS chunk: consecutive lanes, consecutive 16 B
            const int tok = ch >> 4, slot = ch & 15;
            const unsigned off = (unsigned)(((int64_t)(m0 + tok) * p.ldxb) * 2) + (unsigned)u * 256u + (unsigned)((slot ^ swz(tok)) * 16);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(smem + buf * (BM * 256) + (i * kThreads + wave * 64) * 16),
                                                     16, (m0 + tok) < p.M ? off : 0xFFFFFFF0u, 0, 0, 0);
        }
    };
    u32x4 stage[kXDma ? 1 : kXChunks];
    auto xload = [&](int u) {
        if constexpr (kXDma) return;
#pragma unroll
        for (int i = 0; i < kXChunks; ++i) {
            const int ch = i * kThreads + threadIdx.x;
            const int tok = ch >> 4, col = ch & 15;
            const unsigned off = (unsigned)(((int64_t)(m0 + tok) * p.ldxb) * 2) + (unsigned)u * 256u + (unsigned)col * 16u;
            stage[i] = __builtin_bit_cast(
                u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (m0 + tok) < p.M ? off : 0xFFFFFFF0u, 0, 0));
        }
    };
    auto xstore = [&](int buf) {
        if constexpr (kXDma) return;
#pragma unroll
        for (int i = 0; i < kXChunks; ++i) {
            const int ch = i * kThreads + threadIdx.x;
            const int tok = ch >> 4, col = ch & 15;
            *(u32x4*)(smem + buf * (BM * 256) + tok * 256 + ((col ^ swz(tok)) << 4)) = stage[i];
        }
    };

    f32x4 acc[kTPW][kTT];
#pragma unroll
    for (int t = 0; t < kTPW; ++t)
#pragma unroll
        for (int tt = 0; tt < kTT; ++tt) acc[t][tt] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 wcur[kTPW][kWP], wnext[kRefill ? 1 : kTPW][kRefill ? 1 : kWP];
#pragma unroll
    for (int t = 0; t < kTPW; ++t) wload(t, u_lo, wcur[t]);
    if constexpr (kXDma) xdma(u_lo, u_lo & 1);
    xload(u_lo);
    if constexpr (FUSE) {
        if (p.f_in) {
            // (behind the first unit's operand requests, in front of its barrier)
            // partial sums -> per-row factors.  Every thread takes a row and every kP-th partial, four loads in flight
            // (a plain loop over the partials is one memory round trip per partial: 13-15 us per launch, measured), the
            // kP shares are added in share order through LDS: a fixed order, deterministic
            constexpr int kP = kThreads >= BM ? kThreads / BM : 1;
            float* frp = fr + 2 * BM;  // [kP][2][BM]
            for (int idx = threadIdx.x; idx < kP * BM; idx += kThreads) {
                const int t = idx % BM, part = idx / BM;
                const int64_t m = m0 + t < p.M ? m0 + t : p.M - 1;
                float sxs = 0.f, sss = 0.f;
                // (a K-slice of a split launch takes the shares of its own units: the operand sum of ITS columns)
                const int jlo = p.ksplit > 1 ? slice_lo(ks, p.units, p.ksplit) * p.in_ppu : 0;
                const int jhi = p.ksplit > 1 ? slice_lo(ks + 1, p.units, p.ksplit) * p.in_ppu : p.in_sx_n;
                for (int j0 = jlo + part; j0 < jhi; j0 += 4 * kP) {
                    const int j1 = j0 + kP, j2 = j0 + 2 * kP, j3 = j0 + 3 * kP;
                    const float v0 = p.in_sx[j0 * (int64_t)p.M + m];
                    const float v1 = p.in_sx[(j1 < jhi ? j1 : j0) * (int64_t)p.M + m];
                    const float v2 = p.in_sx[(j2 < jhi ? j2 : j0) * (int64_t)p.M + m];
                    const float v3 = p.in_sx[(j3 < jhi ? j3 : j0) * (int64_t)p.M + m];
                    sxs += v0;
                    sxs += j1 < jhi ? v1 : 0.f;
                    sxs += j2 < jhi ? v2 : 0.f;
                    sxs += j3 < jhi ? v3 : 0.f;
                }
                for (int j0 = part; j0 < p.in_ss_n; j0 += 4 * kP) {
                    const int j1 = j0 + kP, j2 = j0 + 2 * kP, j3 = j0 + 3 * kP;
                    const float v0 = p.in_ss[j0 * (int64_t)p.M + m];
                    const float v1 = p.in_ss[(j1 < p.in_ss_n ? j1 : j0) * (int64_t)p.M + m];
                    const float v2 = p.in_ss[(j2 < p.in_ss_n ? j2 : j0) * (int64_t)p.M + m];
                    const float v3 = p.in_ss[(j3 < p.in_ss_n ? j3 : j0) * (int64_t)p.M + m];
                    sss += v0;
                    sss += j1 < p.in_ss_n ? v1 : 0.f;
                    sss += j2 < p.in_ss_n ? v2 : 0.f;
                    sss += j3 < p.in_ss_n ? v3 : 0.f;
                }
                frp[(part * 2 + 0) * BM + t] = sss;
                frp[(part * 2 + 1) * BM + t] = sxs;
            }
        }
    }
    xstore(u_lo & 1);
    __syncthreads();
    if constexpr (FUSE) {
        if (p.f_in) {
            constexpr int kP = kThreads >= BM ? kThreads / BM : 1;
            const float* frp = fr + 2 * BM;
            for (int t = threadIdx.x; t < BM; t += kThreads) {
                float sss = 0.f, sxs = 0.f;
#pragma unroll
                for (int part = 0; part < kP; ++part) {
                    sss += frp[(part * 2 + 0) * BM + t];
                    sxs += frp[(part * 2 + 1) * BM + t];
                }
                fr[t] = p.in_ss_n > 0 ? rsqrtf(sss / (float)p.K + p.eps) : 1.0f;
                fr[BM + t] = sxs;
            }
        }
    }

    // next unit's operands, requested UNCONDITIONALLY (a load inside `if (more)` makes hipcc drain vmcnt at the join,
    // i.e. wait for these very loads before the first MFMA): past the last unit the offsets fall into the next row /
    // tile or out of the descriptors (zeros) and the values are never used
    auto prefetch = [&](int u) {
#if MI355_GEMM_PIN_W
        // (A / B knob, off: 128-token blocks with only the WEIGHT requests — HBM latency — pinned at the unit's start, the
        // activation requests, L2 hits, free to sink: 116-124 VGPRs, measured neutral — profiles/r04_prefill_fused_chain_ab.txt)
        if constexpr (BM > 64 && GRP == 0 && FMT == MI355_W_Q4) {
#pragma unroll
            for (int t = 0; t < kTPW; ++t) wload(t, u + 1, wnext[t]);
            __builtin_amdgcn_sched_barrier(0);
            xload(u + 1);
            return;
        }
#endif
        if constexpr (kXDma) {
            // (weights first: HBM latency; both pinned at the unit's start — they cost 8 registers now, not 40)
            if constexpr (!kRefill) {
#pragma unroll
                for (int t = 0; t < kTPW; ++t) wload(t, u + 1, wnext[t]);
            }
            xdma(u + 1, (u & 1) ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            return;
        }
        xload(u + 1);
        if constexpr (!kRefill) {
#pragma unroll
            for (int t = 0; t < kTPW; ++t) wload(t, u + 1, wnext[t]);
        }
        // hipcc sinks these requests to the END of the unit, right in front of the LDS stores that wait for them (the
        // 24 registers they land in would otherwise be live across the MFMAs): a unit pays the memory latency in front
        // of its barrier, covered by the other waves of the SIMD.  Pinned here they fly under the unit's MFMAs — at
        // 141-145 VGPRs for 128-token blocks (one workgroup per CU), so only the 64-token blocks (95) can take it.
        if constexpr (MI355_GEMM_PIN_LOADS && BM <= 64 && GRP == 0) __builtin_amdgcn_sched_barrier(0);
    };
    // the 4 x kTT x kTPW MFMAs of a unit; sub >= 0 (GRP 2): only the lane groups of sub-group `sub` contribute
    // (explicitly software-pipelined B-fragment reads — 16 fragments in registers, pinned with sched_barrier —
    // measured SLOWER, 600-670 vs 760-800 TFLOP/s: with 4 waves per SIMD the hardware hides the LDS latency itself)
#if MI355_GEMM_AND_OR
    uint32_t cmask = 0x000F000Fu, cmagic = 0x43004300u;
    asm volatile("" : "+s"(cmask));
    asm volatile("" : "+v"(cmagic));
#endif
    constexpr int kBPipe = (GRP == 0 && FMT == MI355_W_Q4 && BM == 128) ? MI355_GEMM_BPIPE : 0;
    auto bread = [&](const char* xs, int d, int tt, bool mine) {
        const int tok = tt * 16 + c;
        u32x4 braw = *(const u32x4*)(xs + tok * 256 + (((4 * g + d) ^ swz(tok)) << 4));
        if constexpr (GRP == 2) braw = mine ? braw : u32x4{0u, 0u, 0u, 0u};
        return braw;
    };
    [[maybe_unused]] int u_refill = 0;  // kRefill: the unit whose pieces mfmas() requests
    auto mfmas = [&](int buf, int sub, int sub_shift) {
        const char* xs = smem + buf * (BM * 256);
        const bool mine = GRP != 2 || (g >> sub_shift) == sub;
        [[maybe_unused]] u32x4 bq[kBPipe > 0 ? kBPipe : 1];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            // int4 -> bf16 MFMA A fragments of k-quarter d: (w >> 4i) & 0x000F000F | 0x43004300 = (128 + q_2i, 128 + q_2i+1)
            bf16x8 a[kTPW];
#pragma unroll
            for (int t = 0; t < kTPW; ++t) {
                if constexpr (FMT == MI355_W_BF16) {
                    a[t] = __builtin_bit_cast(bf16x8, wcur[t][d]);
                } else {
                    const uint32_t v = wcur[t][0][d];
                    u32x4 f;
#if MI355_GEMM_AND_OR
                    // (mask and exponent pattern as OPAQUE register values: hipcc then selects one v_and_or_b32 per field
                    // — from literals it emits v_and + v_or, 11 instead of 7 VALU per dword — and pads the VALU -> MFMA
                    // hazard itself, which an inline-asm v_and_or_b32 does not get: csrc/fused_step_ring.hip nib2f16)
                    f[0] = (v & cmask) | cmagic;
                    f[1] = ((v >> 4) & cmask) | cmagic;
                    f[2] = ((v >> 8) & cmask) | cmagic;
                    f[3] = ((v >> 12) & cmask) | cmagic;
#else
                    f[0] = (v & 0x000F000Fu) | 0x43004300u;
                    f[1] = ((v >> 4) & 0x000F000Fu) | 0x43004300u;
                    f[2] = ((v >> 8) & 0x000F000Fu) | 0x43004300u;
                    f[3] = ((v >> 12) & 0x000F000Fu) | 0x43004300u;
#endif
                    a[t] = __builtin_bit_cast(bf16x8, f);
                }
            }
            if constexpr (kBPipe > 0) {
                // fragment s = d * kTT + tt of the unit; bq holds the next kBPipe of them.  The group barriers pin "one LDS read (of
                // fragment s + kBPipe), then the MFMAs of fragment s": without them hipcc sinks every read to its use again
#pragma unroll
                for (int tt = 0; tt < kTT; ++tt) {
                    const int s = d * kTT + tt;
                    if (s == 0) {
#pragma unroll
                        for (int q = 0; q < kBPipe; ++q) bq[q] = bread(xs, q / kTT, q % kTT, mine);
                        __builtin_amdgcn_sched_group_barrier(0x100, kBPipe, 0);  // the unit's first kBPipe reads up front
                    }
                    const bf16x8 b = __builtin_bit_cast(bf16x8, bq[s % kBPipe]);
                    if (s + kBPipe < 4 * kTT) bq[s % kBPipe] = bread(xs, (s + kBPipe) / kTT, (s + kBPipe) % kTT, mine);
#pragma unroll
                    for (int t = 0; t < kTPW; ++t) acc[t][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], b, acc[t][tt], 0, 0, 0);
                    if (s + kBPipe < 4 * kTT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, kTPW, 0);
                }
            } else {
#pragma unroll
                for (int tt = 0; tt < kTT; ++tt) {
                    const bf16x8 b = __builtin_bit_cast(bf16x8, bread(xs, d, tt, mine));
#pragma unroll
                    for (int t = 0; t < kTPW; ++t) acc[t][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], b, acc[t][tt], 0, 0, 0);
                }
            }
            if constexpr (kRefill) {
#pragma unroll
                for (int t = 0; t < kTPW; ++t) wload1(t, u_refill, d, wcur[t][d]);
            }
        }
    };
    auto rotate = [&](int buf) {
        xstore(buf ^ 1);
        if constexpr (!kRefill) {
#pragma unroll
            for (int t = 0; t < kTPW; ++t)
#pragma unroll
                for (int d = 0; d < kWP; ++d) wcur[t][d] = wnext[t][d];
        }
        __syncthreads();
    };

    f32x4 accf[GRP ? kTPW : 1][GRP ? kTT : 1];
    if constexpr (GRP == 0) {
        for (int u = u_lo; u < u_hi; ++u) {
            prefetch(u);
            u_refill = u + 1;
            mfmas(u & 1, 0, 0);
            rotate(u & 1);
        }
    } else {
#pragma unroll
        for (int t = 0; t < kTPW; ++t)
#pragma unroll
            for (int tt = 0; tt < kTT; ++tt) accf[t][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 tabc[kTPW], tabn[kTPW];
        float sxv[kTT];
        auto tload = [&](int grp, u32x4 (&dst)[kTPW]) {  // rows 4 g .. 4 g + 3 of this wave's tiles: one 16-B load each
#pragma unroll
            for (int t = 0; t < kTPW; ++t) {
                const int tl = tile[t] < p.n_tiles ? tile[t] : 0;
                dst[t] = *(const u32x4*)(p.gtab + ((int64_t)grp * (PAIR ? 2 : 1) + rr[t]) * p.tab_ld + tl * 16 + 4 * g);
            }
        };
        auto sxload = [&](int grp) {
#pragma unroll
            for (int tt = 0; tt < kTT; ++tt) {
                const int m = m0 + tt * 16 + c;
                sxv[tt] = p.sxt[(int64_t)grp * p.M + (m < p.M ? m : 0)];
            }
        };
        auto apply = [&]() {  // end of a group: tabc = this group's pairs, tabn = the next group's
            // two FMAs per output element, issued as v_pk_fma_f32 on register pairs (the VALU, not the matrix pipe, bounds
            // the grouped kernel: 64-token blocks halve the MFMAs a weight conversion is amortised over)
            f32x2 ds[kTPW][2], szp[kTPW][2];
#pragma unroll
            for (int t = 0; t < kTPW; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s_c = __uint_as_float(tabc[t][r] << 16), z_c = __uint_as_float(tabc[t][r] & 0xffff0000u);
                    const float s_n = __uint_as_float(tabn[t][r] << 16);
                    ds[t][r >> 1][r & 1] = s_c - s_n;
                    szp[t][r >> 1][r & 1] = -(s_c * (128.f + z_c));
                }
#pragma unroll
            for (int tt = 0; tt < kTT; ++tt) {
                const f32x2 sx2 = {sxv[tt], sxv[tt]};
#pragma unroll
                for (int t = 0; t < kTPW; ++t)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x2 a2 = {acc[t][tt][2 * h], acc[t][tt][2 * h + 1]};
                        f32x2 f2 = {accf[t][tt][2 * h], accf[t][tt][2 * h + 1]};
                        f2 = __builtin_elementwise_fma(ds[t][h], a2, f2);
                        f2 = __builtin_elementwise_fma(szp[t][h], sx2, f2);
                        accf[t][tt][2 * h] = f2[0];
                        accf[t][tt][2 * h + 1] = f2[1];
                    }
            }
#pragma unroll
            for (int t = 0; t < kTPW; ++t) tabc[t] = tabn[t];
        };
        // (a K-slice is a sum of its own: past its last group the scale is 0 — table row n_groups)
        if constexpr (GRP == 1) {
            tload(g_lo, tabc);
            int u = u_lo;
            for (int grp = g_lo; grp < g_hi; ++grp) {
                tload(grp + 1 < g_hi ? grp + 1 : p.n_groups, tabn);
                sxload(grp);
                const int u_end = (grp + 1) * p.upg < u_hi ? (grp + 1) * p.upg : u_hi;
                for (; u < u_end; ++u) {
                    prefetch(u);
                    mfmas(u & 1, 0, 0);
                    rotate(u & 1);
                }
                apply();
            }
        } else {
            const int sub_shift = p.gq_shift;  // lane groups per sub-group = 1 << gq_shift (0: 32 columns, 1: 64)
            const int nsub = 4 >> sub_shift;
            tload(u_lo * nsub, tabc);
            for (int u = u_lo; u < u_hi; ++u) {
                prefetch(u);
#pragma unroll 1
                for (int sub = 0; sub < nsub; ++sub) {
                    const int grp = u * nsub + sub < p.n_groups ? u * nsub + sub : p.n_groups - 1;
                    tload((u == u_hi - 1 && sub == nsub - 1) ? p.n_groups : grp + 1, tabn);
                    sxload(grp);
                    mfmas(u & 1, sub, sub_shift);
                    apply();
                }
                rotate(u & 1);
            }
        }
    }

    // ---- epilogue: lane (g, c) holds rows 4 g .. 4 g + 3 of its tiles for token tt * 16 + c
    float sc[kTPW][4], zp[kTPW][4];
    [[maybe_unused]] float ns[kTPW][4];  // FUSE, residual epilogue: the next RMSNorm's scales of this lane's rows
    const bool emit = FUSE && EPI == MI355_EPI_ACCUM && p.out_xb != nullptr;
    const bool sums = FUSE && EPI != MI355_EPI_STORE && p.out_sx != nullptr;
#pragma unroll
    for (int t = 0; t < kTPW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = tile[t] * 16 + 4 * g + r;
            const bool ok = n < p.N;
            const void* sp = (PAIR && rr[t] == 1) ? p.scales2 : p.scales;
            const void* zq = (PAIR && rr[t] == 1) ? p.zeros2 : p.zeros;
            if constexpr (FMT == MI355_W_BF16 || GRP != 0) {
                sc[t][r] = ok ? 1.f : 0.f;
                zp[t][r] = 0.f;
            } else {
                sc[t][r] = ok ? ldsz(sp, n, p.sz_dtype) : 0.f;
                zp[t][r] = ok ? 128.f + ldsz(zq, n, p.sz_dtype) : 0.f;
            }
            if constexpr (FUSE && EPI == MI355_EPI_ACCUM) ns[t][r] = (emit && ok) ? ldsz(p.next_norm, n, p.next_norm_dtype) : 0.f;
        }
    [[maybe_unused]] float es1[kTT], es2[kTT];  // FUSE: this lane's share of the rows' partial sums (operand values, squares)
#pragma unroll
    for (int tt = 0; tt < kTT; ++tt) {
        if constexpr (FUSE) es1[tt] = es2[tt] = 0.f;
        const int m = m0 + tt * 16 + c;
        if (m >= p.M) continue;
        float sxm, ri;
        if (FUSE && p.f_in) {
            ri = fr[tt * 16 + c];
            sxm = fr[BM + tt * 16 + c];
        } else {
            sxm = GRP ? 0.f : p.sx[(int64_t)m * p.ksplit + ks];
            ri = p.rinv[m];
        }
        float v[kTPW][4];
#pragma unroll
        for (int t = 0; t < kTPW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (GRP != 0)
                    v[t][r] = accf[t][tt][r] * ri;
                else
                    v[t][r] = sc[t][r] * (acc[t][tt][r] - zp[t][r] * sxm) * ri;
            }
        if (p.ksplit > 1) {  // partial sums of this K-slice; the SwiGLU pair keeps c_fc1 at column n, c_fc2 at N + n
            const int64_t ldp = PAIR ? 2 * (int64_t)p.N : (int64_t)p.N;
#pragma unroll
            for (int t = 0; t < kTPW; ++t) {
                const int n = tile[t] * 16 + 4 * g;
                if (n < p.N)
                    *(f32x4*)(p.part + ((int64_t)ks * p.M + m) * ldp + (PAIR && rr[t] == 1 ? p.N : 0) + n) =
                        f32x4{v[t][0], v[t][1], v[t][2], v[t][3]};
            }
            continue;
        }
        if constexpr (EPI == MI355_EPI_SWIGLU) {
#pragma unroll
            for (int t = 0; t < kTPW; t += 2) {
                const int n = tile[t] * 16 + 4 * g;
                if (n < p.N) {  // N % 4 == 0 (host check)
                    bf16_t* dst = (bf16_t*)p.y + (int64_t)m * p.ldy + n;
                    bf16_t ob[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) ob[r] = bf16_hw(swiglu_fast(v[t][r], v[t + 1][r]));
                    u32x2 o;
                    o[0] = (uint32_t)ob[0] | ((uint32_t)ob[1] << 16);
                    o[1] = (uint32_t)ob[2] | ((uint32_t)ob[3] << 16);
                    *(u32x2*)dst = o;
                    if constexpr (FUSE) es1[tt] += (bf16_to_f32(ob[0]) + bf16_to_f32(ob[1])) + (bf16_to_f32(ob[2]) + bf16_to_f32(ob[3]));
                }
            }
        } else {
            [[maybe_unused]] int posm = 0;
            if constexpr (FUSE && EPI == MI355_EPI_STORE) {
                if (p.rope != nullptr) posm = p.pos[m];
            }
#pragma unroll
            for (int t = 0; t < kTPW; ++t) {
                const int n = tile[t] * 16 + 4 * g;
                if (n >= p.N) continue;
                if constexpr (FUSE && EPI == MI355_EPI_STORE) {
                    // c_attn (model.py:197-221): rows [C, 2C) are k — rotated in f32 (model.py:306-323; the lane's four rows
                    // are two interleaved pairs) — rows [2C, 3C) are v; both go to their bf16 cache rows, q goes to y
                    const int sec = p.rope != nullptr ? (n >= p.C) + (n >= 2 * p.C) : 0;
                    if (sec != 0) {
                        const int cn = n - sec * p.C, h = cn >> 7, d = cn & 127;
                        float o0 = v[t][0], o1 = v[t][1], o2 = v[t][2], o3 = v[t][3];
                        if (sec == 1) {
                            const f32x4 cs = *(const f32x4*)(p.rope + (int64_t)(p.rope_gathered ? m : posm) * 128 + d);
                            const float k0 = o0 * cs[0] - o1 * cs[1], k1 = o1 * cs[0] + o0 * cs[1];
                            const float k2 = o2 * cs[2] - o3 * cs[3], k3 = o3 * cs[2] + o2 * cs[3];
                            o0 = k0, o1 = k1, o2 = k2, o3 = k3;
                        }
                        const int slot = posm < p.S - 1 ? posm : p.S - 1;
                        bf16_t* dst = (sec == 1 ? p.kcache : p.vcache) + ((int64_t)h * p.S + slot) * 128 + d;
                        u32x2 pk;
                        pk[0] = (uint32_t)bf16_hw(o0) | ((uint32_t)bf16_hw(o1) << 16);
                        pk[1] = (uint32_t)bf16_hw(o2) | ((uint32_t)bf16_hw(o3) << 16);
                        *(u32x2*)dst = pk;
                        continue;
                    }
                    if (p.rope != nullptr && p.q_scale != 0.f) {
                        store_q_ready(p, m, n, posm, v[t][0], v[t][1], v[t][2], v[t][3]);
                        continue;
                    }
                }
                if (p.y_dtype == MI355_F32) {
                    float* dst = (float*)p.y + (int64_t)m * p.ldy + n;
                    f32x4 o = {v[t][0], v[t][1], v[t][2], v[t][3]};
                    if constexpr (EPI == MI355_EPI_ACCUM) o += *(const f32x4*)dst;
                    *(f32x4*)dst = o;
                    if constexpr (FUSE && EPI == MI355_EPI_ACCUM) {
                        if (emit) {  // the next linear's operand: bf16(norm scale * x), as stage_rows_kernel rounds it
                            bf16_t ob[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) ob[r] = bf16_hw(o[r] * ns[t][r]);
                            u32x2 pk;
                            pk[0] = (uint32_t)ob[0] | ((uint32_t)ob[1] << 16);
                            pk[1] = (uint32_t)ob[2] | ((uint32_t)ob[3] << 16);
                            *(u32x2*)(p.out_xb + (int64_t)m * p.out_ld + n) = pk;
                            es1[tt] += (bf16_to_f32(ob[0]) + bf16_to_f32(ob[1])) + (bf16_to_f32(ob[2]) + bf16_to_f32(ob[3]));
                            es2[tt] += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
                        }
                    }
                } else {
                    bf16_t* dst = (bf16_t*)p.y + (int64_t)m * p.ldy + n;
                    float o[4] = {v[t][0], v[t][1], v[t][2], v[t][3]};
                    if constexpr (EPI == MI355_EPI_ACCUM) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] += bf16_to_f32(dst[r]);
                    }
                    u32x2 pk;
                    pk[0] = (uint32_t)bf16_hw(o[0]) | ((uint32_t)bf16_hw(o[1]) << 16);
                    pk[1] = (uint32_t)bf16_hw(o[2]) | ((uint32_t)bf16_hw(o[3]) << 16);
                    *(u32x2*)dst = pk;
                }
            }
        }
    }
    if constexpr (FUSE && EPI != MI355_EPI_STORE) {
        // partial sums of the block's rows per token: lane groups (permlane swaps), then the waves through LDS in wave
        // order (the activation buffers are free: every wave is past its last unit's barrier), one entry per row block
        if (sums) {
            // (a block of 256 rows — 8 waves, no pair — writes one share per 128 rows: the granularity a K-split consumer's
            // slices are cut at; smaller blocks are shares of their own)
            constexpr int kHalves = (!PAIR && kWaves == 8) ? 2 : 1;
            float* red = (float*)smem;  // [kWaves][2][BM]
#pragma unroll
            for (int tt = 0; tt < kTT; ++tt) {
                float a = es1[tt], b = es2[tt];
                a += lane_xor16(a);
                a += lane_xor32(a);
                b += lane_xor16(b);
                b += lane_xor32(b);
                if (g == 0) {
                    red[(wave * 2 + 0) * BM + tt * 16 + c] = a;
                    red[(wave * 2 + 1) * BM + tt * 16 + c] = b;
                }
            }
            __syncthreads();
            for (int idx = threadIdx.x; idx < kHalves * BM; idx += kThreads) {
                const int t = idx % BM, half = idx / BM;
                const int m = m0 + t;
                const int share = nb * kHalves + half;
                if (m < p.M && share * (kHalves == 2 ? 128 : (PAIR ? kSlots / 2 : kSlots) * 16) < p.N) {
                    float a = 0.f, b = 0.f;
#pragma unroll
                    for (int w = 0; w < kWaves / kHalves; ++w) {
                        a += red[((half * (kWaves / kHalves) + w) * 2 + 0) * BM + t];
                        b += red[((half * (kWaves / kHalves) + w) * 2 + 1) * BM + t];
                    }
                    p.out_sx[(int64_t)share * p.M + m] = a;
                    if (p.out_ss != nullptr) p.out_ss[(int64_t)share * p.M + m] = b;
                }
            }
        }
    }
}

template <int EPI, bool PAIR, int FMT, int kWaves, int BM, int GRP = 0, bool FUSE = false>
int launch_gemm_w(const GemmParams& p, hipStream_t s) {  // p.n_tiles: 16-row tiles (pair tiles for the SwiGLU stream)
    constexpr int kLdsW = 2 * BM * 256 + (FUSE ? 2 * BM * 4 + 4096 : 0);  // two activation buffers (+ the rows' factors and the shares they are summed from)
    static hipError_t attr_err = hipFuncSetAttribute((const void*)gemm_q4_kernel<EPI, PAIR, FMT, kWaves, BM, GRP, FUSE>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLdsW);
    if (attr_err != hipSuccess) {
        mi355_set_error("hipFuncSetAttribute(gemm) failed: %s", hipGetErrorString(attr_err));
        return (int)attr_err;
    }
    constexpr int kSlots = kWaves * kTPW;
    const int per_block = PAIR ? kSlots / 2 : kSlots;
    GemmParams q = p;
    q.n_blocks = (p.n_tiles + per_block - 1) / per_block;
    q.total_blocks = q.n_blocks * ((p.M + BM - 1) / BM);
    q.per_xcd = (q.total_blocks * q.ksplit + 7) / 8;
    if constexpr (FUSE) {
        if (q.ksplit > 1) {  // the producer roles move to the reduction of the K-slices
            q.out_xb = nullptr;
            q.out_ss = q.out_sx = nullptr;
            q.rope = nullptr;
        }
    }
    hipLaunchKernelGGL((gemm_q4_kernel<EPI, PAIR, FMT, kWaves, BM, GRP, FUSE>), dim3(8 * q.per_xcd), dim3(64 * kWaves), kLdsW, s, q);
    MI355_LAUNCH_CHECK();
    if constexpr (FUSE) {
        if (q.ksplit > 1) {
            const int64_t groups = (int64_t)p.M * (p.N >> 7);
            const int grid = (int)((groups + 7) / 8 < 4096 ? (groups + 7) / 8 : 4096);
            hipLaunchKernelGGL(splitk_fused_reduce_kernel<EPI>, dim3(grid), dim3(256), 0, s, p);
            MI355_LAUNCH_CHECK();
            return 0;
        }
    }
    if (q.ksplit > 1) {
        const int64_t n = (int64_t)p.M * (p.N >> 2);
        const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel<EPI>, dim3(grid), dim3(256), 0, s, q.part, q.ksplit, p.M, p.N, p.y, p.y_dtype, p.ldy);
        MI355_LAUNCH_CHECK();
    }
    return 0;
}
// The shape of an ungrouped launch: blocks of 16 row tiles x 128 tokens when that fills the chip (256 CUs x 2 workgroups);
// anything smaller — the N = 4096 outputs of a long prompt (16 row blocks: at T = 2048 one workgroup per CU, matrix pipes
// 37 % busy against 53 % for the c_fc1 / c_fc2 pair), and every launch of a short prompt, K-split or not — takes 64-token
// blocks: twice the workgroups, half the LDS each.  Sweep over (waves, tokens per block, K-slices) at 128 / 256 / 512 tokens
// (scripts/sweep_gemm_shapes.sh, profiles/r04_gemm_shape_sweep.txt): 64-token blocks win every linear (one layer's linears
// 216 -> 183 / 269 -> 226 / 361 -> 320 us), the 2-wave blocks of round 3 never do; those stay for launches too small to split.
struct GemmShape {
    int waves, bm;
};
// tuning override for sweeps (scripts/sweep_gemm_shapes.sh): MI355_GEMM_FORCE="waves:bm:ksplit[:bf16_blocks[:q4_blocks]]", 0 = the rule's choice
// (fields 4 / 5: the block counts below which the BF16 / int4 streams take 64-token blocks)
struct GemmForce {
    int waves, bm, ksplit, bf16_blocks, q4_blocks;  // *_blocks: the 64-token-block thresholds of the two formats (sweeps)
};
GemmForce gemm_force() {
    // parsed ONCE per process (advisor r4: a getenv + sscanf per launch sat on the production path, and a change of the variable
    // between mi355_linear_gemm_plan and the launch would have desynchronised the share layout of the fused chain; the sweep starts
    // one process per setting)
    static const GemmForce f = [] {
        GemmForce g = {0, 0, 0, 0, 0};
        if (const char* e = getenv("MI355_GEMM_FORCE")) sscanf(e, "%d:%d:%d:%d:%d", &g.waves, &g.bm, &g.ksplit, &g.bf16_blocks, &g.q4_blocks);
        return g;
    }();
    return f;
}
GemmShape gemm_shape(int n_tiles, int M, bool pair, int ksplit, bool bf16 = false) {
    const GemmForce fo = gemm_force();
    if ((fo.waves == 8 && (fo.bm == 64 || fo.bm == kBM)) || ((fo.waves == 2 || fo.waves == 1) && (fo.bm == 0 || fo.bm == kBM)))
        return {fo.waves, fo.bm ? fo.bm : kBM};
    const int per_block8 = pair ? 8 * kTPW / 2 : 8 * kTPW;
    const int blocks8 = ((n_tiles + per_block8 - 1) / per_block8) * ((M + kBM - 1) / kBM);
    if (ksplit == 1 && blocks8 < 96) {  // (K < 1024: nothing to split)
        if (blocks8 * 4 >= 128) return {2, kBM};
        return {1, kBM};
    }
    // (only the WAVES of the answer enter mi355_linear_gemm_plan's share layout: the format may pick its own token block)
    // BF16 streams are bound by the weight bytes a CU requests per MFMA (pieces four times an int4 one's, a 64-token block requests twice
    // a 128-token block's): they keep 128 tokens as long as those blocks cover 5/8 of the CUs, K-slices or not (7B: 2048 tokens 28.5 -> 25.6 ms, 1536 23.2 -> 22.0,
    // 512 10.6 -> 9.8; at 128 blocks — N = 4096 at 1024 tokens — 64 tokens win, 15.2 vs 16.7 ms; profiles/r06_bf16_gemm_tilings.txt)
#ifndef MI355_GEMM_NO_BM64
    // int4 streams: 64-token blocks below 384 blocks — unless the 128-token blocks are exactly one per CU (N = 4096 at 2048 tokens:
    // no tail; 2048-token prompt 21.85 -> 21.5 ms, where a plain threshold of 256 cost 2 % at 384 / 512 tokens: profiles/r06_q4_gemm_tilings.txt)
    const bool one_per_cu = blocks8 * ksplit == 256 && MI355_GEMM_ONE_PER_CU;
    if (bf16 ? blocks8 < (fo.bf16_blocks ? fo.bf16_blocks : 160) : (blocks8 * ksplit < (fo.q4_blocks ? fo.q4_blocks : 384) && !one_per_cu)) return {8, 64};
#endif
    return {8, kBM};
}
// split-K of an ungrouped launch: fewer than 96 blocks (a 128-token prompt against N = 4096 is 16) cut K into up to 8
// slices, as many as bring the launch to ~128 workgroups and fit the partial buffer
int gemm_ksplit(int M, int N, int K, bool swiglu) {
    const int units = (K + 127) / 128;
    if (const int fk = gemm_force().ksplit) {
        if (fk >= 1 && fk <= kMaxSplit && units >= fk && (size_t)2 * fk * M * (size_t)N * (swiglu ? 2 : 1) * 4 <= kSplitBudget) return fk;
    }
    const int rows_per_block = swiglu ? 16 * 8 * kTPW / 2 : 16 * 8 * kTPW;
    const int blocks8 = ((N + rows_per_block - 1) / rows_per_block) * ((M + kBM - 1) / kBM);
    const size_t row_floats = (size_t)N * (swiglu ? 2 : 1);
    int ksplit = 1;
    while (ksplit < kMaxSplit && blocks8 * ksplit < 96 && units >= 8 * ksplit && (size_t)2 * ksplit * M * row_floats * 4 <= kSplitBudget)
        ksplit *= 2;
    return ksplit;
}
template <int EPI, bool PAIR, int FMT, bool FUSE>
int launch_gemm(const GemmParams& p, hipStream_t s) {
    const GemmShape sh = gemm_shape(p.n_tiles, p.M, PAIR, p.ksplit, FMT == MI355_W_BF16);
    if (sh.waves == 8) {
        if (sh.bm == 64) return launch_gemm_w<EPI, PAIR, FMT, 8, 64, 0, FUSE>(p, s);
        return launch_gemm_w<EPI, PAIR, FMT, 8, kBM, 0, FUSE>(p, s);
    }
    if (sh.waves == 2) return launch_gemm_w<EPI, PAIR, FMT, 2, kBM, 0, FUSE>(p, s);
    return launch_gemm_w<EPI, PAIR, FMT, 1, kBM, 0, FUSE>(p, s);
}
// grouped scales: blocks of 64 tokens x 4 waves (the second accumulator set costs the registers of more tokens or waves);
// from kGrpWideM tokens on, 128 tokens x 8 waves (2048-token g128 prompt 41.9 -> 39.4 ms; at 512 tokens it is the slower
// one, 16.3 -> 17.6 ms, at 1024 22.5 -> 27.3 ms, and 64 x 8 / 128 x 4 lose everywhere: profiles/r06_ab_grouped_prompt.txt)
constexpr int kGrpBM = 64;
constexpr int kGrpWideM = 2048;
template <int EPI, bool PAIR, int GRP>
int launch_gemm_grouped(const GemmParams& p, hipStream_t s) {
    const int per_block8 = PAIR ? 8 * kTPW / 2 : 8 * kTPW;
    const int blocks8 = ((p.n_tiles + per_block8 - 1) / per_block8) * ((p.M + kGrpBM - 1) / kGrpBM);
    if (p.ksplit > 1) return launch_gemm_w<EPI, PAIR, MI355_W_Q4, 4, 64, GRP>(p, s);
    if constexpr (GRP == 1) {  // groups below a unit (GRP 2) carry masked passes: 128 x 8 does not fit their registers
        if (blocks8 >= 128 && p.M >= kGrpWideM) return launch_gemm_w<EPI, PAIR, MI355_W_Q4, 8, 128, GRP>(p, s);
    }
    if (blocks8 >= 128) return launch_gemm_w<EPI, PAIR, MI355_W_Q4, 4, kGrpBM, GRP>(p, s);
    if (blocks8 * 4 >= 128) return launch_gemm_w<EPI, PAIR, MI355_W_Q4, 2, 64, GRP>(p, s);
    return launch_gemm_w<EPI, PAIR, MI355_W_Q4, 1, 64, GRP>(p, s);
}
template <int GRP>
int launch_gemm_grouped_epi(const GemmParams& p, int epi, hipStream_t s) {
    if (epi == MI355_EPI_SWIGLU) return launch_gemm_grouped<MI355_EPI_SWIGLU, true, GRP>(p, s);
    if (epi == MI355_EPI_ACCUM) return launch_gemm_grouped<MI355_EPI_ACCUM, false, GRP>(p, s);
    return launch_gemm_grouped<MI355_EPI_STORE, false, GRP>(p, s);
}
template <int FMT, bool FUSE>
int launch_gemm_epi(const GemmParams& p, int epi, hipStream_t s) {
    if (epi == MI355_EPI_SWIGLU) return launch_gemm<MI355_EPI_SWIGLU, true, FMT, FUSE>(p, s);
    if (epi == MI355_EPI_ACCUM) return launch_gemm<MI355_EPI_ACCUM, false, FMT, FUSE>(p, s);
    return launch_gemm<MI355_EPI_STORE, false, FMT, FUSE>(p, s);
}

}  // namespace

extern "C" size_t mi355_linear_gemm_workspace_bytes(int M, int K) {
    if (M <= 0 || K <= 0) return 0;
    const size_t kp = ((size_t)K + 127) / 128 * 128;
    // staged operands, 1/rms, per-slice operand sums, split-K partials (used when a launch would be a few dozen blocks)
    // ... and, at the very end, the partial sums of the fused prompt chain (gemm_fuse.h)
    return (size_t)M * kp * 2 + (size_t)M * 4 * (1 + kMaxSplit) + 256 + kSplitBudget + kFuseScratch;
}

size_t mi355_linear_gemm_fuse_scratch_bytes() { return kFuseScratch; }

extern "C" int mi355_linear_gemm(const mi355_linear_args* a, void* workspace, size_t workspace_bytes,
                                 mi355_stream_t stream) {
    return mi355_linear_gemm_fused(a, nullptr, workspace, workspace_bytes, stream);
}

void mi355_linear_gemm_plan(int M, int N, int K, int R, int* ksplit, int* shares, int* shares_per_unit) {
    const bool pair = R == 2;
    const int ks = gemm_ksplit(M, N, K, pair);
    const GemmShape sh = gemm_shape((N + 15) / 16, M, pair, ks);
    // rows (hidden rows of a pair stream) behind one share of a producer's partial sums: a 128-column unit when the K-slices'
    // reduction writes them or a block holds 256 rows, else the block's rows
    int rows = (pair ? sh.waves * kTPW / 2 : sh.waves * kTPW) * 16;
    if (ks > 1 || rows > 128) rows = 128;
    *ksplit = ks;
    *shares = (N + rows - 1) / rows;
    *shares_per_unit = 128 / rows;
}

int mi355_linear_gemm_fused(const mi355_linear_args* a, const mi355_gemm_fuse* f, void* workspace, size_t workspace_bytes,
                            mi355_stream_t stream) {
    MI355_CHECK_ARG(a != nullptr && workspace != nullptr, MI355_E_ARG, "linear_gemm: null argument");
    MI355_CHECK_ARG(a->fmt == MI355_W_Q4 || a->fmt == MI355_W_BF16, MI355_E_ARG,
                    "linear_gemm: the wide path handles the Q4 and BF16 streams (fmt %d)", a->fmt);
    const bool q4 = a->fmt == MI355_W_Q4;
    MI355_CHECK_ARG(a->w && a->x && a->y && (!q4 || (a->scales && a->zeros)), MI355_E_ARG, "linear_gemm: null w/x/y/scales/zeros");
    // grouped scales: group_cols = input columns per (scale, zero) pair; 0 or >= K: one pair per output row
    const bool grouped = q4 && a->group_cols > 0 && a->group_cols < a->K;
    int gq_shift = 0;
    if (grouped) {
        while ((32 << gq_shift) < a->group_cols) ++gq_shift;
        MI355_CHECK_ARG((32 << gq_shift) == a->group_cols && a->K % 128 == 0, MI355_E_SHAPE,
                        "linear_gemm: group size %d is not 32 * 2^n or K %% 128 != 0 (use the generic kernel)", a->group_cols);
        MI355_CHECK_ARG(a->sz_dtype == MI355_BF16, MI355_E_DTYPE, "linear_gemm: grouped scales / zeros must be bf16");
    }
    MI355_CHECK_ARG(a->M >= 1 && a->N > 0 && a->K > 0, MI355_E_SHAPE, "linear_gemm: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
    MI355_CHECK_ARG(a->attn_partials == nullptr && a->bias == nullptr, MI355_E_ARG, "linear_gemm: no bias / attention prologue");
    const bool swiglu = a->epi == MI355_EPI_SWIGLU;
    MI355_CHECK_ARG(a->epi >= MI355_EPI_STORE && a->epi <= MI355_EPI_SWIGLU, MI355_E_ARG, "linear_gemm: bad epi");
    MI355_CHECK_ARG(swiglu ? (a->R == 2 && (!q4 || (a->scales2 && a->zeros2)) && a->y_dtype == MI355_BF16) : a->R == 1, MI355_E_ARG,
                    "linear_gemm: STORE / ACCUM take the R = 1 stream, SWIGLU the interleaved R = 2 stream with bf16 output");
    MI355_CHECK_ARG(a->N % 4 == 0 && a->ldy % 4 == 0, MI355_E_SHAPE, "linear_gemm: N and ldy must be multiples of 4");
    auto two = [](int d) { return d == MI355_F32 || d == MI355_BF16; };
    MI355_CHECK_ARG(two(a->x_dtype) && two(a->y_dtype) && (!q4 || two(a->sz_dtype)), MI355_E_DTYPE, "linear_gemm: dtypes must be f32 or bf16");
    MI355_CHECK_ARG(a->norm_scale == nullptr || two(a->norm_dtype), MI355_E_DTYPE, "linear_gemm: norm scale dtype");
    MI355_CHECK_ARG(workspace_bytes >= mi355_linear_gemm_workspace_bytes(a->M, a->K) && (uintptr_t)workspace % 16 == 0,
                    MI355_E_SHAPE, "linear_gemm: workspace of %zu bytes is too small (need %zu)", workspace_bytes,
                    mi355_linear_gemm_workspace_bytes(a->M, a->K));
    hipStream_t s = (hipStream_t)stream;
    const int units = (a->K + 127) / 128, kp = units * 128;
    bf16_t* xb = (bf16_t*)workspace;
    float* rinv = (float*)((char*)workspace + (size_t)a->M * kp * 2);
    float* sx = rinv + a->M;
    // split-K: launches of fewer than 96 blocks (a 128-token prompt against N = 4096 is 16) cut K into up to 8
    // slices, as many as bring the launch to ~128 workgroups and fit the partial buffer
    int ksplit = 1;
    size_t tab_bytes = 0, sxt_bytes = 0;
    int n_groups = 0, tab_ld = 0;
    if (grouped) {
        n_groups = (a->K + a->group_cols - 1) / a->group_cols;
        tab_ld = (a->N + 15) / 16 * 16;
        tab_bytes = ((size_t)(n_groups + 1) * a->R * tab_ld * 4 + 15) & ~(size_t)15;
        sxt_bytes = (size_t)n_groups * a->M * 4;
        MI355_CHECK_ARG(tab_bytes + sxt_bytes <= kSplitBudget, MI355_E_SHAPE,
                        "linear_gemm: group tables of %zu B exceed the workspace region (%zu B)", tab_bytes + sxt_bytes, kSplitBudget);
        // blocks of 8 row tiles x 64 tokens; slices of whole groups (>= 4 per slice), partials in the first half of the
        // region, the tables in the second
        const int rows_per_block = swiglu ? 16 * 4 * kTPW / 2 : 16 * 4 * kTPW;
        const int blocks = ((a->N + rows_per_block - 1) / rows_per_block) * ((a->M + 63) / 64);
        const size_t row_floats = (size_t)a->N * (swiglu ? 2 : 1);
        const int cuts = a->group_cols >= 128 ? n_groups : units;
        while (ksplit < kMaxSplit && blocks * ksplit < 256 && cuts >= 8 * ksplit && tab_bytes + sxt_bytes <= kSplitBudget / 2 &&
               (size_t)2 * ksplit * a->M * row_floats * 4 <= kSplitBudget / 2)
            ksplit *= 2;
    } else {
        ksplit = gemm_ksplit(a->M, a->N, a->K, swiglu);
    }
    // producer / consumer fusion (gemm_fuse.h): whole-K launches over per-row scales only — the caller plans with
    // mi355_linear_gemm_plan
    const bool fused = f != nullptr;
    const bool f_in = fused && f->prestaged;
    if (fused) {
        MI355_CHECK_ARG(!grouped, MI355_E_STATE, "linear_gemm: fusion over a grouped launch (M=%d N=%d K=%d)", a->M, a->N, a->K);
        const bool produces = f->out_xb != nullptr || f->out_sx != nullptr || f->rope != nullptr;
        MI355_CHECK_ARG(!produces || a->N % 128 == 0, MI355_E_SHAPE, "linear_gemm: a producer of the fused chain has N %% 128 == 0 (N=%d)", a->N);
        MI355_CHECK_ARG(ksplit == 1 || !produces || a->epi == MI355_EPI_SWIGLU || a->y_dtype == MI355_F32, MI355_E_DTYPE,
                        "linear_gemm: the fused reduction of a K-split launch writes f32 rows (or the SwiGLU pair's bf16)");
        MI355_CHECK_ARG(!f_in || ksplit == 1 || (f->in_ppu >= 1 && f->in_sx_n == units * f->in_ppu), MI355_E_ARG,
                        "linear_gemm: a K-split consumer needs %d x in_ppu partial operand sums (got %d, in_ppu %d)", units, f->in_sx_n, f->in_ppu);
        MI355_CHECK_ARG(!f_in || (a->x_dtype == MI355_BF16 && a->norm_scale == nullptr && a->K % 128 == 0 && a->ldx % 8 == 0 &&
                                  (uintptr_t)a->x % 16 == 0 && f->in_sx != nullptr && f->in_sx_n > 0 && (f->in_ss == nullptr || f->in_ss_n > 0)),
                        MI355_E_ARG, "linear_gemm: a pre-staged operand is bf16 [M, ldx] with K %% 128 == 0 and partial operand sums");
        MI355_CHECK_ARG(f->out_xb == nullptr || (a->epi == MI355_EPI_ACCUM && a->y_dtype == MI355_F32 && f->next_norm && f->out_ss &&
                                                 f->out_sx && f->out_ld % 4 == 0 && (void*)f->out_xb != a->x &&
                                                 (f_in || (void*)f->out_xb != workspace)),
                        MI355_E_ARG, "linear_gemm: the next operand is emitted by the f32 residual epilogue, next to its partial sums");
        MI355_CHECK_ARG(f->out_sx == nullptr || a->epi == MI355_EPI_SWIGLU || f->out_xb != nullptr, MI355_E_ARG,
                        "linear_gemm: partial operand sums come from the SwiGLU or the emitting residual epilogue");
        MI355_CHECK_ARG(f->rope == nullptr || (a->epi == MI355_EPI_STORE && f->pos && f->kcache && f->vcache && f->hs == 128 && f->S > 0 &&
                                               a->N == 3 * f->n_head * f->hs),
                        MI355_E_ARG, "linear_gemm: the K / V cache epilogue is c_attn's (N = 3 n_head x 128, STORE)");
    }
    float* part = (float*)((char*)workspace + (size_t)a->M * kp * 2 + (size_t)a->M * 4 * (1 + kMaxSplit) + 256);
    part = (float*)(((uintptr_t)part + 15) & ~(uintptr_t)15);
    GemmParams p;
    memset(&p, 0, sizeof(p));
    if (grouped) {
        // the group tables live in the split-K region of the workspace (its second half when the launch is split)
        uint32_t* gtab = (uint32_t*)((char*)part + (ksplit > 1 ? kSplitBudget / 2 : 0));
        float* sxt = (float*)((char*)gtab + tab_bytes);
        {
            const int64_t total = (int64_t)(n_groups + 1) * a->R * tab_ld;
            const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
            hipLaunchKernelGGL(group_table_kernel, dim3(grid), dim3(256), 0, s, (const bf16_t*)a->scales, (const bf16_t*)a->zeros,
                               (const bf16_t*)a->scales2, (const bf16_t*)a->zeros2, a->N, tab_ld, n_groups, a->R, gtab);
            MI355_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(stage_rows_grouped_kernel, dim3(a->M), dim3(256), 0, s, a->x, a->x_dtype, a->ldx, a->norm_scale,
                           a->norm_dtype, a->eps, a->K, kp, xb, (int64_t)kp, rinv, sxt, a->M, a->group_cols, n_groups);
        MI355_LAUNCH_CHECK();
        p.gtab = gtab;
        p.sxt = sxt;
        p.n_groups = n_groups;
        p.gq_shift = gq_shift;
        p.upg = a->group_cols >= 128 ? a->group_cols / 128 : 1;
        p.tab_ld = tab_ld;
    } else if (!f_in) {
        hipLaunchKernelGGL(stage_rows_kernel, dim3(a->M), dim3(256), 0, s, a->x, a->x_dtype, a->ldx, a->norm_scale, a->norm_dtype,
                           a->eps, a->K, kp, xb, (int64_t)kp, rinv, sx, ksplit);
        MI355_LAUNCH_CHECK();
    }
    p.w = (const uint8_t*)a->w;
    {
        const size_t wb = mi355_packed_bytes(a->fmt, a->N, a->K, a->R, swiglu ? 1 : 0);
        MI355_CHECK_ARG(wb > 0 && wb < 0xFFFFFFF0ull, MI355_E_SHAPE, "linear_gemm: weight stream of %zu B", wb);
        p.w_bytes = (unsigned)wb;
    }
    MI355_CHECK_ARG((size_t)a->M * (f_in ? (size_t)a->ldx : (size_t)kp) * 2 < 0xFFFFFFF0ull, MI355_E_SHAPE,
                    "linear_gemm: M x K too large for one launch");
    p.xb = f_in ? (const bf16_t*)a->x : xb;
    p.ldxb = f_in ? a->ldx : (int64_t)kp;
    p.rinv = rinv;
    p.sx = sx;
    p.scales = a->scales;
    p.zeros = a->zeros;
    p.scales2 = a->scales2;
    p.zeros2 = a->zeros2;
    p.y = a->y;
    p.ldy = a->ldy;
    p.M = a->M;
    p.N = a->N;
    p.K = a->K;
    p.units = units;
    p.n_tiles = (a->N + 15) / 16;
    p.sz_dtype = a->sz_dtype;
    p.y_dtype = a->y_dtype;
    p.ksplit = ksplit;
    p.part = part;
    if (grouped) return a->group_cols >= 128 ? launch_gemm_grouped_epi<1>(p, a->epi, s) : launch_gemm_grouped_epi<2>(p, a->epi, s);
    if (fused) {
        p.f_in = f_in ? 1 : 0;
        p.in_sx = f->in_sx;
        p.in_sx_n = f_in ? f->in_sx_n : 0;
        p.in_ppu = f_in ? f->in_ppu : 0;
        p.in_ss = f->in_ss;
        p.in_ss_n = f_in && f->in_ss != nullptr ? f->in_ss_n : 0;
        p.eps = a->eps;
        p.out_xb = f->out_xb;
        p.out_ld = f->out_ld;
        p.next_norm = f->next_norm;
        p.next_norm_dtype = f->next_norm_dtype;
        p.out_ss = f->out_ss;
        p.out_sx = f->out_sx;
        p.rope = f->rope;
        p.pos = f->pos;
        p.kcache = f->kcache;
        p.vcache = f->vcache;
        p.S = f->S;
        p.C = f->n_head * f->hs;
        p.rope_gathered = f->rope_gathered;
        p.q_scale = f->rope != nullptr ? f->q_scale : 0.f;
        if (q4) return launch_gemm_epi<MI355_W_Q4, true>(p, a->epi, s);
        return launch_gemm_epi<MI355_W_BF16, true>(p, a->epi, s);
    }
    if (q4) return launch_gemm_epi<MI355_W_Q4, false>(p, a->epi, s);
    return launch_gemm_epi<MI355_W_BF16, false>(p, a->epi, s);
}

// LLM.int8 linear (Linear8bitLt) for gfx950, M <= 16.
//
// Replaces bitsandbytes' MatMul8bitLt forward behind lit_llama.quantization.Linear8bitLt
// (/root/reference lit_llama/quantization.py:38-77; has_fp16_weights=False, threshold=6.0) — five CUDA
// launches there (double_quant, transform, igemmlt, mm_dequant, outlier fp16 matmul) — by ONE kernel with
// the same streaming skeleton as gemv.hip:
//   prologue : x -> f16; outlier columns {k : |x[m,k]| >= threshold for some m}; SCA[m] = max |x[m,k]|
//              over sub-threshold entries; CA = rint(x * (127 / SCA)) with outlier columns zeroed  (LDS)
//   stream   : int8 weights, 1-KiB coalesced wave loads, v_mfma_i32_16x16x64_i8 (exact int32)
//   epilogue : f16(((acc * 1/127^2) * SCA[m]) * SCB[n] + bias)  +  f16(sum_{k in outliers} x[m,k] *
//              f16(CB[n,k] * SCB[n] / 127)), added in f16, then cast to the output dtype.
// bitsandbytes is not vendored, pinned or tested by the reference: this arithmetic is the restatement
// written down in oracle/oracle.py (`int8_quant_rows`, `llm_int8_linear`; "parity unpinned").
#include <mutex>

#include "common.h"

namespace {

constexpr int kUnitK = 128;
constexpr int kMaxM = 16;
constexpr int kHdr = 512;
constexpr int kMaxLds = 160 * 1024;

struct I8Params {
    const uint8_t* w;
    const float* scb;
    const float* scb2;
    const void* x;
    const void* norm_scale;
    const void* bias;
    void* y;
    int64_t ldx, ldy;
    int N, K, M, n_tiles, units;
    int u_q, u_r, t_q, t_r;  // units / waves, units % waves, n_tiles / grid, n_tiles % grid
    int x_dtype, norm_dtype, bias_dtype, y_dtype, epi;
    int xq_stride;  // bytes per int8 activation row in LDS
    unsigned w_bytes;  // size of the weight stream (buffer descriptor bound)
    float eps, threshold;
    uint64_t* dbg;
    const float* attn_part;  // split-attention partial records feeding the row (M = 1), or nullptr
    int attn_splits, attn_heads, attn_hs;
    int vec;  // 1: 16-B / 8-B vector staging of x (aligned rows, bf16 norm scale), see mi355_linear_int8
};

__device__ __forceinline__ float f16r(float v) { return f16_to_f32(f32_to_f16(v)); }

constexpr int kFastOut = 8;  // outlier columns whose weight bytes are prefetched one tile ahead (registers)
constexpr int kStageVec = 8;  // 4-column activation vectors a thread keeps in registers (decode: K <= 32 * threads)

// LDS map: [hdr: red[64] | sca[16] | sinv[16] | ototal] [part 2*W*R KiB] [xq M*xq_stride] [xh M*Kp f16]
//          [olist Kp u16: outlier columns, ascending] [obits Kp/8 B: outlier bit set]
// VNV: 0 element loop, else 4-column vectors per thread kept in registers.  FAST: the decode step (one row that
// fits the registers) as its own kernel, without the row / chunk loops of the prompt path (cold instruction cache:
// every instruction of a launch's prologue is paid at memory latency).
template <int R, int P, int VNV, bool FAST>
__global__ __launch_bounds__(512) void int8_gemv_kernel(const I8Params p) {
    const int M = FAST ? 1 : p.M;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)smem;
    float* sca = (float*)(smem + 256);
    float* sinv = (float*)(smem + 320);
    int* ototal = (int*)(smem + 384);
    char* part = smem + kHdr;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = blockDim.x >> 6;
#define MI355_STAMP(i)                                                                          \
    do {                                                                                        \
        if (p.dbg != nullptr && threadIdx.x == 0) p.dbg[blockIdx.x * 8 + (i)] = wall_clock64(); \
    } while (0)
    MI355_STAMP(0);
    const int units = p.units, Kp = units * kUnitK;
    char* xq = part + 2 * W * R * 1024;
    f16_t* xh = (f16_t*)(xq + (size_t)M * p.xq_stride);
    uint16_t* olist = (uint16_t*)(xh + (size_t)M * Kp);
    unsigned* obits = (unsigned*)(olist + Kp);

    // K split over the waves / tiles over the workgroups from host-computed quotients (see gemv.hip)
    const int nu = p.u_q + (wave < p.u_r ? 1 : 0);
    const int u0 = wave * p.u_q + (wave < p.u_r ? wave : p.u_r), u1 = u0 + nu;
    const int bid = blockIdx.x, nb = gridDim.x;
    const int my_tiles = p.t_q + (bid < p.t_r ? 1 : 0);
    const int total = my_tiles * nu;

    constexpr int kSlot = R * 2;

    // unconditional refills through a buffer descriptor (out-of-range -> zeros, no memory request), see gemv.hip
    u32x4 ring[P][kSlot];
    int pf_tile = bid, pf_u = u0, pf_n = 0;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    const unsigned lane_off = lane * 16;
    const unsigned unit_bytes32 = (unsigned)kSlot * 1024u;
#define MI355_ISSUE(slot)                                                                                       \
    do {                                                                                                        \
        const bool ok__ = pf_n < total;                                                                         \
        const unsigned off__ = ok__ ? ((unsigned)pf_tile * (unsigned)units + (unsigned)pf_u) * unit_bytes32 + lane_off \
                                    : 0xFFFFF000u;                                                              \
        _Pragma("unroll") for (int s__ = 0; s__ < kSlot; ++s__) ring[slot][s__] = __builtin_bit_cast(           \
            u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off__ + s__ * 1024, 0, 2));                      \
        ++pf_n;                                                                                                 \
        if (ok__ && ++pf_u == u1) {                                                                             \
            pf_u = u0;                                                                                          \
            pf_tile += nb;                                                                                      \
        }                                                                                                       \
    } while (0)
    // Vector staging: this thread's first activation vectors (and their norm scales) are requested BEFORE the
    // weight ring, because VMEM returns in order and the prologue must not queue behind ~32 KiB of weights.
    constexpr bool VEC = VNV > 0;
    constexpr int NV = VEC ? VNV : 1;
    const int nthr = blockDim.x;
    const int nvecp = Kp >> 2;
    const bool x32 = p.x_dtype == MI355_F32;
    // loads go through buffer descriptors: vectors past the row end return zeros without a branch (a
    // conditional load is a separate basic block and the compiler drains vmcnt at every join)
    [[maybe_unused]] u32x4 xr[NV];
    [[maybe_unused]] u32x2 nr[NV];
    const int x_esz = x32 ? 4 : 2;
    [[maybe_unused]] auto load_chunk = [&](int m, int c) {
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((const char*)p.x + (int64_t)m * p.ldx * x_esz), 0, p.K * x_esz, 0x00020000);
        if (x32) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                xr[i] = __builtin_bit_cast(
                    u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ((c * NV + i) * nthr + tid) * 16, 0, 0));
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const u32x2 t = __builtin_bit_cast(
                    u32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, ((c * NV + i) * nthr + tid) * 8, 0, 0));
                xr[i] = u32x4{t[0], t[1], 0u, 0u};
            }
        }
    };
    // split-attention input (attn.c_proj of a decode step): per vector the <= 4 partial records of its head
    constexpr bool PARTS = VNV == 2;  // only the short-row variant carries the 48 extra registers
    [[maybe_unused]] f32x4 po[PARTS ? NV : 1][4];
    [[maybe_unused]] float pm[PARTS ? NV : 1][4], pl[PARTS ? NV : 1][4];
    const bool from_parts = PARTS && p.attn_part != nullptr;
    if constexpr (VEC) {
        if (from_parts) {
            if constexpr (PARTS) {
                const int rs = p.attn_hs + 4, last = (p.K >> 2) - 1;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    int v = i * nthr + tid;
                    v = v < last ? v : last;  // clamped, not branched: rows past the end are zeroed in decode()
                    const int k0 = v * 4, h = k0 / p.attn_hs, d0 = k0 - h * p.attn_hs;
                    const float* rec = p.attn_part + (size_t)h * p.attn_splits * rs;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float* q = rec + (j < p.attn_splits ? j : p.attn_splits - 1) * rs;
                        pm[i][j] = q[0];
                        pl[i][j] = q[1];
                        po[i][j] = *(const f32x4*)(q + 4 + d0);
                    }
                }
            }
        } else {
            load_chunk(0, 0);
        }
        const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.norm_scale != nullptr ? p.norm_scale : p.x), 0, p.norm_scale != nullptr ? p.K * 2 : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < NV; ++i)
            nr[i] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rn, (i * nthr + tid) * 8, 0, 0));
    }
#pragma unroll
    for (int j = 0; j < P; ++j) MI355_ISSUE(j);
    MI355_STAMP(1);

    // ---------------- prologue: f16 activations, outlier columns, row scales, int8 quantisation
    // Outlier columns {k : |x[m,k]| >= threshold for some row m} are a bit set in LDS (atomic OR by whichever
    // thread stages the column; there are only a handful), from which wave 0 later writes the ascending list.
    for (int i = tid; i < (Kp >> 5); i += nthr) obits[i] = 0u;
    if (tid < 64) red[tid] = 0.f;
    if (tid == 0) ototal[0] = 0;
    __syncthreads();
    const bool thr_on = p.threshold > 0.f;
    // wave reductions: DPP inside 16-lane rows, two ds_bpermute steps across rows; one barrier per block reduction
    // (alternating scratch slots, so a slot is rewritten only after two later barriers)
    auto blk_reduce = [&](float v, int slot, bool is_max) -> float {
        if (is_max) {
            v = MI355_DPP_MAX(v, 0xB1);
            v = MI355_DPP_MAX(v, 0x4E);
            v = MI355_DPP_MAX(v, 0x141);
            v = MI355_DPP_MAX(v, 0x140);
            v = fmaxf(v, lane_xor16(v));
            v = fmaxf(v, lane_xor32(v));
        } else {
            v = group_sum(v, 64);
        }
        float* r = red + slot * 16;
        if (lane == 0) r[wave] = v;
        __syncthreads();
        // W <= 8; unused slots hold 0 (identity of both reductions: sums of squares and absolute maxima), so the
        // eight reads are unconditional and overlap
        const f32x4 a = *(const f32x4*)r, b = *(const f32x4*)(r + 4);
        if (is_max) return fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3])));
        return ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
    };
    bool quantised = false;
    // raw vectors -> f32
    [[maybe_unused]] auto decode = [&](float (&xf)[NV][4]) {
        if (from_parts) {
            if constexpr (PARTS) {
                // x[h*hs + d] = sum_j e^{m_j - M} o_j[d] / sum_j e^{m_j - M} l_j (flash-decoding combine), rounded
                // to bf16 like the attention kernel's own output
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    float M_ = -1.0e30f, L = 0.f, acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < p.attn_splits) {
                            const float Mn = fmaxf(M_, pm[i][j]);
                            const float c_old = expf(M_ - Mn), c_new = expf(pm[i][j] - Mn);
                            L = L * c_old + pl[i][j] * c_new;
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[e] = acc[e] * c_old + po[i][j][e] * c_new;
                            M_ = Mn;
                        }
                    const float inv = 1.0f / L;
                    const bool live = i * nthr + tid < (p.K >> 2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xf[i][e] = live ? bf16_to_f32(f32_to_bf16(acc[e] * inv)) : 0.f;
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (x32) {
#pragma unroll
                for (int j = 0; j < 4; ++j) xf[i][j] = __uint_as_float(xr[i][j]);
            } else if (p.x_dtype == MI355_BF16) {
                xf[i][0] = __uint_as_float(xr[i][0] << 16);
                xf[i][1] = __uint_as_float(xr[i][0] & 0xffff0000u);
                xf[i][2] = __uint_as_float(xr[i][1] << 16);
                xf[i][3] = __uint_as_float(xr[i][1] & 0xffff0000u);
            } else {
                xf[i][0] = f16_to_f32((f16_t)(xr[i][0] & 0xffffu));
                xf[i][1] = f16_to_f32((f16_t)(xr[i][0] >> 16));
                xf[i][2] = f16_to_f32((f16_t)(xr[i][1] & 0xffffu));
                xf[i][3] = f16_to_f32((f16_t)(xr[i][1] >> 16));
            }
        }
    };
    [[maybe_unused]] auto sum_sq = [&](const float (&xf)[NV][4]) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) ss += xf[i][j] * xf[i][j];
        return ss;
    };
    // f16 copy of one chunk -> LDS, outlier bits, sub-threshold absmax; hf / outm keep the chunk in registers
    [[maybe_unused]] auto emit = [&](int m, int c, const float (&xf)[NV][4], float rinv, float& amax,
                                     float (&hf)[NV][4], unsigned& outm) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = (c * NV + i) * nthr + tid;
            if (v < nvecp) {
                uint32_t hb[4];
                unsigned ob = 0u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float val = xf[i][j];  // zero beyond K (buffer load)
                    if (p.norm_scale != nullptr) {
                        const uint32_t w2 = nr[i][j >> 1];
                        const float sc = __uint_as_float((j & 1) ? (w2 & 0xffff0000u) : (w2 << 16));
                        val = sc * (val * rinv);
                    }
                    const f16_t h = f32_to_f16(val);
                    hb[j] = (uint32_t)h;
                    const float hv = f16_to_f32(h);
                    hf[i][j] = hv;
                    const float a = fabsf(hv);
                    if (thr_on && a >= p.threshold)
                        ob |= 1u << j;
                    else
                        amax = fmaxf(amax, a);
                }
                u32x2 o2;
                o2[0] = hb[0] | (hb[1] << 16);
                o2[1] = hb[2] | (hb[3] << 16);
                *(u32x2*)(xh + (size_t)m * Kp + 4 * (size_t)v) = o2;
                if (ob != 0u) atomicOr(&obits[v >> 3], ob << ((v & 7) * 4));
                outm |= ob << (4 * i);
            }
        }
    };
    if constexpr (VEC) {
        [[maybe_unused]] const int nchunk = FAST ? 1 : (nvecp + NV * nthr - 1) / (NV * nthr);  // 1 with a fused norm
        if constexpr (FAST) {
            // decode: straight-line code on the vectors requested before the ring (a loop around the loads would
            // make the compiler wait for the whole ring prefill first), quantised straight from registers
            float xf[NV][4], hf[NV][4];
            decode(xf);
            float rinv = 1.f;
            if (p.norm_scale != nullptr) rinv = rsqrtf(blk_reduce(sum_sq(xf), 0, false) / (float)p.K + p.eps);
            float amax = 0.f;
            unsigned outm = 0u;
            emit(0, 0, xf, rinv, amax, hf, outm);
            amax = blk_reduce(amax, 1, true);
            const float inv = amax > 0.f ? __fdiv_rn(127.0f, amax) : 0.f;  // IEEE division: parity with the oracle
            if (tid == 0) {
                sca[0] = amax;
                sinv[0] = inv;
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int v = i * nthr + tid;
                if (v < nvecp) {
                    uint32_t q4 = 0u;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float q = ((outm >> (4 * i + j)) & 1u) ? 0.f : rintf(hf[i][j] * inv);
                        q4 |= ((uint32_t)(int)q & 0xffu) << (j * 8);
                    }
                    *(uint32_t*)(xq + 4 * (size_t)v) = q4;
                }
            }
            quantised = true;
        } else {
            for (int m = 0; m < M; ++m) {
                float amax = 0.f;
                for (int c = 0; c < nchunk; ++c) {
                    if ((m | c) != 0) load_chunk(m, c);
                    float xf[NV][4], hf[NV][4];
                    unsigned outm = 0u;
                    decode(xf);
                    float rinv = 1.f;
                    if (p.norm_scale != nullptr)
                        rinv = rsqrtf(blk_reduce(sum_sq(xf), (2 * m) & 3, false) / (float)p.K + p.eps);
                    emit(m, c, xf, rinv, amax, hf, outm);
                }
                amax = blk_reduce(amax, (2 * m + 1) & 3, true);
                if (tid == 0) {
                    sca[m] = amax;
                    sinv[m] = amax > 0.f ? __fdiv_rn(127.0f, amax) : 0.f;
                }
            }
        }
    } else {
        for (int m = 0; m < M; ++m) {
            const int64_t base = (int64_t)m * p.ldx;
            float rinv = 1.f;
            if (p.norm_scale != nullptr) {
                float ss = 0.f;
                for (int k = tid; k < p.K; k += blockDim.x) {
                    const float v = ld_as_f32(p.x, base + k, p.x_dtype);
                    ss += v * v;
                }
                ss = blk_reduce(ss, (2 * m) & 3, false);
                rinv = rsqrtf(ss / (float)p.K + p.eps);
            }
            float amax = 0.f;
            for (int k = tid; k < Kp; k += blockDim.x) {
                f16_t h = 0;
                if (k < p.K) {
                    float v = ld_as_f32(p.x, base + k, p.x_dtype);
                    if (p.norm_scale != nullptr) v = ld_as_f32(p.norm_scale, k, p.norm_dtype) * (v * rinv);
                    h = f32_to_f16(v);
                }
                xh[(size_t)m * Kp + k] = h;
                const float a = fabsf(f16_to_f32(h));
                if (thr_on && a >= p.threshold)
                    atomicOr(&obits[k >> 5], 1u << (k & 31));
                else
                    amax = fmaxf(amax, a);
            }
            amax = blk_reduce(amax, (2 * m + 1) & 3, true);
            if (tid == 0) {
                sca[m] = amax;
                sinv[m] = amax > 0.f ? __fdiv_rn(127.0f, amax) : 0.f;
            }
        }
    }
    MI355_STAMP(2);
    // every atomic OR precedes its thread's last blk_reduce barrier: the bit set is complete here.
    // Wave 0 writes the ascending outlier list while the other waves quantise.
    if (wave == 0 && thr_on) {
        int oc = 0;
        for (int w0 = 0; w0 < (Kp >> 5); w0 += 64) {
            const unsigned word = (w0 + lane < (Kp >> 5)) ? obits[w0 + lane] : 0u;
            unsigned long long live = __ballot(word != 0u);
            while (live != 0ull) {
                const int src = __builtin_ctzll(live);
                live &= live - 1ull;
                unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)word, src);
                while (bits != 0u) {
                    const int b = __builtin_ctz(bits);
                    bits &= bits - 1u;
                    if (lane == 0) olist[oc] = (uint16_t)(((w0 + src) << 5) + b);
                    ++oc;
                }
            }
        }
        if (lane == 0) ototal[0] = oc;
    }
    if constexpr (!FAST) {
      if (!quantised) {
        // general case (several rows / long rows): 8 columns per thread step from the f16 copy in LDS;
        // CA[m,k] = rint(x * (127 / SCA[m])), whole outlier columns zeroed
        __syncthreads();  // sinv of the last row
        for (int v8 = tid; v8 < (Kp >> 3); v8 += blockDim.x) {
            const unsigned outm = thr_on ? ((obits[v8 >> 2] >> ((v8 & 3) * 8)) & 0xffu) : 0u;
            for (int m = 0; m < M; ++m) {
                const u32x4 hh = *(const u32x4*)(xh + (size_t)m * Kp + 8 * (size_t)v8);
                const float inv = sinv[m];
                u32x2 q2 = u32x2{0u, 0u};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f16_t h = (f16_t)((hh[j >> 1] >> ((j & 1) * 16)) & 0xffffu);
                    const float q = ((outm >> j) & 1u) ? 0.f : rintf(f16_to_f32(h) * inv);
                    q2[j >> 2] |= ((uint32_t)(int)q & 0xffu) << ((j & 3) * 8);
                }
                *(u32x2*)(xq + (size_t)m * p.xq_stride + 8 * (size_t)v8) = q2;
            }
        }
      }
    }
    __syncthreads();
    const int n_out = ototal[0];
    MI355_STAMP(3);

    const int e_row = (tid >> 4) & 15, e_col = tid & 15;
    const bool e_owner = tid < 256 && e_col < M;

    // epilogue operands of the NEXT tile are fetched one tile ahead and kept as raw bits (see gemv.hip)
    uint32_t eo_scb[R], eo_bias[R], eo_old[R], eo_out[R][kFastOut];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        eo_scb[r] = eo_bias[r] = eo_old[r] = 0u;
#pragma unroll
        for (int i = 0; i < kFastOut; ++i) eo_out[r][i] = 0u;
    }
    // byte offset of CB[n = row e_row of (tile, r), k] in the weight stream
    auto w_off = [&](int tile, int r, int k) -> int64_t {
        const int u = k >> 7, e = (k >> 6) & 1, g = (k >> 4) & 3, j = k & 15;
        return ((((int64_t)tile * units + u) * R + r) * 2 + e) * 1024 + (g * 16 + e_row) * 16 + j;
    };
    auto load_epi = [&](int tile) {
        if (e_owner && tile < p.n_tiles) {
            const bool sw = p.epi == MI355_EPI_SWIGLU;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int n = sw ? tile * 16 + e_row : (tile * R + r) * 16 + e_row;
                if (n < p.N) {
#pragma unroll
                    for (int i = 0; i < kFastOut; ++i)
                        if (i < n_out) eo_out[r][i] = (uint32_t)p.w[w_off(tile, r, olist[i])];
                    eo_scb[r] = ((const uint32_t*)((sw && r == 1) ? p.scb2 : p.scb))[n];
                    if (!sw) {
                        if (p.bias != nullptr)
                            eo_bias[r] = p.bias_dtype == MI355_F32 ? ((const uint32_t*)p.bias)[n]
                                                                   : (uint32_t)((const uint16_t*)p.bias)[n];
                        if (p.epi == MI355_EPI_ACCUM) {
                            const int64_t yi = (int64_t)e_col * p.ldy + n;
                            eo_old[r] = p.y_dtype == MI355_F32 ? ((const uint32_t*)p.y)[yi]
                                                                : (uint32_t)((const uint16_t*)p.y)[yi];
                        }
                    }
                }
            }
        }
    };
    auto raw_to_f32 = [](uint32_t raw, int dtype) {
        return dtype == MI355_F32 ? __uint_as_float(raw)
                                  : (dtype == MI355_BF16 ? __uint_as_float(raw << 16) : f16_to_f32((f16_t)raw));
    };
    load_epi(bid);

    // combine + dequant + outlier side product + store, for the tile in `buf`
    auto epilogue = [&](int tile, int buf) {
        const int src = ((e_row >> 2) << 4) | e_col;
        const int* base = (const int*)(part + (size_t)(buf * W * R) * 1024) + src * 4 + (e_row & 3);
        float v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool sw = p.epi == MI355_EPI_SWIGLU;
            const int n = sw ? tile * 16 + e_row : (tile * R + r) * 16 + e_row;
            v[r] = 0.f;
            if (n >= p.N) continue;
            int acc = 0;
            for (int w = 0; w < W; ++w) acc += base[(w * R + r) * 256];
            const float scb = __uint_as_float(eo_scb[r]);
            float d = (((float)acc * 6.200012e-05f) * sca[e_col]) * scb;
            if (p.bias != nullptr && !sw) d += raw_to_f32(eo_bias[r], p.bias_dtype);
            d = f16r(d);
            // mixed-precision decomposition: outlier columns in f16, ascending k
            float o = 0.f;
            const bool any = n_out > 0;
#pragma unroll
            for (int i = 0; i < kFastOut; ++i) {
                if (i < n_out) {
                    const float cb = (float)(int8_t)eo_out[r][i];
                    const float sub = f16r(__fdiv_rn(cb * scb, 127.0f));
                    o += f16_to_f32(xh[(size_t)e_col * Kp + olist[i]]) * sub;
                }
            }
            for (int i = kFastOut; i < n_out; ++i) {  // rare: more outlier columns than prefetch registers
                const int k = olist[i];
                const float cb = (float)(int8_t)p.w[w_off(tile, r, k)];
                const float sub = f16r(__fdiv_rn(cb * scb, 127.0f));
                o += f16_to_f32(xh[(size_t)e_col * Kp + k]) * sub;
            }
            if (any) d = f16r(d + f16r(o));
            v[r] = d;
        }
        if (p.epi == MI355_EPI_SWIGLU) {
            if constexpr (R == 2) {
                const int n = tile * 16 + e_row;
                if (n < p.N) st_from_f32(p.y, (int64_t)e_col * p.ldy + n, p.y_dtype, swiglu_f32(v[0], v[1]));
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int n = (tile * R + r) * 16 + e_row;
                if (n < p.N) {
                    float out = v[r];
                    if (p.epi == MI355_EPI_ACCUM) out += raw_to_f32(eo_old[r], p.y_dtype);
                    st_from_f32(p.y, (int64_t)e_col * p.ldy + n, p.y_dtype, out);
                }
            }
        }
    };

    i32x4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = i32x4{0, 0, 0, 0};
    int tile = bid, buf = 0;

    if (nu == 0) {
        for (int i = 0; i < my_tiles; ++i) {
            i32x4* pp = (i32x4*)(part + (size_t)((buf * W + wave) * R) * 1024) + lane;
#pragma unroll
            for (int r = 0; r < R; ++r) pp[r * 64] = acc[r];
            __syncthreads();
            if (e_owner) epilogue(tile, buf);
            tile += nb;
            buf ^= 1;
            load_epi(tile);
        }
        return;
    }

    const int g = lane >> 4, c = lane & 15;
    const int xrow = c < M ? c : M - 1;
    const char* xl = xq + (size_t)xrow * p.xq_stride + g * 16;
    int uu = 0;
    for (int t = 0; t < total; t += P) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            if (t + j < total) {
                const char* xb = xl + (u0 + uu) * kUnitK;
                const i32x4 b0 = *(const i32x4*)(xb), b1 = *(const i32x4*)(xb + 64);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    acc[r] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, ring[j][r * 2]), b0, acc[r],
                                                                   0, 0, 0);
                    acc[r] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, ring[j][r * 2 + 1]), b1,
                                                                   acc[r], 0, 0, 0);
                }
                if (++uu == nu) {
                    uu = 0;
                    i32x4* pp = (i32x4*)(part + (size_t)((buf * W + wave) * R) * 1024) + lane;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        pp[r * 64] = acc[r];
                        acc[r] = i32x4{0, 0, 0, 0};
                    }
                    if (tile == bid) MI355_STAMP(4);
                    __syncthreads();
                    if (e_owner) epilogue(tile, buf);
                    if (tile == bid) MI355_STAMP(5);
                    tile += nb;
                    buf ^= 1;
                    load_epi(tile);
                }
            }
            MI355_ISSUE(j);
        }
    }
    MI355_STAMP(6);
#undef MI355_ISSUE
#undef MI355_STAMP
}

// bnb.functional.double_quant(W) rows: SCB[n] = max_k |f16(W[n,k])|, CB = rint(w * (127 / SCB))
__global__ void int8_quant_rows_kernel(const void* w, int dtype, int K, int8_t* cb, float* scb) {
    __shared__ float red[32];
    const int n = blockIdx.x;
    float amax = 0.f;
    for (int k = threadIdx.x; k < K; k += blockDim.x)
        amax = fmaxf(amax, fabsf(f16_to_f32(f32_to_f16(ld_as_f32(w, (int64_t)n * K + k, dtype)))));
    amax = block_max(amax, red);
    if (threadIdx.x == 0) scb[n] = amax;
    const float inv = amax > 0.f ? __fdiv_rn(127.0f, amax) : 0.f;  // IEEE division: bit parity with the oracle
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float v = f16_to_f32(f32_to_f16(ld_as_f32(w, (int64_t)n * K + k, dtype)));
        cb[(int64_t)n * K + k] = (int8_t)rintf(v * inv);
    }
}

template <int R, int P, int VNV, bool FAST>
int launch_i8v(const I8Params& p, int grid, int waves, size_t lds, hipStream_t stream) {
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute((const void*)int8_gemv_kernel<R, P, VNV, FAST>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    });
    if (attr_err != hipSuccess) {
        mi355_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(attr_err));
        return (int)attr_err;
    }
    hipLaunchKernelGGL((int8_gemv_kernel<R, P, VNV, FAST>), dim3(grid), dim3(waves * 64), lds, stream, p);
    MI355_LAUNCH_CHECK();
    return 0;
}

template <int R, int P>
int launch_i8(const I8Params& p, int grid, int waves, size_t lds, hipStream_t stream) {
    // register staging sized to the row: 2 vectors per thread cover K <= 4096 at 512 threads (every n_embd-wide
    // input of the 7B model) with ~100 VGPRs; the 8-vector variant (K <= 16384) needs ~200 and its workgroups
    // take visibly longer to dispatch
    if (p.vec == 0) return launch_i8v<R, P, 0, false>(p, grid, waves, lds, stream);
    const int nvecp = p.units * kUnitK / 4, nthr = waves * 64;
    if (nvecp <= 2 * nthr) {
        return p.M == 1 ? launch_i8v<R, P, 2, true>(p, grid, waves, lds, stream)
                        : launch_i8v<R, P, 2, false>(p, grid, waves, lds, stream);
    }
    return (p.M == 1 && nvecp <= kStageVec * nthr) ? launch_i8v<R, P, kStageVec, true>(p, grid, waves, lds, stream)
                                                   : launch_i8v<R, P, kStageVec, false>(p, grid, waves, lds, stream);
}

}  // namespace

extern "C" int mi355_int8_quant_rows(const void* w, int dtype, int N, int K, int8_t* cb, float* scb,
                                     mi355_stream_t stream) {
    MI355_CHECK_ARG(w && cb && scb, MI355_E_ARG, "int8_quant_rows: null pointer");
    MI355_CHECK_ARG(N > 0 && K > 0, MI355_E_SHAPE, "int8_quant_rows: bad shape");
    MI355_CHECK_ARG(dtype >= MI355_F32 && dtype <= MI355_F16, MI355_E_DTYPE, "int8_quant_rows: bad dtype");
    hipLaunchKernelGGL(int8_quant_rows_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, w, dtype, K, cb, scb);
    MI355_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi355_linear_int8(const mi355_int8_args* a, mi355_stream_t stream) {
    MI355_CHECK_ARG(a != nullptr && a->w && a->scb && a->x && a->y, MI355_E_ARG, "linear_int8: null argument");
    MI355_CHECK_ARG(a->R == 1 || a->R == 2, MI355_E_ARG, "linear_int8: R must be 1 or 2");
    MI355_CHECK_ARG(a->M >= 1 && a->M <= kMaxM, MI355_E_SHAPE, "linear_int8: M=%d outside 1..%d", a->M, kMaxM);
    MI355_CHECK_ARG(a->N > 0 && a->K > 0 && a->K <= 65535, MI355_E_SHAPE, "linear_int8: bad N/K");
    MI355_CHECK_ARG(a->epi >= MI355_EPI_STORE && a->epi <= MI355_EPI_SWIGLU, MI355_E_ARG, "linear_int8: bad epi");
    const bool swiglu = a->epi == MI355_EPI_SWIGLU;
    MI355_CHECK_ARG(!swiglu || (a->R == 2 && a->scb2), MI355_E_ARG, "linear_int8: SwiGLU needs R=2 and scb2");

    I8Params p;
    p.w = (const uint8_t*)a->w;
    p.scb = a->scb;
    p.scb2 = a->scb2;
    p.x = a->x;
    p.norm_scale = a->norm_scale;
    p.bias = a->bias;
    p.y = a->y;
    p.ldx = a->ldx;
    p.ldy = a->ldy;
    p.N = a->N;
    p.K = a->K;
    p.M = a->M;
    const int rows_per_tile = swiglu ? 16 : 16 * a->R;
    p.n_tiles = (a->N + rows_per_tile - 1) / rows_per_tile;
    p.units = (a->K + kUnitK - 1) / kUnitK;
    p.x_dtype = a->x_dtype;
    p.norm_dtype = a->norm_dtype;
    p.bias_dtype = a->bias_dtype;
    p.y_dtype = a->y_dtype;
    p.epi = a->epi;
    const int Kp = p.units * kUnitK;
    p.xq_stride = Kp + 16;
    p.eps = a->eps;
    p.threshold = a->threshold;
    p.dbg = a->debug_stamps;
    p.attn_part = a->attn_partials;
    p.attn_splits = a->attn_splits;
    p.attn_heads = a->attn_heads;
    p.attn_hs = a->attn_hs;
    {
        const size_t wb = mi355_packed_bytes(MI355_W_I8, a->N, a->K, a->R, swiglu ? 1 : 0);
        MI355_CHECK_ARG(wb > 0 && wb < 0xFFFFFFF0ull, MI355_E_SHAPE, "linear_int8: weight stream of %zu B exceeds 4 GiB", wb);
        p.w_bytes = (unsigned)wb;
    }

    int waves = a->waves > 0 ? a->waves : 8;
    if (waves > 8) waves = 8;
    if (waves < 4) waves = 4;
    {
        // vector staging: 4 columns per load (16 B of f32 / 8 B of bf16, f16), rows aligned accordingly; with a
        // fused norm the whole row must sit in one register chunk (kStageVec vectors per thread) and the scale be bf16
        const int esz = a->x_dtype == MI355_F32 ? 4 : 2;
        const uintptr_t al = esz == 4 ? 16 : 8;
        bool ok = a->K % 4 == 0 && (uintptr_t)a->x % al == 0 && (a->M == 1 || (a->ldx * esz) % al == 0);
        if (a->norm_scale != nullptr)
            ok = ok && a->norm_dtype == MI355_BF16 && (uintptr_t)a->norm_scale % 8 == 0 && Kp / 4 <= kStageVec * waves * 64;
        p.vec = ok ? 1 : 0;
    }
    if (a->attn_partials != nullptr) {
        MI355_CHECK_ARG(a->M == 1 && a->norm_scale == nullptr && a->attn_splits >= 1 && a->attn_splits <= 4 &&
                            a->attn_hs >= 4 && a->attn_hs % 4 == 0 && a->attn_heads * a->attn_hs == a->K &&
                            a->K % 4 == 0 && Kp / 4 <= 2 * waves * 64 && (uintptr_t)a->attn_partials % 16 == 0,
                        MI355_E_SHAPE, "linear_int8: split-attention input needs M = 1, no norm, <= 4 splits, K <= %d",
                        8 * waves * 64);
        p.vec = 1;  // the partial records replace the row loads
    }
    const size_t lds = kHdr + (size_t)2 * waves * a->R * 1024 + (size_t)a->M * p.xq_stride + (size_t)a->M * Kp * 2 +
                       (size_t)Kp * 2 + (size_t)Kp / 8 + 16;
    MI355_CHECK_ARG(lds <= (size_t)kMaxLds, MI355_E_SHAPE,
                    "linear_int8: M=%d x K=%d activations do not fit LDS (%zu B); chunk M", a->M, a->K, lds);
    int grid = a->grid;
    if (grid <= 0) grid = (mi355_num_cus() > 0 ? mi355_num_cus() : 256) * 2;
    if (grid > p.n_tiles) grid = p.n_tiles;
    p.u_q = p.units / waves;
    p.u_r = p.units % waves;
    p.t_q = p.n_tiles / grid;
    p.t_r = p.n_tiles % grid;
    hipStream_t s = (hipStream_t)stream;
    // ring depth: 4 units in flight per wave measured best for single matrices, 2 for the c_fc1/c_fc2 pair
    const bool deep = a->prefetch > 0 ? a->prefetch >= 4 : a->R == 1;
    if (a->R == 1) return deep ? launch_i8<1, 4>(p, grid, waves, lds, s) : launch_i8<1, 2>(p, grid, waves, lds, s);
    return deep ? launch_i8<2, 4>(p, grid, waves, lds, s) : launch_i8<2, 2>(p, grid, waves, lds, s);
}

int mi355_linear_int8_from_weight(const mi355_weight* w, const mi355_model* m, const void* x, int x_dtype, int M,
                                  int64_t ldx, const void* norm_scale, int epi, void* y, int y_dtype, int64_t ldy,
                                  hipStream_t stream, const float* attn_partials) {
    mi355_int8_args a;
    memset(&a, 0, sizeof(a));
    a.w = (const int8_t*)w->w;
    a.scb = w->scb;
    a.scb2 = w->scb2;
    a.N = w->N;
    a.K = w->K;
    a.x = x;
    a.x_dtype = x_dtype;
    a.M = M;
    a.ldx = ldx;
    a.norm_scale = norm_scale;
    a.norm_dtype = m->param_dtype;
    a.eps = m->eps;
    a.threshold = m->int8_threshold;
    a.R = w->R;
    a.bias = nullptr;
    a.epi = epi;
    a.y = y;
    a.y_dtype = y_dtype;
    a.ldy = ldy;
    a.waves = w->waves;
    a.grid = w->grid;
    a.prefetch = w->prefetch;
    if (attn_partials != nullptr) {
        a.attn_partials = attn_partials;
        a.attn_splits = m->attn_splits;
        a.attn_heads = m->n_head;
        a.attn_hs = m->hs;
    }
    return mi355_linear_int8(&a, stream);
}

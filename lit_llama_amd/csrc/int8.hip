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
// written down in oracle/llm_int8.py ("parity unpinned").
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
    int x_dtype, norm_dtype, bias_dtype, y_dtype, epi;
    int xq_stride;  // bytes per int8 activation row in LDS
    unsigned w_bytes;  // size of the weight stream (buffer descriptor bound)
    float eps, threshold;
};

__device__ __forceinline__ float f16r(float v) { return f16_to_f32(f32_to_f16(v)); }

// LDS map: [hdr: red[64] | sca[16] | ocnt[8]] [part 2*W*R KiB] [xq M*xq_stride] [xh M*Kp f16] [olist Kp u16]
template <int R, int P>
__global__ __launch_bounds__(512) void int8_gemv_kernel(const I8Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)smem;
    float* sca = (float*)(smem + 256);
    int* ocnt = (int*)(smem + 320);
    char* part = smem + kHdr;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = blockDim.x >> 6;
    const int units = p.units, Kp = units * kUnitK;
    char* xq = part + 2 * W * R * 1024;
    f16_t* xh = (f16_t*)(xq + (size_t)p.M * p.xq_stride);
    uint16_t* olist = (uint16_t*)(xh + (size_t)p.M * Kp);

    const int u0 = (units * wave) / W, u1 = (units * (wave + 1)) / W;
    const int nu = u1 - u0;
    const int bid = blockIdx.x, nb = gridDim.x;
    const int my_tiles = (p.n_tiles > bid) ? (p.n_tiles - bid + nb - 1) / nb : 0;
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
#pragma unroll
    for (int j = 0; j < P; ++j) MI355_ISSUE(j);

    // ---------------- prologue: f16 activations, outlier columns, row scales, int8 quantisation
    for (int m = 0; m < p.M; ++m) {
        const int64_t base = (int64_t)m * p.ldx;
        float rinv = 1.f;
        if (p.norm_scale != nullptr) {
            float ss = 0.f;
            for (int k = tid; k < p.K; k += blockDim.x) {
                const float v = ld_as_f32(p.x, base + k, p.x_dtype);
                ss += v * v;
            }
            ss = block_sum(ss, red);
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
            if (!(p.threshold > 0.f) || a < p.threshold) amax = fmaxf(amax, a);
        }
        amax = block_max(amax, red);
        if (tid == 0) sca[m] = amax;
    }
    __syncthreads();
    // outlier columns of this wave's K slice, in ascending k (one ballot per 64 columns)
    {
        uint16_t* mylist = olist + u0 * kUnitK;
        int cnt = 0;
        if (p.threshold > 0.f) {
            for (int k0 = u0 * kUnitK; k0 < u1 * kUnitK; k0 += 64) {
                const int k = k0 + lane;
                bool out = false;
                for (int m = 0; m < p.M; ++m) out |= fabsf(f16_to_f32(xh[(size_t)m * Kp + k])) >= p.threshold;
                const unsigned long long mask = __ballot(out);
                if (out) mylist[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)k;
                cnt += __popcll(mask);
            }
        }
        if (lane == 0) ocnt[wave] = cnt;
    }
    // quantise: CA[m,k] = rint(x * (127 / SCA[m])), whole outlier columns zeroed
    for (int k = tid; k < Kp; k += blockDim.x) {
        bool out = false;
        if (p.threshold > 0.f)
            for (int m = 0; m < p.M; ++m) out |= fabsf(f16_to_f32(xh[(size_t)m * Kp + k])) >= p.threshold;
        for (int m = 0; m < p.M; ++m) {
            const float s = sca[m];
            const float inv = s > 0.f ? __fdiv_rn(127.0f, s) : 0.f;  // IEEE division: bit parity with the oracle
            const float q = out ? 0.f : rintf(f16_to_f32(xh[(size_t)m * Kp + k]) * inv);
            ((int8_t*)(xq + (size_t)m * p.xq_stride))[k] = (int8_t)q;
        }
    }
    __syncthreads();

    const int e_row = (tid >> 4) & 15, e_col = tid & 15;
    const bool e_owner = tid < 256 && e_col < p.M;

    // epilogue operands of the NEXT tile are fetched one tile ahead and kept as raw bits (see gemv.hip)
    uint32_t eo_scb[R], eo_bias[R], eo_old[R];
#pragma unroll
    for (int r = 0; r < R; ++r) eo_scb[r] = eo_bias[r] = eo_old[r] = 0u;
    auto load_epi = [&](int tile) {
        if (e_owner && tile < p.n_tiles) {
            const bool sw = p.epi == MI355_EPI_SWIGLU;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int n = sw ? tile * 16 + e_row : (tile * R + r) * 16 + e_row;
                if (n < p.N) {
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
            bool any = false;
            for (int w = 0; w < W; ++w) {
                const int wu0 = (units * w) / W;
                const uint16_t* lst = olist + wu0 * kUnitK;
                const int c = ocnt[w];
                for (int i = 0; i < c; ++i) {
                    const int k = lst[i];
                    const int u = k >> 7, e = (k >> 6) & 1, g = (k >> 4) & 3, j = k & 15;
                    const int64_t off =
                        ((((int64_t)tile * units + u) * R + r) * 2 + e) * 1024 + (g * 16 + e_row) * 16 + j;
                    const float cb = (float)(int8_t)p.w[off];
                    const float sub = f16r(__fdiv_rn(cb * scb, 127.0f));
                    o += f16_to_f32(xh[(size_t)e_col * Kp + k]) * sub;
                    any = true;
                }
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
    const int xrow = c < p.M ? c : p.M - 1;
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
                    __syncthreads();
                    if (e_owner) epilogue(tile, buf);
                    tile += nb;
                    buf ^= 1;
                    load_epi(tile);
                }
            }
            MI355_ISSUE(j);
        }
    }
#undef MI355_ISSUE
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

template <int R, int P>
int launch_i8(const I8Params& p, int grid, int waves, size_t lds, hipStream_t stream) {
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute((const void*)int8_gemv_kernel<R, P>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       kMaxLds);
    });
    if (attr_err != hipSuccess) {
        mi355_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(attr_err));
        return (int)attr_err;
    }
    hipLaunchKernelGGL((int8_gemv_kernel<R, P>), dim3(grid), dim3(waves * 64), lds, stream, p);
    MI355_LAUNCH_CHECK();
    return 0;
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
    {
        const size_t wb = mi355_packed_bytes(MI355_W_I8, a->N, a->K, a->R, swiglu ? 1 : 0);
        MI355_CHECK_ARG(wb > 0 && wb < 0xFFFFFFF0ull, MI355_E_SHAPE, "linear_int8: weight stream of %zu B exceeds 4 GiB", wb);
        p.w_bytes = (unsigned)wb;
    }

    int waves = a->waves > 0 ? a->waves : 8;
    if (waves > 8) waves = 8;
    if (waves < 4) waves = 4;
    const size_t lds = kHdr + (size_t)2 * waves * a->R * 1024 + (size_t)a->M * p.xq_stride + (size_t)a->M * Kp * 2 +
                       (size_t)Kp * 2 + 16;
    MI355_CHECK_ARG(lds <= (size_t)kMaxLds, MI355_E_SHAPE,
                    "linear_int8: M=%d x K=%d activations do not fit LDS (%zu B); chunk M", a->M, a->K, lds);
    int grid = a->grid;
    if (grid <= 0) grid = (mi355_num_cus() > 0 ? mi355_num_cus() : 256) * 2;
    if (grid > p.n_tiles) grid = p.n_tiles;
    hipStream_t s = (hipStream_t)stream;
    const bool deep = a->prefetch >= 4;
    if (a->R == 1) return deep ? launch_i8<1, 4>(p, grid, waves, lds, s) : launch_i8<1, 2>(p, grid, waves, lds, s);
    return deep ? launch_i8<2, 4>(p, grid, waves, lds, s) : launch_i8<2, 2>(p, grid, waves, lds, s);
}

int mi355_linear_int8_from_weight(const mi355_weight* w, const mi355_model* m, const void* x, int x_dtype, int M,
                                  int64_t ldx, const void* norm_scale, int epi, void* y, int y_dtype, int64_t ldy,
                                  hipStream_t stream) {
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
    return mi355_linear_int8(&a, stream);
}

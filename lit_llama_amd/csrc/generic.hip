// Generic (any shape / any M / f32, bf16, f16) operators of libmi355llama, plus the small glue
// kernels of the decode step (embedding gather, greedy argmax, step parameters).
//
// These follow the reference arithmetic op by op in f32 and are what the f32 "plumbing"
// configuration (BASELINE.json configs[0]) and unusual layouts (gptq.int8, grouped scales) run on.
// They are correct-first kernels: one wave per output element with coalesced reads along K.
#include <stdarg.h>

#include "common.h"

// ------------------------------------------------------------------------------------ error plumbing
static thread_local char g_err[512] = "";

void mi355_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mi355_last_error(void) { return g_err; }
extern "C" int mi355_version(void) { return MI355_ABI_VERSION; }
extern "C" int mi355_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(mi355_linear_args);
        case 1: return (int)sizeof(mi355_attn_args);
        case 2: return (int)sizeof(mi355_int8_args);
        case 3: return (int)sizeof(mi355_weight);
        case 4: return (int)sizeof(mi355_layer);
        case 5: return (int)sizeof(mi355_model);
        case 6: return (int)sizeof(mi355_fused_step_args);
        case 7: return (int)sizeof(mi355_tp_comm);
        default: return -1;
    }
}

namespace {

inline bool dtype_ok(int d) { return d == MI355_F32 || d == MI355_BF16 || d == MI355_F16; }

// ------------------------------------------------------------------------------------ dense linear
// grid (ceil(N / 4), M), 256 threads: one wave per output n, lanes stride K (W row-major -> coalesced).
__global__ void linear_dense_kernel(const void* x, int64_t ldx, const void* w, const void* bias, void* y, int64_t ldy,
                                    int N, int K, int dtype) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    const int m = blockIdx.y;
    if (n >= N) return;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64)
        acc += ld_as_f32(x, (int64_t)m * ldx + k, dtype) * ld_as_f32(w, (int64_t)n * K + k, dtype);
    acc = wave_sum(acc);
    if (lane == 0) {
        if (bias != nullptr) acc += ld_as_f32(bias, n, dtype);
        st_from_f32(y, (int64_t)m * ldy + n, dtype, acc);
    }
}

// ------------------------------------------------------------------------------------ ColBlock linear
// W[n,k] = (q[n,k] - zeros[n,g]) * scales[n,g], g = k / tile_cols  (lit_llama/quantization.py:398-410)
__device__ __forceinline__ float colblock_w(const uint8_t* qw, int64_t sn, int64_t skb, const void* scales,
                                            const void* zeros, int sz_dtype, int n_groups, int tile_cols, int bits,
                                            int n, int k) {
    float q;
    if (bits == 4) {
        const uint8_t b = qw[(int64_t)n * sn + (int64_t)(k >> 1) * skb];
        q = (float)((k & 1) ? (b >> 4) : (b & 0xF));
    } else {
        q = (float)qw[(int64_t)n * sn + (int64_t)k * skb];
    }
    const int g = k / tile_cols;
    const float z = ld_as_f32(zeros, (int64_t)n * n_groups + g, sz_dtype);
    const float s = ld_as_f32(scales, (int64_t)n * n_groups + g, sz_dtype);
    return (q - z) * s;
}

__global__ void linear_colblock_kernel(const void* x, int64_t ldx, const uint8_t* qw, int64_t sn, int64_t skb,
                                       const void* scales, const void* zeros, int sz_dtype, int n_groups,
                                       int tile_cols, int bits, const void* bias, void* y, int64_t ldy, int N, int K,
                                       int dtype) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    const int m = blockIdx.y;
    if (n >= N) return;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) {
        // the reference materialises W in the activation dtype (get_weight(dtype=inp.dtype))
        const float wv = round_to(colblock_w(qw, sn, skb, scales, zeros, sz_dtype, n_groups, tile_cols, bits, n, k), dtype);
        acc += ld_as_f32(x, (int64_t)m * ldx + k, dtype) * wv;
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        if (bias != nullptr) acc += ld_as_f32(bias, n, sz_dtype);
        st_from_f32(y, (int64_t)m * ldy + n, dtype, acc);
    }
}

__global__ void colblock_dequant_kernel(const uint8_t* qw, int64_t sn, int64_t skb, const void* scales,
                                        const void* zeros, int sz_dtype, int n_groups, int tile_cols, int bits,
                                        void* out, int out_dtype, int N, int K) {
    const int64_t total = (int64_t)N * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / K), k = (int)(i % K);
        st_from_f32(out, i, out_dtype, colblock_w(qw, sn, skb, scales, zeros, sz_dtype, n_groups, tile_cols, bits, n, k));
    }
}

// ------------------------------------------------------------------------------------ RMSNorm
__global__ void rmsnorm_kernel(const void* x, int64_t ldx, const void* scale, int scale_dtype, float eps, void* y,
                               int64_t ldy, int C, int x_dtype, int y_dtype) {
    __shared__ float red[32];
    const int m = blockIdx.x;
    float ss = 0.f;
    for (int k = threadIdx.x; k < C; k += blockDim.x) {
        const float v = ld_as_f32(x, (int64_t)m * ldx + k, x_dtype);
        ss += v * v;
    }
    ss = block_sum(ss, red);
    const float rinv = rsqrtf(ss / (float)C + eps);
    for (int k = threadIdx.x; k < C; k += blockDim.x) {
        const float v = ld_as_f32(x, (int64_t)m * ldx + k, x_dtype);
        st_from_f32(y, (int64_t)m * ldy + k, y_dtype, ld_as_f32(scale, k, scale_dtype) * (v * rinv));
    }
}

// ------------------------------------------------------------------------------------ RoPE (standalone)
// x [B, T, n_head, hs]; rope [T, hs/2, 2]; interleaved pairs rotated in f32 (lit_llama/model.py:312-320)
__global__ void rope_kernel(const void* x, const float* rope, void* y, int T, int n_head, int hs, int dtype,
                            int64_t n_pairs) {
    const int half = hs / 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pairs; i += (int64_t)gridDim.x * blockDim.x) {
        const int pi = (int)(i % half);
        const int64_t rest = i / half;  // (b * T + t) * n_head + h
        const int t = (int)((rest / n_head) % T);
        const float c = rope[((int64_t)t * half + pi) * 2], s = rope[((int64_t)t * half + pi) * 2 + 1];
        const float a = ld_as_f32(x, 2 * i, dtype), b = ld_as_f32(x, 2 * i + 1, dtype);
        st_from_f32(y, 2 * i, dtype, a * c - b * s);
        st_from_f32(y, 2 * i + 1, dtype, b * c + a * s);
    }
}

__global__ void swiglu_kernel(const void* a, const void* b, void* y, int64_t n, int dtype) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        // F.silu(a) is rounded to the tensor dtype before the product in the reference (model.py:252)
        const float av = ld_as_f32(a, i, dtype);
        const float sil = round_to(av / (1.0f + expf(-av)), dtype);
        st_from_f32(y, i, dtype, sil * ld_as_f32(b, i, dtype));
    }
}

__global__ void add_kernel(const void* a, const void* b, void* y, int64_t n, int dtype) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        st_from_f32(y, i, dtype, ld_as_f32(a, i, dtype) + ld_as_f32(b, i, dtype));
}

__global__ void embedding_kernel(const void* idx, int idx_is_i64, const void* wte, int w_dtype, void* y, int y_dtype,
                                 int C, int vocab) {
    const int m = blockIdx.x;
    int64_t tok = idx_is_i64 ? ((const int64_t*)idx)[m] : (int64_t)((const int32_t*)idx)[m];
    if (tok < 0) tok = 0;
    if (tok >= vocab) tok = vocab - 1;
    if (w_dtype == MI355_BF16 && y_dtype == MI355_F32 && (C & 7) == 0 && ((uintptr_t)wte & 15) == 0 &&
        ((uintptr_t)y & 15) == 0) {
        // decode path: one 16-B load (8 bf16) -> two 16-B stores (8 f32) per thread step
        const u32x4* src = (const u32x4*)((const bf16_t*)wte + tok * C);
        f32x4* dst = (f32x4*)((float*)y + (int64_t)m * C);
        for (int v = threadIdx.x; v < (C >> 3); v += blockDim.x) {
            const u32x4 r = src[v];
            f32x4 a, b;
            a[0] = __uint_as_float(r[0] << 16);
            a[1] = __uint_as_float(r[0] & 0xffff0000u);
            a[2] = __uint_as_float(r[1] << 16);
            a[3] = __uint_as_float(r[1] & 0xffff0000u);
            b[0] = __uint_as_float(r[2] << 16);
            b[1] = __uint_as_float(r[2] & 0xffff0000u);
            b[2] = __uint_as_float(r[3] << 16);
            b[3] = __uint_as_float(r[3] & 0xffff0000u);
            dst[2 * v] = a;
            dst[2 * v + 1] = b;
        }
        return;
    }
    for (int k = threadIdx.x; k < C; k += blockDim.x)
        st_from_f32(y, (int64_t)m * C + k, y_dtype, ld_as_f32(wte, tok * C + k, w_dtype));
}

// first index of the maximum (torch.topk(.., 1) / argmax tie rule is "lowest index" for the CPU oracle)
__global__ void argmax_kernel(const float* logits, int V, int32_t* out, int32_t* out2, const int32_t* out2_pos) {
    const int bi = block_argmax_first(logits, V);
    if (threadIdx.x == 0) {
        out[0] = bi;
        if (out2 != nullptr) out2[out2_pos != nullptr ? out2_pos[0] + 1 : 0] = bi;
    }
}

inline int ew_grid(int64_t n) {
    const int64_t b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int mi355_linear_dense(const void* x, int64_t ldx, const void* w, const void* bias, void* y, int64_t ldy,
                                  int M, int N, int K, int dtype, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && w && y, MI355_E_ARG, "linear_dense: null pointer");
    MI355_CHECK_ARG(M > 0 && N > 0 && K > 0 && M <= 65535, MI355_E_SHAPE, "linear_dense: bad shape M=%d N=%d K=%d", M, N, K);
    MI355_CHECK_ARG(dtype_ok(dtype), MI355_E_DTYPE, "linear_dense: bad dtype %d", dtype);
    hipLaunchKernelGGL(linear_dense_kernel, dim3((N + 3) / 4, M), dim3(256), 0, (hipStream_t)stream, x, ldx, w, bias, y,
                       ldy, N, K, dtype);
    MI355_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi355_linear_colblock(const void* x, int64_t ldx, const uint8_t* qweight, int64_t stride_n,
                                     int64_t stride_kb, const void* scales, const void* zeros, int sz_dtype,
                                     int n_groups, int tile_cols, int bits, const void* bias, void* y, int64_t ldy,
                                     int M, int N, int K, int dtype, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && qweight && scales && zeros && y, MI355_E_ARG, "linear_colblock: null pointer");
    MI355_CHECK_ARG(bits == 4 || bits == 8, MI355_E_ARG, "linear_colblock: bits must be 4 or 8 (got %d)", bits);
    MI355_CHECK_ARG(M > 0 && N > 0 && K > 0 && M <= 65535 && tile_cols > 0, MI355_E_SHAPE, "linear_colblock: bad shape");
    MI355_CHECK_ARG(n_groups == (K + tile_cols - 1) / tile_cols, MI355_E_SHAPE,
                    "linear_colblock: n_groups=%d does not match ceil(K=%d / tile_cols=%d)", n_groups, K, tile_cols);
    MI355_CHECK_ARG(dtype_ok(dtype) && dtype_ok(sz_dtype), MI355_E_DTYPE, "linear_colblock: bad dtype");
    hipLaunchKernelGGL(linear_colblock_kernel, dim3((N + 3) / 4, M), dim3(256), 0, (hipStream_t)stream, x, ldx, qweight,
                       stride_n, stride_kb, scales, zeros, sz_dtype, n_groups, tile_cols, bits, bias, y, ldy, N, K,
                       dtype);
    MI355_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi355_colblock_dequant(const uint8_t* qweight, int64_t stride_n, int64_t stride_kb, const void* scales,
                                      const void* zeros, int sz_dtype, int n_groups, int tile_cols, int bits, void* out,
                                      int out_dtype, int N, int K, mi355_stream_t stream) {
    MI355_CHECK_ARG(qweight && scales && zeros && out, MI355_E_ARG, "colblock_dequant: null pointer");
    MI355_CHECK_ARG(bits == 4 || bits == 8, MI355_E_ARG, "colblock_dequant: bits must be 4 or 8");
    MI355_CHECK_ARG(N > 0 && K > 0 && tile_cols > 0 && n_groups == (K + tile_cols - 1) / tile_cols, MI355_E_SHAPE,
                    "colblock_dequant: bad shape");
    MI355_CHECK_ARG(dtype_ok(out_dtype) && dtype_ok(sz_dtype), MI355_E_DTYPE, "colblock_dequant: bad dtype");
    hipLaunchKernelGGL(colblock_dequant_kernel, dim3(ew_grid((int64_t)N * K)), dim3(256), 0, (hipStream_t)stream,
                       qweight, stride_n, stride_kb, scales, zeros, sz_dtype, n_groups, tile_cols, bits, out, out_dtype,
                       N, K);
    MI355_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi355_rmsnorm(const void* x, int64_t ldx, const void* scale, int scale_dtype, float eps, void* y,
                             int64_t ldy, int M, int C, int x_dtype, int y_dtype, mi355_stream_t stream) {
    MI355_CHECK_ARG(x && scale && y, MI355_E_ARG, "rmsnorm: null pointer");
    MI355_CHECK_ARG(M > 0 && C > 0, MI355_E_SHAPE, "rmsnorm: bad shape");
    MI355_CHECK_ARG(dtype_ok(x_dtype) && dtype_ok(y_dtype) && dtype_ok(scale_dtype), MI355_E_DTYPE, "rmsnorm: bad dtype");
    hipLaunchKernelGGL(rmsnorm_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, x, ldx, scale, scale_dtype, eps, y,
                       ldy, C, x_dtype, y_dtype);
    MI355_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi355_apply_rope(const void* x, const float* rope, void* y, int B, int T, int n_head, int hs, int dtype,
                                mi355_stream_t stream) {
    MI355_CHECK_ARG(x && rope && y, MI355_E_ARG, "apply_rope: null pointer");
    MI355_CHECK_ARG(B > 0 && T > 0 && n_head > 0 && hs > 0 && hs % 2 == 0, MI355_E_SHAPE, "apply_rope: bad shape");
    MI355_CHECK_ARG(dtype_ok(dtype), MI355_E_DTYPE, "apply_rope: bad dtype");
    const int64_t n_pairs = (int64_t)B * T * n_head * (hs / 2);
    hipLaunchKernelGGL(rope_kernel, dim3(ew_grid(n_pairs)), dim3(256), 0, (hipStream_t)stream, x, rope, y, T, n_head, hs,
                       dtype, n_pairs);
    MI355_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi355_swiglu(const void* a, const void* b, void* y, int64_t n, int dtype, mi355_stream_t stream) {
    MI355_CHECK_ARG(a && b && y && n > 0, MI355_E_ARG, "swiglu: bad argument");
    MI355_CHECK_ARG(dtype_ok(dtype), MI355_E_DTYPE, "swiglu: bad dtype");
    hipLaunchKernelGGL(swiglu_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, y, n, dtype);
    MI355_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi355_add(const void* a, const void* b, void* y, int64_t n, int dtype, mi355_stream_t stream) {
    MI355_CHECK_ARG(a && b && y && n > 0, MI355_E_ARG, "add: bad argument");
    MI355_CHECK_ARG(dtype_ok(dtype), MI355_E_DTYPE, "add: bad dtype");
    hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, y, n, dtype);
    MI355_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi355_embedding(const void* idx, int idx_is_i64, const void* wte, int w_dtype, void* y, int y_dtype,
                               int M, int C, int vocab, mi355_stream_t stream) {
    MI355_CHECK_ARG(idx && wte && y, MI355_E_ARG, "embedding: null pointer");
    MI355_CHECK_ARG(M > 0 && C > 0 && vocab > 0, MI355_E_SHAPE, "embedding: bad shape");
    MI355_CHECK_ARG(dtype_ok(w_dtype) && dtype_ok(y_dtype), MI355_E_DTYPE, "embedding: bad dtype");
    hipLaunchKernelGGL(embedding_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, idx, idx_is_i64, wte, w_dtype, y,
                       y_dtype, C, vocab);
    MI355_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi355_argmax(const float* logits, int V, int32_t* out, int32_t* out2, const int32_t* out2_pos,
                            mi355_stream_t stream) {
    MI355_CHECK_ARG(logits && out && V > 0, MI355_E_ARG, "argmax: bad argument");
    hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, V, out, out2, out2_pos);
    MI355_LAUNCH_CHECK();
    return 0;
}

// GPTQ block step for gfx950: the sequential inner loop of the reference's weight quantiser
// (/root/reference lit_llama/quantization.py:573-592, GPTQQuantizer.quantize; E. Frantar et al., arXiv:2210.17323).
//
// Within one block of <= 128 columns the rows of the weight matrix are independent: for row n and column i
//     q_i   = scale * (clamp(rint(w_i / scale) + zero, 0, maxq) - zero)
//     e_i   = (w_i - q_i) / d_i                       d_i = Hinv1[i, i]
//     w_j  -= e_i * Hinv1[i, j]     for j >= i        (product rounded, then subtracted: no fused multiply-add,
//     L_i   = (w_i - q_i)^2 / d_i^2                     so the result is bit-identical to the f32 CPU reference)
// The reference runs this as ~10 tiny PyTorch launches per column (40 000 per 4096-column linear).  Here one
// 64-thread workgroup owns 64 rows for the whole block: its slice of W1 sits TRANSPOSED in LDS ([column][row], so
// the 64 lanes of a step touch 64 consecutive words — no bank conflicts), the block of Hinv is read as LDS
// broadcasts, and the O(count^2) update runs out of LDS without touching HBM.
#include "common.h"

// hipcc contracts a * b - c into one fused multiply-add by default (-ffp-contract=fast); the reference rounds the
// product and the difference separately, and so must this file.
#pragma clang fp contract(off)

namespace {

// a product the optimiser cannot fold into a later add / subtract (the pragma above does not reach the inlined
// device-library bodies of __fmul_rn / __fsub_rn: hipcc still emitted v_fma_f32 for w - s * t)
__device__ __forceinline__ float mul_rounded(float a, float b) {
    float p = a * b;
    asm volatile("" : "+v"(p));
    return p;
}

constexpr int kRows = 64;       // rows (= threads) per workgroup
constexpr int kMaxCount = 128;  // columns per block (the reference's blocksize)

__global__ __launch_bounds__(kRows) void gptq_block_kernel(const float* W1, int64_t ldw, int N, int count,
                                                           const float* Hinv1, int64_t ldh, const float* scale,
                                                           const float* zero, int64_t sz_rs, int64_t sz_cs,
                                                           float maxq, float* Q1, float* E1, float* L1,
                                                           int64_t ldo) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Hs = smem;                  // [count][count]
    float* Wt = smem + count * count;  // [count][kRows]
    const int tid = threadIdx.x;
    const int64_t row = (int64_t)blockIdx.x * kRows + tid;
    const bool valid = row < N;
    for (int idx = tid; idx < count * count; idx += kRows) Hs[idx] = Hinv1[(int64_t)(idx / count) * ldh + idx % count];
    // W1 slice: a wave reads one row at a time (coalesced), each lane scatters its column entries into Wt
    const int64_t row0 = (int64_t)blockIdx.x * kRows;
    for (int r = 0; r < kRows; ++r) {
        const bool rv = row0 + r < N;
        for (int j = tid; j < count; j += kRows) Wt[j * kRows + r] = rv ? W1[(row0 + r) * ldw + j] : 0.f;
    }
    __syncthreads();
    for (int i = 0; i < count; ++i) {
        const float w = Wt[i * kRows + tid];
        const float d = Hs[i * count + i];
        const float s = valid ? scale[row * sz_rs + i * sz_cs] : 1.f;
        const float z = valid ? zero[row * sz_rs + i * sz_cs] : 0.f;
        const float lvl = fminf(fmaxf(rintf(__fdiv_rn(w, s)) + z, 0.f), maxq);
        const float q = mul_rounded(s, lvl - z);
        const float diff = w - q;
        const float e = __fdiv_rn(diff, d);
        if (valid) {
            Q1[row * ldo + i] = q;
            E1[row * ldo + i] = e;
            L1[row * ldo + i] = __fdiv_rn(mul_rounded(diff, diff), mul_rounded(d, d));
        }
        const float* hrow = Hs + i * count;
        for (int j = i; j < count; ++j) {
            float* p = Wt + j * kRows + tid;
            *p = *p - mul_rounded(e, hrow[j]);
        }
    }
}

// Row parameters of a [N, cols] slice (GPTQQuantizer.find_params_weight, quantization.py:472-513, perchannel):
// one wave per row; IEEE divisions and round-half-even as the CPU reference (torch's GPU division by a scalar is a
// reciprocal multiplication and lands one ulp off).
__global__ __launch_bounds__(64) void gptq_row_params_kernel(const float* W, int64_t ldw, int N, int cols, float maxq,
                                                             int sym, float* scale, float* zero) {
    const int row = blockIdx.x, lane = threadIdx.x;
    float lo = INFINITY, hi = -INFINITY;
    for (int j = lane; j < cols; j += 64) {
        const float v = W[(int64_t)row * ldw + j];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o, 64));
        hi = fmaxf(hi, __shfl_xor(hi, o, 64));
    }
    if (lane == 0) {
        lo = fminf(lo, 0.f);
        hi = fmaxf(hi, 0.f);
        if (sym) {
            hi = fmaxf(fabsf(lo), hi);
            if (lo < 0.f) lo = -hi;
        }
        if (lo == 0.f && hi == 0.f) {
            lo = -1.f;
            hi = 1.f;
        }
        const float sc = __fdiv_rn(hi - lo, maxq);
        scale[row] = sc;
        zero[row] = sym ? (maxq + 1.f) * 0.5f : rintf(__fdiv_rn(-lo, sc));
    }
}

}  // namespace

extern "C" int mi355_gptq_row_params(const float* W, int64_t ldw, int N, int cols, int maxq, int sym, float* scale,
                                     float* zero, mi355_stream_t stream) {
    MI355_CHECK_ARG(W && scale && zero, MI355_E_ARG, "gptq_row_params: null pointer");
    MI355_CHECK_ARG(N > 0 && cols > 0 && ldw >= cols && maxq > 0 && maxq <= 255, MI355_E_SHAPE, "gptq_row_params: bad shape");
    hipLaunchKernelGGL(gptq_row_params_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, W, ldw, N, cols, (float)maxq, sym,
                       scale, zero);
    MI355_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi355_gptq_block(const float* W1, int64_t ldw, int N, int count, const float* Hinv1, int64_t ldh,
                                const float* scale, const float* zero, int64_t sz_row_stride, int64_t sz_col_stride,
                                int maxq, float* Q1, float* Err1, float* Loss1, int64_t ldo, mi355_stream_t stream) {
    MI355_CHECK_ARG(W1 && Hinv1 && scale && zero && Q1 && Err1 && Loss1, MI355_E_ARG, "gptq_block: null pointer");
    MI355_CHECK_ARG(N > 0 && count > 0 && count <= kMaxCount, MI355_E_SHAPE, "gptq_block: N=%d count=%d (count <= %d)", N,
                    count, kMaxCount);
    MI355_CHECK_ARG(ldw >= count && ldh >= count && ldo >= count && maxq > 0 && maxq <= 255, MI355_E_ARG,
                    "gptq_block: bad strides / maxq");
    const size_t lds = (size_t)(count * count + count * kRows) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        MI355_HIP(hipFuncSetAttribute((const void*)gptq_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL(gptq_block_kernel, dim3((N + kRows - 1) / kRows), dim3(kRows), lds, (hipStream_t)stream, W1, ldw, N,
                       count, Hinv1, ldh, scale, zero, sz_row_stride, sz_col_stride, (float)maxq, Q1, Err1, Loss1, ldo);
    MI355_LAUNCH_CHECK();
    return 0;
}

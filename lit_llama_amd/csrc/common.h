// Shared device/host helpers for libmi355llama (gfx950 only; no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mi355_llama.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

typedef uint16_t bf16_t;  // raw bf16 bits
typedef uint16_t f16_t;   // raw f16 bits

// ---------------------------------------------------------------- error plumbing (host)
void mi355_set_error(const char* fmt, ...);

#define MI355_CHECK_ARG(cond, code, ...)          \
    do {                                          \
        if (!(cond)) {                            \
            mi355_set_error(__VA_ARGS__);         \
            return (code);                        \
        }                                         \
    } while (0)

#define MI355_HIP(expr)                                                                    \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess) {                                                           \
            mi355_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return (int)e__;                                                               \
        }                                                                                  \
    } while (0)

#define MI355_LAUNCH_CHECK()                                                               \
    do {                                                                                   \
        hipError_t e__ = hipGetLastError();                                                \
        if (e__ != hipSuccess) {                                                           \
            mi355_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
            return (int)e__;                                                               \
        }                                                                                  \
    } while (0)

// Events armed by mi355_debug_time_next_launch: the next launch of a timed kernel (gemv_kernel, fused_step_kernel)
// hands them to hipExtLaunchKernel, so they receive the dispatch's own begin / end timestamps (gemv.hip).
extern thread_local hipEvent_t t_time_start, t_time_stop;

// ---------------------------------------------------------------- scalar conversions (device)
__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// round-to-nearest-even, NaN preserved (same rule as torch's float -> bfloat16)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

__device__ __forceinline__ float f16_to_f32(f16_t h) {
    _Float16 v;
    __builtin_memcpy(&v, &h, 2);
    return (float)v;
}
__device__ __forceinline__ f16_t f32_to_f16(float f) {
    _Float16 v = (_Float16)f;  // v_cvt_f16_f32: RNE
    f16_t h;
    __builtin_memcpy(&h, &v, 2);
    return h;
}

// element load/store by runtime dtype code (MI355_F32 / BF16 / F16); dtype is wave-uniform
__device__ __forceinline__ float ld_as_f32(const void* p, int64_t i, int dtype) {
    if (dtype == MI355_F32) return ((const float*)p)[i];
    if (dtype == MI355_BF16) return bf16_to_f32(((const bf16_t*)p)[i]);
    return f16_to_f32(((const f16_t*)p)[i]);
}
__device__ __forceinline__ void st_from_f32(void* p, int64_t i, int dtype, float v) {
    if (dtype == MI355_F32)
        ((float*)p)[i] = v;
    else if (dtype == MI355_BF16)
        ((bf16_t*)p)[i] = f32_to_bf16(v);
    else
        ((f16_t*)p)[i] = f32_to_f16(v);
}
// round a float through the storage dtype (what a torch tensor of that dtype would hold)
__device__ __forceinline__ float round_to(float v, int dtype) {
    if (dtype == MI355_F32) return v;
    if (dtype == MI355_BF16) return bf16_to_f32(f32_to_bf16(v));
    return f16_to_f32(f32_to_f16(v));
}

__host__ __device__ __forceinline__ int dtype_size(int dtype) { return dtype == MI355_F32 ? 4 : 2; }

// ---------------------------------------------------------------- wave / block reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum; `red` is LDS scratch of >= 32 floats; result broadcast to all threads.
// Fixed lane->wave->serial order, so the result is reproducible run to run.
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();  // protect `red` against a previous use
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < nw; ++w) t += red[w];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = red[0];
    for (int w = 1; w < nw; ++w) t = fmaxf(t, red[w]);
    return t;
}

// Sum over aligned groups of `lpr` lanes (power of two); every lane of a group ends up with the group's sum.
// Steps 1, 2, 4, 8 are DPP lane permutations inside quads / half rows / rows (xor-butterfly equivalents:
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror — full-rate VALU).  `__shfl_xor` lowers
// to ds_bpermute, an LDS round trip of ~120 cycles per step; a dependent chain of those per cached row was most of
// this kernel's time.  Only groups wider than a 16-lane row use ds_bpermute for the last step(s).
#define MI355_DPP_ADD(v, ctrl) \
    ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, false)))
// xor-16 / xor-32 lane exchanges on the VALU: gfx950's v_permlane16_swap / v_permlane32_swap swap the odd 16-lane
// rows (upper 32 lanes) of one register with the even rows (lower 32 lanes) of another; fed the same value twice,
// one of the two results holds the partner lane's value (semantics checked by scripts/micro/permlane.hip).
// ds_bpermute (what __shfl_xor lowers to) is an LDS round trip of ~100 cycles.
__device__ __forceinline__ float lane_xor16(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(((threadIdx.x >> 4) & 1) ? r[0] : r[1]);
}
__device__ __forceinline__ float lane_xor32(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}
// value of lane (lane ^ o), o a wave-uniform power of two
__device__ __forceinline__ float lane_xor(float v, int o) {
    if (o == 16) return lane_xor16(v);
    if (o == 32) return lane_xor32(v);
    return __shfl_xor(v, o, 64);
}

#define MI355_DPP_MAX(v, ctrl) \
    fmaxf((v), __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, false)))
__device__ __forceinline__ float group_sum(float v, int lpr) {
    if (lpr >= 2) v = MI355_DPP_ADD(v, 0xB1);
    if (lpr >= 4) v = MI355_DPP_ADD(v, 0x4E);
    if (lpr >= 8) v = MI355_DPP_ADD(v, 0x141);
    if (lpr >= 16) v = MI355_DPP_ADD(v, 0x140);
    if (lpr >= 32) v += lane_xor16(v);
    if (lpr >= 64) v += lane_xor32(v);
    return v;
}

// silu(a) * b in f32 (F.silu: a * sigmoid(a))
__device__ __forceinline__ float swiglu_f32(float a, float b) { return (a / (1.0f + expf(-a))) * b; }

// Block-wide index of the first maximum of logits[0..V) (lowest index wins ties, as torch.topk(.., 1) /
// argmax do on the CPU oracle); every thread gets the result.  An all-NaN row gives 0.
__device__ inline int block_argmax_first(const float* logits, int V) {
    __shared__ float sv[16];
    __shared__ int si[16];
    float best = -INFINITY;
    int bi = 0x7fffffff;
    auto upd = [&](float v, int i) {
        if (v > best || (v == best && i < bi)) {
            best = v;
            bi = i;
        }
    };
    if ((V & 3) == 0 && ((uintptr_t)logits & 15) == 0) {
        // 8 x 16-B loads per thread in flight (a dependent scalar chain took ~14 us for 32000 logits)
        const int nv = V >> 2;
        for (int base = 0; base < nv; base += blockDim.x * 8) {
            f32x4 r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int v = base + u * blockDim.x + threadIdx.x;
                if (v < nv) r[u] = ((const f32x4*)logits)[v];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int v = base + u * blockDim.x + threadIdx.x;
                if (v < nv) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) upd(r[u][q], 4 * v + q);
                }
            }
        }
    } else {
        for (int i = threadIdx.x; i < V; i += blockDim.x) upd(logits[i], i);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) {
        sv[wave] = best;
        si[wave] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
                best = sv[w];
                bi = si[w];
            }
        if (bi == 0x7fffffff) bi = 0;
        si[0] = bi;
    }
    __syncthreads();
    return si[0];
}

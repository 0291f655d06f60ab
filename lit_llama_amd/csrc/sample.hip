// Sampling of the next token on the device: temperature, exact top-k threshold, softmax, inverse-CDF multinomial from a
// caller-supplied uniform — the tail of the reference's generate loop, /root/reference generate.py:68-85:
//     logits = logits[0, -1] / temperature
//     v, _ = torch.topk(logits, min(top_k, V));  logits = torch.where(logits < v[[-1]], -inf, logits)
//     probs = softmax(logits);  idx_next = torch.multinomial(probs, 1)
// as ONE launch that ends the decode step the way the greedy chain does (next token id, position + 1, output slot),
// so a sampled run needs no device->host read and no torch op per token (the reference-style loop over model.forward
// ran 604 tok/s against 725 greedy in round 1).
//
// torch.multinomial draws its own noise, so a sample cannot be reproduced bit for bit; what is pinned instead
// (tests/test_sampling_gpu.py): the kept set {i : logit_i >= k-th largest}, the probabilities, and the inverse-CDF rule
//     token = min { i : sum_{j <= i} p_j > u }      (index order, u in [0, 1) from the caller's generator).
// One workgroup of 1024 threads: the k-th largest value by a 4-pass MSB radix select over order-preserving keys
// (exact, ties included, as `logits < v[-1]` keeps them), masked max / sum, a block scan of per-thread partial sums.
#include "common.h"

namespace {

constexpr int kT = 1024;

__device__ __forceinline__ unsigned fkey(float f) {  // ascending float order -> ascending unsigned order
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float unkey(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(kT) void sample_kernel(const float* logits, int V, float temperature, int top_k,
                                                    const float* uniforms, int32_t* next_token, int32_t* out_tokens,
                                                    int32_t* tokens, int32_t* pos, int advance, float* probs_out,
                                                    int use_lds) {
    extern __shared__ __attribute__((aligned(16))) float vals[];  // the scaled logits, when the vocabulary fits LDS
    __shared__ unsigned hist[16][256];  // one histogram per wave: a shared one serialises on the hot bins
    __shared__ unsigned sel_prefix, sel_rank;
    __shared__ float red[32];
    __shared__ float part[kT];
    __shared__ int winner;
    const int tid = threadIdx.x;
    const int ps = pos[0];
    const float u = uniforms[ps];
    const bool cached = use_lds != 0;
    if (cached) {
        for (int i = tid; i < V; i += kT) vals[i] = logits[i] / temperature;  // IEEE division, as `logits / temperature`
        __syncthreads();
    }
    auto lg = [&](int i) { return cached ? vals[i] : logits[i] / temperature; };

    // ---- threshold = the k-th largest scaled logit (none when top_k covers the vocabulary)
    float thr = -INFINITY;
    if (top_k > 0 && top_k < V) {
        if (tid == 0) {
            sel_prefix = 0u;
            sel_rank = (unsigned)(V - top_k);  // 0-based rank, ascending, of the k-th largest
        }
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            for (int b = tid; b < 16 * 256; b += kT) (&hist[0][0])[b] = 0u;
            __syncthreads();
            const unsigned prefix = sel_prefix, mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
            for (int i = tid; i < V; i += kT) {
                const unsigned k = fkey(lg(i));
                if ((k & mask) == prefix) atomicAdd(&hist[tid >> 6][(k >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid < 256) {
                unsigned t = 0;
#pragma unroll
                for (int w = 0; w < 16; ++w) t += hist[w][tid];
                hist[0][tid] = t;
            }
            __syncthreads();
            if (tid == 0) {
                unsigned r = sel_rank, b = 0;
                for (; b < 256; ++b) {
                    if (r < hist[0][b]) break;
                    r -= hist[0][b];
                }
                sel_prefix = prefix | (b << shift);
                sel_rank = r;
            }
            __syncthreads();
        }
        thr = unkey(sel_prefix);
    }
    // ---- masked max and sum of exponentials
    float mx = -INFINITY;
    for (int i = tid; i < V; i += kT) {
        const float v = lg(i);
        if (v >= thr) mx = fmaxf(mx, v);
    }
    mx = block_max(mx, red);
    // per-thread partial sums over CONTIGUOUS index ranges (the CDF runs in index order)
    const int chunk = (V + kT - 1) / kT;
    const int i0 = tid * chunk, i1 = i0 + chunk < V ? i0 + chunk : V;
    float local = 0.f;
    for (int i = i0; i < i1; ++i) {
        const float v = lg(i);
        if (v >= thr) local += expf(v - mx);
    }
    part[tid] = local;
    __syncthreads();
    for (int off = 1; off < kT; off <<= 1) {  // inclusive scan
        const float add = tid >= off ? part[tid - off] : 0.f;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    const float total = part[kT - 1];
    if (probs_out != nullptr) {
        for (int i = tid; i < V; i += kT) {
            const float v = lg(i);
            probs_out[i] = v >= thr ? expf(v - mx) / total : 0.f;
        }
    }
    // ---- inverse CDF: smallest i with cumulative mass > u * total; u * total >= total (rounding) -> last kept index
    const float target = u * total;
    if (tid == 0) winner = -1;
    __syncthreads();
    const float before = tid ? part[tid - 1] : 0.f;
    if (target >= before && target < part[tid] && local > 0.f) {
        float acc = before;
        int pick = -1;
        for (int i = i0; i < i1; ++i) {
            const float v = lg(i);
            if (v >= thr) {
                acc += expf(v - mx);
                pick = i;
                if (acc > target) break;
            }
        }
        winner = pick;  // exactly one thread's interval contains the target
    }
    __syncthreads();
    if (winner < 0) {  // target fell on / past the total: the last kept index
        int last = -1;
        for (int i = i1 - 1; i >= i0; --i)
            if (lg(i) >= thr) {
                last = i;
                break;
            }
        __shared__ int lastk;
        if (tid == 0) lastk = -1;
        __syncthreads();
        if (last >= 0) atomicMax(&lastk, last);
        __syncthreads();
        if (tid == 0) winner = lastk < 0 ? 0 : lastk;
        __syncthreads();
    }
    if (tid == 0) {
        const int w = winner;
        next_token[0] = w;
        if (out_tokens != nullptr) out_tokens[ps + 1] = w;
        if (advance) {
            tokens[0] = w;
            pos[0] = ps + 1;
        }
    }
}

}  // namespace

extern "C" int mi355_sample(const float* logits, int V, float temperature, int top_k, const float* uniforms,
                            int32_t* next_token, int32_t* out_tokens, int32_t* tokens, int32_t* pos, int advance,
                            float* probs_out, mi355_stream_t stream) {
    MI355_CHECK_ARG(logits && uniforms && next_token && pos, MI355_E_ARG, "sample: null argument");
    MI355_CHECK_ARG(V >= 1, MI355_E_SHAPE, "sample: V=%d", V);
    MI355_CHECK_ARG(temperature > 0.f, MI355_E_ARG, "sample: temperature must be positive (greedy decoding is top_k = 1)");
    MI355_CHECK_ARG(!advance || tokens != nullptr, MI355_E_ARG, "sample: advance needs the token slot");
    const size_t lds = (size_t)V * sizeof(float);
    const int use_lds = lds <= 120 * 1024;
    if (use_lds) {
        static hipError_t attr_err =
            hipFuncSetAttribute((const void*)sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
        MI355_CHECK_ARG(attr_err == hipSuccess, (int)attr_err, "sample: hipFuncSetAttribute failed: %s", hipGetErrorString(attr_err));
    }
    hipLaunchKernelGGL(sample_kernel, dim3(1), dim3(kT), use_lds ? lds : 0, (hipStream_t)stream, logits, V, temperature,
                       top_k, uniforms, next_token, out_tokens, tokens, pos, advance, probs_out, use_lds);
    MI355_LAUNCH_CHECK();
    return 0;
}

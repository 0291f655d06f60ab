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
// One workgroup of 1024 threads.  Vocabularies up to 32768 (sample_kernel) live in LDS, 32 consecutive scaled
// logits per thread (the CDF runs in index order).  The k-th largest value is found exactly in two cheap steps: a
// 255-bin LINEAR histogram of (max - v) over the above-mean half (per-wave LDS histograms; bins are monotone in v, so
// the bin holding the k-th largest and its rank inside that bin are exact), then an all-pairs rank among the handful of
// values of that bin (ties included, as `logits < v[-1]` keeps them).  Degenerate inputs (k beyond the above-mean half,
// a crowded bin, non-finite range) take the 4-pass MSB radix select over order-preserving keys instead, as does
// sample_kernel_large (vocabularies past 32768, values re-read from LDS / memory).  Then masked sum of exponentials
// (the maximum is always kept), a block scan of the per-thread partial sums, and the walk inside the one thread whose
// interval holds the target.  The sampled run has to stay within 5 % of the greedy rate: see the note on code size
// at sample_kernel.
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

__global__ __launch_bounds__(kT) void sample_kernel_large(const float* logits, int V, float temperature, int top_k,
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


constexpr int kCH = 32;  // scaled logits per thread of the LDS-resident kernel: vocabularies up to 32768

__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// The code of this kernel is fetched cold on every token (the decode step before it streams 3.4 GB through the L2, and
// ONE workgroup runs it): every loop below stays rolled (`unroll 1`) on purpose.  A fully unrolled, register-resident
// version was 27 us back to back and 157 us behind a decode step: 10k straight-line instructions at instruction-miss
// latency.  The scaled logits live in LDS with one pad word per 32, so that both the coalesced fill (consecutive
// threads, consecutive words) and the thread-contiguous passes (thread t owns indices [t ch, (t + 1) ch): the CDF runs
// in index order) are free of bank conflicts.
__global__ __launch_bounds__(kT) void sample_kernel(const float* logits, int V, float temperature, int top_k,
                                                    const float* uniforms, int32_t* next_token, int32_t* out_tokens,
                                                    int32_t* tokens, int32_t* pos, int advance, float* probs_out) {
    extern __shared__ __attribute__((aligned(16))) float vals[];  // [V + V / 32]
    __shared__ unsigned hist[16][256];  // one histogram per wave: a shared one serialises on the hot bins
    __shared__ float red[32];
    __shared__ float cand[kT];
    __shared__ unsigned sh_bin, sh_rank, sh_n, sh_cnt, sh_thr, sel_prefix, sel_rank;
    __shared__ int winner, lastk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ps = pos[0];
    const float u = uniforms[ps];
    const int ch = (V + kT - 1) / kT;  // <= kCH (host check)
    const int i0 = tid * ch;
    const int i1 = i0 + ch < V ? i0 + ch : V;  // this thread's indices: [i0, i1) (empty past the vocabulary)
    auto at = [&](int i) -> float& { return vals[i + (i >> 5)]; };
    // fill: 8 coalesced loads in flight per thread and round (the row was just written by another kernel: memory latency)
#pragma unroll 1
    for (int b = 0; b < V; b += 8 * kT) {
        float t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = b + q * kT + tid;
            t[q] = i < V ? logits[i] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = b + q * kT + tid;
            if (i < V) at(i) = t[q] / temperature;  // IEEE division, as `logits / temperature`
        }
    }
    if (tid == 0) {
        sh_bin = 255u;
        sh_cnt = 0u;
        sh_thr = __float_as_uint(-INFINITY);
        winner = -1;
        lastk = -1;
    }
    __syncthreads();
    float mx = -INFINITY, sum = 0.f;
#pragma unroll 1
    for (int i = i0; i < i1; ++i) {
        const float x = at(i);
        mx = fmaxf(mx, x);
        sum += x;
    }
    mx = block_max(mx, red);  // the maximum is kept by every top-k threshold: also the softmax maximum

    // ---- threshold = the k-th largest scaled logit (none when top_k covers the vocabulary)
    float thr = -INFINITY;
    if (top_k > 0 && top_k < V) {
        const unsigned k = (unsigned)top_k;
        const float range = mx - block_sum(sum, red) / (float)V;  // max - mean
        bool fast = range > 0.f && range < INFINITY;                // (NaN compares false)
        const float sc = 255.0f / range;
        if (fast) {
#pragma unroll 1
            for (int b = tid; b < 16 * 256; b += kT) (&hist[0][0])[b] = 0u;
            __syncthreads();
#pragma unroll 1
            for (int i = i0; i < i1; ++i) {
                const int b = (int)fminf((mx - at(i)) * sc, 255.f);  // monotone in the value; below the mean -> 255
                if (b < 255) atomicAdd(&hist[wave][b], 1u);
            }
            __syncthreads();
            unsigned t = 0, incl = 0;
            if (tid < 256) {
#pragma unroll 1
                for (int w = 0; w < 16; ++w) t += hist[w][tid];
                incl = t;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned x = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += x;
                }
                if (lane == 63) red[wave] = __uint_as_float(incl);
            }
            __syncthreads();
            if (tid < 255) {
#pragma unroll 1
                for (int w = 0; w < wave; ++w) incl += __float_as_uint(red[w]);
                if (incl >= k && incl - t < k) {  // the bin holding the k-th largest value
                    sh_bin = (unsigned)tid;
                    sh_rank = k - (incl - t);      // its 1-based rank, from the top, inside the bin
                    sh_n = t;
                }
            }
            __syncthreads();
            const unsigned bsel = sh_bin, n = sh_n, r = sh_rank;
            fast = bsel < 255u && n <= (unsigned)kT;
            if (fast) {
#pragma unroll 1
                for (int i = i0; i < i1; ++i) {
                    const float x = at(i);
                    if ((int)fminf((mx - x) * sc, 255.f) == (int)bsel) cand[atomicAdd(&sh_cnt, 1u)] = x;
                }
                __syncthreads();
                if ((unsigned)tid < n) {
                    const float c = cand[tid];
                    unsigned g = 0, ge = 0;
#pragma unroll 1
                    for (unsigned q = 0; q < n; ++q) {
                        const float x = cand[q];
                        g += x > c;
                        ge += x >= c;
                    }
                    if (g < r && r <= ge) sh_thr = __float_as_uint(c);  // (ties write the same value)
                }
                __syncthreads();
                thr = __uint_as_float(sh_thr);
            }
        }
        if (!fast) {  // exact 4-pass MSB radix select over order-preserving keys
            if (tid == 0) {
                sel_prefix = 0u;
                sel_rank = (unsigned)(V - top_k);  // 0-based rank, ascending, of the k-th largest
            }
#pragma unroll 1
            for (int pass = 0; pass < 4; ++pass) {
                const int shift = 24 - 8 * pass;
                __syncthreads();
#pragma unroll 1
                for (int b = tid; b < 16 * 256; b += kT) (&hist[0][0])[b] = 0u;
                __syncthreads();
                const unsigned prefix = sel_prefix, mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
#pragma unroll 1
                for (int i = i0; i < i1; ++i) {
                    const unsigned key = fkey(at(i));
                    if ((key & mask) == prefix) atomicAdd(&hist[wave][(key >> shift) & 255u], 1u);
                }
                __syncthreads();
                if (tid < 256) {
                    unsigned t = 0;
#pragma unroll 1
                    for (int w = 0; w < 16; ++w) t += hist[w][tid];
                    hist[0][tid] = t;
                }
                __syncthreads();
                if (tid == 0) {
                    unsigned r = sel_rank, b = 0;
#pragma unroll 1
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
    }
    // ---- masked exponentials (kept in place of the values; dropped entries become 0), per-thread partial sums over
    // CONTIGUOUS index ranges, inclusive scan over the threads
    float local = 0.f;
#pragma unroll 1
    for (int i = i0; i < i1; ++i) {
        const float x = at(i);
        const float e = x >= thr ? expf(x - mx) : 0.f;
        at(i) = x >= thr ? e : -1.f;  // -1: dropped (a kept entry may underflow to 0 and stays kept)
        local += e;
    }
    float after = wave_incl_scan(local, lane);
    __syncthreads();
    if (lane == 63) red[wave] = after;
    __syncthreads();
    float wbase = 0.f, total = 0.f;
#pragma unroll 1
    for (int w = 0; w < 16; ++w) {
        if (w == wave) wbase = total;
        total += red[w];
    }
    after += wbase;
    if (probs_out != nullptr) {
#pragma unroll 1
        for (int i = tid; i < V; i += kT) probs_out[i] = fmaxf(at(i), 0.f) / total;
    }
    // ---- inverse CDF: smallest i with cumulative mass > u * total; u * total >= total (rounding) -> last kept index
    const float target = u * total;
    {
        const float lo = __shfl_up(after, 1, 64);
        const float prev = lane ? lo : wbase;  // inclusive sum of the previous thread, bit for bit
        if (target >= prev && target < after && local > 0.f) {
            // the walk re-adds this thread's terms in the order of its partial sum
            float acc = prev;
            int pick = -1;
#pragma unroll 1
            for (int i = i0; i < i1; ++i) {
                const float e = at(i);
                if (e >= 0.f) {
                    acc += e;
                    pick = i;
                    if (acc > target) break;
                }
            }
            winner = pick;  // exactly one thread's interval contains the target
        }
    }
    __syncthreads();
    if (winner < 0) {  // target fell on / past the total: the last kept index
        int last = -1;
#pragma unroll 1
        for (int i = i0; i < i1; ++i)
            if (at(i) >= 0.f) last = i;
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
    if (V <= kT * kCH) {
        static hipError_t attr_err =
            hipFuncSetAttribute((const void*)sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (kT * kCH + kT * kCH / 32 + 1) * 4);  // = the request below at V = kT * kCH, the documented bound
        MI355_CHECK_ARG(attr_err == hipSuccess, (int)attr_err, "sample: hipFuncSetAttribute failed: %s", hipGetErrorString(attr_err));
        hipLaunchKernelGGL(sample_kernel, dim3(1), dim3(kT), (size_t)(V + V / 32 + 1) * 4, (hipStream_t)stream, logits, V,
                           temperature, top_k, uniforms, next_token, out_tokens, tokens, pos, advance, probs_out);
    } else {
        const size_t lds = (size_t)V * sizeof(float);
        const int use_lds = lds <= 120 * 1024;
        if (use_lds) {
            static hipError_t attr_err = hipFuncSetAttribute((const void*)sample_kernel_large,
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
            MI355_CHECK_ARG(attr_err == hipSuccess, (int)attr_err, "sample: hipFuncSetAttribute failed: %s",
                            hipGetErrorString(attr_err));
        }
        hipLaunchKernelGGL(sample_kernel_large, dim3(1), dim3(kT), use_lds ? lds : 0, (hipStream_t)stream, logits, V,
                           temperature, top_k, uniforms, next_token, out_tokens, tokens, pos, advance, probs_out, use_lds);
    }
    MI355_LAUNCH_CHECK();
    return 0;
}

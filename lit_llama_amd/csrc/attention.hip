// RoPE + KV-cache write + causal attention for gfx950.
//
// Replaces the body of CausalSelfAttention.forward between the two linears
// (/root/reference lit_llama/model.py:199-232): q/k/v split, apply_rope (:306-323), the out-of-place
// cache index_copy (:219-220, here an in-place row write), and F.scaled_dot_product_attention with the
// boolean causal mask (:230) — which the reference evaluates over all S cache rows; here only rows
// [0, slot] are read.
//
// One workgroup per (head, query token, batch row).  K/V rows are streamed with 16-B lane loads
// (LPR = row_bytes / 16 lanes per row, 64 / LPR rows per wave instruction), scores reduced with
// wavefront shuffles, softmax in LDS, PV accumulated per lane and combined across waves in a fixed
// order.  For the decode step (T == 1) the kernel also applies RoPE to the new key, writes the new
// K/V row into the cache and uses its LDS copy for the current position, so no other kernel (and no
// global round trip) sits between the qkv projection and the attention output.
#include "common.h"

namespace {

struct AttnParams {
    const void* qkv;
    const float* rope;
    const int32_t* pos;
    void* kcache;
    void* vcache;
    void* y;
    int64_t ld_qkv, ldy;
    int qkv_dtype, y_dtype;
    int B, T, n_head, hs, S;
    int fused;  // 1: this kernel writes the (single) new K/V row itself
    int rope_gathered;  // rope row of token t is t (rows pre-selected by the caller), not pos[t]
    float scale;
};

template <typename CT>
__device__ __forceinline__ float ct_to_f32(CT v);
template <>
__device__ __forceinline__ float ct_to_f32<float>(float v) {
    return v;
}
template <>
__device__ __forceinline__ float ct_to_f32<bf16_t>(bf16_t v) {
    return bf16_to_f32(v);
}
template <typename CT>
__device__ __forceinline__ CT f32_to_ct(float v);
template <>
__device__ __forceinline__ float f32_to_ct<float>(float v) {
    return v;
}
template <>
__device__ __forceinline__ bf16_t f32_to_ct<bf16_t>(float v) {
    return f32_to_bf16(v);
}

template <typename CT>
struct Vec16 {
    static constexpr int kN = 16 / sizeof(CT);
};

// unpack a 16-B piece into kN floats
template <typename CT>
__device__ __forceinline__ void unpack16(const u32x4& raw, float (&out)[Vec16<CT>::kN]) {
    if constexpr (sizeof(CT) == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = __uint_as_float(raw[i]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            out[2 * i] = __uint_as_float(raw[i] << 16);
            out[2 * i + 1] = __uint_as_float(raw[i] & 0xffff0000u);
        }
    }
}

// RoPE of one interleaved pair at absolute position `pos` (lit_llama/model.py:314-318)
__device__ __forceinline__ void rope_pair(const float* rope, int pos, int half, int pi, float a, float b, float& oa,
                                          float& ob) {
    const float c = rope[((int64_t)pos * half + pi) * 2], s = rope[((int64_t)pos * half + pi) * 2 + 1];
    oa = a * c - b * s;
    ob = b * c + a * s;
}

// grid (n_head, T, B): write RoPE'd K and V rows of the T new tokens into the cache (T > 1 / no-cache path)
template <typename CT>
__global__ void rope_kv_write_kernel(const AttnParams p) {
    const int h = blockIdx.x, t = blockIdx.y, b = blockIdx.z;
    const int hs = p.hs, half = hs >> 1, C = p.n_head * hs;
    const int pos = p.pos ? p.pos[t] : t;
    const int slot = pos < p.S - 1 ? pos : p.S - 1;
    const int64_t row = ((int64_t)b * p.T + t) * p.ld_qkv;
    CT* kc = (CT*)p.kcache + (((int64_t)b * p.n_head + h) * p.S + slot) * hs;
    CT* vc = (CT*)p.vcache + (((int64_t)b * p.n_head + h) * p.S + slot) * hs;
    for (int pi = threadIdx.x; pi < half; pi += blockDim.x) {
        const float a = ld_as_f32(p.qkv, row + C + h * hs + 2 * pi, p.qkv_dtype);
        const float bb = ld_as_f32(p.qkv, row + C + h * hs + 2 * pi + 1, p.qkv_dtype);
        float oa, ob;
        rope_pair(p.rope, p.rope_gathered ? t : pos, half, pi, a, bb, oa, ob);
        kc[2 * pi] = f32_to_ct<CT>(oa);
        kc[2 * pi + 1] = f32_to_ct<CT>(ob);
    }
    for (int d = threadIdx.x; d < hs; d += blockDim.x)
        vc[d] = f32_to_ct<CT>(ld_as_f32(p.qkv, row + 2 * C + h * hs + d, p.qkv_dtype));
}

// Dynamic LDS: qs[hs] kcur[hs] vcur[hs] opart[nw][hs] red[32] scores[len]
template <typename CT>
__global__ __launch_bounds__(512) void attn_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int VEC = Vec16<CT>::kN;
    const int h = blockIdx.x, t = blockIdx.y, b = blockIdx.z;
    const int hs = p.hs, half = hs >> 1, C = p.n_head * hs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;

    float* qs = (float*)smem;
    float* kcur = qs + hs;
    float* vcur = kcur + hs;
    float* opart = vcur + hs;
    float* red = opart + nw * hs;
    float* scores = red + 32;

    const int pos = p.pos ? p.pos[t] : t;
    const int slot = pos < p.S - 1 ? pos : p.S - 1;
    const int len = slot + 1;
    const int n_glob = p.fused ? slot : len;  // rows read from the cache in global memory

    const int64_t row = ((int64_t)b * p.T + t) * p.ld_qkv;
    const CT* kc = (const CT*)p.kcache + ((int64_t)b * p.n_head + h) * p.S * hs;
    const CT* vc = (const CT*)p.vcache + ((int64_t)b * p.n_head + h) * p.S * hs;

    // ---- phase 0: q (and, fused, the new k / v row) through RoPE into LDS
    for (int pi = tid; pi < half; pi += blockDim.x) {
        const float a = ld_as_f32(p.qkv, row + h * hs + 2 * pi, p.qkv_dtype);
        const float bb = ld_as_f32(p.qkv, row + h * hs + 2 * pi + 1, p.qkv_dtype);
        float oa, ob;
        const int rrow = p.rope_gathered ? t : pos;
        rope_pair(p.rope, rrow, half, pi, a, bb, oa, ob);
        qs[2 * pi] = oa;
        qs[2 * pi + 1] = ob;
        if (p.fused) {
            const float ka = ld_as_f32(p.qkv, row + C + h * hs + 2 * pi, p.qkv_dtype);
            const float kb = ld_as_f32(p.qkv, row + C + h * hs + 2 * pi + 1, p.qkv_dtype);
            rope_pair(p.rope, rrow, half, pi, ka, kb, oa, ob);
            const CT ca = f32_to_ct<CT>(oa), cb = f32_to_ct<CT>(ob);
            CT* kw = (CT*)p.kcache + (((int64_t)b * p.n_head + h) * p.S + slot) * hs;
            kw[2 * pi] = ca;
            kw[2 * pi + 1] = cb;
            kcur[2 * pi] = ct_to_f32<CT>(ca);
            kcur[2 * pi + 1] = ct_to_f32<CT>(cb);
        }
    }
    if (p.fused) {
        for (int d = tid; d < hs; d += blockDim.x) {
            const CT cv = f32_to_ct<CT>(ld_as_f32(p.qkv, row + 2 * C + h * hs + d, p.qkv_dtype));
            ((CT*)p.vcache)[(((int64_t)b * p.n_head + h) * p.S + slot) * hs + d] = cv;
            vcur[d] = ct_to_f32<CT>(cv);
        }
    }
    __syncthreads();

    const int row_bytes = hs * (int)sizeof(CT);
    const bool vec_ok = (row_bytes % 16 == 0) && ((row_bytes / 16) <= 64) && (((row_bytes / 16) & ((row_bytes / 16) - 1)) == 0);
    const int LPR = vec_ok ? row_bytes / 16 : 64;  // lanes per row
    const int rpw = 64 / LPR;                      // rows per wave instruction
    const int li = lane % LPR, lr = lane / LPR;

    // ---- phase 1: scores[s] = scale * <q, K[s]>
    if (vec_ok) {
        float qf[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) qf[j] = qs[li * VEC + j];
        const int stride = nw * rpw;
        // wave-uniform trip count: the shuffles below need every lane of a row group in the loop
        for (int base = 0; base < n_glob; base += stride * 4) {
            const int s0 = base + wave * rpw + lr;
            u32x4 raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = s0 + u * stride;
                if (s < n_glob) raw[u] = *(const u32x4*)((const char*)kc + (int64_t)s * row_bytes + li * 16);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = s0 + u * stride;
                float dot = 0.f;
                if (s < n_glob) {
                    float kf[VEC];
                    unpack16<CT>(raw[u], kf);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) dot += qf[j] * kf[j];
                }
                for (int o = LPR >> 1; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
                if (s < n_glob && li == 0) scores[s] = dot * p.scale;
            }
        }
    } else {
        for (int s = wave; s < n_glob; s += nw) {
            float dot = 0.f;
            for (int d = lane; d < hs; d += 64) dot += qs[d] * ct_to_f32<CT>(kc[(int64_t)s * hs + d]);
            dot = wave_sum(dot);
            if (lane == 0) scores[s] = dot * p.scale;
        }
    }
    if (p.fused && wave == nw - 1) {
        float dot = 0.f;
        for (int d = lane; d < hs; d += 64) dot += qs[d] * kcur[d];
        dot = wave_sum(dot);
        if (lane == 0) scores[slot] = dot * p.scale;
    }
    __syncthreads();

    // ---- phase 2: softmax over [0, len)
    float mx = -INFINITY;
    for (int s = tid; s < len; s += blockDim.x) mx = fmaxf(mx, scores[s]);
    mx = block_max(mx, red);
    float sum = 0.f;
    for (int s = tid; s < len; s += blockDim.x) {
        const float e = expf(scores[s] - mx);
        scores[s] = e;
        sum += e;
    }
    sum = block_sum(sum, red);  // includes the barriers that publish scores[]
    const float inv = 1.0f / sum;

    // ---- phase 3: o[d] = sum_s p[s] V[s][d]
    if (vec_ok) {
        float of[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) of[j] = 0.f;
        const int stride = nw * rpw;
        for (int base = 0; base < n_glob; base += stride * 4) {
            const int s0 = base + wave * rpw + lr;
            u32x4 raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = s0 + u * stride;
                if (s < n_glob) raw[u] = *(const u32x4*)((const char*)vc + (int64_t)s * row_bytes + li * 16);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = s0 + u * stride;
                if (s < n_glob) {
                    float vf[VEC];
                    unpack16<CT>(raw[u], vf);
                    const float pr = scores[s];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) of[j] += pr * vf[j];
                }
            }
        }
        // combine the rpw row groups of the wave (lanes with equal li)
        for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) of[j] += __shfl_xor(of[j], o, 64);
        }
        if (lr == 0) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) opart[wave * hs + li * VEC + j] = of[j];
        }
    } else {
        for (int d = lane; d < hs; d += 64) {
            float o = 0.f;
            for (int s = wave; s < n_glob; s += nw) o += scores[s] * ct_to_f32<CT>(vc[(int64_t)s * hs + d]);
            opart[wave * hs + d] = o;
        }
    }
    __syncthreads();
    for (int d = tid; d < hs; d += blockDim.x) {
        float o = 0.f;
        for (int w = 0; w < nw; ++w) o += opart[w * hs + d];
        if (p.fused) o += scores[slot] * vcur[d];
        st_from_f32(p.y, ((int64_t)b * p.T + t) * p.ldy + h * hs + d, p.y_dtype, o * inv);
    }
}

template <typename CT>
__global__ void kv_roll_kernel(CT* kcache, CT* vcache, int S, int hs) {
    // blockIdx.x enumerates (b, head); blockIdx.y: 0 = K, 1 = V.  Column d is owned by one thread, rows
    // are shifted in ascending order, so the in-place shift needs no extra storage.
    CT* base = (blockIdx.y == 0 ? kcache : vcache) + (int64_t)blockIdx.x * S * hs;
    for (int d = threadIdx.x; d < hs; d += blockDim.x)
        for (int s = 0; s + 1 < S; ++s) base[(int64_t)s * hs + d] = base[(int64_t)(s + 1) * hs + d];
}

}  // namespace

extern "C" int mi355_attention(const mi355_attn_args* a, mi355_stream_t stream) {
    MI355_CHECK_ARG(a != nullptr, MI355_E_ARG, "attention: null args");
    MI355_CHECK_ARG(a->qkv && a->rope && a->y, MI355_E_ARG, "attention: null qkv/rope/y");
    MI355_CHECK_ARG(a->B > 0 && a->T > 0 && a->n_head > 0 && a->hs > 0 && a->hs % 2 == 0, MI355_E_SHAPE,
                    "attention: bad shape B=%d T=%d n_head=%d hs=%d", a->B, a->T, a->n_head, a->hs);
    MI355_CHECK_ARG(a->B <= 65535 && a->T <= 65535, MI355_E_SHAPE, "attention: B/T too large for the grid");
    MI355_CHECK_ARG(a->cache_dtype == MI355_F32 || a->cache_dtype == MI355_BF16, MI355_E_DTYPE,
                    "attention: cache dtype must be f32 or bf16");
    const bool has_cache = a->kcache != nullptr && a->vcache != nullptr;
    MI355_CHECK_ARG(has_cache || a->kv_tmp != nullptr, MI355_E_ARG, "attention: no cache and no kv_tmp scratch");
    MI355_CHECK_ARG(!has_cache || a->pos != nullptr, MI355_E_ARG, "attention: cache given without positions");
    MI355_CHECK_ARG(!has_cache || a->S > 0, MI355_E_SHAPE, "attention: S must be positive");

    AttnParams p;
    p.qkv = a->qkv;
    p.rope = a->rope;
    p.y = a->y;
    p.ld_qkv = a->ld_qkv;
    p.ldy = a->ldy;
    p.qkv_dtype = a->qkv_dtype;
    p.y_dtype = a->y_dtype;
    p.B = a->B;
    p.T = a->T;
    p.n_head = a->n_head;
    p.hs = a->hs;
    p.scale = 1.0f / sqrtf((float)a->hs);
    const int esz = a->cache_dtype == MI355_F32 ? 4 : 2;
    if (has_cache) {
        p.pos = a->pos;
        p.kcache = a->kcache;
        p.vcache = a->vcache;
        p.S = a->S;
    } else {
        p.pos = nullptr;  // position t, slot t
        p.kcache = a->kv_tmp;
        p.vcache = (char*)a->kv_tmp + (size_t)a->B * a->n_head * a->T * a->hs * esz;
        p.S = a->T;
    }
    p.fused = (has_cache && a->T == 1) ? 1 : 0;
    p.rope_gathered = a->rope_gathered;

    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(a->n_head, a->T, a->B);
    if (!p.fused) {
        const int thr = a->hs >= 128 ? 128 : 64;
        if (esz == 4)
            hipLaunchKernelGGL(rope_kv_write_kernel<float>, grid, dim3(thr), 0, s, p);
        else
            hipLaunchKernelGGL(rope_kv_write_kernel<bf16_t>, grid, dim3(thr), 0, s, p);
        MI355_LAUNCH_CHECK();
    }
    const int threads = 512, nw = threads / 64;
    const size_t lds = (size_t)(3 * a->hs + nw * a->hs + 32 + p.S) * sizeof(float) + 16;
    MI355_CHECK_ARG(lds <= 160 * 1024, MI355_E_SHAPE, "attention: S=%d hs=%d needs %zu B of LDS", p.S, a->hs, lds);
    static bool attr_done = false;
    if (!attr_done) {
        MI355_HIP(hipFuncSetAttribute((const void*)attn_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024));
        MI355_HIP(hipFuncSetAttribute((const void*)attn_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024));
        attr_done = true;
    }
    if (esz == 4)
        hipLaunchKernelGGL(attn_kernel<float>, grid, dim3(threads), lds, s, p);
    else
        hipLaunchKernelGGL(attn_kernel<bf16_t>, grid, dim3(threads), lds, s, p);
    MI355_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi355_kv_roll(void* kcache, void* vcache, int cache_dtype, int B, int n_head, int S, int hs,
                             mi355_stream_t stream) {
    MI355_CHECK_ARG(kcache && vcache, MI355_E_ARG, "kv_roll: null cache");
    MI355_CHECK_ARG(B > 0 && n_head > 0 && S > 0 && hs > 0, MI355_E_SHAPE, "kv_roll: bad shape");
    MI355_CHECK_ARG(cache_dtype == MI355_F32 || cache_dtype == MI355_BF16, MI355_E_DTYPE, "kv_roll: bad dtype");
    const dim3 grid(B * n_head, 2);
    const int thr = hs >= 256 ? 256 : (hs >= 128 ? 128 : 64);
    if (cache_dtype == MI355_F32)
        hipLaunchKernelGGL(kv_roll_kernel<float>, grid, dim3(thr), 0, (hipStream_t)stream, (float*)kcache,
                           (float*)vcache, S, hs);
    else
        hipLaunchKernelGGL(kv_roll_kernel<bf16_t>, grid, dim3(thr), 0, (hipStream_t)stream, (bf16_t*)kcache,
                           (bf16_t*)vcache, S, hs);
    MI355_LAUNCH_CHECK();
    return 0;
}

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

// flash_prefill.hip: declared in gemm_fuse.h
#include "gemm_fuse.h"

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
    int n_split;        // workgroups per head (flash-decoding, a power of two); > 1 writes partial records to `part`
    int ns_shift;       // log2(n_split)
    int lpr_shift;      // log2(lanes per cache row) when a row is 16 B x a power of two <= 64 lanes, else -1
    float* part;
    unsigned long long* dbg;
    float scale;
    // LLaMA-Adapter prefix term folded into the decode kernel (aT > 0): see the tail of attn_kernel
    const float* ak;
    const float* av;
    const float* gate;
    int aT;
};

constexpr int kMaxPrefix = 64;  // prefix rows the folded adapter term handles (adapter_prompt_length is 10)

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

// Dynamic LDS: qs[hs] kcur[hs] vcur[hs] | per wave: m, l | scur | opart[nw][hs]
//
// Flash-decoding: grid (head, token, batch x n_split).  A head's cached rows are cut into n_split contiguous
// chunks so that several CUs pull one head's K/V (one CU sustains only ~50-100 GB/s; 32 heads on 32 CUs left
// 7/8 of the chip idle and cost ~12-16 us per layer).  Inside a workgroup every wave walks a strided subset
// of its chunk with the K row and the V row of a position loaded together (16 x 16-B loads in flight per lane),
// keeps a running (max, sum, weighted V) per row group, merges row groups with wavefront shuffles and waves
// through LDS.  The first batch of K/V loads is issued BEFORE the q / RoPE prologue: they do not depend on it,
// so the two memory round trips overlap.
//   n_split == 1: the workgroup normalises and writes y.
//   n_split  > 1: it writes its un-normalised partial (m, l, o[hs]) to `part`; the consumer combines them
//                 (mi355_attn_combine, or the prologue of the following c_proj linear — see gemv.hip).
// LEAN: the engine's decode shape (f32 qkv row, cache rows of 16 B x a power of two lanes) with the generic
// fall-backs compiled out — a launch starts with a cold instruction cache, so dead code between the live paths costs.
template <typename CT, bool LEAN>
__global__ __launch_bounds__(512) void attn_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int VEC = Vec16<CT>::kN;
    constexpr float kNegBig = -1.0e30f;
    constexpr int U = 8;
    const int h = blockIdx.x, t = blockIdx.y;
    const int ns = p.n_split;
    const int b = blockIdx.z >> p.ns_shift, sj = blockIdx.z & (ns - 1);  // host-computed shifts: runtime integer
                                                                          // divisions cost ~25 instructions each
    const int hs = p.hs, half = hs >> 1, C = p.n_head * hs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;

    float* qs = (float*)smem;
    float* kcur = qs + hs;
    float* vcur = kcur + hs;
    float* wm = vcur + hs;        // [nw] running max per wave
    float* wl = wm + nw;          // [nw] running sum per wave
    float* scur = wl + nw;        // [4] score of the current position (fused)
    float* opart = scur + 4;      // [nw][hs]
    float* pdot = opart + nw * hs;  // [kMaxPrefix] scores of the adapter prefix rows (aT > 0)
    const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
#define MI355_STAMP(i)                                                                  \
    do {                                                                                \
        if (p.dbg != nullptr && threadIdx.x == 0) p.dbg[wg * 8 + (i)] = wall_clock64(); \
    } while (0)
    MI355_STAMP(0);

    const int pos = p.pos ? p.pos[t] : t;
    const int slot = pos < p.S - 1 ? pos : p.S - 1;
    const int len = slot + 1;
    const int n_glob = p.fused ? slot : len;  // rows read from the cache in global memory
    const int chunk = (n_glob + ns - 1) >> p.ns_shift;
    const int s_begin = sj * chunk;
    const int s_end = (s_begin + chunk < n_glob) ? s_begin + chunk : n_glob;
    const bool own_cur = p.fused && sj == ns - 1;  // the split that folds in the new token

    const int64_t row = ((int64_t)b * p.T + t) * p.ld_qkv;
    const CT* kc = (const CT*)p.kcache + ((int64_t)b * p.n_head + h) * p.S * hs;
    const CT* vc = (const CT*)p.vcache + ((int64_t)b * p.n_head + h) * p.S * hs;

    const int row_bytes = hs * (int)sizeof(CT);
    const bool vec_ok = LEAN || p.lpr_shift >= 0;
    const int LPR = vec_ok ? (1 << p.lpr_shift) : 64;  // lanes per row
    const int rpw = vec_ok ? (64 >> p.lpr_shift) : 1;  // rows per wave instruction
    const int li = vec_ok ? (lane & (LPR - 1)) : lane, lr = vec_ok ? (lane >> p.lpr_shift) : 0;
    const int stride = nw * rpw;

    // ---- K/V rows stream through two register batches (A, B) of U row groups per wave: while one batch is
    // reduced the other one (and the refill of the first) is in flight, so a long context keeps ~2 x 16 KiB per
    // wave outstanding instead of paying the full HBM latency once per batch.  Loads go through buffer
    // descriptors bounded at this split's last row: rows past the end return zeros without a branch (a guarded
    // load is its own basic block and makes the compiler drain vmcnt at the join).  Both batches are requested
    // before q is prepared.
    u32x4 krA[U], vrA[U], krB[U], vrB[U];
    const __amdgpu_buffer_rsrc_t rk =
        __builtin_amdgcn_make_buffer_rsrc((void*)kc, 0, vec_ok ? s_end * row_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv =
        __builtin_amdgcn_make_buffer_rsrc((void*)vc, 0, vec_ok ? s_end * row_bytes : 0, 0x00020000);
    auto issue = [&](u32x4 (&kr)[U], u32x4 (&vr)[U], int base) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = base + wave * rpw + lr + u * stride;
            const unsigned off = s < s_end ? (unsigned)s * (unsigned)row_bytes + (unsigned)li * 16u : 0xFFFFFFF0u;
            kr[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, off, 0, 0));
            vr[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, off, 0, 0));
        }
    };
    // A split whose rows fit one batch (every decode step of a short context) takes a straight-line path with
    // half the load instructions; the choice is made once, around the whole staging + reduction code, so that the
    // compiler's vmcnt bookkeeping in the long loop never has to cover "batch B was not requested".
    const int bstep = stride * U;
    const bool one_batch = s_begin + bstep >= s_end;

    // ---- q (and, fused, the new k / v row) through RoPE into LDS
    const int rrow = p.rope_gathered ? t : pos;
    // engine path (f32 qkv): the few q / rope / new-k / new-v loads are requested FIRST, unconditionally (clamped
    // indices, no branch -> no wait at a join), then the K/V batches; VMEM returns in order, so q is staged while
    // the rows are still in flight instead of queueing behind them
    const bool fastq = LEAN || (p.qkv_dtype == MI355_F32 && half <= (int)blockDim.x && hs <= (int)blockDim.x);
    float2 qv = {0.f, 0.f}, kv = {0.f, 0.f}, cs = {0.f, 0.f};
    float vv = 0.f;
    auto q_load = [&]() {
        if (fastq) {
            const float* qrow = (const float*)p.qkv + row;
            const int pt = tid < half ? tid : 0, dt = tid < hs ? tid : 0;
            qv = *(const float2*)(qrow + h * hs + 2 * pt);
            cs = *(const float2*)(p.rope + ((int64_t)rrow * half + pt) * 2);
            kv = *(const float2*)(qrow + (own_cur ? C : 0) + h * hs + 2 * pt);
            vv = qrow[(own_cur ? 2 * C : 0) + h * hs + dt];
        }
    };
    auto stage_q = [&]() {
    MI355_STAMP(1);
    if (fastq) {
        if (tid < half) {
            qs[2 * tid] = qv.x * cs.x - qv.y * cs.y;
            qs[2 * tid + 1] = qv.y * cs.x + qv.x * cs.y;
            if (own_cur) {
                const CT ca = f32_to_ct<CT>(kv.x * cs.x - kv.y * cs.y), cb = f32_to_ct<CT>(kv.y * cs.x + kv.x * cs.y);
                CT* kw = (CT*)p.kcache + (((int64_t)b * p.n_head + h) * p.S + slot) * hs;
                kw[2 * tid] = ca;
                kw[2 * tid + 1] = cb;
                kcur[2 * tid] = ct_to_f32<CT>(ca);
                kcur[2 * tid + 1] = ct_to_f32<CT>(cb);
            }
        }
        if (own_cur && tid < hs) {
            const CT cv = f32_to_ct<CT>(vv);
            ((CT*)p.vcache)[(((int64_t)b * p.n_head + h) * p.S + slot) * hs + tid] = cv;
            vcur[tid] = ct_to_f32<CT>(cv);
        }
    } else {
        for (int pi = tid; pi < half; pi += blockDim.x) {
            const float a = ld_as_f32(p.qkv, row + h * hs + 2 * pi, p.qkv_dtype);
            const float bb = ld_as_f32(p.qkv, row + h * hs + 2 * pi + 1, p.qkv_dtype);
            float oa, ob;
            rope_pair(p.rope, rrow, half, pi, a, bb, oa, ob);
            qs[2 * pi] = oa;
            qs[2 * pi + 1] = ob;
            if (own_cur) {
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
        if (own_cur) {
            for (int d = tid; d < hs; d += blockDim.x) {
                const CT cv = f32_to_ct<CT>(ld_as_f32(p.qkv, row + 2 * C + h * hs + d, p.qkv_dtype));
                ((CT*)p.vcache)[(((int64_t)b * p.n_head + h) * p.S + slot) * hs + d] = cv;
                vcur[d] = ct_to_f32<CT>(cv);
            }
        }
    }
    __syncthreads();
    MI355_STAMP(2);
    };  // stage_q

    if (vec_ok) {
        float qf[VEC], m_run = kNegBig, l_run = 0.f, of[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) of[j] = 0.f;
        auto load_q = [&]() {
#pragma unroll
            for (int j = 0; j < VEC; ++j) qf[j] = qs[li * VEC + j];
        };
        // one batch: all U partial dot products first, then the U lane-group reductions (independent ->
        // interleaved), then the sequential online-softmax updates.  Steps whose rows are all past the end are
        // skipped with a wave-uniform test (a 150-row context over 4 splits uses 3 of the 8 steps); the
        // shuffles need every lane of a row group, so every test here is wave-uniform.
        auto process = [&](const u32x4 (&kr)[U], const u32x4 (&vr)[U], int base) {
            const int s0 = base + wave * rpw + lr;
            const int sw0 = base + wave * rpw;  // first row of this wave's step 0
            float dots[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int s = s0 + u * stride;
                float dot = 0.f;
                if (sw0 + u * stride < s_end) {
                    if (s < s_end) {
                        float kf[VEC];
                        unpack16<CT>(kr[u], kf);
#pragma unroll
                        for (int j = 0; j < VEC; ++j) dot += qf[j] * kf[j];
                    }
                    dot = group_sum(dot, LPR);
                }
                dots[u] = dot;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (sw0 + u * stride >= s_end) break;  // wave-uniform
                const int s = s0 + u * stride;
                const bool valid = s < s_end;
                float vf[VEC];
                if (valid) {
                    unpack16<CT>(vr[u], vf);
                } else {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) vf[j] = 0.f;
                }
                const float sc = valid ? dots[u] * p.scale : kNegBig;
                const float m_new = fmaxf(m_run, sc);
                const float corr = __expf(m_run - m_new);
                const float pr = valid ? __expf(sc - m_new) : 0.f;
                l_run = l_run * corr + pr;
#pragma unroll
                for (int j = 0; j < VEC; ++j) of[j] = of[j] * corr + pr * vf[j];
                m_run = m_new;
            }
        };
        if (one_batch) {
            q_load();
            issue(krA, vrA, s_begin);
            stage_q();
            load_q();
            process(krA, vrA, s_begin);
        } else {
            q_load();
            issue(krA, vrA, s_begin);
            issue(krB, vrB, s_begin + bstep);
            stage_q();
            load_q();
            for (int base = s_begin; base < s_end; base += 2 * bstep) {
                process(krA, vrA, base);
                issue(krA, vrA, base + 2 * bstep);
                if (base + bstep < s_end) process(krB, vrB, base + bstep);
                issue(krB, vrB, base + 3 * bstep);
            }
        }
        // merge the rpw row groups of the wave (lanes with equal li)
        for (int o = LPR; o < 64; o <<= 1) {
            const float m_o = __shfl_xor(m_run, o, 64), l_o = __shfl_xor(l_run, o, 64);
            const float m_new = fmaxf(m_run, m_o);
            const float ca = __expf(m_run - m_new), cb = __expf(m_o - m_new);
            l_run = l_run * ca + l_o * cb;
#pragma unroll
            for (int j = 0; j < VEC; ++j) of[j] = of[j] * ca + __shfl_xor(of[j], o, 64) * cb;
            m_run = m_new;
        }
        if (lr == 0) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) opart[wave * hs + li * VEC + j] = of[j];
            if (li == 0) {
                wm[wave] = m_run;
                wl[wave] = l_run;
            }
        }
    } else {
        q_load();
        stage_q();
        // odd head sizes (tiny test models): one row per wave step, lanes stride the head dimension
        float m_run = kNegBig, l_run = 0.f;
        float oacc[4] = {0.f, 0.f, 0.f, 0.f};  // d = lane + 64 j, hs <= 256
        for (int s = s_begin + wave; s < s_end; s += nw) {
            float dot = 0.f;
            for (int d = lane; d < hs; d += 64) dot += qs[d] * ct_to_f32<CT>(kc[(int64_t)s * hs + d]);
            dot = wave_sum(dot) * p.scale;
            const float m_new = fmaxf(m_run, dot);
            const float corr = expf(m_run - m_new), pr = expf(dot - m_new);
            l_run = l_run * corr + pr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = lane + 64 * j;
                if (d < hs) oacc[j] = oacc[j] * corr + pr * ct_to_f32<CT>(vc[(int64_t)s * hs + d]);
            }
            m_run = m_new;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = lane + 64 * j;
            if (d < hs) opart[wave * hs + d] = oacc[j];
        }
        if (lane == 0) {
            wm[wave] = m_run;
            wl[wave] = l_run;
        }
    }
    if (own_cur && wave == nw - 1) {
        float dot = 0.f;
        for (int d = lane; d < hs; d += 64) dot += qs[d] * kcur[d];
        dot = group_sum(dot, 64);
        if (lane == 0) scur[0] = dot * p.scale;
    }
    MI355_STAMP(3);
    __syncthreads();

    // ---- combine the waves (fixed order) and, fused, the current position from its LDS copy
    // nw <= 8: fixed-trip loops, so the LDS reads of a loop overlap and each wave's weight e^{m_w - M} is
    // computed once (v_exp_f32) instead of once per output element
    float m_all = kNegBig;
    float wmv[8], wgt[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        wmv[w] = w < nw ? wm[w] : kNegBig;
        m_all = fmaxf(m_all, wmv[w]);
    }
    float s_cur = kNegBig;
    if (own_cur) {
        s_cur = scur[0];
        m_all = fmaxf(m_all, s_cur);
    }
    float l_all = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        wgt[w] = w < nw ? __expf(wmv[w] - m_all) : 0.f;
        l_all += (w < nw ? wl[w] : 0.f) * wgt[w];
    }
    const float p_cur = own_cur ? __expf(s_cur - m_all) : 0.f;
    l_all += p_cur;
    // LLaMA-Adapter (adapter.py:134-151): y += P,  P = gate[h] * softmax(q . ak[h]^T * scale) av[h]  over the aT prefix rows
    // (no RoPE on them, no mask, a softmax of their own).  With the rows split over ns workgroups every split adds
    // l_split * P to its un-normalised record: sum_s w_s (o_s + l_s P) / sum_s w_s l_s = y + P, so the consumers of the
    // records (mi355_attn_combine, the c_proj prologue) need not know about the prefix.
    float pfx_scale = 0.f, pfx_max = 0.f;  // gate / sum of the prefix softmax weights, their maximum (aT == 0: no adapter)
    if (p.aT > 0) {
        for (int s2 = wave; s2 < p.aT; s2 += nw) {
            const float* kr = p.ak + ((int64_t)h * p.aT + s2) * hs;
            float dot = 0.f;
            for (int d = lane; d < hs; d += 64) dot += qs[d] * kr[d];
            dot = group_sum(dot, 64);
            if (lane == 0) pdot[s2] = dot * p.scale;
        }
        __syncthreads();
        pfx_max = kNegBig;
        for (int s2 = 0; s2 < p.aT; ++s2) pfx_max = fmaxf(pfx_max, pdot[s2]);
        float pl = 0.f;
        for (int s2 = 0; s2 < p.aT; ++s2) pl += __expf(pdot[s2] - pfx_max);
        pfx_scale = p.gate[h] / pl;
    }
    auto prefix = [&](int d) {  // P[d]
        float acc = 0.f;
        for (int s2 = 0; s2 < p.aT; ++s2) acc += __expf(pdot[s2] - pfx_max) * p.av[((int64_t)h * p.aT + s2) * hs + d];
        return acc * pfx_scale;
    };
    if (ns == 1) {
        const float inv = 1.0f / l_all;
        for (int d = tid; d < hs; d += blockDim.x) {
            float o = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w)
                if (w < nw) o += opart[w * hs + d] * wgt[w];
            if (own_cur) o += p_cur * vcur[d];
            st_from_f32(p.y, ((int64_t)b * p.T + t) * p.ldy + h * hs + d, p.y_dtype, o * inv + prefix(d));
        }
    } else {
        // partial record: [m, l, 0, 0, o[hs]] (un-normalised), one per (token row, head, split)
        float* rec = p.part + ((((int64_t)b * p.T + t) * p.n_head + h) * ns + sj) * (hs + 4);
        if (tid == 0) {
            rec[0] = m_all;
            rec[1] = l_all;
            rec[2] = 0.f;
            rec[3] = 0.f;
        }
        for (int d = tid; d < hs; d += blockDim.x) {
            float o = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w)
                if (w < nw) o += opart[w * hs + d] * wgt[w];
            if (own_cur) o += p_cur * vcur[d];
            rec[4 + d] = o + l_all * prefix(d);
        }
    }
    MI355_STAMP(4);
#undef MI355_STAMP
}

// y[row, h*hs + d] = sum_j e^{m_j - M} o_j[d] / sum_j e^{m_j - M} l_j  over the n_split partial records
__global__ void attn_combine_kernel(const float* part, int n_split, int n_head, int hs, void* y, int y_dtype,
                                    int64_t ldy) {
    const int r = blockIdx.y, h = blockIdx.x;
    const float* rec = part + (((int64_t)r * n_head + h) * n_split) * (hs + 4);
    float M = -1.0e30f;
    for (int j = 0; j < n_split; ++j) M = fmaxf(M, rec[j * (hs + 4)]);
    float L = 0.f;
    for (int j = 0; j < n_split; ++j) L += rec[j * (hs + 4) + 1] * expf(rec[j * (hs + 4)] - M);
    const float inv = 1.0f / L;
    for (int d = threadIdx.x; d < hs; d += blockDim.x) {
        float o = 0.f;
        for (int j = 0; j < n_split; ++j) o += rec[j * (hs + 4) + 4 + d] * expf(rec[j * (hs + 4)] - M);
        st_from_f32(y, (int64_t)r * ldy + h * hs + d, y_dtype, o * inv);
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
    // LLaMA-Adapter: the gated prefix term rides in the decode kernel; behind the flash kernel it is its own launch
    const bool adapter = a->adapter_len > 0;
    MI355_CHECK_ARG(!adapter || (a->adapter_k && a->adapter_v && a->adapter_gate && a->adapter_len <= kMaxPrefix), MI355_E_ARG,
                    "attention: adapter prefix of %d rows (at most %d) needs keys, values and gates", a->adapter_len, kMaxPrefix);
    p.ak = a->adapter_k;
    p.av = a->adapter_v;
    p.gate = a->adapter_gate;
    p.aT = adapter ? a->adapter_len : 0;

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
    // many query tokens against a bf16 cache at the LLaMA head size: flash-style MFMA kernel (flash_prefill.hip)
    // (with a cache: rows [0, pos[t]]; without: the T tokens themselves, K / V in kv_tmp, token t at position t)
    if (a->T >= 32 && a->B == 1 && esz == 2 && a->hs == 128 && (a->qkv_dtype == MI355_F32 || a->qkv_dtype == MI355_BF16) &&
        a->y_dtype == MI355_BF16 && a->n_split <= 1 && a->ld_qkv % 8 == 0 && a->ldy % 4 == 0 &&
        (int64_t)p.S * 256 < 0x7fffffffLL) {
        if (int rc = mi355_flash_prefill(a->qkv, a->qkv_dtype, a->ld_qkv, a->rope, a->rope_gathered, p.pos, p.kcache, p.vcache,
                                         a->T, a->n_head, p.S, a->y, a->ldy, p.scale, nullptr, s))
            return rc;
        if (!adapter) return 0;
        mi355_adapter_args b;
        memset(&b, 0, sizeof(b));
        b.qkv = a->qkv;
        b.qkv_dtype = a->qkv_dtype;
        b.B = a->B;
        b.ld_qkv = a->ld_qkv;
        b.rope = a->rope;
        b.pos = p.pos;
        b.rope_gathered = a->rope_gathered;
        b.T = a->T;
        b.n_head = a->n_head;
        b.hs = a->hs;
        b.aT = a->adapter_len;
        b.y_dtype = a->y_dtype;
        b.ak = a->adapter_k;
        b.av = a->adapter_v;
        b.gate = a->adapter_gate;
        b.y = a->y;
        b.ldy = a->ldy;
        return mi355_adapter_prefix(&b, stream);
    }
    int ns = a->n_split > 1 ? a->n_split : 1;
    MI355_CHECK_ARG(ns == 1 || a->partials != nullptr, MI355_E_ARG, "attention: n_split > 1 needs a partials buffer");
    MI355_CHECK_ARG(ns <= 64 && (int64_t)a->B * ns <= 65535, MI355_E_SHAPE, "attention: n_split too large");
    MI355_CHECK_ARG((ns & (ns - 1)) == 0, MI355_E_ARG, "attention: n_split must be a power of two (got %d)", ns);
    p.n_split = ns;
    p.ns_shift = __builtin_ctz((unsigned)ns);
    {
        const int row_bytes = a->hs * esz, n16 = row_bytes / 16;
        const bool vec_ok = (row_bytes % 16 == 0) && n16 >= 1 && n16 <= 64 && (n16 & (n16 - 1)) == 0;
        p.lpr_shift = vec_ok ? __builtin_ctz((unsigned)n16) : -1;
    }
    p.part = (float*)a->partials;
    p.dbg = (unsigned long long*)a->debug_stamps;
    const int threads = ns > 1 ? 256 : 512, nw = threads / 64;
    const size_t lds = (size_t)(3 * a->hs + 2 * nw + 4 + nw * a->hs + kMaxPrefix) * sizeof(float) + 16;
    MI355_CHECK_ARG(a->hs <= 256 || (a->hs * esz) % 16 == 0, MI355_E_SHAPE, "attention: head size %d unsupported", a->hs);
    MI355_CHECK_ARG(lds <= 160 * 1024, MI355_E_SHAPE, "attention: hs=%d needs %zu B of LDS", a->hs, lds);
    static bool attr_done = false;
    if (!attr_done) {
        MI355_HIP(hipFuncSetAttribute((const void*)attn_kernel<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024));
        MI355_HIP(hipFuncSetAttribute((const void*)attn_kernel<bf16_t, false>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MI355_HIP(hipFuncSetAttribute((const void*)attn_kernel<bf16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024));
        attr_done = true;
    }
    const dim3 agrid(a->n_head, a->T, a->B * ns);
    const bool lean = esz == 2 && p.lpr_shift >= 0 && a->qkv_dtype == MI355_F32 && a->hs <= threads;
    if (esz == 4)
        hipLaunchKernelGGL((attn_kernel<float, false>), agrid, dim3(threads), lds, s, p);
    else if (lean)
        hipLaunchKernelGGL((attn_kernel<bf16_t, true>), agrid, dim3(threads), lds, s, p);
    else
        hipLaunchKernelGGL((attn_kernel<bf16_t, false>), agrid, dim3(threads), lds, s, p);
    MI355_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi355_attn_combine(const float* partials, int n_split, int rows, int n_head, int hs, void* y,
                                  int y_dtype, int64_t ldy, mi355_stream_t stream) {
    MI355_CHECK_ARG(partials && y, MI355_E_ARG, "attn_combine: null pointer");
    MI355_CHECK_ARG(n_split >= 1 && rows >= 1 && rows <= 65535 && n_head >= 1 && hs >= 1, MI355_E_SHAPE,
                    "attn_combine: bad shape");
    hipLaunchKernelGGL(attn_combine_kernel, dim3(n_head, rows), dim3(hs >= 128 ? 128 : 64), 0, (hipStream_t)stream,
                       partials, n_split, n_head, hs, y, y_dtype, ldy);
    MI355_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ LLaMA-Adapter prefix term
// y[row, h, :] += gate[h] * softmax(rope(q[row, h]) . ak[h]^T / sqrt(hs)) av[h]      (/root/reference lit_llama/adapter.py:134-151:
// the adaption prompt's keys / values carry no RoPE and no mask, the softmax is its own, the gate is per head).
// One wave per (row, head): a lane holds the interleaved pairs lane, lane + 64 of the query (RoPE is lane-local), the
// aT (ten) prefix rows are walked with a running (max, sum, weighted values) — f32 throughout.
struct AdapterParams {
    const void* qkv;
    int qkv_dtype;
    int64_t ld_qkv;
    const float* rope;
    const int* pos;
    int rope_gathered, T, rows, n_head, hs, aT;
    const float* ak;    // [n_head, aT, hs]
    const float* av;
    const float* gate;  // [n_head]
    void* y;
    int y_dtype;
    int64_t ldy;
};

__global__ __launch_bounds__(256) void adapter_prefix_kernel(const AdapterParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = blockIdx.x, row = blockIdx.y * 4 + wave;
    if (row >= p.rows) return;
    const int hs = p.hs, half = hs >> 1;
    const int t = row % p.T;
    const int rrow = p.rope_gathered ? t : (p.pos ? p.pos[t] : t);
    float qa[2], qb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pi = lane + 64 * j;
        qa[j] = qb[j] = 0.f;
        if (pi < half) {
            const float a = ld_as_f32(p.qkv, (int64_t)row * p.ld_qkv + h * hs + 2 * pi, p.qkv_dtype);
            const float b = ld_as_f32(p.qkv, (int64_t)row * p.ld_qkv + h * hs + 2 * pi + 1, p.qkv_dtype);
            rope_pair(p.rope, rrow, half, pi, a, b, qa[j], qb[j]);
        }
    }
    const float scale = rsqrtf((float)hs);
    float m = -INFINITY, l = 0.f, oa[2] = {0.f, 0.f}, ob[2] = {0.f, 0.f};
    for (int s = 0; s < p.aT; ++s) {
        const float* kr = p.ak + ((int64_t)h * p.aT + s) * hs;
        const float* vr = p.av + ((int64_t)h * p.aT + s) * hs;
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pi = lane + 64 * j;
            if (pi < half) dot += qa[j] * kr[2 * pi] + qb[j] * kr[2 * pi + 1];
        }
        dot = wave_sum(dot) * scale;
        const float mn = fmaxf(m, dot);
        const float corr = __expf(m - mn), pw = __expf(dot - mn);
        l = l * corr + pw;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pi = lane + 64 * j;
            if (pi < half) {
                oa[j] = oa[j] * corr + pw * vr[2 * pi];
                ob[j] = ob[j] * corr + pw * vr[2 * pi + 1];
            }
        }
        m = mn;
    }
    const float gl = p.gate[h] / l;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pi = lane + 64 * j;
        if (pi < half) {
            const int64_t yi = (int64_t)row * p.ldy + h * hs + 2 * pi;
            st_from_f32(p.y, yi, p.y_dtype, ld_as_f32(p.y, yi, p.y_dtype) + gl * oa[j]);
            st_from_f32(p.y, yi + 1, p.y_dtype, ld_as_f32(p.y, yi + 1, p.y_dtype) + gl * ob[j]);
        }
    }
}

extern "C" int mi355_adapter_prefix(const mi355_adapter_args* a, mi355_stream_t stream) {
    MI355_CHECK_ARG(a != nullptr && a->qkv && a->rope && a->ak && a->av && a->gate && a->y, MI355_E_ARG,
                    "adapter_prefix: null argument");
    MI355_CHECK_ARG(a->B > 0 && a->T > 0 && a->n_head > 0 && a->hs > 0 && a->hs % 2 == 0 && a->hs <= 256 && a->aT > 0,
                    MI355_E_SHAPE, "adapter_prefix: bad shape (B=%d T=%d heads=%d hs=%d prefix rows=%d)", a->B, a->T, a->n_head,
                    a->hs, a->aT);
    auto three = [](int d) { return d == MI355_F32 || d == MI355_BF16 || d == MI355_F16; };
    MI355_CHECK_ARG(three(a->qkv_dtype) && three(a->y_dtype), MI355_E_DTYPE, "adapter_prefix: qkv / y dtype");
    MI355_CHECK_ARG(a->rope_gathered || a->pos != nullptr || a->T > 0, MI355_E_ARG, "adapter_prefix: no positions");
    const int64_t rows = (int64_t)a->B * a->T;
    MI355_CHECK_ARG((rows + 3) / 4 <= 65535, MI355_E_SHAPE, "adapter_prefix: too many rows for the grid");
    AdapterParams p;
    p.qkv = a->qkv;
    p.qkv_dtype = a->qkv_dtype;
    p.ld_qkv = a->ld_qkv;
    p.rope = a->rope;
    p.pos = a->pos;
    p.rope_gathered = a->rope_gathered;
    p.T = a->T;
    p.rows = (int)rows;
    p.n_head = a->n_head;
    p.hs = a->hs;
    p.aT = a->aT;
    p.ak = a->ak;
    p.av = a->av;
    p.gate = a->gate;
    p.y = a->y;
    p.y_dtype = a->y_dtype;
    p.ldy = a->ldy;
    hipLaunchKernelGGL(adapter_prefix_kernel, dim3(a->n_head, (unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
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

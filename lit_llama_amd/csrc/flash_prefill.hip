// Causal attention over many query tokens (prompt prefill, no-cache evaluation) for gfx950: flash-style, on MFMA.
//
// Replaces F.scaled_dot_product_attention with the boolean causal mask of /root/reference lit_llama/model.py:93-99,
// :230 for T >= 32 query tokens (evaluate/full.py:120-129 runs T = 2048): the one-workgroup-per-(head, query) kernel
// of attention.hip re-reads a head's K / V once per query.
//
// One workgroup = 64 queries of one head (4 waves x 16), walking the keys 32 at a time up to the causal limit:
//   * S^T = K Q^T (keys x queries) rather than Q K^T: the MFMA result then has a QUERY per lane column and 4 + 4 keys
//     per lane in registers, which is exactly the B-operand shape of the next product O^T = V^T P^T (32 keys x 16
//     queries) — up to a fixed permutation of the 32 keys, which a sum over keys does not care about as long as V^T
//     uses the same one.  So the probabilities never leave their registers (no LDS round trip, no transposition of
//     P), and the online-softmax rescale is a per-lane scalar.  Row maxima / sums run over the 8 registers and two
//     permlane swaps (lane ^ 16, lane ^ 32).
//   * K tiles go to LDS as they are (XOR-swizzled 16-B chunks); V tiles are TRANSPOSED on the way in (d-major, keys in
//     the permuted order, rows padded to 80 B: conflict-free 16-B fragment reads), both double buffered.
//   * q is RoPE'd in registers while it is loaded (f32 qkv rows, rope row = the token's position); the new K / V rows
//     were written to the cache by rope_kv_write_kernel before this launch.
#include "common.h"

namespace {

constexpr int kHs = 128;
constexpr int kBQ = 64;    // queries per workgroup
constexpr int kBK = 32;    // keys per step
constexpr int kVtRow = 80; // bytes per d-row of the transposed V tile (32 keys x 2 B, padded)
constexpr int kKTile = kBK * 256, kVTile = kHs * kVtRow;
constexpr int kLds = 2 * (kKTile + kVTile);

struct FlashParams {
    const void* qkv;     // [T, ld_qkv] f32 or bf16, q of head h at column h * 128
    const float* rope;   // [block_size, 64, 2]; row = the token's position, or its index when rope_gathered
    const int32_t* pos;  // [T], or NULL: token t sits at position t (no-cache forward)
    const bf16_t* kcache;
    const bf16_t* vcache;  // [n_head, S, 128]
    bf16_t* y;           // [T, ldy]
    int64_t ld_qkv, ldy;
    int T, n_head, S, qkv_dtype, rope_gathered;
    float scale;
};

__global__ __launch_bounds__(256) void flash_prefill_kernel(const FlashParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int h = blockIdx.y, qb = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int q_idx = qb * kBQ + wave * 16 + c;  // this lane's query (column of every MFMA result below)
    const bool q_ok = q_idx < p.T;
    const int q_row = q_ok ? q_idx : p.T - 1;
    const int q_abs = p.pos != nullptr ? p.pos[q_row] : q_row;
    // the workgroup's last query bounds the keys it walks
    const int q_lrow = (qb * kBQ + kBQ - 1 < p.T) ? qb * kBQ + kBQ - 1 : p.T - 1;
    const int q_last = p.pos != nullptr ? p.pos[q_lrow] : q_lrow;
    const int n_keys = q_last + 1 < p.S ? q_last + 1 : p.S;
    const int n_kb = (n_keys + kBK - 1) / kBK;

    // ---- Q^T as B operands: bq[dc] = q[32 dc + 8 g .. + 8), RoPE'd (model.py:306-323), bf16
    bf16x8 bq[4];
    {
        const int64_t qoff = (int64_t)q_row * p.ld_qkv + h * kHs;
        const float* rrow = p.rope + (int64_t)(p.rope_gathered ? q_row : q_abs) * kHs;  // 64 pairs x (cos, sin)
#pragma unroll
        for (int dc = 0; dc < 4; ++dc) {
            const int d0 = 32 * dc + 8 * g;
            f32x4 a, b;
            if (p.qkv_dtype == MI355_F32) {
                a = *(const f32x4*)((const float*)p.qkv + qoff + d0);
                b = *(const f32x4*)((const float*)p.qkv + qoff + d0 + 4);
            } else {
                const u32x4 raw = *(const u32x4*)((const bf16_t*)p.qkv + qoff + d0);
                a = f32x4{__uint_as_float(raw[0] << 16), __uint_as_float(raw[0] & 0xffff0000u),
                          __uint_as_float(raw[1] << 16), __uint_as_float(raw[1] & 0xffff0000u)};
                b = f32x4{__uint_as_float(raw[2] << 16), __uint_as_float(raw[2] & 0xffff0000u),
                          __uint_as_float(raw[3] << 16), __uint_as_float(raw[3] & 0xffff0000u)};
            }
            const f32x4 r0 = *(const f32x4*)(rrow + d0), r1 = *(const f32x4*)(rrow + d0 + 4);  // (c, s, c, s)
            u32x4 o;
            o[0] = (uint32_t)f32_to_bf16(a[0] * r0[0] - a[1] * r0[1]) | ((uint32_t)f32_to_bf16(a[1] * r0[0] + a[0] * r0[1]) << 16);
            o[1] = (uint32_t)f32_to_bf16(a[2] * r0[2] - a[3] * r0[3]) | ((uint32_t)f32_to_bf16(a[3] * r0[2] + a[2] * r0[3]) << 16);
            o[2] = (uint32_t)f32_to_bf16(b[0] * r1[0] - b[1] * r1[1]) | ((uint32_t)f32_to_bf16(b[1] * r1[0] + b[0] * r1[1]) << 16);
            o[3] = (uint32_t)f32_to_bf16(b[2] * r1[2] - b[3] * r1[3]) | ((uint32_t)f32_to_bf16(b[3] * r1[2] + b[2] * r1[3]) << 16);
            bq[dc] = __builtin_bit_cast(bf16x8, o);
        }
    }

    const bf16_t* kc = p.kcache + (int64_t)h * p.S * kHs;
    const bf16_t* vc = p.vcache + (int64_t)h * p.S * kHs;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)kc, 0, n_keys * 256, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)vc, 0, n_keys * 256, 0x00020000);
    // tile staging: 512 chunks of 16 B per tile, 2 per thread: chunk = (key, 16-B column)
    u32x4 ks[2], vs[2];
    auto tload = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = i * 256 + threadIdx.x;
            const int key = kb * kBK + (ch >> 4);
            const unsigned off = key < n_keys ? (unsigned)key * 256u + (unsigned)(ch & 15) * 16u : 0xFFFFFFF0u;
            ks[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, off, 0, 0));
            vs[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, off, 0, 0));
        }
    };
    auto tstore = [&](int buf) {
        char* kt = smem + buf * (kKTile + kVTile);
        char* vt = kt + kKTile;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = i * 256 + threadIdx.x;
            const int key = ch >> 4, col = ch & 15;
            *(u32x4*)(kt + key * 256 + ((col ^ (key & 7)) << 4)) = ks[i];
            // V transposed: d-major rows, the key at position 8 g' + j with key = (j < 4 ? 4 g' + j : 16 + 4 g' + j - 4)
            const int kt16 = key >> 4, w16 = key & 15;
            const int ppos = 8 * (w16 >> 2) + (kt16 ? 4 : 0) + (w16 & 3);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t v = vs[i][e];
                *(bf16_t*)(vt + (col * 8 + 2 * e) * kVtRow + ppos * 2) = (bf16_t)(v & 0xffffu);
                *(bf16_t*)(vt + (col * 8 + 2 * e + 1) * kVtRow + ppos * 2) = (bf16_t)(v >> 16);
            }
        }
    };

    f32x4 acc[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -1.0e30f, l_run = 0.f;

    tload(0);
    tstore(0);
    __syncthreads();
    for (int kb = 0; kb < n_kb; ++kb) {
        const int buf = kb & 1;
        const bool more = kb + 1 < n_kb;
        if (more) tload(kb + 1);
        const char* kt = smem + buf * (kKTile + kVTile);
        const char* vt = kt + kKTile;
        // ---- S^T[key][q] for the two 16-key tiles
        f32x4 st[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            st[t2] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int key = t2 * 16 + c;  // A-operand row of this lane
#pragma unroll
            for (int dc = 0; dc < 4; ++dc) {
                const bf16x8 ka = *(const bf16x8*)(kt + key * 256 + (((4 * dc + g) ^ (key & 7)) << 4));
                st[t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, bq[dc], st[t2], 0, 0, 0);
            }
        }
        // ---- causal mask, online softmax over this lane's query (8 keys here, the rest in lanes ^ 16, ^ 32)
        float sv[8];
        float m_blk = -1.0e30f;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key_abs = kb * kBK + t2 * 16 + 4 * g + r;
                const float s = key_abs <= q_abs ? st[t2][r] * p.scale : -1.0e30f;
                sv[t2 * 4 + r] = s;
                m_blk = fmaxf(m_blk, s);
            }
        m_blk = fmaxf(m_blk, lane_xor16(m_blk));
        m_blk = fmaxf(m_blk, lane_xor32(m_blk));
        const float m_new = fmaxf(m_run, m_blk);
        const float corr = __expf(m_run - m_new);
        float psum = 0.f;
        u32x4 pb;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float p0 = sv[2 * i] > -1.0e29f ? __expf(sv[2 * i] - m_new) : 0.f;
            const float p1 = sv[2 * i + 1] > -1.0e29f ? __expf(sv[2 * i + 1] - m_new) : 0.f;
            const bf16_t h0 = f32_to_bf16(p0), h1 = f32_to_bf16(p1);
            psum += bf16_to_f32(h0) + bf16_to_f32(h1);  // the sum of what the MFMA will multiply
            pb[i] = (uint32_t)h0 | ((uint32_t)h1 << 16);
        }
        l_run = l_run * corr + psum;
        m_run = m_new;
        // ---- O^T[d][q] = O^T * corr + V^T P^T
        const bf16x8 pfrag = __builtin_bit_cast(bf16x8, pb);
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            const bf16x8 va = *(const bf16x8*)(vt + (dt * 16 + c) * kVtRow + g * 16);
            f32x4 a = acc[dt];
            a[0] *= corr;
            a[1] *= corr;
            a[2] *= corr;
            a[3] *= corr;
            acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pfrag, a, 0, 0, 0);
        }
        if (more) tstore(buf ^ 1);
        __syncthreads();
    }
    float l = l_run + lane_xor16(l_run);
    l += lane_xor32(l);
    if (q_ok) {
        const float inv = 1.0f / l;
        bf16_t* yrow = p.y + (int64_t)q_idx * p.ldy + h * kHs;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            u32x2 o;
            o[0] = (uint32_t)f32_to_bf16(acc[dt][0] * inv) | ((uint32_t)f32_to_bf16(acc[dt][1] * inv) << 16);
            o[1] = (uint32_t)f32_to_bf16(acc[dt][2] * inv) | ((uint32_t)f32_to_bf16(acc[dt][3] * inv) << 16);
            *(u32x2*)(yrow + dt * 16 + 4 * g) = o;
        }
    }
}

}  // namespace

// y[t, h * 128 + d] for T query tokens against cache rows [0, pos[t]] (K / V rows of the T tokens already written).
int mi355_flash_prefill(const void* qkv, int qkv_dtype, int64_t ld_qkv, const float* rope, int rope_gathered,
                        const int32_t* pos, const void* kcache, const void* vcache, int T, int n_head, int S, void* y,
                        int64_t ldy, float scale, hipStream_t s) {
    static hipError_t attr_err =
        hipFuncSetAttribute((const void*)flash_prefill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (attr_err != hipSuccess) {
        mi355_set_error("hipFuncSetAttribute(flash_prefill) failed: %s", hipGetErrorString(attr_err));
        return (int)attr_err;
    }
    FlashParams p;
    p.qkv = qkv;
    p.rope = rope;
    p.pos = pos;
    p.kcache = (const bf16_t*)kcache;
    p.vcache = (const bf16_t*)vcache;
    p.y = (bf16_t*)y;
    p.ld_qkv = ld_qkv;
    p.ldy = ldy;
    p.T = T;
    p.n_head = n_head;
    p.S = S;
    p.qkv_dtype = qkv_dtype;
    p.rope_gathered = rope_gathered;
    p.scale = scale;
    hipLaunchKernelGGL(flash_prefill_kernel, dim3((T + kBQ - 1) / kBQ, n_head), dim3(256), kLds, s, p);
    MI355_LAUNCH_CHECK();
    return 0;
}

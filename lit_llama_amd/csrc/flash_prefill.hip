// Causal attention over many query tokens (prompt prefill, no-cache evaluation) for gfx950: flash-style, on MFMA.
//
// Replaces F.scaled_dot_product_attention with the boolean causal mask of /root/reference lit_llama/model.py:93-99,
// :230 for T >= 32 query tokens (evaluate/full.py:120-129 runs T = 2048): the one-workgroup-per-(head, query) kernel
// of attention.hip re-reads a head's K / V once per query.
//
// One workgroup = 128 queries of one head (8 waves x 16), walking the keys 32 at a time up to the causal limit:
//   * S^T = K Q^T (keys x queries) rather than Q K^T: the MFMA result then has a QUERY per lane column and 4 + 4 keys
//     per lane in registers, which is exactly the B-operand shape of the next product O^T = V^T P^T (32 keys x 16
//     queries) — up to a fixed permutation of the 32 keys, which a sum over keys does not care about as long as V^T
//     uses the same one.  So the probabilities never leave their registers (no LDS round trip, no transposition of
//     P), and the online-softmax rescale is a per-lane scalar.  Row maxima / sums run over the 8 registers and two
//     permlane swaps (lane ^ 16, lane ^ 32).
//   * K tiles go to LDS as they are (16-B chunks XOR-swizzled by key & 15: conflict-free fragment reads); V tiles are TRANSPOSED on the way in (d-major, keys in
//     the permuted order, rows padded to 96 B: conflict-free 16-B fragment reads), both double buffered.  The
//     transposition happens in registers: a thread loads the same 8 dimensions of 4 consecutive keys and writes 8-byte
//     groups of 4 keys (v_perm_b32), the row it writes rotated by its column so that the 16 columns of an instruction
//     hit 16 different bank groups (2-byte stores were 8-way bank conflicted and bounded the kernel: 339 us per layer
//     at T = 2048).
//   * q is RoPE'd in registers while it is loaded (f32 qkv rows, rope row = the token's position); the new K / V rows
//     were written to the cache by rope_kv_write_kernel before this launch.
#include "common.h"

namespace {

constexpr int kHs = 128;
constexpr int kBQ = 128;   // queries per workgroup (8 waves x 16)
constexpr int kThreadsF = 512;
constexpr int kBK = 32;    // keys per step
constexpr int kVtRow = 96; // bytes per d-row of the transposed V tile (32 keys x 2 B, padded: conflict-free b128 reads)
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
    int T, n_head, S, qkv_dtype, rope_gathered, q_blocks;
    float scale;
};

__global__ __launch_bounds__(kThreadsF) void flash_prefill_kernel(const FlashParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware order (workgroup b runs on XCD b % 8): all query blocks of a head on one XCD, whose L2 then holds that
    // head's K / V once; the longest (last) query blocks first
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int h = xcd + 8 * (idx / p.q_blocks), qb = p.q_blocks - 1 - idx % p.q_blocks;
    if (h >= p.n_head) return;
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
    // tile staging.  K: 512 chunks of 16 B, one per thread: chunk = (key, 16-B column).  V: threads 0..127 take
    // (group of 4 consecutive keys, 16-B column): 4 loads, transposed in registers.
    u32x4 ks, vs[4];
    const int vgrp = threadIdx.x >> 4, vcol = threadIdx.x & 15;  // key group 0..7 (threads < 128), column 0..15
    auto tload = [&](int kb) {
        {
            const int key = kb * kBK + (threadIdx.x >> 4);
            const unsigned off = key < n_keys ? (unsigned)key * 256u + (unsigned)(threadIdx.x & 15) * 16u : 0xFFFFFFF0u;
            ks = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, off, 0, 0));
        }
        if (threadIdx.x < 128) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kb * kBK + vgrp * 4 + r;
                const unsigned off = key < n_keys ? (unsigned)key * 256u + (unsigned)vcol * 16u : 0xFFFFFFF0u;
                vs[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, off, 0, 0));
            }
        }
    };
    auto tstore = [&](int buf) {
        char* kt = smem + buf * (kKTile + kVTile);
        char* vt = kt + kKTile;
        {
            const int key = threadIdx.x >> 4, col = threadIdx.x & 15;
            *(u32x4*)(kt + key * 256 + ((col ^ (key & 15)) << 4)) = ks;
        }
        if (threadIdx.x < 128) {
            // keys 4 vg' .. + 3 of 16-key tile kt16 sit at positions 8 g' + (kt16 ? 4 : 0) + 0..3 of a d-row: 8 bytes
            const int kt16 = vgrp >> 2, gq = vgrp & 3;
            const int pbyte = (8 * gq + (kt16 ? 4 : 0)) * 2;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = (i + vcol) & 7;  // rotate the row by the column: spreads an instruction over the banks
                const int dw = e >> 1;
                u32x2 o;
                if (e & 1) {
                    o[0] = __builtin_amdgcn_perm(vs[1][dw], vs[0][dw], 0x07060302u);
                    o[1] = __builtin_amdgcn_perm(vs[3][dw], vs[2][dw], 0x07060302u);
                } else {
                    o[0] = __builtin_amdgcn_perm(vs[1][dw], vs[0][dw], 0x05040100u);
                    o[1] = __builtin_amdgcn_perm(vs[3][dw], vs[2][dw], 0x05040100u);
                }
                *(u32x2*)(vt + (vcol * 8 + e) * kVtRow + pbyte) = o;
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
        tload(kb + 1);  // unconditional (past the last block: offsets beyond n_keys read zeros): no vmcnt drain at a join
        const char* kt = smem + buf * (kKTile + kVTile);
        const char* vt = kt + kKTile;
        // every fragment of the block is requested up front: K (8 x 16 B) for the scores, V^T (8 x 16 B) lands while the
        // softmax arithmetic runs
        bf16x8 ka[2][4], va[8];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const int key = t2 * 16 + c;  // A-operand row of this lane
#pragma unroll
            for (int dc = 0; dc < 4; ++dc) ka[t2][dc] = *(const bf16x8*)(kt + key * 256 + (((4 * dc + g) ^ (key & 15)) << 4));
        }
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) va[dt] = *(const bf16x8*)(vt + (dt * 16 + c) * kVtRow + g * 16);
        __builtin_amdgcn_sched_barrier(0);
        // ---- S^T[key][q] for the two 16-key tiles
        f32x4 st[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            st[t2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dc = 0; dc < 4; ++dc) st[t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[t2][dc], bq[dc], st[t2], 0, 0, 0);
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
            f32x4 a = acc[dt];
            a[0] *= corr;
            a[1] *= corr;
            a[2] *= corr;
            a[3] *= corr;
            acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[dt], pfrag, a, 0, 0, 0);
        }
        tstore(buf ^ 1);
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
    p.q_blocks = (T + kBQ - 1) / kBQ;
    hipLaunchKernelGGL(flash_prefill_kernel, dim3(p.q_blocks * ((n_head + 7) / 8 * 8)), dim3(kThreadsF), kLds, s, p);
    MI355_LAUNCH_CHECK();
    return 0;
}

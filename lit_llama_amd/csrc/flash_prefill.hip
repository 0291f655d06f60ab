// Causal attention over many query tokens (prompt prefill, no-cache evaluation) for gfx950: flash-style, on MFMA.
//
// Replaces F.scaled_dot_product_attention with the boolean causal mask of /root/reference lit_llama/model.py:93-99,
// :230 for T >= 32 query tokens (evaluate/full.py:120-129 runs T = 2048): the one-workgroup-per-(head, query) kernel
// of attention.hip re-reads a head's K / V once per query.
//
// One workgroup = 128 queries of one head (4 waves x 32), walking the keys 64 at a time up to the causal limit, on
// v_mfma_f32_32x32x16_bf16 (round 3; the round-2 kernel gave a wave 16 queries on 16x16x32 tiles and a barrier per 32
// keys: every wave re-read the whole K and V^T tiles from LDS for 16 queries, and the PMC pass showed the matrix pipes
// 7.5 % busy — profiles/r03_rocprofv3_pmc_mfma_util_prefill.txt):
//   * S^T = K Q^T (keys x queries) rather than Q K^T: the MFMA result then has a QUERY per lane column and 16 keys per
//     lane in registers, which is the B-operand shape of the next product O^T = V^T P^T (16 keys x 32 queries per
//     instruction) — up to a fixed permutation of the keys inside a group of 16, which a sum over keys does not care
//     about as long as V^T uses the same one (result register i of lane half h is key (i & 3) + 8 (i >> 2) + 4 h of the
//     32-key tile; as a B operand the same registers are k-slots 8 h + (i & 7): the runs of 4 keys 0-3 | 4-7 | 8-11 |
//     12-15 sit at slots 0-3 | 8-11 | 4-7 | 12-15).  So the probabilities never leave their registers, and the
//     online-softmax rescale is a per-lane scalar; row maxima / sums run over the 16 registers and one permlane32 swap.
//   * K tiles go to LDS as they are (16-B chunks XOR-swizzled by key & 15: conflict-free fragment reads); V tiles are
//     TRANSPOSED on the way in (d-major, keys in the permuted order, rows padded to 144 B: conflict-free 16-B fragment
//     reads), both double buffered.  The transposition happens in registers: a thread loads the same 8 dimensions of 4
//     consecutive keys and writes 8-byte groups of 4 keys (v_perm_b32), the row it writes rotated by its column.
//   * q is RoPE'd in registers while it is loaded (f32 qkv rows, rope row = the token's position) — or, inside the fused prompt chain,
//     arrives as the finished operand from the c_attn epilogue (MI355_Q_READY, gemm_fuse.h: rotated, scaled, bf16); the new K / V rows
//     were written to the cache by rope_kv_write_kernel / that epilogue before this launch.
//   * Round 6: the key step is a written-out software pipeline (see `step`): 32 MFMAs in turn — the NEXT step's 16 score MFMAs, then
//     this step's 16 value MFMAs — each followed by one piece of VALU work and a sched_barrier, LDS fragments requested three MFMAs
//     ahead.  Timing-only builds had shown the kernel slow per step (1.47 us solo / 2.35 us paired against 0.43 us of MFMA work), not
//     for lack of company on its CU; 68 -> 52-60 us per launch at T = 2048 (profiles/r06_prefill_flash_pipeline_ab.txt).
#include "common.h"
#include "gemm_fuse.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32
// two f32 -> packed bf16, round to nearest even (one v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    const bf16x2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

constexpr int kHs = 128;
constexpr int kBQ = 128;    // queries per workgroup (4 waves x 32)
constexpr int kThreadsF = 256;
constexpr int kBK = 64;     // keys per step (two 32-key score tiles)
constexpr int kVtRow = 144; // bytes per d-row of the transposed V tile (64 keys x 2 B, padded: conflict-free b128 reads)
constexpr int kKTile = kBK * 256, kVTile = kHs * kVtRow;
constexpr int kLds = 2 * (kKTile + kVTile);

struct FlashParams {
    const void* qkv;     // [T, ld_qkv] f32 or bf16, q of head h at column h * 128
    const float* rope;   // [block_size, 64, 2]; row = the token's position, or its index when rope_gathered
    const int32_t* pos;  // [T], or NULL: token t sits at position t (no-cache forward)
    const bf16_t* kcache;
    const bf16_t* vcache;  // [n_head, S, 128]
    bf16_t* y;           // [T, ldy]
    float* sx_part;      // [n_head][T] sums of this head's bf16 outputs per token (partial operand sums of attn.c_proj, gemm_fuse.h) or NULL
    int64_t ld_qkv, ldy;
    int T, n_head, S, qkv_dtype, rope_gathered, q_blocks;
    float scale;
};

__global__ __launch_bounds__(kThreadsF, 2) void flash_prefill_kernel(const FlashParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware order (workgroup b runs on XCD b % 8): all query blocks of a head on one XCD, whose L2 then holds that
    // head's K / V once; the longest (last) query blocks first
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int h = xcd + 8 * (idx / p.q_blocks);
    if (h >= p.n_head) return;
    // Causal work grows with the query block (block qb walks 2 (qb + 1) key steps), and at T = 2048 every workgroup of
    // the launch is resident at once, two per CU: what counts is WHICH two share a CU.  An XCD hands its workgroups to
    // its 32 CUs in order, so workgroups idx and idx + 32 meet: every second group of 32 walks the query blocks upwards
    // instead of downwards, and a long block sits next to a short one (34 key steps per CU instead of 4 .. 64).
    const int jq = idx % p.q_blocks;
    const bool up = (32 % p.q_blocks == 0) && ((idx >> 5) & 1);
    const int qb = up ? jq : p.q_blocks - 1 - jq;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hh = lane >> 5, c = lane & 31;      // lane half (k-slots 8 hh .. + 7 of an operand), column / row 0 .. 31
    const int q_idx = qb * kBQ + wave * 32 + c;  // this lane's query (column of every MFMA result below)
    const bool q_ok = q_idx < p.T;
    const int q_row = q_ok ? q_idx : p.T - 1;
    const int q_abs = p.pos != nullptr ? p.pos[q_row] : q_row;
    // the workgroup's last query bounds the keys it walks
    const int q_lrow = (qb * kBQ + kBQ - 1 < p.T) ? qb * kBQ + kBQ - 1 : p.T - 1;
    const int q_last = p.pos != nullptr ? p.pos[q_lrow] : q_lrow;
    const int n_keys = q_last + 1 < p.S ? q_last + 1 : p.S;
    const int n_kb = (n_keys + kBK - 1) / kBK;
    // the smallest position among the wave's queries: key steps entirely at or below it need no mask
    int q_min = q_abs;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q_min = min(q_min, __shfl_xor(q_min, o, 64));
    q_min = __builtin_amdgcn_readfirstlane(q_min);

    // ---- Q^T as B operands: bq[ks] = q[16 ks + 8 hh .. + 8), RoPE'd (model.py:306-323), bf16
    bf16x8 bq[8];
    {
        const int64_t qoff = (int64_t)q_row * p.ld_qkv + h * kHs;
        const float* rrow = p.rope + (int64_t)(p.rope_gathered ? q_row : q_abs) * kHs;  // 64 pairs x (cos, sin)
        const float qs = p.scale * 1.44269504088896340736f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int d0 = 16 * ks + 8 * hh;
            if (p.qkv_dtype == MI355_Q_READY) {  // the c_attn epilogue did the rotation, the scale and the rounding (gemm_fuse.h: q_scale)
                bq[ks] = *(const bf16x8*)((const bf16_t*)p.qkv + qoff + d0);
                continue;
            }
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
            // the softmax scale and log2(e) ride on q (f32, before the one rounding to bf16): the scores leave the MFMA in
            // the exp2 domain and the 16 scores per lane and tile need no multiply
            u32x4 o;  // (v_cvt_pk_bf16_f32: the same round-to-nearest-even as f32_to_bf16, one instruction per pair)
            o[0] = pack_bf16x2(qs * (a[0] * r0[0] - a[1] * r0[1]), qs * (a[1] * r0[0] + a[0] * r0[1]));
            o[1] = pack_bf16x2(qs * (a[2] * r0[2] - a[3] * r0[3]), qs * (a[3] * r0[2] + a[2] * r0[3]));
            o[2] = pack_bf16x2(qs * (b[0] * r1[0] - b[1] * r1[1]), qs * (b[1] * r1[0] + b[0] * r1[1]));
            o[3] = pack_bf16x2(qs * (b[2] * r1[2] - b[3] * r1[3]), qs * (b[3] * r1[2] + b[2] * r1[3]));
            bq[ks] = __builtin_bit_cast(bf16x8, o);
        }
    }

    const bf16_t* kc = p.kcache + (int64_t)h * p.S * kHs;
    const bf16_t* vc = p.vcache + (int64_t)h * p.S * kHs;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)kc, 0, n_keys * 256, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)vc, 0, n_keys * 256, 0x00020000);
    // tile staging.  K: 1024 chunks of 16 B, four per thread: chunk = (key, 16-B column).  V: one (run of 4 consecutive
    // keys, 16-B column) per thread: 4 loads, transposed in registers.
    u32x4 vs[4];
    // a wave covers all 16 runs of the 64-key tile for 4 of the 16 columns: one store instruction of the transposition
    // below then writes 16 different 8-byte slots of the SAME 4 d-rows (2-way bank conflicts at most) with compile-time
    // register indices (the round-2 mapping, 16 columns per instruction, needed a per-lane rotation of the rows: dynamic
    // register selects, ~100 VALU instructions per step)
    const int vrun = threadIdx.x & 15, vcol = (threadIdx.x >> 6) * 4 + ((threadIdx.x >> 4) & 3);
    // Round 6: K runs ONE key step ahead of V.  The scores of step kb + 1 are issued in front of the softmax of step kb, so that a
    // wave's own MFMAs run under its VALU work (round 3-5: scores -> softmax -> weighted values strictly in turn, the matrix pipe 20 %
    // busy): K tile j lives in K buffer j & 1, V tile j in V buffer j & 1, and step kb stages K(kb + 2) and V(kb + 1).
    // K tiles go global -> LDS by LDS-DMA (16 B per lane, one instruction fills 1 KiB = four key rows; the XOR swizzle is applied on the
    // SOURCE side: LDS slot s of key k takes column s ^ (k & 15)): no staging registers (the second set of scores needs them)
    // (keys past n_keys: the VECTOR offset leaves the descriptor's n_keys x 256 bytes — the loads return zeros, the DMA writes zeros — so a
    // step's eight requests cost one v_add each: chunk i of the K tile is 16 keys = 4096 B behind chunk 0 with the same swizzle term)
    const unsigned koff0 = (unsigned)(threadIdx.x >> 4) * 256u + (unsigned)(((threadIdx.x & 15) ^ ((threadIdx.x >> 4) & 15)) * 16);
    const unsigned voff0 = (unsigned)(vrun * 4) * 256u + (unsigned)vcol * 16u;
    auto dmaK = [&](int kb) {
        char* kt = smem + (kb & 1) * kKTile;
#pragma unroll
        for (int i = 0; i < 4; ++i)  // LDS chunk i * 256 + thread: consecutive lanes, consecutive 16 B
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (__attribute__((address_space(3))) void*)(kt + (i * kThreadsF + wave * 64) * 16), 16,
                                                     koff0 + (unsigned)(kb * kBK * 256 + i * 4096), 0, 0, 0);
    };
    auto tloadV = [&](int kb) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            vs[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, voff0 + (unsigned)(kb * kBK * 256 + r * 256), 0, 0));
    };
    auto tstoreV = [&](int buf) {
        char* vt = smem + 2 * kKTile + buf * kVTile;
        // run r of 16-key group G sits at k-slots 4 perm[r] .. + 3 of the group (perm = 0, 2, 1, 3): 8 bytes
        const int G = vrun >> 2, r = vrun & 3;
        const int pbyte = (G * 16 + 4 * ((r & 1) * 2 + (r >> 1))) * 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
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
    };
    // S^T[key][q] of the two 32-key tiles of K buffer `buf`: 2 x 8 MFMAs over the 128 dimensions, two independent accumulators
    auto scores = [&](int buf, f32x16 (&st)[2]) {
        const char* kt = smem + buf * kKTile;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int i = 0; i < 16; ++i) st[t2][i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const int key = t2 * 32 + c;  // A-operand row of this lane
                const bf16x8 ka = *(const bf16x8*)(kt + key * 256 + (((2 * ks + hh) ^ (key & 15)) << 4));
                st[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, bq[ks], st[t2], 0, 0, 0);
            }
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[dt][i] = 0.f;
    float m_run = -1.0e30f, l_run = 0.f;

    // (v_max3_f32: 16 instructions for the 32 scores of a lane; fmaxf() would canonicalise every MFMA result first)
    auto smax = [&](const f32x16 (&st)[2]) {
        float m = max3f(st[0][0], st[0][1], st[0][2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) m = max3f(m, st[0][i], st[0][i + 1]);
        m = max3f(m, st[0][15], st[1][0]);
#pragma unroll
        for (int i = 1; i < 15; i += 2) m = max3f(m, st[1][i], st[1][i + 1]);
        const float mh = max3f(m, st[1][15], st[1][15]);
        return max3f(mh, mh, lane_xor32(mh));
    };
    float m_blk;  // the maximum of the step's scores: taken under the value MFMAs of the step before (unmasked; a diagonal step takes it again)
    f32x16 sa[2], sb[2];  // the scores of this step and of the next one, swapping roles (the loop is unrolled by two: no copies)
    dmaK(0);
    dmaK(1);
    tloadV(0);
    __syncthreads();  // (waits vmcnt(0): both K tiles have landed)
    scores(0, sa);
    tstoreV(0);
    m_blk = smax(sa);
    __syncthreads();
    auto step = [&](const int kb, f32x16 (&st)[2], f32x16 (&sn)[2]) {
        // unconditional (past the last block: offsets beyond n_keys read zeros): no vmcnt drain at a join
        // (V first: the transposition then waits for its four loads with the four K requests still in flight — vmcnt retires in order)
        tloadV(kb + 1);
        dmaK(kb + 2);  // into the buffer K(kb) left during step kb - 1; landed at this step's barrier
        const char* vt = smem + 2 * kKTile + (kb & 1) * kVTile;
        // ---- causal mask (only where the step reaches the wave's diagonal: wave-uniform), online softmax over this
        // lane's query in the exp2 domain (32 keys here, the other 32 in lane ^ 32)
        if (kb * kBK + kBK - 1 > q_min) {
            asm volatile("" ::: "memory");  // (keeps the mask a branch: if-converted it costs ~110 VALU instructions in EVERY step)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key_abs = kb * kBK + t2 * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    st[t2][i] = key_abs <= q_abs ? st[t2][i] : -1.0e30f;
                }
            m_blk = smax(st);
        }
        // Lazy rescale: the running reference maximum m_run of a query moves only when the step's maximum exceeds it by
        // more than 2^8 (exp2 domain) — probabilities then stay below 256, which bf16 and the f32 sums hold easily — and
        // the 64 accumulator registers are rescaled only in the steps where some query of the wave moves: the first one
        // and a handful after it.
        const bool move = m_blk > m_run + 8.0f;
        if (__any(move)) {
            const float m_new = move ? m_blk : m_run;
            const float corr = exp2_fast(m_run - m_new);  // (first step: exp2(-1e30 - m) = 0 on all-zero accumulators)
            l_run *= corr;
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[dt][i] *= corr;
        }
        // ---- the block's software pipeline, written out and pinned with sched_barrier (hipcc's own order is fragile: with the loop
        // unrolled by two it issued the 16 score MFMAs, then all 32 exponentials, then the 16 value MFMAs, every MFMA behind the wait for
        // its own fragment read).  32 MFMAs in turn: the NEXT step's scores (16: K buffer (kb + 1) & 1), then O^T += V^T P^T (16).  Every
        // MFMA's LDS fragment is requested kAhead MFMAs earlier (ring of kAhead + 1 registers); under MFMA i runs one piece of VALU work:
        // a pair of probabilities (subtract, v_exp, packed bf16, sum) under score MFMAs 0..11 and value MFMAs 0..3 — value group sg needs
        // pairs 4 sg .. 4 sg + 3 only — the next step's row maximum under value MFMA 4, and the transposition of V(kb + 1) under value MFMAs 8..15 (its loads were
        // requested at the step's start).
        constexpr int kAhead = 3;
        const char* kt = smem + ((kb + 1) & 1) * kKTile;
        auto frag = [&](int i) {  // operand A of MFMA i
            if (i < 16) {
                const int ks = i >> 1, key = (i & 1) * 32 + c;
                return *(const bf16x8*)(kt + key * 256 + (((2 * ks + hh) ^ (key & 15)) << 4));
            }
            const int sg = (i - 16) >> 2, dt = (i - 16) & 3;
            return *(const bf16x8*)(vt + (dt * 32 + c) * kVtRow + sg * 32 + hh * 16);
        };
        bf16x8 fr[kAhead + 1];
#pragma unroll
        for (int i = 0; i < kAhead; ++i) fr[i] = frag(i);
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int i = 0; i < 16; ++i) sn[t2][i] = 0.f;
        float psum = 0.f, m_next = 0.f;
        u32x4 pb[2][2];
        auto pair = [&](int pr) {
            // masked scores are -1e30: exp2 underflows to exactly 0.  The denominator sums the f32 probabilities, the
            // numerator their bf16 roundings (v_cvt_pk_bf16_f32): zero-mean relative differences of 2^-9 per key
            const int t2 = pr >> 3, i = pr & 7;
            const float p0 = exp2_fast(st[t2][2 * i] - m_run), p1 = exp2_fast(st[t2][2 * i + 1] - m_run);
            psum += p0 + p1;
            uint32_t w = pack_bf16x2(p0, p1);
            asm volatile("" : "+v"(psum), "+v"(w));  // (inside its piece: hipcc otherwise keeps all 32 probabilities live to sum them at the end)
            pb[t2][i >> 2][i & 3] = w;
        };
        char* vtn = smem + 2 * kKTile + ((kb + 1) & 1) * kVTile;  // V(kb + 1): V(kb - 1) was read during step kb - 1
        const int vG = vrun >> 2, vr = vrun & 3;
        const int pbyte = (vG * 16 + 4 * ((vr & 1) * 2 + (vr >> 1))) * 2;
        auto vpiece = [&](int e) {  // one 8-byte group of the transposition (tstoreV)
            const int dw = e >> 1;
            u32x2 o;
            if (e & 1) {
                o[0] = __builtin_amdgcn_perm(vs[1][dw], vs[0][dw], 0x07060302u);
                o[1] = __builtin_amdgcn_perm(vs[3][dw], vs[2][dw], 0x07060302u);
            } else {
                o[0] = __builtin_amdgcn_perm(vs[1][dw], vs[0][dw], 0x05040100u);
                o[1] = __builtin_amdgcn_perm(vs[3][dw], vs[2][dw], 0x05040100u);
            }
            *(u32x2*)(vtn + (vcol * 8 + e) * kVtRow + pbyte) = o;
        };
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (i + kAhead < 32) fr[(i + kAhead) % (kAhead + 1)] = frag(i + kAhead);
            if (i < 16) {
                sn[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % (kAhead + 1)], bq[i >> 1], sn[i & 1], 0, 0, 0);
                if (i < 12) pair(i);
            } else {
                const int sg = (i - 16) >> 2, dt = (i - 16) & 3;
                const bf16x8 pfrag = __builtin_bit_cast(bf16x8, pb[sg >> 1][sg & 1]);
                acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % (kAhead + 1)], pfrag, acc[dt], 0, 0, 0);
                if (i < 20) pair(i - 4);
                else if (i == 20) m_next = smax(sn);
                else if (i >= 24) vpiece(i - 24);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        l_run += psum;
        m_blk = m_next;
        __syncthreads();
    };
    for (int kb = 0; kb < n_kb; kb += 2) {
        step(kb, sa, sb);
        if (kb + 1 < n_kb) step(kb + 1, sb, sa);
    }
    float l = l_run + lane_xor32(l_run);
    float osum = 0.f;  // this lane's 64 of the query's 128 outputs, as the bf16 values the consumer multiplies
    if (q_ok) {
        const float inv = 1.0f / l;
        bf16_t* yrow = p.y + (int64_t)q_idx * p.ldy + h * kHs;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                u32x2 o;
                o[0] = pack_bf16x2(acc[dt][4 * gq] * inv, acc[dt][4 * gq + 1] * inv);
                o[1] = pack_bf16x2(acc[dt][4 * gq + 2] * inv, acc[dt][4 * gq + 3] * inv);
                *(u32x2*)(yrow + dt * 32 + 8 * gq + 4 * hh) = o;
                osum += (__uint_as_float(o[0] << 16) + __uint_as_float(o[0] & 0xffff0000u)) +
                        (__uint_as_float(o[1] << 16) + __uint_as_float(o[1] & 0xffff0000u));
            }
    }
    if (p.sx_part != nullptr) {  // (wave-uniform; the other half of the dimensions sits in lane ^ 32)
        osum += lane_xor32(osum);
        if (q_ok && hh == 0) p.sx_part[(int64_t)h * p.T + q_idx] = osum;
    }
}

}  // namespace

// y[t, h * 128 + d] for T query tokens against cache rows [0, pos[t]] (K / V rows of the T tokens already written).
int mi355_flash_prefill(const void* qkv, int qkv_dtype, int64_t ld_qkv, const float* rope, int rope_gathered,
                        const int32_t* pos, const void* kcache, const void* vcache, int T, int n_head, int S, void* y,
                        int64_t ldy, float scale, float* sx_part, hipStream_t s) {
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
    p.sx_part = sx_part;
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

// The whole T = 1 decode step of a 7B-class gptq.int4 LLaMA as ONE persistent launch on gfx950 (round-3 version).
//
// Replaces, per generated token, the 161 operator calls of /root/reference lit_llama/model.py:76-122 (Block.forward
// :165-168, CausalSelfAttention.forward :194-237, MLP.forward :251-254, RMSNorm :274-277, apply_rope :306-323) and the
// greedy sampling of generate.py:68-85 — and this repository's own 162-launch step (engine.hip).
//
// Structure:
//  * 256 workgroups, one per CU, all resident for the whole step: 8 STREAMER waves (weights, MFMA) and 2 GATHERER
//    waves (every other global access: hand-offs, scales, cache rows, logits).
//  * WEIGHTS.  Every streamer wave owns a ring of 16 1-KiB pieces in LDS (8 x 16 KiB of the CU's 160 KiB) that it fills
//    itself by LDS-DMA (`buffer_load_dwordx4 ... lds`, non-temporal: the data never passes through registers) and
//    drains with one ds_read_b128 per lane and piece.  The pieces of a wave form ONE flat sequence over the phases
//    and layers of the step (c_attn | attn.c_proj | c_fc1/c_fc2 | mlp.c_proj | next layer ... | lm_head), so the ring
//    runs AHEAD ACROSS phase boundaries: while a phase computes, publishes, and while the next hand-off is in flight,
//    the weights of the following phases keep landing.  The round-2 version (fused_step_ring.hip) held the ring in
//    registers and could request a phase's weights only after the previous phase's publish: HBM idled during every
//    compute / publish / attention span (10 of a layer's 27 us) and the hand-offs then waited for the ring turn to land.
//  * THIN WINDOW.  A CU's memory pipeline serves its waves in order, so whatever is in flight stands in front of the
//    gatherers' publish stores and sweeps.  A wave keeps at most kWRun (while it computes) / kWPoll (while it waits for
//    a hand-off) pieces in flight; how many have landed is read without blocking from the wave's own vmcnt
//    (s_getreg_b32 HW_REG_IB_STS) while it waits, and known exactly from its `s_waitcnt vmcnt(n)` while it computes.
//    After a phase the streamers resume issuing only once gatherer 0 has issued the publish stores ("publish, then
//    refill", measured in round 2: 6.2 -> 4.2 us per 96-KiB phase).
//    Measured first as a protocol (scripts/micro/ldsdma.hip, profiles/r03_ldsdma_microbench.txt): LDS-DMA lands lane i's
//    16 B at M0 + 16 i up to the last KiB of the 160-KiB LDS, IB_STS carries the live vmcnt, 256 x 8 such rings stream
//    6.15 TB/s at 4 pieces per wave in flight, 5.1 at 2 (12-piece register rings: 6.85).
//  * SYNCHRONISATION inside the workgroup: the streamers learn that the activation vector of a phase is staged from
//    two LDS words (one per gatherer wave) they poll while they top up their rings — not from a barrier, which would
//    park them; s_barrier remains where streamers hand partial tiles to gatherer 0 (everybody is there at once).
//  * Activations move between the phases of a layer as 8-byte {tag, value} granules written with ONE sc1 store and
//    swept with sc1 loads until every tag equals the phase's epoch (unchanged from round 2): the data is the flag, no
//    fence, no grid barrier; tags are unique per (step, edge), nothing is zeroed between launches.
//  * c_attn, RoPE, the KV-cache row write and the attention of a head are local to the 8 workgroups of that head; the
//    residual stream never leaves the chip (workgroup b owns rows 16 b .. 16 b + 15 in registers).
// Every spin is bounded; a time-out raises the abort word, all other spins then give up at once and the host reports
// MI355_E_STATE (mi355_fused_step_status).  fp16 edges that had to be clipped are counted in state[2].
#include <math.h>
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include <hip/hip_ext.h>

#include "common.h"
#include "fused_step_common.h"

namespace {

constexpr int kG = 256;         // workgroups
constexpr int kSW = 8;          // streamer waves
constexpr int kGW = 2;          // gatherer waves
constexpr int kThreads = 64 * (kSW + kGW);
constexpr int kD = 16;          // ring pieces (1 KiB each) per streamer wave, in LDS
constexpr int kQ = kD / 4;      // = 4 quads of 4 consecutive pieces: the unit of requesting and of the in-flight window
#ifndef MI355_FUSED_WRUN
#define MI355_FUSED_WRUN 2      // quads (4 KiB) per wave in flight while the wave computes
#endif
#ifndef MI355_FUSED_WPOLL
#define MI355_FUSED_WPOLL 1     // ... while it waits for a hand-off (the gatherers' sweep is in the same memory pipeline)
#endif
#ifndef MI355_FUSED_PUBGATE
#define MI355_FUSED_PUBGATE 0   // after a phase, request again only once gatherer 0 has issued the publish stores
#endif
#ifndef MI355_FUSED_AHEAD
#define MI355_FUSED_AHEAD 1     // while a phase computes, request beyond its own quads
#endif
#ifndef MI355_FUSED_GPRIO
#define MI355_FUSED_GPRIO 1
#endif
#ifndef MI355_FUSED_POLL_SLEEP
#define MI355_FUSED_POLL_SLEEP 2
#endif
#ifndef MI355_FUSED_ATTN_ISSUE
#define MI355_FUSED_ATTN_ISSUE 1
#endif
#ifndef MI355_FUSED_HSWEEP
#define MI355_FUSED_HSWEEP 2
#endif
constexpr int kWRun = MI355_FUSED_WRUN, kWPoll = MI355_FUSED_WPOLL;
constexpr int kC = 4096;        // n_embd
constexpr int kHeads = 32;
constexpr int kHs = 128;
constexpr int kGs = kG / kHeads;  // workgroups per head
constexpr int kUnitsC = kC / 128;
constexpr int kMaxFcTiles = 3;    // c_fc1/c_fc2 pair tiles per workgroup (n_hidden <= 12288)
constexpr int kMaxHeadTiles = 8;  // lm_head tiles per workgroup (vocab <= 32768)
constexpr unsigned kSpinLimit = 400000u;

// LDS map (bytes).  The kernel has no static __shared__ data: the dynamic segment starts at LDS byte 0, which the
// LDS-DMA destination (an absolute byte address in M0) relies on (checked against the pointer at kernel entry).
constexpr int kOffMisc = 0;       // f32: [0] 1/rms, [4..7] operand sums, [16..23] / [24..31] per-wave softmax max / sum
constexpr int kOffFlag = 256;     // u32: [0] / [1] stages staged by gatherer 0 / 1, [2] publishes issued by gatherer 0
constexpr int kOffXs = 512;       // activation vector, fp16, <= 96 units of 128 values
constexpr int kOffPart = kOffXs + 96 * 256;             // [2][8 waves][4 row groups][16 rows] f32 partial outputs
constexpr int kOffQ = kOffPart + 2 * kSW * 4 * 64;      // q[128] knew[128] vnew[128] f32
constexpr int kOffOpart = kOffQ + 3 * 512;              // [8 waves][16] f32
constexpr int kOffRing = 31 * 1024;                     // [8 waves][kD][1 KiB]
constexpr int kMaxS = 32768;      // cache rows (the attention keeps no per-row state in LDS)
constexpr int kLdsBytes = kOffRing + kSW * kD * 1024;
static_assert(kOffOpart + 512 <= kOffRing, "LDS map: the small regions overlap the rings");
static_assert(kLdsBytes <= 160 * 1024, "LDS map exceeds the CU");

// ------------------------------------------------------------------------------------------------ granules
__device__ __forceinline__ void gr_store(u64* p, unsigned tag, unsigned val) {
    __hip_atomic_store(p, ((u64)tag << 32) | val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // one 8-B sc1 store
}
__device__ __forceinline__ bool aborted(const FusedParams& p) {
    return __hip_atomic_load(p.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}
__device__ __forceinline__ void raise_abort(const FusedParams& p, unsigned code) {
    __hip_atomic_store(p.state, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One wave sweeps 16-B loads (two granules each) number first + k * 64 + lane, k < NL, of the granule buffer behind
// `rs` (load i covers bytes base + 16 i ..) until every tag equals `epoch`; loads at or past `end` are skipped.
// Returns false after a time-out / abort (the values are then garbage, the caller keeps going so that the barrier
// counts of the workgroup stay balanced).
template <int NL>
__device__ __forceinline__ void sweep_issue(__amdgpu_buffer_rsrc_t rs, unsigned base, int first, int end, u32x4 (&v)[NL],
                                            int lane) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int i = first + k * 64 + lane;
        const unsigned off = i < end ? base + (unsigned)i * 16u : 0xFFFFFFF0u;
        v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));  // sc1
    }
}
// `preissued`: the caller has requested v already (sweep_issue) — several chunks of one edge in flight at once
template <int NL>
__device__ __forceinline__ bool sweep(const FusedParams& p, __amdgpu_buffer_rsrc_t rs, unsigned base, int first, int end,
                                      unsigned epoch, u32x4 (&v)[NL], unsigned code, int lane, bool preissued = false) {
    // (lane: the caller's per-layer opaque copy of the lane id — from threadIdx the offsets of every sweep site are
    // loop invariants, which hipcc computes once in the kernel prologue and then spills)
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
        if (!(preissued && spins == 0)) sweep_issue<NL>(rs, base, first, end, v, lane);
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = first + k * 64 + lane;
            ok &= i >= end || (v[k][1] == epoch && v[k][3] == epoch);
        }
        if (__all(ok)) return true;
        if (spins > kSpinLimit || aborted(p)) {
            if (lane == 0) raise_abort(p, code);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// ------------------------------------------------------------------------------------------------ LDS-DMA ring
// One 1-KiB piece: lane i's 16 B at rs[voff + soff] land at LDS byte lds_dst + 16 i (measured, scripts/micro/ldsdma.hip).
// hipcc neither counts this load in its vmcnt bookkeeping nor knows that it writes LDS: completion is waited for with
// wait_vmcnt_dyn below, and hipcc's own waits for ITS loads only ever come out stricter than needed (VMEM returns in
// order).  The s_nop pads the SALU-write -> VMEM-read hazard of soff / M0, which hipcc does not do inside an asm
// statement; the leading lgkmcnt(0) makes sure every ds_read of the slot's previous piece has returned.
__device__ __forceinline__ void dma_piece(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 2\n\t"
        "buffer_load_dwordx4 %1, %2, %4 offen nt lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(rs), "s"(lds_dst), "s"(soff)
        : "memory");
}
// Four consecutive pieces at once: lane i's 16 B of piece j (rs[v_j + soff], v_j = 16 i + 1024 j) at LDS byte
// lds_dst + 1024 j + 16 i.
__device__ __forceinline__ void dma_quad(__amdgpu_buffer_rsrc_t rs, unsigned v0, unsigned v1, unsigned v2, unsigned v3,
                                         unsigned soff, unsigned lds_dst) {
    unsigned keep;
    const unsigned d1 = lds_dst + 1024u, d2 = lds_dst + 2048u, d3 = lds_dst + 3072u;
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %7\n\t"
        "s_nop 2\n\t"
        "buffer_load_dwordx4 %1, %5, %6 offen nt lds\n\t"
        "s_mov_b32 m0, %8\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %5, %6 offen nt lds\n\t"
        "s_mov_b32 m0, %9\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %3, %5, %6 offen nt lds\n\t"
        "s_mov_b32 m0, %10\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %4, %5, %6 offen nt lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(rs), "s"(soff), "s"(lds_dst), "s"(d1), "s"(d2), "s"(d3)
        : "memory");
}
// this wave's outstanding vector-memory operations, without waiting (VM_CNT of IB_STS: bits 3:0 and 23:22)
__device__ __forceinline__ int vmcnt_now() {
    const unsigned ib = __builtin_amdgcn_s_getreg(7 | (0 << 6) | (31 << 11));
    return (int)((ib & 15u) | ((ib >> 18) & 0x30u));
}
// wait until at most n (wave-uniform) of this wave's vector-memory operations are outstanding
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
    switch (n) {
#define W_(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        W_(0) W_(1) W_(2) W_(3) W_(4) W_(5) W_(6) W_(7) W_(8) W_(9) W_(10) W_(11) W_(12) W_(13) W_(14)
#undef W_
        default: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    }
}
// a fresh 16-B read of LDS byte `addr` (the flag words): asm, so that no cached copy and no flat load can take its place
__device__ __forceinline__ u32x4 lds_peek128(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_poke32(unsigned addr, unsigned val) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(val) : "memory");
}

// int4 -> MFMA operand, 5 VALU ops per 8 weights.  The operands are fp16, whose 10-bit mantissa holds TWO nibble
// positions under one exponent pattern:
//   (x & 0x000F000F) | 0x64006400 = the fp16 pair (1024 + nibble 0, 1024 + nibble 4)
//   (x & 0x00F000F0) | 0x64006400 = the fp16 pair (1024 + 16 nibble 1, 1024 + 16 nibble 5)
// and the same two masks on x >> 8 give nibbles 2 / 6 and 16 x nibbles 3 / 7: one shift + four v_and_or_b32.  The
// factor 16 is undone on the activation side: the producers publish every ODD pair of the activation vector divided
// by 16 (exact in fp16), and the epilogue subtracts 1024 (S_even + S_odd) + zero (S_even + 16 S_odd) with the two
// sums taken while the vector is staged.  With the masks in SGPRs and the exponent pattern in a VGPR whose values the
// compiler cannot see, hipcc selects v_and_or_b32 itself (and pads the VALU -> MFMA hazard, which an inline-asm
// v_and_or_b32 does not get: that variant produced NaNs).
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
__device__ __forceinline__ uint32_t nib2f16(uint32_t x, uint32_t mask_s, uint32_t magic_v) { return (x & mask_s) | magic_v; }

#define FS_STAMP(i)                                                                   \
    do {                                                                              \
        if (p.dbg != nullptr && (threadIdx.x & 63) == 0) p.dbg[bid * 64 + (i)] = wall_clock64(); \
    } while (0)

}  // namespace

__global__ __launch_bounds__(kThreads) void fused_step_kernel(const FusedParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* misc = (float*)(smem + kOffMisc);
    char* xs = smem + kOffXs;
    char* part = smem + kOffPart;
    float* qs = (float*)(smem + kOffQ);
    float* knew = qs + kHs;
    float* vnew = knew + kHs;
    float* opart = (float*)(smem + kOffOpart);

    // workgroup -> head group: the 8 workgroups of a head sit on one XCD (blocks are dealt round-robin to the 8 XCDs;
    // a speed matter only — the protocol does not depend on placement)
    const int xcd = bid & 7, slot = bid >> 3;
    const int head = xcd * (kHeads / 8) + slot / kGs;
    const int hj = slot % kGs;  // which 16 dimensions of the head

    const int pos = p.pos[0];
    const int token = p.tokens[0];
    const unsigned step_id = p.state[1];
    const unsigned ebase = step_id * 1024u + 1u;
    const int n_fc = (p.fc_tiles - bid + kG - 1) / kG;       // this workgroup's pair tiles (2 or 3 for 7B)
    const int n_head_t = (p.head_tiles - bid + kG - 1) / kG;  // lm_head tiles (7 or 8)
    const int head_tiles_max = (p.head_tiles + kG - 1) / kG;

    // entered outside the cache (the host takes the cache-roll regime of model.py:214-218 elsewhere) or with a token id
    // outside the embedding table: refuse before anything is written.  Uniform over the grid, so no hand-off hangs.
    // (the LDS-DMA destinations are absolute LDS addresses: refuse as well if the dynamic segment does not start at 0)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    if (pos < 0 || pos >= p.S || token < 0 || token >= p.V || lds0 != 0u) {
        if (bid == 0 && threadIdx.x == 0) raise_abort(p, lds0 != 0u ? 0x11u : 0x10u);
        return;
    }
    if (threadIdx.x < 4) ((unsigned*)(smem + kOffFlag))[threadIdx.x] = 0u;
    FS_STAMP(0);
    __syncthreads();  // B0: the flag words are zero

    if (wave < kSW) {
        // =========================================================================================== streamers
        unsigned lane_off = lane * 16;
        const int g = lane >> 4;
        uint32_t magic = 0x64006400u;
        uint32_t nmask = 0x000F000Fu, nmask16 = 0x00F000F0u;
        asm volatile("" : "+v"(magic));  // opaque register values (see nib2f16)
        asm volatile("" : "+s"(nmask));
        asm volatile("" : "+s"(nmask16));
        int buf = 0;

        // ---- this wave's flat sequence of QUADS (4 consecutive 1-KiB pieces = 4 activation units of one row group):
        // per layer c_attn 3 (q, k, v of units wave * 4 ..), attn.c_proj 1, c_fc1/c_fc2 2 per pair tile (units 0-1, 2-3 of
        // the wave, fc1 / fc2 interleaved per unit), mlp.c_proj 3 (the wave's 10 or 11 units; the rest of the last quad is
        // requested through a zero-sized descriptor: no traffic); then lm_head 1 per tile.  A quad is ONE asm statement
        // of 4 loads: issuing costs ~20 instructions per 4 KiB instead of a loop iteration per piece.
        const int tile_attn = head * 8 + hj;
        const int mp_q = p.units_h / kSW, mp_r = p.units_h % kSW;
        const int mp_u0 = wave * mp_q + (wave < mp_r ? wave : mp_r), mp_n = mp_q + (wave < mp_r ? 1 : 0);
        const int fc_ring = n_fc < 2 ? n_fc : 2;  // pair tiles that go through the LDS ring (a third one: registers, below)
        const int q_layer = 3 + 1 + 2 * fc_ring + 3;
        const int total = p.n_layer * q_layer + n_head_t;
        const unsigned ring0 = (unsigned)(kOffRing + wave * kD * 1024);
        int issued = 0, cons = 0, landed = 0;  // QUADS requested / consumed / known to have landed
        int i_layer = 0, i_seg = 0, i_idx = 0, i_cnt = 3;
        __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.layer_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_h =
            __builtin_amdgcn_make_buffer_rsrc((void*)p.w_head, 0, (int)p.head_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_null = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0, 0x00020000);
        unsigned lo1 = lane_off + 1024u, lo2 = lane_off + 2048u, lo3 = lane_off + 3072u;

        auto issue_quad = [&]() {
            const unsigned dst = ring0 + (unsigned)(issued & (kQ - 1)) * 4096u;
            if (i_layer < p.n_layer) {
                unsigned off;
                if (i_seg == 0)
                    off = p.off_attn + (unsigned)((tile_attn + i_idx * (kC / 16)) * kUnitsC + wave * 4) * 1024u;
                else if (i_seg == 1)
                    off = p.off_proj + (unsigned)(bid * kUnitsC + wave * 4) * 1024u;
                else if (i_seg == 2)
                    off = p.off_fc + (unsigned)(((bid + (i_idx >> 1) * kG) * kUnitsC + wave * 4 + 2 * (i_idx & 1)) * 2) * 1024u;
                else
                    off = p.off_mproj + (unsigned)(bid * p.units_h + mp_u0 + 4 * i_idx) * 1024u;
                if (i_seg == 3 && 4 * i_idx + 4 > mp_n) {
                    // the partial last quad of mlp.c_proj: 4 loads all the same (vmcnt counts quads of 4), the ones past
                    // this wave's units through the zero-sized descriptor
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        dma_piece(4 * i_idx + j < mp_n ? rs_i : rs_null, lane_off, off + (unsigned)j * 1024u, dst + (unsigned)j * 1024u);
                } else {
                    dma_quad(rs_i, lane_off, lo1, lo2, lo3, off, dst);
                }
            } else {
                dma_quad(rs_h, lane_off, lo1, lo2, lo3, (unsigned)((bid + i_idx * kG) * kUnitsC + wave * 4) * 1024u, dst);
            }
            ++issued;
            if (++i_idx == i_cnt) {
                i_idx = 0;
                if (++i_seg == 4) {
                    i_seg = 0;
                    ++i_layer;
                    if (i_layer < p.n_layer)
                        rs_i = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (size_t)i_layer * p.layer_stride), 0,
                                                                 (int)p.layer_bytes, 0x00020000);
                }
                i_cnt = i_layer >= p.n_layer ? n_head_t : i_seg == 0 ? 3 : i_seg == 1 ? 1 : i_seg == 2 ? 2 * fc_ring : 3;
            }
        };
        // request quads: the next `must` ones of the sequence unconditionally (they are about to be waited for), then what
        // the ring (kQ quads) and the in-flight window (W quads) allow, up to quad `upto` of the sequence
        auto pump = [&](int must, int W, int upto) {
            while (issued < cons + must || (issued < upto && issued - cons < kQ && issued - landed < W)) issue_quad();
        };
        auto refresh_landed = [&]() {  // non-blocking: what this wave's vmcnt says has landed (4 loads per quad)
            const int l = issued - ((vmcnt_now() + 3) >> 2);
            if (landed < l) landed = l;
        };
        // block until the next n (<= kQ) quads of the sequence, all requested already, have landed
        auto wait_quads = [&](int n) {
            const int after = issued - (cons + n);  // quads requested behind the last one needed: 0 .. 3
            if (after <= 0)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (after == 1)
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (after == 2)
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            if (landed < cons + n) landed = cons + n;
        };
        // wait until both gatherer waves have staged stage `st`; top the ring up meanwhile (MI355_FUSED_PUBGATE: only once
        // gatherer 0 has issued the publish stores of this workgroup, so that nothing of ours queues in front of them)
        bool dbg_on = false;
#define FS_SSTAMP(i)                                                                      \
    do {                                                                                  \
        if (dbg_on && threadIdx.x == 0) p.dbg[bid * 64 + (i)] = wall_clock64();            \
    } while (0)
        auto poll_stage = [&](unsigned st, int must) {
            for (unsigned spins = 0;; ++spins) {
                const u32x4 fv = lds_peek128((unsigned)kOffFlag);  // (every lane reads the same words: wave-uniform)
                const unsigned f0 = __builtin_amdgcn_readfirstlane(fv[0]), f1 = __builtin_amdgcn_readfirstlane(fv[1]),
                               f2 = __builtin_amdgcn_readfirstlane(fv[2]);
                const bool ready = f0 >= st && f1 >= st;
                if (must > 0 && (ready || !MI355_FUSED_PUBGATE || f2 >= st)) {
                    refresh_landed();
                    pump(must, kWPoll, total);  // the coming phase's quads regardless of the window, the rest within it
                }
                if (ready) break;
                if (spins > kSpinLimit) {
                    if (lane == 0) raise_abort(p, 0x800u + st);
                    break;
                }
                __builtin_amdgcn_s_sleep(MI355_FUSED_POLL_SLEEP);
            }
        };
        f32x4 acc[3][2];
        auto zero_acc = [&]() {
#pragma unroll
            for (int r = 0; r < 3; ++r) acc[r][0] = acc[r][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        };
        // The weights of a tile leave LDS in ONE burst of ds_read_b128 (w[]: piece i of the tile, as it lies in the ring)
        // and the activation unit of step s + 1 is read before the MFMAs of step s: read per step, every step began with
        // an LDS round trip on the step's critical path (+0.45 us per phase against the register-ring kernel).
        auto read_b = [&](int unit, f16x8 (&b)[4]) {
            const char* xb = xs + unit * 256 + g * 64;
#pragma unroll
            for (int d = 0; d < 4; ++d) b[d] = *(const f16x8*)(xb + 16 * d);
        };
        auto mma = [&](auto Rc, const u32x4* v, const f16x8 (&b)[4]) {  // one step: R pieces against one activation unit
            constexpr int R = decltype(Rc)::value;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t x = v[r][d];
                    const uint32_t x8 = x >> 8;
                    u32x4 a;
                    a[0] = nib2f16(x, nmask, magic);
                    a[1] = nib2f16(x, nmask16, magic);
                    a[2] = nib2f16(x8, nmask, magic);
                    a[3] = nib2f16(x8, nmask16, magic);
                    acc[r][d & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), b[d],
                                                                           acc[r][d & 1], 0, 0, 0);
                }
            }
        };
        // NS steps of R pieces: w[s * R + r], activation units unit0 + s
        auto tile_mma = [&](auto Rc, auto NSc, const u32x4* w, int unit0, int nsteps) {
            constexpr int R = decltype(Rc)::value, NS = decltype(NSc)::value;
            f16x8 b[4], bn[4];
            read_b(unit0, bn);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s < nsteps) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) b[d] = bn[d];
                    if (s + 1 < NS) read_b(unit0 + (s + 1 < nsteps ? s + 1 : 0), bn);
                    mma(Rc, w + s * R, b);
                }
            }
        };
        auto ring_piece = [&](unsigned addr) { return *(const u32x4*)(smem + addr + lane_off); };
        auto qaddr = [&](int q) { return ring0 + (unsigned)(q & (kQ - 1)) * 4096u; };  // LDS byte address of quad q
        // tile done: this wave's partial outputs (column 0 of the 16 x 16 result: every column is the same vector)
        auto put_partials = [&](int R) {
            if ((lane & 15) == 0) {
                f32x4* pp = (f32x4*)(part + (size_t)((buf * kSW + wave) * 4) * 64) + (lane >> 4);
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    if (r < R) pp[r * 4] = acc[r][0] + acc[r][1];
            }
        };
        using R1 = std::integral_constant<int, 1>;
        using R2 = std::integral_constant<int, 2>;
        using R3 = std::integral_constant<int, 3>;
        using N4 = std::integral_constant<int, 4>;
        using N12 = std::integral_constant<int, 12>;

        // The phases of the step in order: per layer c_attn (0), attention (1), attn.c_proj (2), c_fc1/c_fc2 (3),
        // mlp.c_proj (4); then lm_head (5).  ONE loop, ONE copy of the polling / issuing / waiting code: the kernel has to
        // stay inside the 64-KiB instruction cache (the first version inlined them per phase, 81 KB, and every phase
        // started on instruction-cache misses).
        unsigned stage = 0;  // stages waited for so far (the gatherers count the same way)
        const bf16_t* kv_l = (const bf16_t*)p.kv;
        const int n_ph = p.n_layer * 5 + 1;
        int kind = 0, layer = 0;
#pragma unroll 1
        for (int ph = 0; ph < n_ph; ++ph) {
            if (ph == n_ph - 1) kind = 5;
            dbg_on = p.dbg != nullptr && layer == p.dbg_layer && kind < 5;
            asm volatile("" : "+v"(lane_off));  // per-lane addresses are recomputed per phase, not hoisted and spilled
            if (kind == 1) {
                // ---------------- attention: scores over the whole context, then this workgroup's 16 output dims
                {
                    const bf16_t* kc = kv_l + (size_t)head * p.S * kHs;
                    const bf16_t* vc = kc + (size_t)kHeads * p.S * kHs;
                    const int li = (lane_off >> 4) & 15, lr = lane_off >> 8;
                    const int half = lane_off >> 9, rl = (lane_off >> 4) & 31;
                    const __amdgpu_buffer_rsrc_t rk =
                        __builtin_amdgcn_make_buffer_rsrc((void*)kc, 0, pos * (kHs * 2), 0x00020000);
                    const __amdgpu_buffer_rsrc_t rv =
                        __builtin_amdgcn_make_buffer_rsrc((void*)vc, 0, pos * (kHs * 2), 0x00020000);
                    const int n_blocks = (pos + 255) >> 8;  // blocks of 256 cached rows: 32 per wave and block
                    u32x4 kr[8], vr;
                    // rows of block 0: requested before q is known.  A wave scores the SAME 32 rows it then weighs the
                    // values of (row wave * 32 + u * 4 + lr for the scores, 16 lanes per row; row wave * 32 + rl for the
                    // values, 8 of the workgroup's 16 output dimensions per lane): no score leaves the wave, the softmax is
                    // a per-wave partial (running maximum, sum, weighted values) that gatherer 0 merges.
                    // No weight piece is requested between these loads and their use: VMEM returns in order, so the rows
                    // come back behind at most the pieces that are in flight now.
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int t = wave * 32 + u * 4 + lr;
                        kr[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                              rk, t < pos ? (unsigned)t * 256u + li * 16u : 0xFFFFFFF0u, 0, 0));
                    }
                    {
                        const int t = wave * 32 + rl;
                        vr = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                           rv, t < pos ? (unsigned)t * 256u + hj * 32u + half * 16u : 0xFFFFFFF0u, 0, 0));
                    }
                    poll_stage(++stage, 0);  // q / new k / new v of the head are in LDS
                    FS_SSTAMP(23);
                    float qf[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) qf[j] = qs[li * 8 + j];
                    // lane L (value row rl = L & 31) takes its row's score from the lane group that computed it
                    const int pull = ((((lane_off >> 4) & 3) << 4) | (rl >> 2)) * 4;
                    float m_run = -1.0e30f, l_run = 0.f;
                    float of[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) of[j] = 0.f;
                    for (int blk = 0; blk < n_blocks; ++blk) {
                        u32x4 vv = vr;
                        if (blk > 0) {
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const int t = blk * 256 + wave * 32 + u * 4 + lr;
                                kr[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                      rk, t < pos ? (unsigned)t * 256u + li * 16u : 0xFFFFFFF0u, 0, 0));
                            }
                            const int t = blk * 256 + wave * 32 + rl;
                            vv = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                               rv, t < pos ? (unsigned)t * 256u + hj * 32u + half * 16u : 0xFFFFFFF0u, 0, 0));
                        }
                        float sel = 0.f;
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            float dot = 0.f;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                dot += qf[2 * i] * __uint_as_float(kr[u][i] << 16);
                                dot += qf[2 * i + 1] * __uint_as_float(kr[u][i] & 0xffff0000u);
                            }
                            dot = group_sum(dot, 16);
                            if ((li & 7) == u) sel = dot;
                        }
                        const int t = blk * 256 + wave * 32 + rl;
                        float sc = __int_as_float(__builtin_amdgcn_ds_bpermute(pull, __float_as_int(sel))) * p.scale;
                        sc = t < pos ? sc : -1.0e30f;
                        float bm = fmaxf(sc, lane_xor16(sc));  // maximum over the wave's 32 rows (both halves hold them)
                        bm = MI355_DPP_MAX(bm, 0x140);
                        bm = MI355_DPP_MAX(bm, 0x141);
                        bm = MI355_DPP_MAX(bm, 0x4E);
                        bm = MI355_DPP_MAX(bm, 0xB1);
                        float s_new = -1.0e30f;
                        if (blk == 0 && wave == 0) {  // the new token's own score, from the LDS copy of its key
                            float dot = qs[lane] * knew[lane] + qs[lane + 64] * knew[lane + 64];
                            s_new = group_sum(dot, 64) * p.scale;
                            bm = fmaxf(bm, s_new);
                        }
                        const float m_new = fmaxf(m_run, bm);
                        const float corr = __expf(m_run - m_new);
                        const float pr = t < pos ? __expf(sc - m_new) : 0.f;
                        l_run = l_run * corr + pr;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            of[2 * i] = of[2 * i] * corr + pr * __uint_as_float(vv[i] << 16);
                            of[2 * i + 1] = of[2 * i + 1] * corr + pr * __uint_as_float(vv[i] & 0xffff0000u);
                        }
                        if (blk == 0 && wave == 0 && rl == 0) {  // the new token's value row
                            const float pn = __expf(s_new - m_new);
                            l_run += pn;
#pragma unroll
                            for (int j = 0; j < 8; ++j) of[j] += pn * vnew[hj * 16 + half * 8 + j];
                        }
                        m_run = m_new;
                    }
                    if (n_blocks == 0 && wave == 0) {  // position 0: the new token attends to itself only
                        float dot = qs[lane] * knew[lane] + qs[lane + 64] * knew[lane + 64];
                        m_run = group_sum(dot, 64) * p.scale;
                        if (rl == 0) {
                            l_run = 1.f;
#pragma unroll
                            for (int j = 0; j < 8; ++j) of[j] = vnew[hj * 16 + half * 8 + j];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) of[j] = group_sum(of[j], 32);
                    l_run = group_sum(l_run, 32);
                    if (rl == 0) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) opart[wave * 16 + half * 8 + j] = of[j];
                    }
                    if ((threadIdx.x & 63) == 0) {
                        misc[16 + wave] = m_run;
                        misc[24 + wave] = l_run;
                    }
                    FS_SSTAMP(25);
                    __syncthreads();  // Ba3: partial outputs of the 8 waves
                }

            } else {
                // ---------------- a linear phase.  Its first quads (as many as the ring holds) were requested while the
                // wave waited for the hand-off (poll_stage: `must`), so the compute path only waits — straight-line code
                // per phase, no requesting except where a phase is longer than the ring (c_fc1/c_fc2 of a 3-tile
                // workgroup, lm_head).  (A generic loop over "groups of quads" with the requesting code inside cost 0.5 us
                // of scalar instructions per group: 3 us per layer.)
                const int ph_quads = kind == 0 ? 3 : kind == 2 ? 1 : kind == 3 ? 2 * fc_ring : kind == 4 ? 3 : n_head_t;
                // c_fc1/c_fc2 of a workgroup with three pair tiles is 6 quads, the ring holds 4: the third tile (8 KiB per
                // wave) is requested into REGISTERS before the wait for the hand-off, so that all 192 KiB of the phase
                // stream under the hand-off (a ring-only version streamed the third tile during the phase: 4.7 us instead
                // of 2.2).  Unconditional loads through a zero-sized descriptor elsewhere (a load in a branch makes hipcc
                // wait with vmcnt(0) at the join).
                u32x4 rq[8];
                {
                    const bool third = kind == 3 && n_fc == 3;
                    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(
                        (void*)(p.w + (size_t)(layer < p.n_layer ? layer : 0) * p.layer_stride), 0, third ? (int)p.layer_bytes : 0, 0x00020000);
                    const unsigned off3 = p.off_fc + (unsigned)(((bid + 2 * kG) * kUnitsC + wave * 4) * 2) * 1024u;
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        rq[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_c, lane_off, third ? off3 + j * 1024u : 0u, 2));
                }
                poll_stage(++stage, ph_quads < kQ ? ph_quads : kQ);
                FS_SSTAMP(kind == 0 ? 20 : 22 + 2 * kind);
                if (kind == 0) {
                    zero_acc();
                    if (landed < cons + 3) wait_quads(3);
                    const unsigned q0 = qaddr(cons), q1 = qaddr(cons + 1), q2 = qaddr(cons + 2);
                    u32x4 w[12];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        w[3 * s] = ring_piece(q0 + s * 1024u);
                        w[3 * s + 1] = ring_piece(q1 + s * 1024u);
                        w[3 * s + 2] = ring_piece(q2 + s * 1024u);
                    }
                    tile_mma(R3{}, N4{}, w, wave * 4, 4);
                    cons += 3;
                    FS_SSTAMP(21);
                    put_partials(3);
                    __syncthreads();  // Bt
                    buf ^= 1;
                } else if (kind == 2) {
                    zero_acc();
                    if (landed < cons + 1) wait_quads(1);
                    const unsigned q0 = qaddr(cons);
                    u32x4 w[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) w[s] = ring_piece(q0 + s * 1024u);
                    tile_mma(R1{}, N4{}, w, wave * 4, 4);
                    cons += 1;
                    FS_SSTAMP(27);
                    put_partials(1);
                    __syncthreads();  // Bt
                    buf ^= 1;
                } else if (kind == 3) {
#pragma unroll 1
                    for (int ti = 0; ti < 2; ++ti) {
                        if (ti < n_fc) {
                            zero_acc();
                            if (landed < cons + 2) wait_quads(2);
                            const unsigned q0 = qaddr(cons), q1 = qaddr(cons + 1);
                            u32x4 w[8];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                w[j] = ring_piece(q0 + j * 1024u);
                                w[4 + j] = ring_piece(q1 + j * 1024u);
                            }
                            tile_mma(R2{}, N4{}, w, wave * 4, 4);
                            cons += 2;
                            if (ti + 1 == n_fc) FS_SSTAMP(29);
                            put_partials(2);
                        }
                        __syncthreads();  // Bt
                        buf ^= 1;
                    }
                    if (n_fc == 3) {  // the third pair tile, from registers
                        zero_acc();
                        tile_mma(R2{}, N4{}, rq, wave * 4, 4);
                        FS_SSTAMP(29);
                        put_partials(2);
                    }
                    __syncthreads();  // Bt (third tile)
                    buf ^= 1;
                } else if (kind == 4) {
                    zero_acc();
                    if (landed < cons + 3) wait_quads(3);
                    u32x4 w[12];
#pragma unroll
                    for (int s = 0; s < 12; ++s) w[s] = ring_piece(qaddr(cons + (s >> 2)) + (unsigned)(s & 3) * 1024u);
                    tile_mma(R1{}, N12{}, w, mp_u0, mp_n);
                    cons += 3;
                    FS_SSTAMP(31);
                    put_partials(1);
                    __syncthreads();  // Bt
                    buf ^= 1;
                } else {  // lm_head
#pragma unroll 1
                    for (int ti = 0; ti < head_tiles_max; ++ti) {
                        if (ti < n_head_t) {
                            zero_acc();
                            if (issued < total) pump(1, kWRun, total);
                            if (landed < cons + 1) wait_quads(1);
                            const unsigned q0 = qaddr(cons);
                            u32x4 w[4];
#pragma unroll
                            for (int s = 0; s < 4; ++s) w[s] = ring_piece(q0 + s * 1024u);
                            tile_mma(R1{}, N4{}, w, wave * 4, 4);
                            cons += 1;
                            put_partials(1);
                        }
                        __syncthreads();  // Bt
                        buf ^= 1;
                    }
                }
            }
            if (++kind == 5) {
                kind = 0;
                ++layer;
                kv_l += (size_t)2 * kHeads * p.S * kHs;
            }
        }
        dbg_on = false;
#undef FS_SSTAMP
    } else {
        // =========================================================================================== gatherers
        const int gw = wave - kSW;  // 0: combines / publishes, 1: helps with the sweeps
#if MI355_FUSED_GPRIO
        // the gatherers are the youngest waves of the workgroup: at equal priority they lose every issue arbitration to
        // the streamer waves that poll beside them, and everything they do is on the step's critical path
        __builtin_amdgcn_s_setprio(3);
#endif
        unsigned edge = 0;   // edges published so far in this step (the epoch of the next one is ebase + edge)
        unsigned stage = 0;  // stages staged so far (flag word of this wave)
        unsigned pubs = 0;   // publishes issued so far (gatherer 0)
        int xpar = 0, apar = 0, hpar = 0, qpar = 0;
        int buf = 0;
        const __amdgpu_buffer_rsrc_t rs_gx = __builtin_amdgcn_make_buffer_rsrc((void*)p.gx, 0, 2 * 2304 * 8, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_ga = __builtin_amdgcn_make_buffer_rsrc((void*)p.ga, 0, 2 * 2048 * 8, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_gh = __builtin_amdgcn_make_buffer_rsrc((void*)p.gh, 0, 2 * (p.H / 2) * 8, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_gq =
            __builtin_amdgcn_make_buffer_rsrc((void*)p.gq, 0, 2 * kHeads * 256 * 8, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_gm = __builtin_amdgcn_make_buffer_rsrc((void*)p.gm, 0, 512 * 8, 0x00020000);

        // the streamers wait for these words (poll_stage): "this wave has staged its share of stage n" and, from
        // gatherer 0, "the publish stores of publish n are in the memory pipeline".  LDS operations of a wave execute
        // in order, so the staging writes before the flag are visible before it.
        auto set_stage = [&]() {
            ++stage;
            if (lane == 0) lds_poke32((unsigned)(kOffFlag + 4 * gw), stage);
        };
        auto set_pub = [&]() {
            ++pubs;
            if (gw == 0 && lane == 0) lds_poke32((unsigned)(kOffFlag + 8), pubs);
        };

        // ---- epilogue mapping of gatherer 0: lane = (pair pg = lane >> 3, streamer wave w8 = lane & 7).  A lane reads
        // rows 2 pg, 2 pg + 1 of ONE wave's partial outputs (8 B), the 8 lanes of a pair are summed with DPP (fixed
        // order), and every lane then holds both outputs of its pair: RoPE pairs, fp16 pair granules and the residual
        // rows stay in registers.
        int lane_v = lane;  // made opaque once per layer: per-lane pointers are otherwise hoisted out of the layer loop
                            // (a few dozen 64-bit addresses) and spilled to scratch, i.e. to VMEM on the hand-off path
        int pg = lane >> 3, w8 = lane & 7;
        auto tile_pair = [&](int r) {
            float2 t = *(const float2*)(part + (size_t)((buf * kSW + w8) * 4 + r) * 64 + pg * 8);
            t.x = group_sum(t.x, 8);
            t.y = group_sum(t.y, 8);
            return t;
        };
        auto ldpair = [&](const bf16_t* q) {  // two consecutive bf16 (4-byte aligned) as floats
            const unsigned v = *(const unsigned*)q;
            return float2{__uint_as_float(v << 16), __uint_as_float(v & 0xffff0000u)};
        };
        auto bfpair = [&](float a, float b) { return (unsigned)f32_to_bf16(a) | ((unsigned)f32_to_bf16(b) << 16); };
        // activation pair granule: fp16 (a, b); ODD pairs of a vector carry a / 16, b / 16 (see nib2f16).  pg is the
        // pair's index inside its 8-pair row, the rows start at even pair indices.
        // fp16 has 5 exponent bits: the conversion saturates (a finite, if clipped, operand instead of an inf that the
        // +1024 offsets would turn into NaN) and COUNTS the clip in state[2] (mi355_fused_step_status reports it: the
        // step's outputs then differ from the unclipped arithmetic); the residual stream, whose size nothing bounds, is
        // published times a power of two that brings its rms near 1 (publish_x).
        auto hpair = [&](float a, float b) {
            const float k = (pg & 1) ? 0.0625f : 1.0f;
            const float ak = a * k, bk = b * k;
            if (fmaxf(fabsf(ak), fabsf(bk)) > 65504.f) atomicAdd(p.state + 2, 1u);
            const f16x2 h = {(_Float16)__builtin_amdgcn_fmed3f(ak, -65504.f, 65504.f),
                             (_Float16)__builtin_amdgcn_fmed3f(bk, -65504.f, 65504.f)};
            return __builtin_bit_cast(unsigned, h);
        };
        // sums of the staged operands, even pairs in .x and odd pairs in .y (one v_dot2_f32_f16 per dword).  They undo
        // the +1024 / zero-point offsets of the int4 operands:
        //   y = scale (acc - 1024 (S_even + S_odd) - zero (S_even + 16 S_odd));
        // every workgroup needs the same sums, so they are taken while the vector is staged.
        const f16x2 ones2 = {(_Float16)1.0f, (_Float16)1.0f};
        auto pair_sums = [&](float2& sx, unsigned even, unsigned odd) {
            sx.x = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, even), ones2, sx.x, false);
            sx.y = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, odd), ones2, sx.y, false);
        };
        // misc[4 + gw] / misc[6 + gw]: this gatherer wave's S_even / S_odd; the epilogue form {A, B}:
        // y = scale (acc - A - zero B).  Written while a stage is staged, read by gatherer 0 after that stage's first
        // tile barrier; the next stage's sums cannot be written before gatherer 0 has published this one (its granules
        // are part of what the next sweep waits for).
        auto put_sums = [&](float2 sx) {
            sx.x = group_sum(sx.x, 64);
            sx.y = group_sum(sx.y, 64);
            if (lane == 0) {
                misc[4 + gw] = sx.x;
                misc[6 + gw] = sx.y;
            }
        };
        auto get_sums = [&]() {
            const float se = misc[4] + misc[5], so = misc[6] + misc[7];
            return float2{1024.f * (se + so), se + 16.f * so};
        };
        bool dbg_on = false;
#define FS_GSTAMP(i)                                                                              \
    do {                                                                                          \
        if (dbg_on && gw == 0 && lane == 0) p.dbg[bid * 64 + (i)] = wall_clock64();               \
    } while (0)

        // publish an x-type edge: fp16(x_scale * norm_scale * x) pairs + the partial sum of squares of this workgroup's
        // rows.  x_scale = the power of two next to 1/rms of the PREVIOUS x edge (the same float in every workgroup: all
        // of them reduce the same 256 partial sums in the same order; 1 for the embedding): the residual stream changes
        // by one sub-layer's output between two edges, so the published values stay O(norm weight), far from the fp16
        // limits both ways.  The consumer folds 1 / x_scale into the 1/rms factor of its epilogue.
        float x_scale = 1.f;       // applied to the edge published last (= the one gathered next)
        float rinv_seen = 1.f;     // 1/rms of the x edge gathered last
        auto publish_x = [&](float2 xv, float2 gsc) {
            const unsigned ep = ebase + edge;
            u64* dst = p.gx + (size_t)xpar * 2304;
            x_scale = __uint_as_float((__float_as_uint(rinv_seen) + 0x00400000u) & 0x7F800000u);
            if (w8 == 0) gr_store(dst + bid * 8 + pg, ep, hpair(x_scale * gsc.x * xv.x, x_scale * gsc.y * xv.y));
            float ss = xv.x * xv.x + xv.y * xv.y;  // the same in the 8 lanes of a pair: sum over the 8 pairs
            ss = MI355_DPP_ADD(ss, 0x140);
            ss += lane_xor16(ss);
            ss += lane_xor32(ss);
            if (lane == 0) gr_store(dst + 2048 + bid, ep, __float_as_uint(ss));
        };
        // gather an x-type edge into xs (fp16), 1/rms into misc[0], the operand sums into misc[4 .. 7]
        auto gather_x = [&]() {
            const unsigned ep = ebase + edge;
            const unsigned base = (unsigned)xpar * 2304u * 8u;
            if (gw == 0) {
                u32x4 v[9];
                // loads 0 .. 447 of the pair region (7 per lane) and the 128 loads of the sums of squares (2 per lane);
                // gatherer 1 takes the other 9 x 64 loads of the pair region: the same number of loads in both waves
                for (unsigned spins = 0;; ++spins) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
                        const unsigned off = k < 7 ? base + (unsigned)(k * 64 + lane_v) * 16u
                                                   : base + 2048u * 8u + (unsigned)((k - 7) * 64 + lane_v) * 16u;
                        v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_gx, off, 0, 16));
                    }
#pragma unroll
                    for (int k = 0; k < 9; ++k) ok &= v[k][1] == ep && v[k][3] == ep;
                    if (__all(ok)) break;
                    if (spins > kSpinLimit || aborted(p)) {
                        if (lane == 0) raise_abort(p, 0x100u + edge);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                float2 sx = {0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    *(u64*)(xs + (size_t)(k * 64 + lane_v) * 8) = ((u64)v[k][2] << 32) | v[k][0];
                    pair_sums(sx, v[k][0], v[k][2]);
                }
                float ss = ((__uint_as_float(v[7][0]) + __uint_as_float(v[7][2])) + __uint_as_float(v[8][0])) +
                           __uint_as_float(v[8][2]);
                ss = group_sum(ss, 64);
                put_sums(sx);
                if (lane == 0) misc[0] = rsqrtf(ss / (float)kC + p.eps);
            } else {
                u32x4 v[9];
                sweep<9>(p, rs_gx, base, 448, 1024, ep, v, 0x200u + edge, lane_v);
                float2 sx = {0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    *(u64*)(xs + (size_t)(448 + k * 64 + lane_v) * 8) = ((u64)v[k][2] << 32) | v[k][0];
                    pair_sums(sx, v[k][0], v[k][2]);
                }
                put_sums(sx);
            }
            xpar ^= 1;
            ++edge;
            set_stage();
        };
        auto deq = [&](float2 t, float2 sc_, float2 z_, float2 sx) {
            return float2{sc_.x * (t.x - sx.x - z_.x * sx.y), sc_.y * (t.y - sx.x - z_.y * sx.y)};
        };

        // ---- the residual rows of this workgroup: embedding of the step's token (model.py:102)
        int r0 = bid * 16 + 2 * pg;  // first row of this lane's pair among the n_embd residual rows
        float2 xres = ldpair(p.wte + (size_t)token * kC + r0);
        const bf16_t* norms_l = p.norms;
        const bf16_t* sz_l = p.sz;
        bf16_t* kv_l = p.kv;
        const float2 cs = *(const float2*)(p.rope + ((size_t)pos * (kHs / 2) + hj * 8 + pg) * 2);
        if (gw == 0) publish_x(xres, ldpair(norms_l + r0));
        set_pub();
        for (int l = 0; l < p.n_layer; ++l) {
            dbg_on = p.dbg != nullptr && l == p.dbg_layer;
            asm volatile("" : "+v"(lane_v));
            pg = lane_v >> 3;
            w8 = lane_v & 7;
            r0 = bid * 16 + 2 * pg;
            // ================= c_attn
            const int nq = (head * 8 + hj) * 16 + 2 * pg;  // q rows of this lane's pair; k at + C, v at + 2 C
            float2 sc[3], zr[3];
            if (gw == 0) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    sc[r] = ldpair(sz_l + nq + r * kC);
                    zr[r] = ldpair(sz_l + 3 * kC + nq + r * kC);
                }
            }
            gather_x();
            FS_GSTAMP(2);
            __syncthreads();  // Bt
            if (gw == 0) {
                rinv_seen = misc[0];
                const float rinv = rinv_seen / x_scale;
                const float2 sx = get_sums();
                float2 y[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    y[r] = deq(tile_pair(r), sc[r], zr[r], sx);
                    y[r].x *= rinv;
                    y[r].y *= rinv;
                }
                // RoPE (model.py:306-323) of the q / k pair, publish to the head group, write the cache row
                const unsigned ep = ebase + edge;
                u64* dst = p.gq + ((size_t)qpar * kHeads + head) * 256 + hj * 32;
                bf16_t* krow = kv_l + ((size_t)head * p.S + pos) * kHs + hj * 16;
                bf16_t* vrow = krow + (size_t)kHeads * p.S * kHs;
                const float qa = y[0].x * cs.x - y[0].y * cs.y, qb = y[0].y * cs.x + y[0].x * cs.y;
                const unsigned kp = bfpair(y[1].x * cs.x - y[1].y * cs.y, y[1].y * cs.x + y[1].x * cs.y);
                const unsigned vp = bfpair(y[2].x, y[2].y);
                if (w8 == 0) gr_store(dst + 2 * pg, ep, __float_as_uint(qa));
                if (w8 == 1) gr_store(dst + 2 * pg + 1, ep, __float_as_uint(qb));
                if (w8 == 2) gr_store(dst + 16 + pg, ep, kp);
                if (w8 == 3) gr_store(dst + 24 + pg, ep, vp);
                if (w8 == 4) ((unsigned*)krow)[pg] = kp;
                if (w8 == 5) ((unsigned*)vrow)[pg] = vp;
            }
            set_pub();
            FS_GSTAMP(3);
            buf ^= 1;
            // ================= attention
            {
                const unsigned ep = ebase + edge;
                if (gw == 0) {
                    u32x4 v[2];
                    sweep<2>(p, rs_gq, (unsigned)((qpar * kHeads + head) * 256) * 8u, 0, 128, ep, v, 0x300u + edge, lane_v);
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
#pragma unroll
                        for (int e2 = 0; e2 < 2; ++e2) {
                            const int gi = (k * 64 + lane_v) * 2 + e2;  // granule index inside the head's 256 (lane_v: not hoisted)
                            const int jj = gi >> 5, e = gi & 31;
                            const unsigned val = v[k][2 * e2];
                            if (e < 16) {
                                qs[jj * 16 + e] = __uint_as_float(val);
                            } else if (e < 24) {
                                knew[jj * 16 + 2 * (e - 16)] = __uint_as_float(val << 16);
                                knew[jj * 16 + 2 * (e - 16) + 1] = __uint_as_float(val & 0xffff0000u);
                            } else {
                                vnew[jj * 16 + 2 * (e - 24)] = __uint_as_float(val << 16);
                                vnew[jj * 16 + 2 * (e - 24) + 1] = __uint_as_float(val & 0xffff0000u);
                            }
                        }
                    }
                }
                qpar ^= 1;
                ++edge;
                set_stage();
                FS_GSTAMP(4);
                __syncthreads();  // Ba3
                FS_GSTAMP(5);
                if (gw == 0) {
                    // merge the 8 per-wave softmax partials (running maximum, sum, weighted values)
                    float2 o = *(const float2*)(opart + w8 * 16 + 2 * pg);
                    const float mw = misc[16 + w8];
                    float mall = MI355_DPP_MAX(mw, 0xB1);
                    mall = MI355_DPP_MAX(mall, 0x4E);
                    mall = MI355_DPP_MAX(mall, 0x141);
                    const float wsc = __expf(mw - mall);
                    o.x = group_sum(o.x * wsc, 8);
                    o.y = group_sum(o.y * wsc, 8);
                    const float inv = 1.0f / group_sum(misc[24 + w8] * wsc, 8);
                    // attention output elements head * 128 + hj * 16 + 2 pg, + 1 -> one pair granule
                    if (w8 == 0)
                        gr_store(p.ga + (size_t)apar * 2048 + head * 64 + hj * 8 + pg, ebase + edge, hpair(o.x * inv, o.y * inv));
                }
                set_pub();
                FS_GSTAMP(6);
            }
            // ================= attn.c_proj (+ residual)
            {
                float2 s1 = {0.f, 0.f}, z1 = {0.f, 0.f}, gn = {0.f, 0.f};
                if (gw == 0) {
                    s1 = ldpair(sz_l + 6 * kC + r0);
                    z1 = ldpair(sz_l + 7 * kC + r0);
                    gn = ldpair(norms_l + kC + r0);  // rms_2
                }
                const unsigned ep = ebase + edge;
                u32x4 v[8];
                sweep<8>(p, rs_ga, (unsigned)apar * 2048u * 8u, gw * 512, gw * 512 + 512, ep, v, 0x400u + edge, lane_v);
                float2 sxp = {0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    *(u64*)(xs + (size_t)(gw * 512 + k * 64 + lane_v) * 8) = ((u64)v[k][2] << 32) | v[k][0];
                    pair_sums(sxp, v[k][0], v[k][2]);
                }
                put_sums(sxp);
                apar ^= 1;
                ++edge;
                set_stage();
                FS_GSTAMP(7);
                __syncthreads();  // Bt
                if (gw == 0) {
                    const float2 d = deq(tile_pair(0), s1, z1, get_sums());
                    xres.x += d.x;
                    xres.y += d.y;
                    publish_x(xres, gn);
                }
                set_pub();
                FS_GSTAMP(8);
                buf ^= 1;
            }
            // ================= c_fc1 / c_fc2 + SwiGLU
            {
                const bf16_t* s_fc = sz_l + 8 * kC;
                float2 fs1[kMaxFcTiles], fz1[kMaxFcTiles], fs2[kMaxFcTiles], fz2[kMaxFcTiles];
                if (gw == 0) {
#pragma unroll
                    for (int t = 0; t < kMaxFcTiles; ++t) {
                        const int n = (bid + (t < n_fc ? t : 0) * kG) * 16 + 2 * pg;
                        fs1[t] = ldpair(s_fc + n);
                        fz1[t] = ldpair(s_fc + p.H + n);
                        fs2[t] = ldpair(s_fc + 2 * p.H + n);
                        fz2[t] = ldpair(s_fc + 3 * p.H + n);
                    }
                }
                gather_x();
                FS_GSTAMP(9);
                const unsigned ep = ebase + edge;
                u64* dst = p.gh + (size_t)hpar * (p.H / 2);
                float rinv = 0.f;
                float2 sx = {0.f, 0.f};
#pragma unroll
                for (int t = 0; t < kMaxFcTiles; ++t) {
                    __syncthreads();  // Bt
                    if (t == 0) {  // (after the first tile barrier both gatherers' sums of this stage are in place)
                        rinv_seen = misc[0];
                        rinv = rinv_seen / x_scale;
                        sx = get_sums();
                    }
                    if (gw == 0 && t < n_fc) {
                        const float2 a = deq(tile_pair(0), fs1[t], fz1[t], sx);
                        const float2 b = deq(tile_pair(1), fs2[t], fz2[t], sx);
                        if (w8 == 0)
                            gr_store(dst + (bid + t * kG) * 8 + pg, ep,
                                     hpair(swiglu_f32(a.x * rinv, b.x * rinv), swiglu_f32(a.y * rinv, b.y * rinv)));
                    }
                    buf ^= 1;
                }
                set_pub();
                FS_GSTAMP(10);
            }
            // ================= mlp.c_proj (+ residual) -> next layer's x edge
            {
                float2 s1 = {0.f, 0.f}, z1 = {0.f, 0.f}, gn = {0.f, 0.f};
                const bf16_t* s_mp = sz_l + 8 * kC + 4 * p.H;
                if (gw == 0) {
                    s1 = ldpair(s_mp + r0);
                    z1 = ldpair(s_mp + kC + r0);
                    gn = ldpair(norms_l + 2 * kC + r0);  // rms_1 of the next layer, or ln_f after the last
                }
                const unsigned ep = ebase + edge;
                const int n_loads = p.H / 4, half_l = (n_loads + 1) / 2;
                const int first = gw * half_l, end = gw == 0 ? half_l : n_loads;
                float2 sxp = {0.f, 0.f};
                {
                    // chunks of 8, 4, 8, 4 loads per lane (24 >= 12288 / 4 / 2 / 64), TWO in flight: only the first one
                    // waits for producers; issued one after the other each later chunk cost its own memory round trip on
                    // the longest hand-off of the layer (44 KB of granules)
                    const unsigned hbase = (unsigned)hpar * (unsigned)(p.H / 2) * 8u;
                    int lh = lane_v;
                    asm volatile("" : "+v"(lh));  // addresses of this block are computed here, not hoisted and spilled
                    u32x4 va[8], vb[4];
                    auto stage_a = [&](int c0) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const int i = c0 + k * 64 + lh;
                            if (i < end) {
                                *(u64*)(xs + (size_t)i * 8) = ((u64)va[k][2] << 32) | va[k][0];
                                pair_sums(sxp, va[k][0], va[k][2]);
                            }
                        }
                    };
                    auto stage_b = [&](int c0) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int i = c0 + k * 64 + lh;
                            if (i < end) {
                                *(u64*)(xs + (size_t)i * 8) = ((u64)vb[k][2] << 32) | vb[k][0];
                                pair_sums(sxp, vb[k][0], vb[k][2]);
                            }
                        }
                    };
                    const int c1 = first + 512, c2 = first + 768, c3 = first + 1280;
                    sweep_issue<8>(rs_gh, hbase, first, end, va, lh);
                    sweep_issue<4>(rs_gh, hbase, c1, end, vb, lh);
                    sweep<8>(p, rs_gh, hbase, first, end, ep, va, 0x500u + edge, lh, true);
                    stage_a(first);
                    sweep_issue<8>(rs_gh, hbase, c2, end, va, lh);
                    sweep<4>(p, rs_gh, hbase, c1, end, ep, vb, 0x500u + edge, lh, true);
                    stage_b(c1);
                    sweep_issue<4>(rs_gh, hbase, c3, end, vb, lh);
                    sweep<8>(p, rs_gh, hbase, c2, end, ep, va, 0x500u + edge, lh, true);
                    stage_a(c2);
                    sweep<4>(p, rs_gh, hbase, c3, end, ep, vb, 0x500u + edge, lh, true);
                    stage_b(c3);
                }
                put_sums(sxp);
                hpar ^= 1;
                ++edge;
                set_stage();
                FS_GSTAMP(11);
                __syncthreads();  // Bt
                if (gw == 0) {
                    const float2 d = deq(tile_pair(0), s1, z1, get_sums());
                    xres.x += d.x;
                    xres.y += d.y;
                    publish_x(xres, gn);
                }
                set_pub();
                FS_GSTAMP(12);
                buf ^= 1;
            }
            norms_l += 2 * kC;
            sz_l += p.sz_layer_stride;
            kv_l += (size_t)2 * kHeads * p.S * kHs;
        }
        dbg_on = false;
        // ================= ln_f + lm_head (+ greedy arg-max, generate.py:68-85 with top_k = 1)
        {
            // scale / zero of a tile's rows are requested one tile ahead
            auto head_sz = [&](int t, float2& sc_, float2& z_) {
                const int n = (bid + t * kG) * 16 + 2 * pg;
                const bool ok = t < n_head_t && n + 1 < p.V;
                sc_ = ok ? ldpair(p.sz_head + n) : float2{0.f, 0.f};
                z_ = ok ? ldpair(p.sz_head + p.V + n) : float2{0.f, 0.f};
            };
            float2 sct = {0.f, 0.f}, zt = {0.f, 0.f};
            if (gw == 0) head_sz(0, sct, zt);
            gather_x();
            float rinv = 0.f;
            float2 sx = {0.f, 0.f};
            float best = -INFINITY;
            int bi = 0x7fffffff;
            for (int t = 0; t < head_tiles_max; ++t) {
                float2 scn = {0.f, 0.f}, zn = {0.f, 0.f};
                if (gw == 0) head_sz(t + 1, scn, zn);
                __syncthreads();  // Bt
                if (t == 0) {
                    rinv_seen = misc[0];
                    rinv = rinv_seen / x_scale;
                    sx = get_sums();
                }
                if (gw == 0 && t < n_head_t) {
                    const int n = (bid + t * kG) * 16 + 2 * pg;
                    float2 y = deq(tile_pair(0), sct, zt, sx);
                    y.x *= rinv;
                    y.y *= rinv;
                    if (n + 1 < p.V) {  // vocab sizes are even (host check): a pair is inside or outside
                        if (w8 == 0) *(float2*)(p.logits + n) = y;
                        if (y.x > best || (y.x == best && n < bi)) {
                            best = y.x;
                            bi = n;
                        }
                        if (y.y > best || (y.y == best && n + 1 < bi)) {
                            best = y.y;
                            bi = n + 1;
                        }
                    }
                }
                sct = scn;
                zt = zn;
                buf ^= 1;
            }
            if ((p.mode & 1) && gw == 0) {
                // best of this workgroup's rows (the 8 lanes of a pair agree), lowest index on ties
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) {
                    const float ov = __shfl_xor(best, o, 64);
                    const int oi = __shfl_xor(bi, o, 64);
                    if (ov > best || (ov == best && oi < bi)) {
                        best = ov;
                        bi = oi;
                    }
                }
                const unsigned ep = ebase + edge;
                if (lane == 0) {
                    gr_store(p.gm + 2 * bid, ep, __float_as_uint(best));
                    gr_store(p.gm + 2 * bid + 1, ep, (unsigned)bi);
                }
                if (bid == 0) {
                    u32x4 v[4];
                    const bool ok = sweep<4>(p, rs_gm, 0u, 0, 256, ep, v, 0x600u + edge, lane_v);
                    float bv = -INFINITY;
                    int bx = 0x7fffffff;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float cv = __uint_as_float(v[k][0]);
                        const int ci = (int)v[k][2];
                        if (cv > bv || (cv == bv && ci < bx)) {
                            bv = cv;
                            bx = ci;
                        }
                    }
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const float ov = __shfl_xor(bv, o, 64);
                        const int oi = __shfl_xor(bx, o, 64);
                        if (ov > bv || (ov == bv && oi < bx)) {
                            bv = ov;
                            bx = oi;
                        }
                    }
                    if (bx == 0x7fffffff) bx = 0;
                    if (lane == 0 && ok && !aborted(p)) {
                        p.next_token[0] = bx;
                        if (p.out_tokens != nullptr) p.out_tokens[pos + 1] = bx;
                        if (p.mode & 2) {
                            p.tokens[0] = bx;
                            p.pos[0] = pos + 1;
                        }
                    }
                }
            }
            if (bid == 0 && gw == 0 && lane == 0) p.state[1] = step_id + 1u;
        }
    }
    FS_STAMP(1);
}

// ------------------------------------------------------------------------------------------------ host side
extern "C" size_t mi355_fused_step_workspace_bytes(int n_hidden) {
    if (n_hidden <= 0) return 0;
    return kFsWsGh + (size_t)2 * (n_hidden / 2) * 8;
}

extern "C" int mi355_fused_step_supported(int n_embd, int n_head, int hs, int n_hidden, int vocab, int S) {
    if (mi355_num_cus() != kG) return 0;  // one resident workgroup per CU, head groups of 8
    if (n_embd != kC || n_head != kHeads || hs != kHs) return 0;
    if (n_hidden <= 0 || n_hidden % 128 != 0 || n_hidden / 16 > kMaxFcTiles * kG || n_hidden / 128 > 96) return 0;
    if (vocab <= 0 || vocab % 2 != 0 || (vocab + 15) / 16 > kMaxHeadTiles * kG) return 0;
    if (S < 1 || S > kMaxS) return 0;
    return 1;
}

namespace {
// The step needs all 256 workgroups resident at once (they wait for each other): one per CU, which the kernel's LDS
// footprint (159 of 160 KiB) and 640 threads allow exactly when nothing else of this process occupies a CU's LDS or
// wave slots.  A plain launch and a cooperative launch get the same residency; the cooperative one costs 15-19 us of
// host time per launch and only adds the launch-time check of the grid against this very query (MI355X_MICROARCH.md,
// "coop-launch"), so the query is made once here and a grid that does not fit is refused up front.  Kernels of OTHER
// streams that hold CUs while a step starts only delay it: workgroups are admitted as CUs drain, every spin is bounded
// by kSpinLimit sweeps (~1 s), and a step that gives up raises the abort word instead of hanging.
int occupancy_ok() {
    static int ok = -1;
    static std::once_flag once;
    std::call_once(once, [] {
        int per_cu = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)fused_step_kernel, kThreads, kLdsBytes);
        ok = (e == hipSuccess && per_cu >= 1) ? 1 : 0;
    });
    return ok;
}
}  // namespace

extern "C" int mi355_fused_step(const mi355_fused_step_args* a, mi355_stream_t stream) {
    MI355_CHECK_ARG(a != nullptr, MI355_E_ARG, "fused_step: null args");
    MI355_CHECK_ARG(mi355_fused_step_supported(a->n_embd, a->n_head, a->hs, a->n_hidden, a->vocab, a->S), MI355_E_SHAPE,
                    "fused_step: needs %d CUs, n_embd %d, %d heads of %d, n_hidden %% 128 == 0 and <= %d, vocab <= %d, "
                    "S <= %d (got %d CUs, C=%d, heads=%d x %d, H=%d, V=%d, S=%d)",
                    kG, kC, kHeads, kHs, kMaxFcTiles * kG * 16, kMaxHeadTiles * kG * 16, kMaxS, mi355_num_cus(), a->n_embd,
                    a->n_head, a->hs, a->n_hidden, a->vocab, a->S);
    const bool grouped = a->group_cols > 0;
    const int fmt = a->weight_fmt;
    MI355_CHECK_ARG(fmt >= 0 && fmt <= 2, MI355_E_ARG, "fused_step: weight_fmt %d (0 = int4 streams, 1 = BF16, 2 = LLM.int8)", fmt);
    MI355_CHECK_ARG(!(grouped && fmt != 0), MI355_E_ARG, "fused_step: grouped scales exist for int4 streams only");
    MI355_CHECK_ARG(a->w && a->w_head && (grouped || fmt == 1 || (a->sz && a->sz_head)) && a->norms && a->wte && a->rope && a->kv && a->tokens &&
                        a->pos && a->logits && a->workspace,
                    MI355_E_ARG, "fused_step: null pointer");
    int gsh = 0;
    if (grouped) {
        while ((128 << gsh) < a->group_cols) ++gsh;
        MI355_CHECK_ARG((128 << gsh) == a->group_cols && a->n_embd % a->group_cols == 0 && a->n_hidden % a->group_cols == 0 &&
                            a->n_hidden % 256 == 0,
                        MI355_E_SHAPE, "fused_step: group size %d must be 128 * 2^n and divide n_embd and n_hidden", a->group_cols);
        MI355_CHECK_ARG(a->gt && a->gt_head && ((uintptr_t)a->gt | (uintptr_t)a->gt_head | a->gt_layer_stride) % 16 == 0, MI355_E_ARG,
                        "fused_step: grouped scales need the 16-B aligned group tables gt / gt_head");
        // a streamer wave keeps its groups side by side in the 16 MFMA token columns
        MI355_CHECK_ARG(((a->n_hidden / 128 + 7) / 8 >> gsh) + 1 <= 16, MI355_E_SHAPE, "fused_step: too many groups per wave");
    }
    MI355_CHECK_ARG(a->n_layer >= 1 && a->n_layer * 6 + 8 < 1024, MI355_E_SHAPE, "fused_step: n_layer %d", a->n_layer);
    MI355_CHECK_ARG(!(a->mode & 1) || a->next_token != nullptr, MI355_E_ARG, "fused_step: arg-max without next_token");
    MI355_CHECK_ARG(a->mode >= 0 && a->mode <= 3 && a->mode != 2, MI355_E_ARG, "fused_step: mode must be 0, 1 or 3");
    MI355_CHECK_ARG(((uintptr_t)a->w | (uintptr_t)a->w_head | (uintptr_t)a->workspace | a->layer_stride | a->off_attn |
                     a->off_proj | a->off_fc | a->off_mproj) % 16 == 0,
                    MI355_E_ARG, "fused_step: streams and workspace must be 16-B aligned");
    static const bool use_ring = [] {  // default: the register-ring kernel (fused_step_ring.hip); "lds": the kernel above
        const char* e = getenv("MI355_FUSED_IMPL");
        return !(e != nullptr && strcmp(e, "lds") == 0);
    }();
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute((const void*)fused_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    });
    MI355_CHECK_ARG(attr_err == hipSuccess, (int)attr_err, "fused_step: hipFuncSetAttribute failed: %s",
                    hipGetErrorString(attr_err));
    MI355_CHECK_ARG((!grouped && fmt == 0) || use_ring, MI355_E_ARG,
                    "fused_step: grouped scales and BF16 streams are implemented by the register-ring kernel only");
    MI355_CHECK_ARG(use_ring ? (fused_step_ring_occupancy_ok() & (fmt == 2 ? 8 : fmt == 1 ? 4 : grouped ? 2 : 1)) != 0 : occupancy_ok(), MI355_E_STATE,
                    "fused_step: the device does not admit one %d-thread workgroup with %d B of LDS per CU", kThreads, kLdsBytes);
    FusedParams p;
    memset(&p, 0, sizeof(p));
    p.w = (const uint8_t*)a->w;
    p.layer_stride = a->layer_stride;
    p.off_attn = a->off_attn;
    p.off_proj = a->off_proj;
    p.off_fc = a->off_fc;
    p.off_mproj = a->off_mproj;
    p.layer_bytes = a->layer_bytes;
    p.head_bytes = a->head_bytes;
    p.w_head = (const uint8_t*)a->w_head;
    p.sz = (const bf16_t*)a->sz;
    p.sz_head = (const bf16_t*)a->sz_head;
    p.norms = (const bf16_t*)a->norms;
    p.wte = (const bf16_t*)a->wte;
    p.rope = a->rope;
    p.kv = (bf16_t*)a->kv;
    p.tokens = a->tokens;
    p.pos = a->pos;
    p.next_token = a->next_token;
    p.out_tokens = a->out_tokens;
    p.logits = a->logits;
    char* ws = (char*)a->workspace;
    p.state = (unsigned*)(ws + kFsWsState);
    p.gx = (u64*)(ws + kFsWsGx);
    p.ga = (u64*)(ws + kFsWsGa);
    p.gq = (u64*)(ws + kFsWsGq);
    p.gm = (u64*)(ws + kFsWsGm);
    p.gp = (u64*)(ws + kFsWsGp);
    p.gh = (u64*)(ws + kFsWsGh);
    p.dbg = (u64*)a->debug_stamps;
    p.dbg_layer = a->reserved0;  // with debug_stamps: the layer whose phases are stamped
    // elements of `sz` per layer: bf16 scales + zeros of the int4 streams, or (weight_fmt 2) the f32 row scales SCB of the int8 ones
    p.sz_layer_stride = fmt == 2 ? (unsigned)(5 * kC + 2 * a->n_hidden) : (unsigned)(10 * kC + 4 * a->n_hidden);
    p.n_layer = a->n_layer;
    p.H = a->n_hidden;
    p.V = a->vocab;
    p.S = a->S;
    p.units_h = a->n_hidden / 128;
    p.fc_tiles = a->n_hidden / 16;
    p.head_tiles = (a->vocab + 15) / 16;
    p.fmt = fmt;
    {
        const int per_wg = (p.head_tiles + kG - 1) / kG;  // tiles of the busiest workgroup (ring version)
        const int steps = per_wg * 4 * (fmt == 1 ? 4 : fmt == 2 ? 2 : 1);  // ring steps per tile and wave: 4 (int4), 16 (BF16), 8 (int8)
        p.head_turns = (steps + 12 - 1) / 12;
    }
    p.mode = a->mode;
    p.eps = a->eps;
    p.scale = 1.0f / sqrtf((float)kHs);
    if (grouped) {
        p.grouped = 1;
        p.gsh = gsh;
        p.ngc = a->n_embd / a->group_cols;
        p.ngh = a->n_hidden / a->group_cols;
        p.gt = (const uint8_t*)a->gt;
        p.gt_head = (const uint8_t*)a->gt_head;
        p.gt_layer_stride = a->gt_layer_stride;
        const size_t lb = ((size_t)(3 * kC / 16 + kC / 16 + 2 * (a->n_hidden / 16)) * p.ngc + (size_t)(kC / 16) * p.ngh) * 64;
        const size_t hb = (size_t)p.head_tiles * p.ngc * 64;
        MI355_CHECK_ARG(lb < 0x7FFFFFF0ull && hb < 0x7FFFFFF0ull && a->gt_layer_stride >= lb, MI355_E_SHAPE,
                        "fused_step: group tables of %zu / %zu B (layer stride %llu)", lb, hb, (unsigned long long)a->gt_layer_stride);
        p.gt_layer_bytes = (unsigned)lb;
        p.gt_head_bytes = (unsigned)hb;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (t_time_start != nullptr) {  // measurement hook: see gemv.hip launch_gemv_m
        e0 = t_time_start;
        e1 = t_time_stop;
        t_time_start = t_time_stop = nullptr;
    }
    if (use_ring) return fused_step_ring_launch(p, (hipStream_t)stream, e0, e1);
    if (e0 != nullptr) {
        hipExtLaunchKernelGGL(fused_step_kernel, dim3(kG), dim3(kThreads), (uint32_t)kLdsBytes, (hipStream_t)stream, e0, e1, 0,
                              p);
    } else {
        hipLaunchKernelGGL(fused_step_kernel, dim3(kG), dim3(kThreads), kLdsBytes, (hipStream_t)stream, p);
    }
    MI355_LAUNCH_CHECK();
    return 0;
}

// The persistent decode step for the WIDE shapes (round 6): mi355_fused_step with weight_fmt 4 — per-row gptq.int4 models whose heads
// do not map 8 workgroups to a head of 32, i.e. every LLaMA size above 7B of /root/reference lit_llama/model.py:43-48: 13B (40 heads,
// n_embd 5120), 30B (52, 6656) and 65B (64, 8192: BASELINE.json configs[4] at TP = 1).  csrc/fused_step_ring.hip is the same protocol
// written for n_embd 4096 / 32 heads with five operand formats; this file carries ONE format (int4 streams -> fp16 operands, its
// weight_fmt 0) and the shape as template parameters:
//
//   NH = heads of 128 dims, GS = workgroups per head, NH x GS workgroups (one per CU), RT = 8 / GS = 16-row tiles per workgroup
//        GS = 4: 40 / 52 / 64 heads = 160 / 208 / 256 workgroups;  GS = 8: 32 heads (the 7B shape, kept as the cross-check against
//        fused_step_ring.hip: MI355_FUSED_WIDE=1 in lit_llama_amd/engine.py)
//
// Replaces, per generated token, the 161 x (n_layer / 32) operator calls of lit_llama/model.py:76-122 (Block.forward :165-168,
// CausalSelfAttention.forward :194-237, MLP.forward :251-254, RMSNorm :274-277, apply_rope :306-323), the greedy sampling of
// generate.py:68-85 and this repository's launch-per-operator step (80 x 7 launches for 65B).  Measured (round 6, bench.py --model):
// 65B 168-170 tok/s = 0.68-0.69 of the int4-weight roofline against 133-138 = 0.54-0.56 on launches, 30B 265 against 209, 13B 565 against 394.
//
// What is shared with fused_step_ring.hip (read its header first): resident workgroups of 8 streamer waves + gatherer waves; weights in
// a 12-piece register ring per streamer wave (1-KiB non-temporal wave loads); activations between phases as 8-byte {tag, value}
// granules (one sc1 store, swept with sc1 loads until every tag equals the edge's epoch); the residual stream in registers for the
// whole step; head-local q / k / v exchange.  What differs:
//   * a workgroup owns 16 RT residual rows and 16 RT dims of its head: c_attn is 3 RT row tiles against one activation operand
//     (R = 6), attn.c_proj / mlp.c_proj RT tiles; FOUR gatherer waves sweep the (twice as large) edges, the first RT of them run the
//     epilogues of the residual / head tiles (tile r belongs to gatherer r) and publish; the c_fc1 / c_fc2 pair tiles alternate between
//     gatherers 0 and 1;
//   * the attention splits cache ROWS over the GS workgroups of a head (the ring kernel does so from position 384 on): chunk c of 32
//     rows belongs to workgroup c % GS, wave (c / GS) % 8, and a second head-local exchange carries (128 weighted values, max, sum)
//     partials — except up to position 256, where every workgroup of the head covers all rows itself (at most one chunk per wave
//     either way; the head group shares an L2) and publishes its own dimensions without that exchange;
//   * a phase's first ring turn is requested in FRONT of the publish barrier of the phase before it (the ring kernel's rule for its
//     bf16 / int8 streams), partial tiles are parked as column 0 only (LDS: the hidden vector alone is 44 KB), the bodies of the pair phase are
//     unrolled (a runtime loop around loads makes hipcc drain the ring at every back edge).
// Where a 65B layer's 77 us go (profiles/r06_wide_65b_first_timeline.txt): the four phases stream 306 MB in 53 us (5.8 TB/s, 7.0 between a
// phase's B1 and wave 0's last tile), the hand-offs take 23 us (x edges 4.2-5.3, attention output 4.2, hidden 7.8: every CU reads every
// edge) during which only the 96 KiB per CU of a first ring turn are in flight.  More in flight across a hand-off costs the sweeps what it
// fetches (LDS-DMA extension ring: -3 %), so does requesting earlier (profiles/r06_ab_wide_knobs.txt).
// Hand-offs are fp16 pairs: the +-65504 range rule, the clip bookkeeping (state[2], state[3]) and the abort word are the ring kernel's.
#include <math.h>

#include <mutex>
#include <type_traits>

#include <hip/hip_ext.h>

#include "common.h"
#include "fused_step_common.h"

namespace {

constexpr int kG = 256;   // workgroups, at most (= CUs)
constexpr int kSW = 8;    // streamer waves
constexpr int kGW = 4;    // gatherer waves: the first RT of them also run the epilogues of the residual / head tiles (two gatherers, the
                          // ring kernel's count, cost 1.3 % at the 65B width, where every edge is twice as large: profiles/r06_ab_wide_knobs.txt)
constexpr int kThreads = 64 * (kSW + kGW);
constexpr int kRing = 12;  // ring pieces (1 KiB each) per streamer wave
#ifndef MI355_WIDE_WINDOW
#define MI355_WIDE_WINDOW 4
#endif
constexpr int kWin = MI355_WIDE_WINDOW;  // pieces per wave in flight while a first ring turn is requested
// (measured and rejected, profiles/r06_ab_wide_knobs.txt — variants in scripts/patches/r06_*: the second ring turn of a phase parked in LDS
// by LDS-DMA across the hand-off (-3 %: what it fetches it costs the sweeps), attn.c_proj's first turn in front of the attention
// (-0.9 %), three instead of two chunks of the hidden edge in flight (-0.7 %), windows of 3 / 6 pieces (0 / +0.5 %))
constexpr int kHs = 128;
constexpr unsigned kSpinLimit = 400000u;
constexpr int kPartStride = 136;  // granules per workgroup partial of the attention: 128 values, max, sum, pad
constexpr int kSoloPos = 256;     // up to this position every workgroup of a head covers ALL cache rows itself (see the attention phase)
constexpr int kMaxUnits = 176;    // units of 128 columns of the widest activation vector (n_hidden <= 22528)
constexpr int kRMax = 6;          // row tiles of a step (c_attn of the 64-head shape)

// LDS map (bytes)
constexpr int kOffMisc = 0;      // [0] 1/rms, [4..11] operand sums (fp16), [16..23] / [24..31] per-wave softmax max / sum, [32..39] the waves' operand sums (F8)
constexpr int kOffZero = 256;    // one all-zero unit (idle ring steps read it)
constexpr int kOffXs = 512;      // activation vector: fp16 (F8 = false), or three E4M3 limb planes in MFMA-B byte order (F8)
// F8: limb plane p of the vector at kF8P0 + p * kF8Plane, one 128-byte unit per 128 values; the planes start 64 B (16 banks) apart modulo
// 256 so that the lanes of MFMA columns 0 / 1 / 2 of one lane group read different banks (fused_step_ring.hip kF8P0 .. kF8P2)
constexpr int kF8Plane = kMaxUnits * 128 + 64, kF8P0 = kOffXs;
constexpr int kXsBytes = 3 * kF8Plane > kMaxUnits * 256 ? 3 * kF8Plane : kMaxUnits * 256;
static_assert(kF8Plane % 256 == 64 && kXsBytes % 16 == 0, "limb planes");
// [2][8 waves][kRMax][3 columns][4 row groups x f32x4]: the first columns of the waves' partial tiles (fp16 operands: column 0 only — at M = 1
// the others are copies; fp8-limb operands: the limb columns 0 / 1 / 2, added up by the gatherer's read)
constexpr int kPartTile = 192;
constexpr int kOffPart = kOffXs + kXsBytes;
constexpr int kPartBytes = 2 * kSW * kRMax * kPartTile;
constexpr int kOffQ = kOffPart + kPartBytes;                // q[128] knew[128] vnew[128] f32
constexpr int kOffO2 = kOffQ + 3 * 512;                     // [8 waves][128] f32: per-wave attention partials
constexpr int kLdsBytes = kOffO2 + kSW * 128 * 4;
static_assert(kLdsBytes <= 160 * 1024, "LDS map");

// smallest number of ring turns of `steps` steps that holds whole tiles of `spt` steps
constexpr int turns_for(int steps, int spt) {
    int t = 1;
    while ((t * steps) % spt != 0) ++t;
    return t;
}

// NH heads of 128 dims on NH x GS workgroups (one per CU, <= 256): GS = 4 for the 13B / 30B / 65B shapes of lit_llama/model.py:43-48
// (40 / 52 / 64 heads: 160 / 208 / 256 workgroups), GS = 8 for the 7B shape (32 heads).
template <int NH_, int GS>
struct Shape {
    static_assert((GS == 4 && (NH_ == 40 || NH_ == 52 || NH_ == 64)) || (GS == 8 && NH_ == 32), "heads x workgroups per head");
    static constexpr int NH = NH_;           // heads
    static constexpr int NWG = NH * GS;      // workgroups
    static_assert(NWG <= kG, "one workgroup per CU");
    static constexpr int C = NH * kHs;       // n_embd
    static constexpr int UC = C / 128;       // units of a C-wide input (= NH)
    static constexpr int RT = 8 / GS;        // 16-row tiles per workgroup (residual rows / head dims)
    static constexpr int DH = 16 * RT;       // dims of its head a workgroup owns
    static constexpr int NUW = (UC + kSW - 1) / kSW;  // units per streamer wave of a C-wide input, at most (8 / 7 / 5 / 4)
    static constexpr int GXS = C / 2 + 2 * NWG;       // granules per parity of an x edge: pairs, then RT (<= 2) sums of squares per workgroup
    // ring steps of mlp.c_proj per streamer wave, at most (n_hidden <= 22528 / 18432 / 18432 / 11264: the host checks)
    static constexpr int MP_STEPS = NH == 64 ? 22 : NH == 32 ? 11 : 18;
    static constexpr int FC_MAX = GS == 4 ? 6 : 3;    // pair tiles of the busiest workgroup, at most (the host checks)
};

// ------------------------------------------------------------------------------------------------ granules
__device__ __forceinline__ void gr_store(u64* p, unsigned tag, unsigned val) {
    __hip_atomic_store(p, ((u64)tag << 32) | val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // one 8-B sc1 store
}
// fp8-limb operands: a granule is {tag: 16 bits, payload: 48 bits} = limb 0 / 1 / 2 of TWO values (fused_step_ring.hip gr_store16, f8_limbs:
// the format, its 16-bit tags and its arithmetic are that kernel's; pre-scale exponents per edge: x 0, attention output 2, SwiGLU output 4)
__device__ __forceinline__ void gr_store16(u64* p, unsigned tag, unsigned lo32, unsigned hi16) {
    __hip_atomic_store(p, ((u64)(((tag & 0xFFFFu) << 16) | hi16) << 32) | lo32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// x -> three OCP E4M3 limbs, x ~ l0 + l1 / 16 + l2 / 256 (every difference is exact in f32, the conversions round to nearest even; past +-448
// v_cvt_pk_fp8_f32 returns NaN, hence the clamps): 12 significant bits for 2^-6 <= |x| <= 448
__device__ __forceinline__ void f8_limbs(float a, float b, unsigned& lo32, unsigned& hi16) {
    const int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(a, -448.f, 448.f), __builtin_amdgcn_fmed3f(b, -448.f, 448.f), 0, false);
    const auto f0 = __builtin_amdgcn_cvt_pk_f32_fp8(w0, false);
    float ra = a - f0[0], rb = b - f0[1];
    const int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(ra * 16.f, -448.f, 448.f),
                                                   __builtin_amdgcn_fmed3f(rb * 16.f, -448.f, 448.f), 0, false);
    const auto f1 = __builtin_amdgcn_cvt_pk_f32_fp8(w1, false);
    ra -= f1[0] * 0.0625f;
    rb -= f1[1] * 0.0625f;
    const int w2 = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(ra * 256.f, -448.f, 448.f),
                                                   __builtin_amdgcn_fmed3f(rb * 256.f, -448.f, 448.f), 0, false);
    lo32 = ((unsigned)w0 & 0xFFFFu) | ((unsigned)w1 << 16);
    hi16 = (unsigned)w2 & 0xFFFFu;
}
constexpr int kF8Ex = 0, kF8Ea = 2, kF8Eh = 4;
__device__ __forceinline__ bool aborted(const FusedParams& p) {
    return __hip_atomic_load(p.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}
__device__ __forceinline__ void raise_abort(const FusedParams& p, unsigned code) {
    __hip_atomic_store(p.state, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One wave requests 16-B loads (two granules each) number first + k * 64 + lane, k < NL, of the granule buffer behind `rs` (load i
// covers bytes base + 16 i ..); loads at or past `end` go out of the descriptor's range (zeros, no memory request).
template <int NL>
__device__ __forceinline__ void sweep_issue(__amdgpu_buffer_rsrc_t rs, unsigned base, int first, int end, u32x4 (&v)[NL], int lane) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int i = first + k * 64 + lane;
        const unsigned off = i < end ? base + (unsigned)i * 16u : 0xFFFFFFF0u;
        v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));  // sc1
    }
}
// ... and repeats them until every tag equals `epoch`.  Returns false after a time-out / abort (the values are then garbage, the
// caller keeps going so that the barrier counts of the workgroup stay balanced).  `preissued`: v was requested already.
// T16: granules with 16-bit tags in the top half of their second dword (gr_store16); `epoch` is then taken modulo 2^16
template <int NL, bool T16 = false>
__device__ __forceinline__ bool sweep(const FusedParams& p, __amdgpu_buffer_rsrc_t rs, unsigned base, int first, int end, unsigned epoch,
                                      u32x4 (&v)[NL], unsigned code, int lane, bool preissued = false) {
    if constexpr (T16) epoch &= 0xFFFFu;
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
        if (!(preissued && spins == 0)) sweep_issue<NL>(rs, base, first, end, v, lane);
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = first + k * 64 + lane;
            if constexpr (T16) ok &= i >= end || ((v[k][1] >> 16) == epoch && (v[k][3] >> 16) == epoch);
            else ok &= i >= end || (v[k][1] == epoch && v[k][3] == epoch);
        }
        if (__all(ok)) return true;
        if (spins > kSpinLimit || aborted(p)) {
            if (lane == 0) raise_abort(p, code);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// ------------------------------------------------------------------------------------------------ streamers
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(8))) int i32x8;

struct PhaseW {  // one phase as a streamer wave sees it (all wave-uniform)
    unsigned base;  // byte offset of the stream inside the layer's descriptor
    int tile0, units, u0, nu, ntiles;
    int tstride;    // M_PAIR / M_SINGLE: tile of the phase's ti-th tile = tile0 + ti * tstride (the number of workgroups)
};
// how the R pieces of a ring step relate: M_SHARED — R row tiles against one activation unit, ONE virtual tile per phase (tile of
// piece r = tile0 + (r / RTK) * kstride + r % RTK: the q / k / v thirds of c_attn, or the RT residual tiles of a projection);
// M_PAIR — the c_fc1 / c_fc2 rows of pair tile tile0 + ti * 256 (stream [tile][unit][2][lane]); M_SINGLE — R = 1, tile tile0 + ti * 256
enum { M_SHARED = 0, M_PAIR = 1, M_SINGLE = 2 };

// Scalar byte offset of piece r of step (tile index ti, step st inside the tile); ok = false for an idle piece (padding of the ring
// turn: the load then goes through a zero-sized descriptor = zeros, no memory request).
template <int MODE, int RTK>
__device__ __forceinline__ unsigned piece_off(const PhaseW& ph, int ti, int st, int r, int kstride, bool& ok) {
    if constexpr (MODE == M_SHARED) {
        const int tile = ph.tile0 + (r / RTK) * kstride + (r % RTK);
        ok = ti == 0 && st < ph.nu;
        return ph.base + (unsigned)(tile * ph.units + ph.u0 + st) * 1024u;
    } else if constexpr (MODE == M_PAIR) {
        const int tile = ph.tile0 + ti * ph.tstride;
        ok = ti < ph.ntiles && st < ph.nu;
        return ph.base + (unsigned)((tile * ph.units + ph.u0 + st) * 2 + r) * 1024u;
    } else {
        const int tile = ph.tile0 + ti * ph.tstride;
        ok = ti < ph.ntiles && st < ph.nu;
        return ph.base + (unsigned)(tile * ph.units + ph.u0 + st) * 1024u;
    }
}
__device__ __forceinline__ u32x4 ring_load(__amdgpu_buffer_rsrc_t rs, __amdgpu_buffer_rsrc_t rs_null, bool ok, unsigned lane_off,
                                           unsigned soff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ok ? rs : rs_null, lane_off, ok ? soff : 0u, 2));
}
// int4 -> centred fp16 operands: fused_step_ring.hip nib2f16 / nib_center (5 + 4 VALU per 8 weights; q - 8 and 16 (q - 8), exact)
__device__ __forceinline__ uint32_t nib2f16(uint32_t x, uint32_t mask_s, uint32_t magic_v) { return (x & mask_s) | magic_v; }
__device__ __forceinline__ uint32_t nib_center(uint32_t pair, f16x2 c) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, pair) - c);
}

struct StreamerCtx {  // per-wave constants of the streamers
    __amdgpu_buffer_rsrc_t rs_null;
    unsigned lane_off;
    int g, wave;
    uint32_t magic, nmask, nmask16;
    char* smem;
    // fp8-limb operands: nibble mask, this lane's limb plane (+ its lane group's 32 bytes of a unit), the step of its block scale
    uint32_t nib8;
    unsigned f8_plane;
    int f8_dsb;
};

// A workgroup barrier for the streamer waves: they publish nothing through global memory, what the barrier has to order is their LDS
// traffic — no fence, no wait for the ring's loads in flight.
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Pieces P0 .. P1 - 1 of a phase's FIRST ring turn (12 pieces), consumption order = issue order (VMEM returns in order).  WAIT: at most
// kWin pieces per wave in flight while they are requested (a deeper queue only stands in front of the gatherers' sweep in the CU's
// in-order memory pipeline).
template <int R, int SPT, int MODE, int RTK, int P0, int P1, bool WAIT>
__device__ __forceinline__ void burst(u32x4 (&ring)[kRing], const PhaseW& ph, int kstride, __amdgpu_buffer_rsrc_t rs, const StreamerCtx& c) {
    constexpr int STEPS = kRing / R;
    static_assert(STEPS * R == kRing, "ring turn");
#pragma unroll
    for (int pc = P0; pc < P1; ++pc) {
        const int step = pc / R, r = pc % R;
        bool ok;
        const unsigned so = piece_off<MODE, RTK>(ph, step / SPT, step % SPT, r, kstride, ok);
        ring[pc] = ring_load(rs, c.rs_null, ok, c.lane_off, so);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (WAIT) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWin - 1) : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// One phase: `nbodies` bodies of TURNS ring turns of 12 / R steps; a step = R pieces against one activation unit; SPT steps per
// tile (TURNS * STEPS is a multiple of SPT: bodies hold whole tiles).  Behind every tile the wave parks column 0 of its R partial
// tiles in LDS and passes the workgroup barrier Bt.
// NB: bodies of the phase when known at compile time (the loop is then unrolled: a runtime loop around loads makes hipcc drain the whole
// ring with vmcnt(0) at every back edge), 0: `nbodies` at run time (lm_head: once per step).
// F8: the int4 stream through fp8 operands (fused_step_ring.hip FMT 3): a byte holding an int4 level IS the E4M3 code of q x 2^-9, ONE
// v_mfma_scale_f32_16x16x128_f8f6f4 per 1-KiB piece (A block scale 2^9), the activation's three limbs in MFMA columns 0 / 1 / 2 under the
// per-lane B block scales 2^(E8 - 0 / 4 / 8), E8 = the pre-scale exponent of the phase's input edge; during a phase's first tile one more
// MFMA per step with an all-ones A operand takes the operand sums (misc[32 + wave]).
template <int R, int SPT, int TURNS, int MODE, int RTK, bool F8, int NB = 1>
__device__ __forceinline__ void run_phase(u32x4 (&ring)[kRing], const PhaseW& ph, int kstride, int nbodies, __amdgpu_buffer_rsrc_t rs,
                                          const StreamerCtx& c, int& buf, u64* stamp, int slot = 0, int e8 = 0) {
    constexpr int STEPS = kRing / R, BSTEPS = TURNS * STEPS, TPB = BSTEPS / SPT;
    static_assert(BSTEPS % SPT == 0 && R <= kRMax, "bodies hold whole tiles");
    constexpr int NACC = R >= 3 ? 1 : 2;  // accumulators per row group: consecutive MFMAs never share one
    const f16x2 zc1 = {(_Float16)1032.0f, (_Float16)1032.0f}, zc16 = {(_Float16)1152.0f, (_Float16)1152.0f};
    if constexpr (NB != 0) nbodies = NB;
    const int total = nbodies * BSTEPS;
    f32x4 acc[R][NACC];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[r][a] = f32x4{0.f, 0.f, 0.f, 0.f};
    [[maybe_unused]] f32x4 accs = f32x4{0.f, 0.f, 0.f, 0.f};  // F8: all-ones rows = the operand sums of this wave's units, per limb column
    [[maybe_unused]] i32x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = 0x38383838;  // E4M3 1.0
    [[maybe_unused]] const int sb = 127 + e8 - c.f8_dsb;  // E8M0 block scale of this lane's 32 operand bytes
    wg_barrier();  // B1: the activation vector is staged
    if (stamp != nullptr && threadIdx.x == 0) stamp[slot] = wall_clock64();
    const char* xs = c.smem + kOffXs;
    [[maybe_unused]] const char* xl = c.smem + c.f8_plane;
    // B operands (activation unit of a step) are read one step ahead
    [[maybe_unused]] f16x8 bn[4];
    [[maybe_unused]] i32x8 bn8;
    if constexpr (F8) {
        bn8 = *(const i32x8*)(xl + ph.u0 * 128);
    } else {
        const char* xb0 = xs + ph.u0 * 256 + c.g * 64;
#pragma unroll
        for (int d = 0; d < 4; ++d) bn[d] = *(const f16x8*)(xb0 + 16 * d);
    }
    constexpr int kUnrollBodies = NB != 0 ? NB : 1;
#pragma unroll kUnrollBodies
    for (int body = 0; body < (NB != 0 ? NB : nbodies); ++body) {
#pragma unroll
        for (int t = 0; t < TURNS; ++t) {
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                const int ls = t * STEPS + s;              // step inside the body (compile time)
                const int ti = body * TPB + ls / SPT, st = ls % SPT;
                [[maybe_unused]] f16x8 b[4];
                [[maybe_unused]] i32x8 b8;
                {
                    const int nst = (st + 1 == SPT) ? 0 : st + 1;
                    const int nun = ph.u0 + (nst < ph.nu ? nst : 0);
                    if constexpr (F8) {
                        b8 = bn8;
                        bn8 = *(const i32x8*)(xl + nun * 128);
                    } else {
#pragma unroll
                        for (int d = 0; d < 4; ++d) b[d] = bn[d];
                        const char* xbn = xs + nun * 256 + c.g * 64;
#pragma unroll
                        for (int d = 0; d < 4; ++d) bn[d] = *(const f16x8*)(xbn + 16 * d);
                    }
                }
                // idle steps (padding of the ring turn) carry no data: skip their MFMAs (wave-uniform)
                if constexpr (F8) {
                    if (st < ph.nu && (MODE == M_SHARED || ti < ph.ntiles)) {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const u32x4 v = ring[s * R + r];
                            i32x8 a;
#pragma unroll
                            for (int d = 0; d < 4; ++d) {
                                a[2 * d] = (int)(v[d] & c.nib8);
                                a[2 * d + 1] = (int)((v[d] >> 4) & c.nib8);
                            }
                            acc[r][s % NACC] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b8, acc[r][s % NACC], 0, 0, 0, 136, 0, sb);
                            // (pinned here: the partial tile's only reader is the `if (colp)` store at the tile end, and hipcc otherwise SINKS
                            // the phase's whole MFMA chain into that divergent branch, behind all of the phase's loads — the ring in scratch)
                            asm volatile("" : "+v"(acc[r][s % NACC]));
                        }
                        if (ti == 0) {
                            accs = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, b8, accs, 0, 0, 0, 127, 0, sb);
                            asm volatile("" : "+v"(accs));
                        }
                    }
                } else if (st < ph.nu && (MODE == M_SHARED || ti < ph.ntiles)) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const uint32_t v = ring[s * R + r][d];
                            const uint32_t v8 = v >> 8;
                            u32x4 a;
                            a[0] = nib_center(nib2f16(v, c.nmask, c.magic), zc1);
                            a[1] = nib_center(nib2f16(v, c.nmask16, c.magic), zc16);
                            a[2] = nib_center(nib2f16(v8, c.nmask, c.magic), zc1);
                            a[3] = nib_center(nib2f16(v8, c.nmask16, c.magic), zc16);
                            acc[r][d % NACC] =
                                __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), b[d], acc[r][d % NACC], 0, 0, 0);
                        }
                    }
                }
                // refill with the same slots of the next turn of THIS phase (nothing past its end)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int nls = ls + STEPS;  // (may run into the next body)
                    const int nti = body * TPB + nls / SPT, nst2 = nls % SPT;
                    bool ok;
                    const unsigned so = piece_off<MODE, RTK>(ph, nti, nst2, r, kstride, ok);
                    ring[s * R + r] = ring_load(rs, c.rs_null, ok && body * BSTEPS + nls < total, c.lane_off, so);
                }
                if ((ls + 1) % SPT == 0) {
                    if (stamp != nullptr && body * BSTEPS + ls + 1 == total) {
                        if (threadIdx.x == 0) stamp[slot + 1] = wall_clock64();
                        // (slots 48 / 51 / 52 / 53: when the LAST streamer wave reaches the phase's last tile end — wave skew)
                        if (c.lane_off == 0u) atomicMax((unsigned long long*)&stamp[48 + (slot - 20) / 2], (unsigned long long)wall_clock64());
                    }
                    if constexpr (F8) {
                        if (ti == 0) {
                            // S of this wave's units: limb columns 0 + 1 + 2 of any row (quad broadcasts of lanes 1 / 2)
                            float ssum = accs[0];
                            ssum += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, accs[0]), 0x55, 0xF, 0xF, false)) +
                                    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, accs[0]), 0xAA, 0xF, 0xF, false));
                            if (c.lane_off == 0u) ((float*)(c.smem + kOffMisc))[32 + c.wave] = ssum;
                        }
                    }
                    // tile done: the first column(s) of this wave's partial 16 x 16 tiles (lane 16 g + n holds rows 4 g .. 4 g + 3 of column n)
                    const unsigned col = (c.lane_off >> 4) & 15u;
                    f32x4* pp = (f32x4*)(c.smem + kOffPart + (size_t)((buf * kSW + c.wave) * kRMax) * kPartTile + col * 64) + (c.lane_off >> 8);
                    const bool colp = col < (F8 ? 3u : 1u);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        f32x4 t4 = acc[r][0];
                        if constexpr (NACC == 2) t4 += acc[r][1];
                        if (colp) pp[r * (kPartTile / 16)] = t4;
#pragma unroll
                        for (int a = 0; a < NACC; ++a) acc[r][a] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    wg_barrier();  // Bt
                    buf ^= 1;
                }
                // keep a step's conversions next to its MFMAs (hipcc otherwise hoists them to the top of the turn and spills)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

#define FW_STAMP(i)                                                                                    \
    do {                                                                                               \
        if (p.dbg != nullptr && (threadIdx.x & 63) == 0) p.dbg[bid * 64 + (i)] = wall_clock64();        \
    } while (0)

}  // namespace

// F8: activations travel as three E4M3 limbs and the int4 levels go to the matrix pipe as E4M3 codes (weight_fmt 5; the ring kernel's FMT 3,
// its default for per-row int4); else fp16 pairs and int4 -> fp16 operands (weight_fmt 4; the rung below, fp16's range)
template <int NH_, int GS, bool F8>
__global__ __launch_bounds__(kThreads) void fused_step_wide_kernel(const FusedParams p) {
    using SH = Shape<NH_, GS>;
    constexpr int C = SH::C, UC = SH::UC, RT = SH::RT, DH = SH::DH, NH = SH::NH, NUW = SH::NUW, NWG = SH::NWG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* misc = (float*)(smem + kOffMisc);
    char* xs = smem + kOffXs;
    float* qs = (float*)(smem + kOffQ);
    float* knew = qs + kHs;
    float* vnew = knew + kHs;
    float* op2 = (float*)(smem + kOffO2);

    // workgroup -> head group: the GS workgroups of a head sit on one XCD (blocks are dealt round-robin to the 8 XCDs; a speed
    // matter only — the protocol does not depend on placement)
    const int xcd = bid & 7, slot = bid >> 3;
    const int head = NH % 8 == 0 ? xcd * (NH / 8) + slot / GS : bid / GS;  // (52 heads: a head group spans XCDs)
    const int hj = NH % 8 == 0 ? slot % GS : bid % GS;                     // which DH dimensions of the head

    const int pos = p.pos[0];
    const int token = p.tokens[0];
    const unsigned step_id = p.state[1];
    const unsigned ebase = step_id * 1024u + 1u;
    const int n_fc = (p.fc_tiles - bid + NWG - 1) / NWG;        // this workgroup's pair tiles
    const int n_head_t = (p.head_tiles - bid + NWG - 1) / NWG;  // lm_head tiles
    int solo_i = pos <= kSoloPos ? 1 : 0;  // short context: no row split, no second head-local exchange (attention phase)

    // entered outside the cache (the host takes the cache-roll regime of model.py:214-218 elsewhere) or with a token id outside
    // the embedding table: refuse before anything is written.  Uniform over the grid, so no hand-off hangs.
    if (pos < 0 || pos >= p.S || token < 0 || token >= p.V) {
        if (bid == 0 && threadIdx.x == 0) raise_abort(p, 0x10u);
        return;
    }
    if (threadIdx.x < 64) ((unsigned*)(smem + kOffZero))[threadIdx.x] = 0u;
    FW_STAMP(0);

    // phase geometry (compile time): steps per tile / ring turns per body
    constexpr int SPT_C = NUW;                                    // tiles over a C-wide input: NUW steps per wave (idle ones where a wave has fewer)
    constexpr int R_ATT = 3 * RT, ST_ATT = kRing / R_ATT;         // c_attn: q / k / v x RT tiles share the operand
    constexpr int TU_ATT = (NUW + ST_ATT - 1) / ST_ATT;
    constexpr int R_PRJ = RT, ST_PRJ = kRing / R_PRJ;             // attn.c_proj, mlp.c_proj: RT residual tiles share the operand
    constexpr int TU_PRJ = (NUW + ST_PRJ - 1) / ST_PRJ;
    constexpr int TU_MP = (SH::MP_STEPS + ST_PRJ - 1) / ST_PRJ;   // mlp.c_proj: up to TU_MP * ST_PRJ units of the hidden vector per wave
    constexpr int TU_FC = turns_for(6, SPT_C), TPB_FC = TU_FC * 6 / SPT_C;      // pair tiles: 6 steps per turn, whole tiles per body
    constexpr int NB_FC = (SH::FC_MAX + TPB_FC - 1) / TPB_FC;                   // bodies of the pair phase (unrolled)
    constexpr int TU_HD = turns_for(12, SPT_C), TPB_HD = TU_HD * 12 / SPT_C;    // lm_head: 12 steps per turn

    if (wave < kSW) {
        // =========================================================================================== streamers
        StreamerCtx c;
        c.lane_off = lane * 16;
        c.g = lane >> 4;
        c.wave = wave;
        c.smem = smem;
        c.magic = 0x64006400u;
        c.nmask = 0x000F000Fu;
        c.nmask16 = 0x00F000F0u;
        asm volatile("" : "+v"(c.magic));  // opaque register values (fused_step_ring.hip nib2f16: hipcc then selects v_and_or_b32)
        asm volatile("" : "+s"(c.nmask));
        asm volatile("" : "+s"(c.nmask16));
        c.nib8 = 0x0F0F0F0Fu;
        if constexpr (F8) asm volatile("" : "+s"(c.nib8));  // (opaque: hipcc then keeps the mask in an SGPR operand)
        {
            const int f8_col = lane & 15;  // MFMA token column of this lane: limb plane 0 / 1 / 2 (columns >= 2 read plane 2: unused copies)
            c.f8_plane = (unsigned)(kF8P0 + (f8_col == 0 ? 0 : f8_col == 1 ? 1 : 2) * kF8Plane) + (unsigned)c.g * 32u;
            c.f8_dsb = f8_col == 0 ? 0 : f8_col == 1 ? 4 : 8;
        }
        u32x4 ring[kRing];
        int buf = 0;

        PhaseW ph_attn, ph_proj, ph_fc, ph_mp, ph_head;
        // this wave's units of a C-wide input (52 heads: 6 or 7) and of the hidden vector
        const int cu0 = wave * (UC / kSW) + (wave < UC % kSW ? wave : UC % kSW), cnu = UC / kSW + (wave < UC % kSW ? 1 : 0);
        ph_attn = {p.off_attn, head * 8 + hj * RT, UC, cu0, cnu, 1, 0};
        ph_proj = {p.off_proj, bid * RT, UC, cu0, cnu, 1, 0};
        ph_fc = {p.off_fc, bid, UC, cu0, cnu, n_fc, NWG};
        {
            const int uq = p.units_h / kSW, ur = p.units_h % kSW;
            ph_mp = {p.off_mproj, bid * RT, p.units_h, wave * uq + (wave < ur ? wave : ur), uq + (wave < ur ? 1 : 0), 1, 0};
        }
        ph_head = {0u, bid, UC, cu0, cnu, n_head_t, NWG};

        __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.layer_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_head, 0, (int)p.head_bytes, 0x00020000);
        c.rs_null = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0, 0x00020000);

        bool dbg_on = false;
#define FW_SSTAMP(i)                                                                  \
    do {                                                                              \
        if (dbg_on && threadIdx.x == 0) p.dbg[bid * 64 + (i)] = wall_clock64();        \
    } while (0)
        // A phase's first ring turn is requested in FRONT of the publish barrier B3 of the phase before it, windowed (kWin pieces per wave in
        // flight): the stream runs through the epilogue, and when the gatherers sweep the edge most of the turn has landed.  Measured
        // against it at the 65B width (profiles/r06_ab_wide_knobs.txt): kWin pieces in front of B3 and the rest behind it -1.6 %; no burst
        // at all but the turn CHAINED to the last turn of the phase before (a consumed slot refilled with the next phase's piece: up to
        // 96 KiB per CU in front of the publish store) -1.2 %; the same two on the ring kernel's bf16 / int8 streams -4 % / -7 %.
#define FW_EDGE(R_, SPT_, MODE_, RTK_, PH_, KS_, RS_)                                                  \
    do {                                                                                              \
        burst<R_, SPT_, MODE_, RTK_, 0, kRing, true>(ring, PH_, KS_, RS_, c);                          \
        wg_barrier(); /* B3 */                                                                        \
    } while (0)
        burst<R_ATT, TU_ATT * ST_ATT, M_SHARED, RT, 0, kRing, true>(ring, ph_attn, C / 16, rs_l, c);
        const bf16_t* kv_l = (const bf16_t*)p.kv;
        for (int l = 0; l < p.n_layer; ++l) {
            dbg_on = p.dbg != nullptr && l == p.dbg_layer;
            asm volatile("" : "+v"(c.lane_off));  // per-lane addresses are recomputed per layer, not hoisted and spilled
            // ---------------- c_attn (q, k, v tiles of this workgroup's DH dimensions of its head)
            run_phase<R_ATT, TU_ATT * ST_ATT, TU_ATT, M_SHARED, RT, F8>(ring, ph_attn, C / 16, 1, rs_l, c, buf, dbg_on ? p.dbg + bid * 64 : nullptr, 20, kF8Ex);
            wg_barrier();  // B3
            // ---------------- attention: this workgroup's chunks of 32 cache rows, all 128 dimensions
            {
                const bf16_t* kc = kv_l + (size_t)head * p.S * kHs;
                const bf16_t* vc = kc + (size_t)NH * p.S * kHs;
                const int li = (c.lane_off >> 4) & 15, lr = c.lane_off >> 8;
                const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)kc, 0, pos * (kHs * 2), 0x00020000);
                const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)vc, 0, pos * (kHs * 2), 0x00020000);
                int n_chunks = (pos + 31) >> 5;
                n_chunks = n_chunks < 1 ? 1 : n_chunks;  // (position 0: chunk 0 carries the new token's own row only)
                // Up to position 256 (eight chunks: at most ONE per wave either way) every workgroup of the head takes ALL rows — chunk c belongs
                // to wave c % 8 of each of them — and so holds the whole head's output: its gatherer publishes its own DH dimensions at once,
                // the exchange of partials (one hand-off, ~1.5 us per layer) is not needed.  The head's GS workgroups sit on one XCD: the rows
                // are read from HBM once and from its L2 GS - 1 times.  Beyond, the rows are split: chunk c -> workgroup c % GS, wave (c / GS) % 8.
                // (written with the loop's compile-time stride: the loop variable counts chunks x GS in that mode — a run-time stride cost 30 VGPRs
                // and scratch)
                const int csh = solo_i ? (GS == 4 ? 2 : 3) : 0, hsel = solo_i ? 0 : hj;
                const int c0 = wave * GS + hsel;          // loop variable: chunk << csh
                const int n_loop = n_chunks << csh;
                u32x4 kr[8], vr[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = (c0 >> csh) * 32 + u * 4 + lr;
                    const unsigned off = t < pos ? (unsigned)t * 256u + li * 16u : 0xFFFFFFF0u;
                    kr[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, off, 0, 0));
                    vr[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, off, 0, 0));
                }
                wg_barrier();  // Ba1: q / new k / new v of the head are in LDS
                FW_SSTAMP(23);
                float qf[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[j] = qs[li * 8 + j];
                float m_run = -1.0e30f, l_run = 0.f;  // l_run: over THIS lane group's rows (u, lr); summed over lr below
                float of[8];                           // dims li * 8 .. + 7, over this lane group's rows
#pragma unroll
                for (int j = 0; j < 8; ++j) of[j] = 0.f;
                for (int chl = c0; chl < n_loop; chl += kSW * GS) {
                    const int ch = chl >> csh;  // the chunk
                    if (chl != c0) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int t = ch * 32 + u * 4 + lr;
                            const unsigned off = t < pos ? (unsigned)t * 256u + li * 16u : 0xFFFFFFF0u;
                            kr[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, off, 0, 0));
                            vr[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, off, 0, 0));
                        }
                    }
                    float sc[8];
                    float bm = -1.0e30f;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        float dot = 0.f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            dot += qf[2 * i] * __uint_as_float(kr[u][i] << 16);
                            dot += qf[2 * i + 1] * __uint_as_float(kr[u][i] & 0xffff0000u);
                        }
                        dot = group_sum(dot, 16) * p.scale;  // every lane of the row's 16 holds the score
                        sc[u] = ch * 32 + u * 4 + lr < pos ? dot : -1.0e30f;
                        bm = fmaxf(bm, sc[u]);
                    }
                    bm = fmaxf(bm, lane_xor16(bm));  // over the 4 row groups lr: the maximum of the wave's 32 rows
                    bm = fmaxf(bm, lane_xor32(bm));
                    float s_new = -1.0e30f;
                    const bool own = ch == 0 && wave == 0 && hsel == 0;  // the new token's own row rides with chunk 0
                    if (own) {
                        float dot = qs[lane] * knew[lane] + qs[lane + 64] * knew[lane + 64];
                        s_new = group_sum(dot, 64) * p.scale;
                        bm = fmaxf(bm, s_new);
                    }
                    const float m_new = fmaxf(m_run, bm);
                    const float corr = __expf(m_run - m_new);
                    l_run *= corr;
#pragma unroll
                    for (int j = 0; j < 8; ++j) of[j] *= corr;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float pr = ch * 32 + u * 4 + lr < pos ? __expf(sc[u] - m_new) : 0.f;
                        l_run += pr;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            of[2 * i] += pr * __uint_as_float(vr[u][i] << 16);
                            of[2 * i + 1] += pr * __uint_as_float(vr[u][i] & 0xffff0000u);
                        }
                    }
                    if (own && lr == 0) {  // (one of the four row groups: they are summed below)
                        const float pn = __expf(s_new - m_new);
                        l_run += pn;
#pragma unroll
                        for (int j = 0; j < 8; ++j) of[j] += pn * vnew[li * 8 + j];
                    }
                    m_run = m_new;
                }
                // sum over the four row groups (lanes that differ in lr hold different rows of the same dimensions)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    of[j] += lane_xor16(of[j]);
                    of[j] += lane_xor32(of[j]);
                }
                l_run += lane_xor16(l_run);
                l_run += lane_xor32(l_run);
                if (lr == 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) op2[wave * 128 + li * 8 + j] = of[j];
                }
                if ((threadIdx.x & 63) == 0) {
                    misc[16 + wave] = m_run;
                    misc[24 + wave] = l_run;
                }
                FW_SSTAMP(25);
                wg_barrier();  // Ba3: partial outputs of the 8 waves
                wg_barrier();  // Ba4: the attention output is published
            }
            // ---------------- attn.c_proj, MLP
            burst<R_PRJ, TU_PRJ * ST_PRJ, M_SHARED, RT, 0, kRing, true>(ring, ph_proj, 0, rs_l, c);
            run_phase<R_PRJ, TU_PRJ * ST_PRJ, TU_PRJ, M_SHARED, RT, F8>(ring, ph_proj, 0, 1, rs_l, c, buf, dbg_on ? p.dbg + bid * 64 : nullptr, 26, kF8Ea);
            FW_EDGE(2, SPT_C, M_PAIR, 1, ph_fc, 0, rs_l);
            run_phase<2, SPT_C, TU_FC, M_PAIR, 1, F8, NB_FC>(ring, ph_fc, 0, NB_FC, rs_l, c, buf, dbg_on ? p.dbg + bid * 64 : nullptr, 28, kF8Ex);
            FW_EDGE(R_PRJ, TU_MP * ST_PRJ, M_SHARED, RT, ph_mp, 0, rs_l);
            run_phase<R_PRJ, TU_MP * ST_PRJ, TU_MP, M_SHARED, RT, F8>(ring, ph_mp, 0, 1, rs_l, c, buf, dbg_on ? p.dbg + bid * 64 : nullptr, 30, kF8Eh);
            // next layer (or the head)
            kv_l += (size_t)2 * NH * p.S * kHs;
            if (l + 1 < p.n_layer) {
                rs_l = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (size_t)(l + 1) * p.layer_stride), 0, (int)p.layer_bytes, 0x00020000);
                FW_EDGE(R_ATT, TU_ATT * ST_ATT, M_SHARED, RT, ph_attn, C / 16, rs_l);
            } else {
                FW_EDGE(1, SPT_C, M_SINGLE, 1, ph_head, 0, rs_h);
            }
        }
        dbg_on = false;
        run_phase<1, SPT_C, TU_HD, M_SINGLE, 1, F8, 0>(ring, ph_head, 0, p.head_turns, rs_h, c, buf, nullptr, 0, kF8Ex);
        wg_barrier();  // B3
        if (p.mode & 1) wg_barrier();  // the arg-max exchange of the gatherers
#undef FW_EDGE
#undef FW_SSTAMP
    } else {
        // =========================================================================================== gatherers
        const int gw = wave - kSW;
        // with RT = 2 both gatherer waves own a residual / head tile (tile `er`) and run its epilogues; with RT = 1 gatherer 0 does
        const bool epi = gw < RT;
        const int er = gw < RT ? gw : 0;
        unsigned edge = 0;  // edges published so far in this step (the epoch of the next one is ebase + edge)
        int xpar = 0, apar = 0, hpar = 0, qpar = 0, ppar = 0;
        int buf = 0;
        // ONE descriptor over the hand-off area of the workspace and compile-time byte offsets of the buffers inside it
        const __amdgpu_buffer_rsrc_t rs_ws =
            __builtin_amdgcn_make_buffer_rsrc((void*)p.gx, 0, (int)(kFwGh - kFwGx) + 2 * (p.H / 2) * 8, 0x00020000);
        constexpr unsigned kOGa = (unsigned)(kFwGa - kFwGx), kOGq = (unsigned)(kFwGq - kFwGx), kOGm = (unsigned)(kFwGm - kFwGx),
                           kOGp = (unsigned)(kFwGp - kFwGx), kOGh = (unsigned)(kFwGh - kFwGx);
        const int gh_stride = p.H / 2;  // granules per parity of the hidden edge

        // ---- epilogue mapping: lane = (pair pg = lane >> 3, streamer wave w8 = lane & 7).  A lane reads rows 2 pg, 2 pg + 1 of ONE
        // wave's partial tile column (8 B), the 8 lanes of a pair are summed with DPP (fixed order), and every lane then holds both
        // outputs of its pair: RoPE pairs, pair granules and the residual rows stay in registers.
        int lane_v = lane;  // made opaque once per layer: per-lane pointers are otherwise hoisted out of the layer loop and spilled
        int pg = lane >> 3, w8 = lane & 7;
        int psrc = (pg >> 1) * 4 + ((2 * pg) & 3);  // float index of D[2 pg][0] in a parked tile column
        auto tile_pair = [&](int r) {
            const float* tp = (const float*)(smem + kOffPart + (size_t)((buf * kSW + w8) * kRMax + r) * kPartTile) + psrc;
            float2 t = *(const float2*)tp;
            if constexpr (F8) {  // limb columns 1 / 2 of the same rows sit 16 / 32 floats on
                const float2 t1 = *(const float2*)(tp + 16), t2 = *(const float2*)(tp + 32);
                t.x += t1.x + t2.x;
                t.y += t1.y + t2.y;
            }
            t.x = group_sum(t.x, 8);
            t.y = group_sum(t.y, 8);
            return t;
        };
        auto ldpair = [&](const bf16_t* q) {  // two consecutive bf16 (4-byte aligned) as floats
            const unsigned v = *(const unsigned*)q;
            return float2{__uint_as_float(v << 16), __uint_as_float(v & 0xffff0000u)};
        };
        // (the epilogues' arithmetic on the ring kernel's instruction diet: hardware bf16 pairs, SiLU and the softmax normalisation through
        // v_exp_f32 / v_rcp_f32, v_rsq_f32 without the denormal guard — profiles/r06_ab1_epilogue_diet.txt)
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        auto bfpair = [&](float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t)); };
        auto swiglu_e = [&](float a, float b) { return a * __builtin_amdgcn_rcpf(1.0f + __expf(-a)) * b; };
        // state[2] counts the clipped pairs, state[3] keeps 0x7FFFFFFF - (the LOWEST position whose step clipped): the host replays
        // from there on the launch-per-operator step (DecodeEngine.check_status)
        auto note_clip = [&]() {
            atomicAdd(p.state + 2, 1u);
            atomicMax(p.state + 3, 0x7FFFFFFFu - (unsigned)pos);
        };
        // activation pair granule: fp16 (a, b); ODD pairs of a vector carry a / 16, b / 16 (nib2f16).  `odd`: parity of the pair's
        // index in its vector.  The conversion saturates and every clip is counted.
        auto hpair = [&](float a, float b, bool odd) {
            const float k = odd ? 0.0625f : 1.0f;
            const float ak = a * k, bk = b * k;
            if (fmaxf(fabsf(ak), fabsf(bk)) > 65504.f) note_clip();
            const f16x2 h = {(_Float16)__builtin_amdgcn_fmed3f(ak, -65504.f, 65504.f), (_Float16)__builtin_amdgcn_fmed3f(bk, -65504.f, 65504.f)};
            return __builtin_bit_cast(unsigned, h);
        };
        // Publish the pair (a, b) = rows 2 q, 2 q + 1 of a 16-row tile whose 8 granules start at `tile` (q = 0 .. 7), epoch `ep`; every lane of the
        // wave must call it, `store`: this lane is the one that writes.  fp16: one {tag, pair} granule at slot q.  fp8 limbs (`e8`: the edge's
        // pre-scale exponent): a granule carries the values at offsets (j, j + 4) of an octet of rows, so the lanes of pairs q and q ^ 2 — XD lanes
        // apart — exchange one value: the lower one publishes (j, j + 4) at slot 4 (octet) + j, the upper one (j + 1, j + 5) (fused_step_ring.hip
        // f8_publish); values past +-448 x 2^e8 are clipped and counted.
        auto xchg = [&](float v, auto xd) {
            constexpr int XD = decltype(xd)::value;
            if constexpr (XD == 16) return lane_xor16(v);
            else if constexpr (XD == 8) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));  // row_ror:8
            else return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
        };
        auto publish_pair = [&](u64* tile, int q, unsigned ep, float a, float b, int e8, bool store, auto xd) {
            if constexpr (F8) {
                const float pre = __uint_as_float((unsigned)(127 - e8) << 23);
                a *= pre;
                b *= pre;
                const bool up = (q & 2) != 0;
                const float got = xchg(up ? a : b, xd);
                const float va = up ? got : a, vb = up ? b : got;
                if (store && fmaxf(fabsf(va), fabsf(vb)) > 448.f) note_clip();
                unsigned lo32, hi16;
                f8_limbs(va, vb, lo32, hi16);
                if (store) gr_store16(tile + 4 * (q >> 2) + 2 * (q & 1) + ((q >> 1) & 1), ep, lo32, hi16);
            } else {
                if (store) gr_store(tile + q, ep, hpair(a, b, (q & 1) != 0));
            }
        };
        constexpr std::integral_constant<int, 16> xd16{};
        // sums of the staged operands, even pairs in .x and odd pairs in .y: y = scale (acc - (zero - 8) (S_even + 16 S_odd))
        const f16x2 ones2 = {(_Float16)1.0f, (_Float16)1.0f};
        auto pair_sums = [&](float2& sx, unsigned even, unsigned odd) {
            sx.x = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, even), ones2, sx.x, false);
            sx.y = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, odd), ones2, sx.y, false);
        };
        auto put_sums = [&](float2 sx) {  // misc[4 + gw] / misc[8 + gw]: this gatherer wave's S_even / S_odd
            if constexpr (F8) return;      // (fp8 operands: the streamers take the operand sums with all-ones MFMAs)
            sx.x = group_sum(sx.x, 64);
            sx.y = group_sum(sx.y, 64);
            if (lane == 0) {
                misc[4 + gw] = sx.x;
                misc[8 + gw] = sx.y;
            }
        };
        auto get_sums = [&]() {
            if constexpr (F8) {
                // the streamer waves' operand sums (valid behind the phase's first tile end); the A block scale made the products q x~
                // themselves, so there is no offset term: y = scale (acc - zero S)
                const f32x4 sa = *(const f32x4*)(misc + 32), sb2 = *(const f32x4*)(misc + 36);
                return ((sa[0] + sa[1]) + (sa[2] + sa[3])) + ((sb2[0] + sb2[1]) + (sb2[2] + sb2[3]));
            }
            float se = misc[4], so = misc[8];
#pragma unroll
            for (int g2 = 1; g2 < kGW; ++g2) {
                se += misc[4 + g2];
                so += misc[8 + g2];
            }
            return se + 16.f * so;
        };
        auto deq = [&](float2 t, float2 sc_, float2 z_, float s) {  // the streamers' operands are q - 8 (fp16) / q itself (fp8)
            constexpr float zc = F8 ? 0.f : 8.f;
            return float2{sc_.x * (t.x - (z_.x - zc) * s), sc_.y * (t.y - (z_.y - zc) * s)};
        };
        // stage one 16-B load (two granules = 4 fp16 values) of an edge into xs and add its operand sums
        auto stage = [&](const u32x4& v, int i, float2& sx) {
            if constexpr (F8) {  // two granules = dword `i` of each limb plane (fused_step_ring.hip f8_stage)
                *(unsigned*)(smem + kF8P0 + (size_t)i * 4) = __builtin_amdgcn_perm(v[2], v[0], 0x05040100u);                 // l0: low halves of dwords 0 / 2
                *(unsigned*)(smem + kF8P0 + kF8Plane + (size_t)i * 4) = __builtin_amdgcn_perm(v[2], v[0], 0x07060302u);      // l1: high halves of dwords 0 / 2
                *(unsigned*)(smem + kF8P0 + 2 * kF8Plane + (size_t)i * 4) = __builtin_amdgcn_perm(v[3], v[1], 0x05040100u);  // l2: low halves of dwords 1 / 3
                return;
            }
            *(u64*)(xs + (size_t)i * 8) = ((u64)v[2] << 32) | v[0];
            pair_sums(sx, v[0], v[2]);
        };
        bool dbg_on = false;
#define FW_GSTAMP(i)                                                                              \
    do {                                                                                          \
        if (dbg_on && gw == 0 && lane == 0) p.dbg[bid * 64 + (i)] = wall_clock64();               \
    } while (0)

        // publish this gatherer's 16 rows of an x-type edge: fp16(x_scale * norm_scale * x) pairs + their partial sum of squares.
        // x_scale = the power of two next to 1/rms of the PREVIOUS x edge (the same float in every workgroup and gatherer).
        float x_scale = 1.f;    // applied to the edge published last (= the one gathered next)
        float rinv_seen = 1.f;  // 1/rms of the x edge gathered last
        auto publish_x = [&](float2 xv, float2 gsc) {
            const unsigned ep = ebase + edge;
            u64* dst = p.gx + (size_t)xpar * SH::GXS;
            x_scale = __uint_as_float((__float_as_uint(rinv_seen) + 0x00400000u) & 0x7F800000u);
            if (!epi) return;  // (RT = 1: gatherer 1 owns no residual tile — it only keeps x_scale in step for its c_fc epilogues)
            publish_pair(dst + (bid * RT + er) * 8, pg, ep, x_scale * gsc.x * xv.x, x_scale * gsc.y * xv.y, kF8Ex, w8 == 0, xd16);
            float ss = xv.x * xv.x + xv.y * xv.y;  // the same in the 8 lanes of a pair: sum over the 8 pairs
            ss = MI355_DPP_ADD(ss, 0x140);
            ss += lane_xor16(ss);
            ss += lane_xor32(ss);
            if (lane == 0) gr_store(dst + C / 2 + bid * RT + er, ep, __float_as_uint(ss));
        };
        // gather an x-type edge into xs (fp16), 1/rms into misc[0], the operand sums into misc[4 .. 7].  The C / 4 pair loads are split
        // NP0 : rest between the two gatherer waves; the 128 RT loads of the sums of squares go to gatherer 0, which also has the
        // serial tail (sums, 1/rms).
        auto gather_x = [&]() {
            const unsigned ep = ebase + edge;
            const unsigned base = (unsigned)xpar * (unsigned)SH::GXS * 8u;
            constexpr int NPL = C / 4 / 64;          // pair loads per lane over both gatherers (32 / 26 / 20 / 16)
            // gatherer 0's share (it also takes the sums of squares and the serial tail behind them)
            constexpr int NP0 = (NPL + (NWG * RT / 2 + 63) / 64 + kGW - 1) / kGW - (NWG * RT / 2 + 63) / 64;
            constexpr int NSL = NWG * RT / 2;        // loads of the per-tile sums of squares (two each)
            constexpr int NS = (NSL + 63) / 64;      // ... per lane
            float2 sx = {0.f, 0.f};
            if (gw == 0) {
                u32x4 v[NP0 + NS];
                for (unsigned spins = 0;; ++spins) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < NP0 + NS; ++k) {
                        const int is = (k - NP0) * 64 + lane_v;  // (k >= NP0: load `is` of the sums; none past NSL)
                        const unsigned off = k < NP0 ? base + (unsigned)(k * 64 + lane_v) * 16u
                                                     : (is < NSL ? base + (unsigned)(C / 2) * 8u + (unsigned)is * 16u : 0xFFFFFFF0u);
                        v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_ws, off, 0, 16));
                    }
#pragma unroll
                    for (int k = 0; k < NP0 + NS; ++k) {
                        if (F8 && k < NP0) ok &= (v[k][1] >> 16) == (ep & 0xFFFFu) && (v[k][3] >> 16) == (ep & 0xFFFFu);  // (pair granules: 16-bit tags)
                        else ok &= (k >= NP0 && (k - NP0) * 64 + lane_v >= NSL) || (v[k][1] == ep && v[k][3] == ep);
                    }
                    if (__all(ok)) break;
                    if (spins > kSpinLimit || aborted(p)) {
                        if (lane == 0) raise_abort(p, 0x100u + edge);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int k = 0; k < NP0; ++k) stage(v[k], k * 64 + lane_v, sx);
                float ss = 0.f;
#pragma unroll
                for (int k = 0; k < NS; ++k)  // (loads past the end returned zeros)
                    ss += __uint_as_float(v[NP0 + k][0]) + __uint_as_float(v[NP0 + k][2]);
                ss = group_sum(ss, 64);
                put_sums(sx);
                if (lane == 0) {
                    const float rv = __builtin_amdgcn_rsqf(ss * (1.0f / (float)C) + p.eps);  // (the argument is >= eps: no denormal guard)
                    misc[0] = rv;
                }
            } else {
                // the other gatherers share the rest: N1 loads per lane each, two chunks in flight, only the first waits for producers
                constexpr int N1 = (NPL - NP0 + kGW - 2) / (kGW - 1);
                constexpr int NA = N1 < 8 ? N1 : 8, NB = N1 - NA;
                constexpr bool EX = (NP0 + (kGW - 1) * N1) * 64 == C / 4;  // the shares tile the edge exactly: no bounds to test
                const int first = (NP0 + (gw - 1) * N1) * 64;
                const int end = first + N1 * 64 < C / 4 ? first + N1 * 64 : C / 4;
                u32x4 va[NA];
                sweep_issue<NA>(rs_ws, base, first, end, va, lane_v);
                if constexpr (NB > 0) {
                    u32x4 vb[NB];
                    sweep_issue<NB>(rs_ws, base, first + NA * 64, end, vb, lane_v);
                    sweep<NA, F8>(p, rs_ws, base, first, end, ep, va, 0x200u + edge, lane_v, true);
#pragma unroll
                    for (int k = 0; k < NA; ++k)
                        if (EX || first + k * 64 + lane_v < end) stage(va[k], first + k * 64 + lane_v, sx);
                    sweep<NB, F8>(p, rs_ws, base, first + NA * 64, end, ep, vb, 0x200u + edge, lane_v, true);
#pragma unroll
                    for (int k = 0; k < NB; ++k)
                        if (EX || first + (NA + k) * 64 + lane_v < end) stage(vb[k], first + (NA + k) * 64 + lane_v, sx);
                } else {
                    sweep<NA, F8>(p, rs_ws, base, first, end, ep, va, 0x200u + edge, lane_v, true);
#pragma unroll
                    for (int k = 0; k < NA; ++k)
                        if (EX || first + k * 64 + lane_v < end) stage(va[k], first + k * 64 + lane_v, sx);
                }
                put_sums(sx);
            }
            xpar ^= 1;
            ++edge;
        };
        // behind the B1 of a phase that gathered an x edge: 1/rms and the factor its epilogue multiplies by (x_scale is a power of two:
        // 1 / x_scale = the float with the mirrored exponent)
        auto x_rinv = [&]() {
            rinv_seen = misc[0];
            return rinv_seen * __uint_as_float(0x7F000000u - __float_as_uint(x_scale));
        };

        // ---- the residual rows of this gatherer's tile: embedding of the step's token (model.py:102)
        int r0 = (bid * RT + er) * 16 + 2 * pg;  // first row of this lane's pair among the n_embd residual rows
        float2 xres = ldpair(p.wte + (size_t)token * C + r0);
        const bf16_t* norms_l = p.norms;
        const bf16_t* sz_l = p.sz;
        bf16_t* kv_l = p.kv;
        // RoPE factors of this lane's q / k pair: dims hj DH + er 16 + 2 pg, + 1 of the head
        const float2 cs = *(const float2*)(p.rope + ((size_t)pos * (kHs / 2) + hj * (DH / 2) + er * 8 + pg) * 2);
        publish_x(xres, ldpair(norms_l + r0));
        for (int l = 0; l < p.n_layer; ++l) {
            dbg_on = p.dbg != nullptr && l == p.dbg_layer;
            asm volatile("" : "+v"(lane_v));
            pg = lane_v >> 3;
            w8 = lane_v & 7;
            psrc = (pg >> 1) * 4 + ((2 * pg) & 3);
            r0 = (bid * RT + er) * 16 + 2 * pg;
            // ================= c_attn
            // (c_attn's epilogue spread over all four gatherers — q rows on gatherers 0 / 1, k and v rows on 2 / 3 — measured flat at the 65B
            // width, profiles/r06_ab_wide_knobs.txt: the tile's owner does all three)
            const int crole = epi ? 3 : 2, cer = er;   // 3: q, k and v; 2: nothing
            const int nq = head * kHs + hj * DH + cer * 16 + 2 * pg;  // q rows of this lane's pair; k at + C, v at + 2 C
            float2 sc[3], zr[3];
            if (crole != 2) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    sc[r] = ldpair(sz_l + nq + r * C);
                    zr[r] = ldpair(sz_l + 3 * C + nq + r * C);
                }
            }
            gather_x();
            FW_GSTAMP(2);
            if (p.dbg != nullptr && l == p.dbg_layer + 1 && gw == 0 && lane == 0) p.dbg[bid * 64 + 46] = wall_clock64();
            __syncthreads();  // B1
            __syncthreads();  // Bt (one virtual tile)
            FW_GSTAMP(56);
            {
                const float rinv = x_rinv();
                if (crole != 2) {
                    const float s = get_sums();
                    // RoPE (model.py:306-323) of the q / k pair, publish to the head group, write the cache rows
                    const unsigned ep = ebase + edge;
                    u64* dst = p.gq + ((size_t)qpar * NH + head) * 256 + hj * (2 * DH);
                    if (crole == 0 || crole == 3) {
                        float2 y = deq(tile_pair(cer), sc[0], zr[0], s);
                        y.x *= rinv;
                        y.y *= rinv;
                        const float qa = y.x * cs.x - y.y * cs.y, qb = y.y * cs.x + y.x * cs.y;
                        if (w8 == 0) gr_store(dst + cer * 16 + 2 * pg, ep, __float_as_uint(qa));
                        if (w8 == 1) gr_store(dst + cer * 16 + 2 * pg + 1, ep, __float_as_uint(qb));
                    }
                    if (crole == 1 || crole == 3) {
                        float2 yk = deq(tile_pair(RT + cer), sc[1], zr[1], s), yv = deq(tile_pair(2 * RT + cer), sc[2], zr[2], s);
                        yk.x *= rinv;
                        yk.y *= rinv;
                        bf16_t* krow = kv_l + ((size_t)head * p.S + pos) * kHs + hj * DH + cer * 16;
                        bf16_t* vrow = krow + (size_t)NH * p.S * kHs;
                        const unsigned kp = bfpair(yk.x * cs.x - yk.y * cs.y, yk.y * cs.x + yk.x * cs.y);
                        const unsigned vp = bfpair(yv.x * rinv, yv.y * rinv);
                        if (w8 == 2) gr_store(dst + DH + cer * 8 + pg, ep, kp);
                        if (w8 == 3) gr_store(dst + DH + DH / 2 + cer * 8 + pg, ep, vp);
                        if (w8 == 4) ((unsigned*)krow)[pg] = kp;
                        if (w8 == 5) ((unsigned*)vrow)[pg] = vp;
                    }
                }
            }
            FW_GSTAMP(3);
            buf ^= 1;
            __syncthreads();  // B3
            // ================= attention
            {
                const unsigned ep = ebase + edge;
                if (gw == 0) {
                    u32x4 v[2];
                    sweep<2>(p, rs_ws, kOGq + (unsigned)((qpar * NH + head) * 256) * 8u, 0, 128, ep, v, 0x300u + edge, lane_v);
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
#pragma unroll
                        for (int e2 = 0; e2 < 2; ++e2) {
                            const int gi = (k * 64 + lane_v) * 2 + e2;  // granule index inside the head's 256
                            const int jj = gi / (2 * DH), e = gi % (2 * DH);
                            const unsigned val = v[k][2 * e2];
                            if (e < DH) {
                                qs[jj * DH + e] = __uint_as_float(val);
                            } else if (e < DH + DH / 2) {
                                knew[jj * DH + 2 * (e - DH)] = __uint_as_float(val << 16);
                                knew[jj * DH + 2 * (e - DH) + 1] = __uint_as_float(val & 0xffff0000u);
                            } else {
                                vnew[jj * DH + 2 * (e - DH - DH / 2)] = __uint_as_float(val << 16);
                                vnew[jj * DH + 2 * (e - DH - DH / 2) + 1] = __uint_as_float(val & 0xffff0000u);
                            }
                        }
                    }
                }
                qpar ^= 1;
                ++edge;
                FW_GSTAMP(4);
                __syncthreads();  // Ba1
                __syncthreads();  // Ba3
                FW_GSTAMP(5);
                solo_i = __builtin_amdgcn_readfirstlane(solo_i);
                asm volatile("" : "+s"(solo_i));  // (see the streamers)
                const bool solo = solo_i != 0;
                // the 8 waves' partials merged: over ALL rows of the head (solo: this workgroup's DH dimensions of it are the attention
                // output — published at once), or over this workgroup's rows (all 128 dimensions of them go to the head group, every
                // workgroup then merges the GS partials for its own DH output dimensions)
                const unsigned ep1 = ebase + edge;
                if (gw == 0) {
                    float mall = misc[16];
#pragma unroll
                    for (int w = 1; w < kSW; ++w) mall = fmaxf(mall, misc[16 + w]);
                    float2 o = {0.f, 0.f};
                    float lsum = 0.f;
#pragma unroll
                    for (int w = 0; w < kSW; ++w) {
                        const float wsc = __expf(misc[16 + w] - mall);
                        const float2 t = *(const float2*)(op2 + w * 128 + 2 * lane_v);
                        o.x += wsc * t.x;
                        o.y += wsc * t.y;
                        lsum += wsc * misc[24 + w];
                    }
                    if (solo) {
                        // lane holds dimensions 2 lane, 2 lane + 1 of the head: pair lane % (DH / 2) of workgroup lane / (DH / 2)
                        const float inv = __builtin_amdgcn_rcpf(lsum);
                        const int px = lane_v % (DH / 2);
                        u64* ga_t = p.ga + (size_t)apar * (C / 2) + head * 64 + hj * (DH / 2);
                        // (the pairs of this workgroup sit in DH / 2 consecutive lanes: pair px ^ 2 is 2 lanes away)
                        publish_pair(ga_t + (px >> 3) * 8, px & 7, ep1, o.x * inv, o.y * inv, kF8Ea, lane_v / (DH / 2) == hj, std::integral_constant<int, 2>{});
                    } else {
                        u64* dstp = p.gp + (((size_t)ppar * NH + head) * GS + hj) * kPartStride;
                        gr_store(dstp + 2 * lane_v, ep1, __float_as_uint(o.x));
                        gr_store(dstp + 2 * lane_v + 1, ep1, __float_as_uint(o.y));
                        if (lane_v == 0) {
                            gr_store(dstp + 128, ep1, __float_as_uint(mall));
                            gr_store(dstp + 129, ep1, __float_as_uint(lsum));
                        }
                    }
                }
                if (!solo) ++edge;
                if (gw == 0 && !solo) {
                    // lane = (pair px of this workgroup's DH output dimensions, partial wq of the head group)
                    const int px = lane_v / GS, wq = lane_v % GS;
                    const unsigned hbase = kOGp + (unsigned)(((ppar * NH + head) * GS) * kPartStride) * 8u;
                    const unsigned off1 = hbase + (unsigned)(wq * kPartStride + hj * DH + 2 * px) * 8u;
                    const unsigned off2 = hbase + (unsigned)(wq * kPartStride + 128) * 8u;
                    u32x4 v1, v2;
                    for (unsigned spins = 0;; ++spins) {
                        v1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_ws, off1, 0, 16));
                        v2 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_ws, off2, 0, 16));
                        const bool ok = v1[1] == ep1 && v1[3] == ep1 && v2[1] == ep1 && v2[3] == ep1;
                        if (__all(ok)) break;
                        if (spins > kSpinLimit || aborted(p)) {
                            if (lane == 0) raise_abort(p, 0x380u + edge);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    const float mj = __uint_as_float(v2[0]), lj = __uint_as_float(v2[2]);
                    float mall = MI355_DPP_MAX(mj, 0xB1);
                    mall = MI355_DPP_MAX(mall, 0x4E);
                    if constexpr (GS == 8) mall = MI355_DPP_MAX(mall, 0x141);
                    const float wsc = __expf(mj - mall);
                    const float ox = group_sum(__uint_as_float(v1[0]) * wsc, GS);
                    const float oy = group_sum(__uint_as_float(v1[2]) * wsc, GS);
                    const float inv = __builtin_amdgcn_rcpf(group_sum(lj * wsc, GS));
                    // attention output elements head * 128 + hj * DH + 2 px, + 1 -> one pair granule
                    u64* ga_t = p.ga + (size_t)apar * (C / 2) + head * 64 + hj * (DH / 2);
                    // (lane = px GS + wq: pair px ^ 2 is 2 GS lanes away)
                    publish_pair(ga_t + (px >> 3) * 8, px & 7, ebase + edge, ox * inv, oy * inv, kF8Ea, wq == 0, std::integral_constant<int, 2 * GS>{});
                }
                if (!solo) ppar ^= 1;
                FW_GSTAMP(6);
                __syncthreads();  // Ba4
            }
            // ================= attn.c_proj (+ residual)
            {
                float2 s1 = {0.f, 0.f}, z1 = {0.f, 0.f}, gn = {0.f, 0.f};
                if (epi) {
                    s1 = ldpair(sz_l + 6 * C + r0);
                    z1 = ldpair(sz_l + 7 * C + r0);
                    gn = ldpair(norms_l + C + r0);  // rms_2
                }
                const unsigned ep = ebase + edge;
                {
                    // C / 4 loads of pair granules, an equal share per gatherer, two chunks in flight
                    constexpr int NL = (C / 4 / 64 + kGW - 1) / kGW;  // per lane
                    constexpr int NA = NL < 8 ? NL : 8, NB = NL - NA;
                    constexpr bool EX = NL * kGW * 64 == C / 4;
                    const unsigned base = kOGa + (unsigned)apar * (unsigned)(C / 2) * 8u;
                    const int first = gw * NL * 64, end = first + NL * 64 < C / 4 ? first + NL * 64 : C / 4;
                    float2 sxp = {0.f, 0.f};
                    u32x4 va[NA];
                    sweep_issue<NA>(rs_ws, base, first, end, va, lane_v);
                    if constexpr (NB > 0) {
                        u32x4 vb[NB];
                        sweep_issue<NB>(rs_ws, base, first + NA * 64, end, vb, lane_v);
                        sweep<NA, F8>(p, rs_ws, base, first, end, ep, va, 0x400u + edge, lane_v, true);
#pragma unroll
                        for (int k = 0; k < NA; ++k)
                            if (EX || first + k * 64 + lane_v < end) stage(va[k], first + k * 64 + lane_v, sxp);
                        sweep<NB, F8>(p, rs_ws, base, first + NA * 64, end, ep, vb, 0x400u + edge, lane_v, true);
#pragma unroll
                        for (int k = 0; k < NB; ++k)
                            if (EX || first + (NA + k) * 64 + lane_v < end) stage(vb[k], first + (NA + k) * 64 + lane_v, sxp);
                    } else {
                        sweep<NA, F8>(p, rs_ws, base, first, end, ep, va, 0x400u + edge, lane_v, true);
#pragma unroll
                        for (int k = 0; k < NA; ++k)
                            if (EX || first + k * 64 + lane_v < end) stage(va[k], first + k * 64 + lane_v, sxp);
                    }
                    put_sums(sxp);
                }
                apar ^= 1;
                ++edge;
                FW_GSTAMP(7);
                __syncthreads();  // B1
                __syncthreads();  // Bt
                {
                    const float2 d = deq(tile_pair(er), s1, z1, get_sums());
                    xres.x += d.x;
                    xres.y += d.y;
                    publish_x(xres, gn);
                }
                FW_GSTAMP(8);
                buf ^= 1;
                __syncthreads();  // B3
            }
            // ================= c_fc1 / c_fc2 + SwiGLU: pair tile t of this workgroup belongs to gatherer t & 1
            {
                const bf16_t* s_fc = sz_l + 8 * C;
                // scales / zeros of the pair tile this gatherer handles next (requested one of its tiles ahead)
                auto fc_sz = [&](int t, float2& a1, float2& b1, float2& a2, float2& b2) {
                    const int n = (bid + (t < n_fc ? t : 0) * NWG) * 16 + 2 * pg;
                    a1 = ldpair(s_fc + n);
                    b1 = ldpair(s_fc + p.H + n);
                    a2 = ldpair(s_fc + 2 * p.H + n);
                    b2 = ldpair(s_fc + 3 * p.H + n);
                };
                float2 fs1, fz1, fs2, fz2;
                fc_sz(gw, fs1, fz1, fs2, fz2);
                gather_x();
                FW_GSTAMP(9);
                __syncthreads();  // B1
                const unsigned ep = ebase + edge;
                u64* dst = p.gh + (size_t)hpar * gh_stride;
                const float rinv = x_rinv();
                float s = 0.f;
                if constexpr (!F8) s = get_sums();
                constexpr int tiles_pad = NB_FC * TPB_FC;  // tile ends the streamers pass
                for (int t = 0; t < tiles_pad; ++t) {
                    float2 ns1 = fs1, nz1 = fz1, ns2 = fs2, nz2 = fz2;
                    if ((t & 1) == gw) fc_sz(t + 2, ns1, nz1, ns2, nz2);
                    __syncthreads();  // Bt
                    if constexpr (F8) {
                        if (t == 0) s = get_sums();  // (the streamers' operand sums exist behind the first tile end)
                    }
                    if ((t & 1) == gw && t < n_fc) {
                        const float2 a = deq(tile_pair(0), fs1, fz1, s);
                        const float2 b = deq(tile_pair(1), fs2, fz2, s);
                        publish_pair(dst + (bid + t * NWG) * 8, pg, ep, swiglu_e(a.x * rinv, b.x * rinv), swiglu_e(a.y * rinv, b.y * rinv), kF8Eh, w8 == 0, xd16);
                    }
                    fs1 = ns1;
                    fz1 = nz1;
                    fs2 = ns2;
                    fz2 = nz2;
                    buf ^= 1;
                }
                FW_GSTAMP(10);
                __syncthreads();  // B3
            }
            // ================= mlp.c_proj (+ residual) -> next layer's x edge
            {
                float2 s1 = {0.f, 0.f}, z1 = {0.f, 0.f}, gn = {0.f, 0.f};
                const bf16_t* s_mp = sz_l + 8 * C + 4 * p.H;
                if (epi) {
                    s1 = ldpair(s_mp + r0);
                    z1 = ldpair(s_mp + C + r0);
                    gn = ldpair(norms_l + 2 * C + r0);  // rms_1 of the next layer, or ln_f after the last
                }
                const unsigned ep = ebase + edge;
                {
                    // H / 4 loads of pair granules, an equal share per gatherer, in chunks of 8 per lane, TWO in flight: only the first one waits
                    // for producers (issued one after the other each later chunk costs its own memory round trip)
                    const int n_loads = p.H / 4, part_l = (n_loads + kGW - 1) / kGW;
                    const int first = gw * part_l, end = first + part_l < n_loads ? first + part_l : n_loads;
                    const unsigned hbase = kOGh + (unsigned)hpar * (unsigned)gh_stride * 8u;
                    int lh = lane_v;
                    asm volatile("" : "+v"(lh));  // addresses of this block are computed here, not hoisted and spilled
                    float2 sxp = {0.f, 0.f};
                    constexpr int NCH = 12 / kGW;  // chunks of 512 loads per gatherer: n_hidden <= 24576
                    constexpr int ND = 2;  // chunks in flight
                    u32x4 vv[ND][8];
#pragma unroll
                    for (int j = 0; j < ND - 1; ++j) sweep_issue<8>(rs_ws, hbase, first + j * 512, end, vv[j], lh);
#pragma unroll
                    for (int ch = 0; ch < NCH; ++ch) {
                        const int c0 = first + ch * 512;
                        if (ch + ND - 1 < NCH) sweep_issue<8>(rs_ws, hbase, c0 + (ND - 1) * 512, end, vv[(ch + ND - 1) % ND], lh);
                        sweep<8, F8>(p, rs_ws, hbase, c0, end, ep, vv[ch % ND], 0x500u + edge, lh, true);
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const int i = c0 + k * 64 + lh;
                            if (i < end) stage(vv[ch % ND][k], i, sxp);
                        }
                    }
                    put_sums(sxp);
                }
                hpar ^= 1;
                ++edge;
                FW_GSTAMP(11);
                __syncthreads();  // B1
                __syncthreads();  // Bt
                FW_GSTAMP(57);
                {
                    const float2 d = deq(tile_pair(er), s1, z1, get_sums());
                    xres.x += d.x;
                    xres.y += d.y;
                    publish_x(xres, gn);
                }
                FW_GSTAMP(12);
                buf ^= 1;
                __syncthreads();  // B3
            }
            norms_l += 2 * C;
            sz_l += p.sz_layer_stride;
            kv_l += (size_t)2 * NH * p.S * kHs;
        }
        dbg_on = false;
        // ================= ln_f + lm_head (+ greedy arg-max, generate.py:68-85 with top_k = 1): gatherer 0
        {
            auto head_sz = [&](int t, float2& sc_, float2& z_) {  // scale / zero of a tile's rows, requested one tile ahead
                const int n = (bid + t * NWG) * 16 + 2 * pg;
                const bool ok = t < n_head_t && n + 1 < p.V;
                sc_ = ok ? ldpair(p.sz_head + n) : float2{0.f, 0.f};
                z_ = ok ? ldpair(p.sz_head + p.V + n) : float2{0.f, 0.f};
            };
            float2 sct = {0.f, 0.f}, zt = {0.f, 0.f};
            if (gw == 0) head_sz(0, sct, zt);
            gather_x();
            __syncthreads();  // B1
            const float rinv = x_rinv();
            float s = 0.f;
            if constexpr (!F8) s = get_sums();
            float best = -INFINITY;
            int bi = 0x7fffffff;
            const int tiles_pad = p.head_turns * TPB_HD;  // tile ends the streamers pass
            for (int t = 0; t < tiles_pad; ++t) {
                float2 scn = {0.f, 0.f}, zn = {0.f, 0.f};
                if (gw == 0) head_sz(t + 1, scn, zn);
                __syncthreads();  // Bt
                if constexpr (F8) {
                    if (t == 0) s = get_sums();
                }
                if (gw == 0 && t < n_head_t) {
                    const int n = (bid + t * NWG) * 16 + 2 * pg;
                    float2 y = deq(tile_pair(0), sct, zt, s);
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
            __syncthreads();  // B3
            if (p.mode & 1) {
                if (gw == 0) {
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
                        const bool ok = sweep<4>(p, rs_ws, kOGm, 0, NWG, ep, v, 0x600u + edge, lane_v);
                        float bv = -INFINITY;
                        int bx = 0x7fffffff;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (k * 64 + lane_v >= NWG) continue;  // (fewer than 256 workgroups: loads past the end returned zeros)
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
                __syncthreads();
            }
            if (bid == 0 && gw == 0 && lane == 0) p.state[1] = step_id + 1u;
        }
#undef FW_GSTAMP
    }
    FW_STAMP(1);
}

// ------------------------------------------------------------------------------------------------ host side
// (residency: see fused_step_ring.hip — the occupancy query is made once, a kernel that does not fit one workgroup per CU is refused)
namespace {
// [0..3]: fp16 operands (weight_fmt 4), [4..7]: E4M3 limb operands (weight_fmt 5)
const void* const kWideFn[8] = {
    (const void*)fused_step_wide_kernel<64, 4, false>, (const void*)fused_step_wide_kernel<52, 4, false>,
    (const void*)fused_step_wide_kernel<40, 4, false>, (const void*)fused_step_wide_kernel<32, 8, false>,
    (const void*)fused_step_wide_kernel<64, 4, true>,  (const void*)fused_step_wide_kernel<52, 4, true>,
    (const void*)fused_step_wide_kernel<40, 4, true>,  (const void*)fused_step_wide_kernel<32, 8, true>};
int wide_index(int n_head) { return n_head == 64 ? 0 : n_head == 52 ? 1 : n_head == 40 ? 2 : n_head == 32 ? 3 : -1; }
}  // namespace

int fused_step_wide_occupancy_ok() {
    static int ok = -1;
    static std::once_flag once;
    std::call_once(once, [] {
        ok = 0;
        for (int i = 0; i < 8; ++i) {
            int per_cu = 0;
            (void)hipFuncSetAttribute(kWideFn[i], hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kWideFn[i], kThreads, kLdsBytes) == hipSuccess && per_cu >= 1) ok |= 1 << i;
        }
    });
    // bit i: instantiation i (64 / 52 / 40 heads x 4 workgroups, 32 heads x 8; + 4: the E4M3-operand twin) fits one workgroup per CU
    return ok;
}

// tiles per body of the pair phase / of lm_head and the ring steps mlp.c_proj may take per wave, for the host's checks and body counts
void fused_step_wide_geometry(int n_head, int* tpb_fc, int* fc_max, int* tpb_head, int* mp_steps) {
    const int spt = (n_head + kSW - 1) / kSW;
    *tpb_fc = turns_for(6, spt) * 6 / spt;
    *tpb_head = turns_for(12, spt) * 12 / spt;
    *fc_max = n_head == 32 ? 3 : 6;
    *mp_steps = n_head == 64 ? 22 : n_head == 32 ? 11 : 18;
}

// launched by mi355_fused_step (fused_step.hip) for weight_fmt 4 / 5 (p.fmt); n_head selects the instantiation, the grid is
// n_head x (4 or 8) workgroups
int fused_step_wide_launch(const FusedParams& p, int n_head, hipStream_t stream, hipEvent_t e0, hipEvent_t e1) {
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        for (int i = 0; i < 8 && attr_err == hipSuccess; ++i)
            attr_err = hipFuncSetAttribute(kWideFn[i], hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    });
    MI355_CHECK_ARG(attr_err == hipSuccess, (int)attr_err, "fused_step: hipFuncSetAttribute failed: %s", hipGetErrorString(attr_err));
    const int grid = n_head * (n_head == 32 ? 8 : 4);
#define FW_LAUNCH(K_)                                                                                                  \
    do {                                                                                                              \
        if (e0 != nullptr) {                                                                                          \
            hipExtLaunchKernelGGL((K_), dim3(grid), dim3(kThreads), (uint32_t)kLdsBytes, stream, e0, e1, 0, p);        \
        } else {                                                                                                      \
            hipLaunchKernelGGL((K_), dim3(grid), dim3(kThreads), kLdsBytes, stream, p);                                \
        }                                                                                                             \
    } while (0)
    const int idx = wide_index(n_head);
    switch (idx < 0 ? -1 : idx + (p.fmt == 5 ? 4 : 0)) {
        case 0: FW_LAUNCH((fused_step_wide_kernel<64, 4, false>)); break;
        case 1: FW_LAUNCH((fused_step_wide_kernel<52, 4, false>)); break;
        case 2: FW_LAUNCH((fused_step_wide_kernel<40, 4, false>)); break;
        case 3: FW_LAUNCH((fused_step_wide_kernel<32, 8, false>)); break;
        case 4: FW_LAUNCH((fused_step_wide_kernel<64, 4, true>)); break;
        case 5: FW_LAUNCH((fused_step_wide_kernel<52, 4, true>)); break;
        case 6: FW_LAUNCH((fused_step_wide_kernel<40, 4, true>)); break;
        case 7: FW_LAUNCH((fused_step_wide_kernel<32, 8, true>)); break;
        default: MI355_CHECK_ARG(false, MI355_E_SHAPE, "fused_step (weight_fmt 4 / 5): %d heads", n_head);
    }
#undef FW_LAUNCH
    MI355_LAUNCH_CHECK();
    return 0;
}

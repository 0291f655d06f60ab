// Host entry of the persistent decode step, mi355_fused_step (include/mi355_llama.h): argument checks and the launch of
// fused_step_ring_kernel<GRP, FMT> (csrc/fused_step_ring.hip: the kernel, its design notes and the reference file:line it replaces —
// lit_llama/model.py:76-122 and generate.py:68-85 per generated token).
//
// Round 3 kept a second implementation of the step in this file — weights through per-wave LDS rings filled by LDS-DMA across phase
// boundaries, `MI355_FUSED_IMPL=lds`: built, parity-tested and 15 % slower than the register-ring kernel (NOTES
// items 23-24).  Round 4 built the bf16 and LLM.int8 persistent steps on the register-ring kernel, so that one has moved out of the
// tree: scripts/patches/r03_fused_step_lds_dma_kernel.hip.txt is its last source.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_ext.h>

#include "common.h"
#include "fused_step_common.h"

namespace {
constexpr int kG = 256;           // workgroups = CUs
constexpr int kThreads = 64 * 10;  // 8 streamer + 2 gatherer waves
constexpr int kC = 4096;          // n_embd
constexpr int kHeads = 32;
constexpr int kHs = 128;
constexpr int kMaxFcTiles = 3;    // c_fc1/c_fc2 pair tiles per workgroup (n_hidden <= 12288)
constexpr int kMaxHeadTiles = 8;  // lm_head tiles per workgroup (vocab <= 32768)
constexpr int kMaxS = 32768;      // cache rows
}  // namespace

// ------------------------------------------------------------------------------------------------ host side
extern "C" size_t mi355_fused_step_workspace_bytes(int n_hidden) {
    if (n_hidden <= 0) return 0;
    // (one size for both kernels: the ring kernel's map and the wide-shape kernel's, fused_step_common.h)
    const size_t ring = kFsWsGh + (size_t)2 * (kFsGhSums + n_hidden / 2) * 8, wide = kFwGh + (size_t)2 * (n_hidden / 2) * 8;
    return ring > wide ? ring : wide;
}

// Does the wide-shape kernel (csrc/fused_step_wide.hip) handle the shape: n_embd = 128 n_head with 40 / 52 / 64 heads (LLaMA-13B / 30B /
// 65B, lit_llama/model.py:43-48: 4 workgroups per head = 160 / 208 / 256 workgroups) or 32 heads (the 7B shape on 8 workgroups per head,
// a cross-check of the ring kernel)
static int wide_ok(int n_embd, int n_head, int hs, int n_hidden, int vocab, int S) {
    if (mi355_num_cus() != kG || hs != kHs || n_embd != n_head * kHs) return 0;
    if (n_head != 32 && n_head != 40 && n_head != 52 && n_head != 64) return 0;
    const int nwg = n_head * (n_head == 32 ? 8 : 4);
    int tpb_fc, fc_max, tpb_head, mp_steps;
    fused_step_wide_geometry(n_head, &tpb_fc, &fc_max, &tpb_head, &mp_steps);
    const int units_h = n_hidden / 128;
    if (n_hidden <= 0 || n_hidden % 128 != 0 || units_h > 176 || (units_h + 7) / 8 > mp_steps || n_hidden / 4 > 2 * 6 * 512) return 0;
    if ((n_hidden / 16 + nwg - 1) / nwg > fc_max) return 0;  // pair tiles of the busiest workgroup (the kernel unrolls its bodies)
    if (vocab <= 0 || vocab % 2 != 0 || S < 1 || S > kMaxS) return 0;
    return 1;
}

extern "C" int mi355_fused_step_supported(int n_embd, int n_head, int hs, int n_hidden, int vocab, int S) {
    if (mi355_num_cus() != kG) return 0;  // one resident workgroup per CU
    if (n_embd == kC && n_head == kHeads && hs == kHs) {  // the 7B shape: csrc/fused_step_ring.hip, head groups of 8
        if (n_hidden <= 0 || n_hidden % 128 != 0 || n_hidden / 16 > kMaxFcTiles * kG || n_hidden / 128 > 96) return 0;
        if (vocab <= 0 || vocab % 2 != 0 || (vocab + 15) / 16 > kMaxHeadTiles * kG) return 0;
        if (S < 1 || S > kMaxS) return 0;
        return 1;
    }
    // wider shapes (round 6): csrc/fused_step_wide.hip, weight_fmt 4 / 5 only
    return wide_ok(n_embd, n_head, hs, n_hidden, vocab, S) ? 2 : 0;
}


// weight_fmt 4 / 5: the wide-shape kernel (per-row int4 streams; fp16 operands / E4M3 limb operands)
static int fused_step_wide(const mi355_fused_step_args* a, mi355_stream_t stream) {
    MI355_CHECK_ARG(wide_ok(a->n_embd, a->n_head, a->hs, a->n_hidden, a->vocab, a->S), MI355_E_SHAPE,
                    "fused_step (weight_fmt 4 / 5): needs %d CUs, n_embd = 128 n_head with 32 / 40 / 52 / 64 heads, n_hidden %% 128 == 0 within the "
                    "shape's limits (<= 6 pair tiles per workgroup, 3 for 32 heads), even vocab, S <= %d (got %d CUs, C=%d, heads=%d x %d, H=%d, V=%d, S=%d)",
                    kG, kMaxS, mi355_num_cus(), a->n_embd, a->n_head, a->hs, a->n_hidden, a->vocab, a->S);
    MI355_CHECK_ARG(a->group_cols == 0, MI355_E_ARG, "fused_step (weight_fmt 4 / 5): per-row scales only");
    MI355_CHECK_ARG(a->w && a->w_head && a->sz && a->sz_head && a->norms && a->wte && a->rope && a->kv && a->tokens && a->pos && a->logits &&
                        a->workspace,
                    MI355_E_ARG, "fused_step: null pointer");
    MI355_CHECK_ARG(a->n_layer >= 1 && a->n_layer * 6 + 8 < 1024, MI355_E_SHAPE, "fused_step: n_layer %d", a->n_layer);
    MI355_CHECK_ARG(!(a->mode & 1) || a->next_token != nullptr, MI355_E_ARG, "fused_step: arg-max without next_token");
    MI355_CHECK_ARG(a->mode >= 0 && a->mode <= 3 && a->mode != 2, MI355_E_ARG, "fused_step: mode must be 0, 1 or 3");
    MI355_CHECK_ARG(((uintptr_t)a->w | (uintptr_t)a->w_head | (uintptr_t)a->workspace | a->layer_stride | a->off_attn | a->off_proj |
                     a->off_fc | a->off_mproj) % 16 == 0,
                    MI355_E_ARG, "fused_step: streams and workspace must be 16-B aligned");
    const int widx = a->n_head == 64 ? 0 : a->n_head == 52 ? 1 : a->n_head == 40 ? 2 : 3;
    MI355_CHECK_ARG((fused_step_wide_occupancy_ok() & (1 << (widx + (a->weight_fmt == 5 ? 4 : 0)))) != 0, MI355_E_STATE,
                    "fused_step: the device does not admit one %d-thread workgroup of the wide-shape kernel per CU", kThreads);
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
    p.state = (unsigned*)ws;
    p.gx = (u64*)(ws + kFwGx);
    p.ga = (u64*)(ws + kFwGa);
    p.gq = (u64*)(ws + kFwGq);
    p.gm = (u64*)(ws + kFwGm);
    p.gp = (u64*)(ws + kFwGp);
    p.gh = (u64*)(ws + kFwGh);
    p.dbg = (u64*)a->debug_stamps;
    p.dbg_layer = a->reserved0;
    p.sz_layer_stride = (unsigned)(10 * a->n_embd + 4 * a->n_hidden);
    p.n_layer = a->n_layer;
    p.H = a->n_hidden;
    p.V = a->vocab;
    p.S = a->S;
    p.units_h = a->n_hidden / 128;
    p.fc_tiles = a->n_hidden / 16;
    p.head_tiles = (a->vocab + 15) / 16;
    p.fmt = a->weight_fmt;
    {
        const int nwg = a->n_head * (a->n_head == 32 ? 8 : 4);
        int tpb_fc, fc_max, tpb_head, mp_steps;
        fused_step_wide_geometry(a->n_head, &tpb_fc, &fc_max, &tpb_head, &mp_steps);
        p.fc_bodies = ((p.fc_tiles + nwg - 1) / nwg + tpb_fc - 1) / tpb_fc;          // (the kernel unrolls them: informational)
        p.head_turns = ((p.head_tiles + nwg - 1) / nwg + tpb_head - 1) / tpb_head;   // bodies of lm_head tiles of the busiest workgroup
    }
    p.mode = a->mode;
    p.eps = a->eps;
    p.scale = 1.0f / sqrtf((float)kHs);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (t_time_start != nullptr) {  // measurement hook: see gemv.hip launch_gemv_m
        e0 = t_time_start;
        e1 = t_time_stop;
        t_time_start = t_time_stop = nullptr;
    }
    return fused_step_wide_launch(p, a->n_head, (hipStream_t)stream, e0, e1);
}

extern "C" int mi355_fused_step(const mi355_fused_step_args* a, mi355_stream_t stream) {
    MI355_CHECK_ARG(a != nullptr, MI355_E_ARG, "fused_step: null args");
    if (a->weight_fmt == 4 || a->weight_fmt == 5) return fused_step_wide(a, stream);
    MI355_CHECK_ARG(mi355_fused_step_supported(a->n_embd, a->n_head, a->hs, a->n_hidden, a->vocab, a->S) == 1, MI355_E_SHAPE,
                    "fused_step: needs %d CUs, n_embd %d, %d heads of %d, n_hidden %% 128 == 0 and <= %d, vocab <= %d, "
                    "S <= %d (got %d CUs, C=%d, heads=%d x %d, H=%d, V=%d, S=%d)",
                    kG, kC, kHeads, kHs, kMaxFcTiles * kG * 16, kMaxHeadTiles * kG * 16, kMaxS, mi355_num_cus(), a->n_embd,
                    a->n_head, a->hs, a->n_hidden, a->vocab, a->S);
    const bool grouped = a->group_cols > 0;
    const int fmt = a->weight_fmt;
    MI355_CHECK_ARG((fmt >= 0 && fmt <= 3) || fmt == 6, MI355_E_ARG,
                    "fused_step: weight_fmt %d (0 = int4 streams, 1 = BF16, 2 = LLM.int8, 3 = int4 streams through fp8 operands; 4 / 5 = the wide-shape "
                    "kernel's fp16 / fp8 operands, 6 = 8-bit ColBlock streams through fp8 operands)", fmt);
    // fp8-limb operands keep three byte planes of the activation vector in LDS: 95 units of 128 columns
    // (and its gatherers sweep the hidden edge — n_hidden / 4 loads of two granules + 128 of operand-sum partials — in at most 24 loads per lane)
    MI355_CHECK_ARG((fmt != 3 && fmt != 6) || a->n_hidden / 128 <= 92, MI355_E_SHAPE, "fused_step: weight_fmt 3 / 6 needs n_hidden <= %d (got %d)", 92 * 128,
                    a->n_hidden);
    MI355_CHECK_ARG(!(grouped && fmt != 0 && fmt != 3), MI355_E_ARG, "fused_step: grouped scales exist for int4 streams only (weight_fmt 0 / 3)");
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
        MI355_CHECK_ARG(((a->n_hidden / 128 + 7) / 8 >> gsh) + 1 <= (fmt == 3 ? 15 : 16), MI355_E_SHAPE, "fused_step: too many groups per wave");
    }
    MI355_CHECK_ARG(a->n_layer >= 1 && a->n_layer * 6 + 8 < 1024, MI355_E_SHAPE, "fused_step: n_layer %d", a->n_layer);
    MI355_CHECK_ARG(!(a->mode & 1) || a->next_token != nullptr, MI355_E_ARG, "fused_step: arg-max without next_token");
    MI355_CHECK_ARG(a->mode >= 0 && a->mode <= 3 && a->mode != 2, MI355_E_ARG, "fused_step: mode must be 0, 1 or 3");
    MI355_CHECK_ARG(((uintptr_t)a->w | (uintptr_t)a->w_head | (uintptr_t)a->workspace | a->layer_stride | a->off_attn |
                     a->off_proj | a->off_fc | a->off_mproj) % 16 == 0,
                    MI355_E_ARG, "fused_step: streams and workspace must be 16-B aligned");
    MI355_CHECK_ARG((fused_step_ring_occupancy_ok() & (fmt == 6 ? 64 : fmt == 3 ? (grouped ? 32 : 16) : fmt == 2 ? 8 : fmt == 1 ? 4 : grouped ? 2 : 1)) != 0, MI355_E_STATE,
                    "fused_step: the device does not admit one %d-thread workgroup of the kernel per CU", kThreads);
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
        const int steps = per_wg * 4 * (fmt == 1 ? 4 : (fmt == 2 || fmt == 6) ? 2 : 1);  // ring steps per tile and wave: 4 (int4), 16 (BF16), 8 (8-bit)
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
    return fused_step_ring_launch(p, (hipStream_t)stream, e0, e1);
}

// Definitions shared by the two implementations of the persistent decode step (fused_step.hip, fused_step_ring.hip).
#pragma once
#include "common.h"

typedef unsigned long long u64;

struct FusedParams {
    const uint8_t* w;        // weight arena: layer l at w + l * layer_stride
    u64 layer_stride;
    unsigned off_attn, off_proj, off_fc, off_mproj, layer_bytes;
    unsigned head_bytes;
    const uint8_t* w_head;
    const bf16_t* sz;        // per layer: s_attn[3C] z_attn[3C] s_proj[C] z_proj[C] s_fc1[H] z_fc1[H] s_fc2[H] z_fc2[H] s_mp[C] z_mp[C]
    const bf16_t* sz_head;   // s[V] z[V]
    const bf16_t* norms;     // [n_layer][2][C], then ln_f[C]
    const bf16_t* wte;
    const float* rope;       // [block_size][hs/2][2]
    bf16_t* kv;              // [n_layer][2][n_head][S][hs]
    int32_t* tokens;
    int32_t* pos;
    int32_t* next_token;
    int32_t* out_tokens;
    float* logits;
    u64* gx;                 // [2][kFsGxStride]  x-type edges: 2048 pair granules, then per workgroup the partial sum of squares (weight_fmt
                             //                   3: {sum of squares, operand sum} as two adjacent granules)
    u64* ga;                 // [2][kFsGaStride]  attention output: 256 operand-sum partials (weight_fmt 3), then 2048 pair granules
    u64* gh;                 // [2][256 + H / 2]  MLP hidden: 256 operand-sum partials (weight_fmt 3), then H / 2 pair granules
    u64* gq;                 // [2][n_head][8][32] q / new k / new v of a head
    u64* gm;                 // [512]             arg-max candidates
    u64* gp;                 // [2][n_head][8][136] row-split attention: a workgroup's (128 weighted values, max, sum)
    unsigned* state;         // [0] abort code, [1] step counter
    u64* dbg;                // optional [kG][64] wall-clock stamps
    unsigned sz_layer_stride;
    int n_layer, H, V, S;
    int units_h, fc_tiles, head_tiles, head_turns;
    int mode;                // bit 0: greedy arg-max, bit 1: chained (advance tokens[0] / pos[0])
    int dbg_layer;           // layer whose phases are stamped into dbg
    float eps, scale;
    // grouped scales (register-ring kernel, GRP instantiation): gsh = log2(units of 128 columns per group), ngc / ngh = groups per
    // row for K = n_embd / K = n_hidden; gt: per layer the tables [tile][group][16 rows] of (bf16 scale | bf16 zero << 16) of
    // c_attn, attn.c_proj, c_fc1, c_fc2, mlp.c_proj in this order; gt_head: lm_head's
    const uint8_t* gt;
    const uint8_t* gt_head;
    u64 gt_layer_stride;
    unsigned gt_layer_bytes, gt_head_bytes;
    int grouped, gsh, ngc, ngh;
    int fmt;                 // 0: int4 streams, 1: BF16 streams, 2: LLM.int8 streams, 3: int4 streams through fp8-limb operands
    int fc_bodies;           // wide-shape kernel (fused_step_wide.hip): bodies of 3 pair tiles the busiest workgroup streams; its
                             // head_turns counts bodies of 3 lm_head tiles
};


// granules per parity of the x / attention-output edges and in front of the hidden edge's pairs (the 256 granules in front of the
// attention-output / hidden pairs carried per-publisher operand sums in round 5's rejected variant; the map keeps the room)
constexpr int kFsGxStride = 2048 + 512, kFsGaSums = 256, kFsGaStride = kFsGaSums + 2048, kFsGhSums = 256;
// workspace map (bytes), see mi355_fused_step_workspace_bytes
constexpr size_t kFsWsState = 0, kFsWsGx = 256, kFsWsGa = kFsWsGx + 2 * kFsGxStride * 8, kFsWsGq = kFsWsGa + 2 * kFsGaStride * 8,
                 kFsWsGm = kFsWsGq + 2 * 32 * 256 * 8, kFsWsGp = kFsWsGm + 512 * 8,
                 kFsWsGh = kFsWsGp + 2 * 32 * 8 * 136 * 8;

// workspace map of the wide-shape kernel (csrc/fused_step_wide.hip, weight_fmt 4), sized for its widest shape (n_embd 8192, 64 heads):
// x edges [2][C / 2 pairs + 512 sums of squares], attention output [2][C / 2], q / k / v of a head [2][n_head][256], arg-max
// candidates [512], attention partials [2][n_head][workgroups per head][136], MLP hidden [2][H / 2] — granules of 8 bytes
constexpr size_t kFwGx = 256, kFwGa = kFwGx + (size_t)2 * (4096 + 512) * 8, kFwGq = kFwGa + (size_t)2 * 4096 * 8,
                 kFwGm = kFwGq + (size_t)2 * 64 * 256 * 8, kFwGp = kFwGm + 512 * 8, kFwGh = kFwGp + (size_t)2 * 256 * 136 * 8;

int fused_step_wide_launch(const FusedParams& p, int n_head, hipStream_t stream, hipEvent_t e0, hipEvent_t e1);
int fused_step_wide_occupancy_ok();  // bit i: instantiation i (64 / 52 / 40 heads x 4 workgroups, 32 heads x 8) fits one workgroup per CU
void fused_step_wide_geometry(int n_head, int* tpb_fc, int* fc_max, int* tpb_head, int* mp_steps);
int fused_step_ring_launch(const FusedParams& p, hipStream_t stream, hipEvent_t e0, hipEvent_t e1);
int fused_step_ring_occupancy_ok();  // the device admits one workgroup of the kernel per CU (queried once)

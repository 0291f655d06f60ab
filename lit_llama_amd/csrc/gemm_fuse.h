// Internal (NOT part of the C ABI): producer / consumer fusion of the wide-GEMM chain of a prompt pass.
//
// Replaces the per-linear staging pass
// (stage_rows_kernel: f32 row -> bf16 operand + 1/rms + operand sum, four launches per layer) and the K / V cache
// write (rope_kv_write_kernel) of /root/reference lit_llama/model.py:185-237,251-254 as this library ran them in
// rounds 2-4: 14 % of a 2048-token 7B prompt (profiles/r04_prefill_kernel_stats_staged_chain.csv).
//   * a producer's epilogue writes what the next linear consumes: the residual epilogue of attn.c_proj / mlp.c_proj
//     emits bf16(next_norm_scale * x_new) next to the f32 residual row, the SwiGLU epilogue's bf16 output and the
//     flash-attention output ARE operands already;
//   * the per-row reductions a consumer needs (sum of squares for RMSNorm's 1/rms, the operand sum that undoes the
//     +128 / zero-point offset of the int4 operands) travel as PARTIAL sums, one per producer block and row,
//     [partial][row] f32, summed by the consumer in index order (deterministic);
//   * the c_attn epilogue rotates k (RoPE, f32) and writes the bf16 K / V cache rows itself; only q goes to `qkv`.
#pragma once
#include "common.h"

struct mi355_gemm_fuse {
    // ---- consumer side: a->x is the bf16 operand itself ([M, ldx], K % 128 == 0), no staging pass
    int prestaged;
    const float* in_sx;   // [in_sx_n][M] partial operand sums ("shares"), in column order
    int in_sx_n;
    int in_ppu;           // shares per 128-column unit (a K-split consumer's slice adds the shares of its own units)
    const float* in_ss;   // [in_ss_n][M] partial sums of squares of the f32 row behind the operand, or NULL (1/rms = 1)
    int in_ss_n;
    // ---- producer side, MI355_EPI_ACCUM with f32 y: also emit the next linear's operand
    bf16_t* out_xb;       // [M, out_ld] = bf16(next_norm[n] * y_new[m, n]), or NULL
    int64_t out_ld;
    const void* next_norm;
    int next_norm_dtype;
    float* out_ss;        // [shares][M] sums of y_new^2 over a share's rows (with out_xb)
    // ACCUM: sums of the out_xb values; MI355_EPI_SWIGLU: sums of the bf16 outputs; NULL: none
    float* out_sx;        // [shares][M]     (shares: mi355_linear_gemm_plan)
    // ---- producer side, MI355_EPI_STORE of c_attn: rows [C, 2C) are rotated and written to kcache, [2C, 3C) to vcache
    const float* rope;    // [block_size, hs / 2, 2], or NULL: plain store
    const int32_t* pos;   // [M]
    bf16_t* kcache;       // [n_head, S, hs]
    bf16_t* vcache;
    int S, n_head, hs, rope_gathered;
    // != 0 (with rope, f32 y): q leaves as the prompt attention's MFMA operand — rotated, multiplied by q_scale (softmax scale x log2 e),
    // rounded once to bf16 — in the first 2 C bytes of its y row (row pitch unchanged: 2 * ldy bf16); mi355_flash_prefill takes it
    // with qkv_dtype MI355_Q_READY.  0: plain f32 q rows.
    float q_scale;
};
#define MI355_Q_READY 100  // qkv_dtype of mi355_flash_prefill: bf16 q rows, rotated and scaled by the producer (q_scale above)

// how a launch of the wide GEMM is cut: K-slices (1 = none), the partial sums ("shares") a producer writes per row and how
// many of them make a 128-column unit.  A launch split over K produces through the reduction of its slices
// (splitk_fused_reduce_kernel), one share per unit.
void mi355_linear_gemm_plan(int M, int N, int K, int R, int* ksplit, int* shares, int* shares_per_unit);
// bytes mi355_linear_gemm_workspace_bytes reserves at the END of a workspace for the chain's partial sums
size_t mi355_linear_gemm_fuse_scratch_bytes();
// as mi355_linear_gemm, with the fusion described by `f` (NULL: none)
int mi355_linear_gemm_fused(const mi355_linear_args* a, const mi355_gemm_fuse* f, void* workspace, size_t workspace_bytes,
                            mi355_stream_t stream);
// flash_prefill.hip; sx_part: [n_head][T] sums of each head's bf16 outputs per token, or NULL
int mi355_flash_prefill(const void* qkv, int qkv_dtype, int64_t ld_qkv, const float* rope, int rope_gathered,
                        const int32_t* pos, const void* kcache, const void* vcache, int T, int n_head, int S, void* y,
                        int64_t ldy, float scale, float* sx_part, hipStream_t s);

/*
 * mi355_llama.h — C ABI of libmi355llama.so, the MI355X (gfx950) native hot path for
 * lit-llama's quantized single-batch decode.
 *
 * The reference (Lightning-AI/lit-llama) has no FFI: its operator boundary is the Python
 * module API (lit_llama/quantization.py, lit_llama/model.py).  This header is the boundary
 * a maintainer would bind with ctypes from those modules; every entry
 * point names the reference code it replaces.  Conventions:
 *
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless the
 *     name ends in _host; the caller owns every buffer and keeps it alive until the stream
 *     has drained;
 *   - every function is asynchronous on `stream` (a hipStream_t passed as void*), performs
 *     no allocation and no synchronisation (graph instantiate/destroy excepted, see below);
 *   - return 0 on success, a positive hipError_t for runtime errors, a negative MI355_E_*
 *     for argument errors; mi355_last_error() returns a thread-local message;
 *   - dtype codes: MI355_F32 / MI355_BF16 / MI355_F16.
 */
#ifndef MI355_LLAMA_H
#define MI355_LLAMA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_ABI_VERSION 1

enum { MI355_F32 = 0, MI355_BF16 = 1, MI355_F16 = 2 };
enum { MI355_E_ARG = -1, MI355_E_SHAPE = -2, MI355_E_DTYPE = -3, MI355_E_STATE = -4 };

/* weight stream formats of the fast (MFMA, M<=16) linear */
enum { MI355_W_Q4 = 0, MI355_W_BF16 = 1, MI355_W_I8 = 2 };
/* epilogues of the fast linear */
enum { MI355_EPI_STORE = 0, MI355_EPI_ACCUM = 1, MI355_EPI_SWIGLU = 2 };

typedef void* mi355_stream_t; /* hipStream_t */

int mi355_version(void);
const char* mi355_last_error(void);
/* number of compute units of the current device (grid sizing on the host side) */
int mi355_num_cus(void);

/* ------------------------------------------------------------------------------------------
 * Load-time weight repack into the wave-linear stream layout consumed by mi355_linear_fast.
 *
 * Stream layout (all formats): [tile][unit][r][lane 0..63][16 B]; a tile is R row-groups of
 * 16 output rows, a unit is 128 input columns.  Lane l = (g = l >> 4, row = l & 15).
 *   Q4  : one 16-B piece per (unit, r): dword d holds k = 128u + 32g + 8d + j, j = 0..7, with
 *         slot j stored in nibble p = (j >> 1) + 4 (j & 1) (so that `(w >> 4i) & 0x000F000F`
 *         yields the bf16 pair (j = 2i, j = 2i + 1) of an MFMA A fragment directly).
 *   BF16: four 16-B pieces per (unit, r), piece d holds bf16 k = 128u + 32g + 8d + j.
 *   I8  : two 16-B pieces per (unit, r), piece e holds int8 k = 128u + 64e + 16g + j, j < 16.
 * ---------------------------------------------------------------------------------------- */

/* bytes of the repacked stream for an [N, K] matrix (rows padded to whole tiles, K to whole units);
 * pair != 0: two [N, K] matrices interleaved per tile (R must be 2) */
size_t mi355_packed_bytes(int fmt, int N, int K, int R, int pair);

/* Repack ColBlockQuantizedLinear.quant_weight (reference layout: lit_llama/quantization.py:350-359,
 * byte (n, kb) = q[n, 2kb] | q[n, 2kb+1] << 4 at q + n*stride_n + kb*stride_kb, :387-390).
 * If q1 != NULL the stream interleaves two matrices per tile (r = 0 -> q0, r = 1 -> q1; R must be 2):
 * the SwiGLU pair c_fc1 / c_fc2 of lit_llama/model.py:251-253. */
int mi355_q4_repack(const uint8_t* q0, const uint8_t* q1, int64_t stride_n, int64_t stride_kb,
                    int N, int K, int R, uint8_t* out, mi355_stream_t stream);

/* Repack a dense row-major [N, K] weight (nn.Linear.weight) of dtype `dtype` (bf16 or f32; f32 is
 * rounded to bf16 RNE) into the BF16 stream; w1 as above. */
int mi355_bf16_repack(const void* w0, const void* w1, int dtype, int N, int K, int R, void* out,
                      mi355_stream_t stream);

/* Repack the uint8 levels of an 8-bit ColBlockQuantizedLinear (quant_weight of lit_llama/quantization.py:340-423 with bits = 8: level (n, k)
 * at q + n * stride_n + k * stride_k; the reference stores it with stride_n = 1, stride_k = N) into the stream mi355_fused_step reads with
 * weight_fmt 6 (N x K bytes; N % 16 == 0, K % 128 == 0).  q1 != NULL: the c_fc1 / c_fc2 pair, R must be 2; else R = 1. */
int mi355_u8_repack(const uint8_t* q0, const uint8_t* q1, int64_t stride_n, int64_t stride_k, int N, int K, int R, uint8_t* out,
                    mi355_stream_t stream);

/* Repack a row-major int8 [N, K] matrix (Linear8bitLt weight.CB, lit_llama/quantization.py:75-77). */
int mi355_i8_repack(const int8_t* cb0, const int8_t* cb1, int N, int K, int R, int8_t* out,
                    mi355_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fast linear, M <= 16 rows of activations:  y[m, n] = epi( sum_k x'[m, k] * W[n, k] )
 *   x' = x, or RMSNorm(x) * norm_scale when norm_scale != NULL (lit_llama/model.py:270-277),
 *   rounded once to bf16 (the MFMA operand type).
 * Replaces, for the decode shapes: ColBlockQuantizedLinear.forward -> qlinear_4bit_weight ->
 * Triton linear_kernel_4bit_weight (lit_llama/quantization.py:413-423, 284-333, 187-282) for
 * MI355_W_Q4 (per-row scale/zero, i.e. tile_cols = -1), torch.nn.Linear.forward / F.linear
 * (lit_llama/model.py:197,235,252-253,120) for MI355_W_BF16.
 * ---------------------------------------------------------------------------------------- */
typedef struct mi355_linear_args {
    int32_t fmt;        /* MI355_W_* */
    int32_t R;          /* row-groups per tile of the stream (1 or 2) */
    const void* w;      /* repacked stream */
    int32_t N;          /* output features (per matrix when interleaved) */
    int32_t K;          /* input features */
    const void* x;      /* [M, ldx] activations */
    int32_t x_dtype;
    int32_t M;          /* 1..16 */
    int64_t ldx;        /* elements */
    const void* norm_scale; /* [K] or NULL */
    int32_t norm_dtype;
    float eps;
    /* Q4: per-output-row scale and zero ([N], dtype sz_dtype; [N, groups] with group_cols); *2 for the second
     * interleaved matrix */
    const void* scales;
    const void* zeros;
    const void* scales2;
    const void* zeros2;
    int32_t sz_dtype;
    int32_t epi;        /* MI355_EPI_* */
    const void* bias;   /* [N] of sz_dtype or NULL (STORE/ACCUM only) */
    void* y;            /* [M, ldy] */
    int32_t y_dtype;
    int32_t group_cols; /* Q4, mi355_linear_fast: input columns per (scale, zero) pair ("groupsize", quantization.py:284-333
                         * with tile_cols > 0): scales / zeros are then [N, ceil(K / group_cols)] row-major bf16 and
                         * group_cols must be 32 * 2^n with 16 * groups <= 2048; 0 or >= K: one pair per output row */
    int64_t ldy;
    /* launch tuning; 0 = library default */
    int32_t waves;      /* waves per workgroup (split of K) */
    int32_t grid;       /* workgroups (persistent loop over tiles) */
    int32_t prefetch;   /* accepted, ignored: one ring depth per format (4 units Q4, 2 units bf16) */
    int32_t flags;      /* reserved, pass 0 (a retired tuning knob) */
    /* activations given as split-attention partial records instead of x (x may be NULL): the prologue combines
     * them (K = attn_heads * attn_hs); layout as written by mi355_attention with n_split = attn_splits.  M = 1 */
    const float* attn_partials;
    int32_t attn_splits;
    int32_t attn_heads;
    int32_t attn_hs;
    int32_t reserved1;
    /* optional profiling aid: uint64 [grid][8] wall-clock stamps (100 MHz) written by wave 0 of every workgroup:
     * 0 entry, 1 ring issued, 2 activations staged, 3 first tile's loop done, 4 first epilogue done, 5 exit */
    uint64_t* debug_stamps;
} mi355_linear_args;

int mi355_linear_fast(const mi355_linear_args* a, mi355_stream_t stream);
/* `count` launches back to back from one host call (tuning / measurement loops that must not be host bound) */
/* Measurement hook: the NEXT mi355_linear_fast launch of the calling thread records its dispatch begin / end
 * timestamps into the two hipEvent_t (hipExtLaunchKernel start / stop events): hipEventElapsedTime(start, stop) is
 * then the launch duration as rocprofv3's kernel trace reports it.  NULL, NULL disarms. */
int mi355_debug_time_next_launch(void* start_event, void* stop_event);

/* Rows (M) one launch of mi355_linear_fast / mi355_linear_int8 can stage in the 160 KiB LDS of a workgroup for
 * this format, input width, R and wave count (<= 16); callers chunk larger M. */
int mi355_linear_max_rows(int fmt, int K, int R, int waves);

int mi355_linear_fast_batch(const mi355_linear_args* a, int count, mi355_stream_t stream);

/* The same linear for WIDE inputs (prompt prefill, no-cache evaluation: lit_llama/quantization.py:284-333 at M >= 32,
 * evaluate/full.py:120-129): LDS-tiled MFMA GEMM over the same Q4 stream (csrc/gemm.hip), any M >= 1.  Takes the
 * mi355_linear_args of mi355_linear_fast (fmt Q4; R = 1 for STORE / ACCUM, the interleaved R = 2 stream for SWIGLU; no
 * bias, no attention prologue; N and ldy multiples of 4) plus a scratch buffer of
 * mi355_linear_gemm_workspace_bytes(M, K) bytes (16-B aligned) for the staged bf16 operands and row statistics. */
size_t mi355_linear_gemm_workspace_bytes(int M, int K);
int mi355_linear_gemm(const mi355_linear_args* a, void* workspace, size_t workspace_bytes, mi355_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Generic (any shape, any M, f32 / bf16 / f16 activations) operators.  They follow the
 * reference arithmetic step by step in f32 and are used for the f32 "plumbing" configuration,
 * odd shapes (K % 128 != 0), gptq.int8 and grouped scales.
 * ---------------------------------------------------------------------------------------- */

/* y[M,N] = x[M,K] . W[N,K]^T + bias  — torch.nn.Linear (lit_llama/model.py:57,177,179,247-249) */
int mi355_linear_dense(const void* x, int64_t ldx, const void* w, const void* bias, void* y, int64_t ldy,
                       int M, int N, int K, int dtype, mi355_stream_t stream);

/* ColBlockQuantizedLinear.forward via get_weight + F.linear (lit_llama/quantization.py:392-423):
 * W[n,k] = (q[n,k] - zeros[n, k / tile_cols]) * scales[n, k / tile_cols]; bits in {4, 8}. */
int mi355_linear_colblock(const void* x, int64_t ldx, const uint8_t* qweight, int64_t stride_n, int64_t stride_kb,
                          const void* scales, const void* zeros, int sz_dtype, int n_groups, int tile_cols, int bits,
                          const void* bias, void* y, int64_t ldy, int M, int N, int K, int dtype,
                          mi355_stream_t stream);

/* ColBlockQuantizedLinear.get_weight (lit_llama/quantization.py:392-411): out[N,K] row-major */
int mi355_colblock_dequant(const uint8_t* qweight, int64_t stride_n, int64_t stride_kb, const void* scales,
                           const void* zeros, int sz_dtype, int n_groups, int tile_cols, int bits, void* out,
                           int out_dtype, int N, int K, mi355_stream_t stream);

/* RMSNorm.forward (lit_llama/model.py:270-277): y = scale * (x * rsqrt(mean(x^2) + eps)) */
int mi355_rmsnorm(const void* x, int64_t ldx, const void* scale, int scale_dtype, float eps, void* y, int64_t ldy,
                  int M, int C, int x_dtype, int y_dtype, mi355_stream_t stream);

/* apply_rope (lit_llama/model.py:306-323): x [B, T, n_head, hs] (contiguous), rope [T, hs/2, 2] f32 */
int mi355_apply_rope(const void* x, const float* rope, void* y, int B, int T, int n_head, int hs, int dtype,
                     mi355_stream_t stream);

/* MLP gate: y = silu(a) * b  (lit_llama/model.py:252) */
int mi355_swiglu(const void* a, const void* b, void* y, int64_t n, int dtype, mi355_stream_t stream);

/* residual add: y = a + b (lit_llama/model.py:166-167) */
int mi355_add(const void* a, const void* b, void* y, int64_t n, int dtype, mi355_stream_t stream);

/* nn.Embedding gather (lit_llama/model.py:102): y[m, :] = wte[idx[m], :]; idx is int32 or int64 */
int mi355_embedding(const void* idx, int idx_is_i64, const void* wte, int w_dtype, void* y, int y_dtype, int M, int C,
                    int vocab, mi355_stream_t stream);

/* greedy sampling (generate.py:68-76 with top_k = 1): out[0] = argmax(logits[0..V)) (first maximal index);
 * optionally also stored at out2[out2_pos ? out2_pos[0] + 1 : 0] (the `idx.index_copy(0, input_pos, idx_next)`
 * of generate.py:85 with the position read from device memory); pass NULL to skip */
int mi355_argmax(const float* logits, int V, int32_t* out, int32_t* out2, const int32_t* out2_pos,
                 mi355_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Attention with KV cache — CausalSelfAttention.forward without the two linears
 * (lit_llama/model.py:199-232): split q,k,v ([Q;K;V] along the feature dim, :197), RoPE on q,k in f32
 * (:204-205), cache write at slot min(pos, S-1) (:217-220; the roll of :214-218 is mi355_kv_roll),
 * softmax(q k^T / sqrt(hs)) v over slots [0, slot] (:230 with the causal mask of :93-99).
 *   rope   f32 [block_size, hs/2, 2] (cos, sin) table indexed by position, or see rope_gathered
 *   qkv    [B*T, ld_qkv]: q at column h*hs, k at C + h*hs, v at 2C + h*hs (C = n_head*hs)
 *   pos    device int32 [T]: absolute positions (RoPE row, cache slot)
 *   cache  [B, n_head, S, hs] each, dtype cache_dtype; may be NULL (no cache: keys are the T new tokens)
 *   y      [B*T, ldy] of y_dtype, head h at column h*hs
 * ---------------------------------------------------------------------------------------- */
typedef struct mi355_attn_args {
    const void* qkv;
    int32_t qkv_dtype;
    int32_t B;
    int64_t ld_qkv;
    const float* rope;  /* [block_size, hs/2, 2] */
    const int32_t* pos; /* device [T] */
    void* kcache;
    void* vcache;
    int32_t cache_dtype;
    int32_t T;
    int32_t n_head;
    int32_t hs;
    int32_t S;          /* cache length (max_seq_length) */
    int32_t y_dtype;
    void* y;
    int64_t ldy;
    void* kv_tmp;       /* when cache == NULL: scratch [2, B, n_head, T, hs] of cache_dtype */
    int32_t rope_gathered; /* != 0: `rope` holds the T rows already selected (rope_cache.index_select(0, input_pos),
                              lit_llama/model.py:94), row t is used for token t instead of row pos[t] */
    int32_t n_split;    /* > 1: flash-decoding, n_split workgroups per head write un-normalised partial records
                           [B*T][n_head][n_split][hs + 4] f32 = (max, sum, 0, 0, o[hs]) to `partials` and y is NOT
                           written: combine with mi355_attn_combine or let the following mi355_linear_fast do it
                           (attn_partials); 0 / 1: single workgroup per head writes y */
    float* partials;
    uint64_t* debug_stamps; /* optional: uint64 [workgroups][8] wall-clock stamps (100 MHz): 0 entry, 1 K/V loads
                               issued, 2 q staged, 3 rows done, 4 exit */
    /* LLaMA-Adapter (lit_llama/adapter.py:134-151), adapter_len > 0: y (or, with n_split > 1, every partial record, so
     * that the combined result is the same) additionally receives gate[h] * softmax(q ak[h]^T / sqrt(hs)) av[h] — operands as
     * mi355_adapter_args takes them, at most 64 prefix rows.  The decode kernel computes the term itself; behind the
     * many-token flash kernel it is a mi355_adapter_prefix launch. */
    const float* adapter_k;
    const float* adapter_v;
    const float* adapter_gate;
    int32_t adapter_len;
    int32_t reserved0;
} mi355_attn_args;

int mi355_attention(const mi355_attn_args* a, mi355_stream_t stream);

/* y[r, h*hs + d] from the partial records of a split attention (rows = B*T) */
int mi355_attn_combine(const float* partials, int n_split, int rows, int n_head, int hs, void* y, int y_dtype,
                       int64_t ldy, mi355_stream_t stream);

/* LLaMA-Adapter prefix term — CausalSelfAttention.forward of lit_llama/adapter.py:134-151 after the causal attention:
 *     y[row, h, :] += gate[h] * softmax(rope(q[row, h]) . ak[h]^T / sqrt(hs)) av[h]
 * q is read from the same qkv rows as mi355_attention (column h * hs) and RoPE'd the same way; ak / av are the k / v
 * projections of the adapter_prompt_length prefix rows (no RoPE, adapter.py:136-141), f32 [n_head, aT, hs]; gate is
 * gating_factor, f32 [n_head]; y ([B*T, ldy] of y_dtype) is updated in place. */
typedef struct mi355_adapter_args {
    const void* qkv;
    int32_t qkv_dtype;
    int32_t B;
    int64_t ld_qkv;
    const float* rope;     /* as mi355_attn_args.rope */
    const int32_t* pos;    /* device [T] or NULL (positions 0 .. T - 1) */
    int32_t rope_gathered; /* as mi355_attn_args.rope_gathered */
    int32_t T;
    int32_t n_head;
    int32_t hs;
    int32_t aT;            /* prefix rows (adapter_prompt_length) */
    int32_t y_dtype;
    const float* ak;
    const float* av;
    const float* gate;
    void* y;
    int64_t ldy;
} mi355_adapter_args;

int mi355_adapter_prefix(const mi355_adapter_args* a, mi355_stream_t stream);

/* torch.roll(cache, -1, dims=2) of lit_llama/model.py:217-218, in place, for both caches */
int mi355_kv_roll(void* kcache, void* vcache, int cache_dtype, int B, int n_head, int S, int hs,
                  mi355_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * LLM.int8 (Linear8bitLt, lit_llama/quantization.py:38-77 + bitsandbytes MatMul8bitLt).
 * ---------------------------------------------------------------------------------------- */

/* bnb.functional.double_quant(W) as used at lit_llama/quantization.py:69-77:
 * SCB[n] = max_k |W[n,k]| (f32), CB[n,k] = rint(127 * W[n,k] / SCB[n]); W is [N,K] of dtype (rounded to f16 first) */
int mi355_int8_quant_rows(const void* w, int dtype, int N, int K, int8_t* cb, float* scb, mi355_stream_t stream);

typedef struct mi355_int8_args {
    const int8_t* w;    /* I8 stream (mi355_i8_repack) */
    const float* scb;   /* [N] */
    int32_t N, K;
    const void* x;      /* [M, ldx] */
    int32_t x_dtype;
    int32_t M;
    int64_t ldx;
    const void* norm_scale;
    int32_t norm_dtype;
    float eps;
    float threshold;    /* 6.0 */
    int32_t R;
    const void* bias;   /* [N] f16-valued, dtype bias_dtype, or NULL */
    int32_t bias_dtype;
    int32_t epi;        /* STORE / ACCUM / SWIGLU (scb2 for the second matrix) */
    const float* scb2;
    void* y;
    int32_t y_dtype;
    int32_t waves;
    int64_t ldy;
    int32_t grid;
    int32_t prefetch;
    uint64_t* debug_stamps; /* optional: uint64 [workgroups][8] wall-clock stamps (100 MHz): 0 entry, 1 ring issued,
                               2 rows staged (f16 + scales), 3 quantised + outlier list, 4 first tile streamed,
                               5 first tile stored, 6 exit */
    /* optional: the activation row is the combine of split-attention partial records (mi355_attn_args.partials),
     * rounded to bf16 like the attention kernel's own output.  M = 1, no norm, <= 4 splits, K <= 4096. */
    const float* attn_partials;
    int32_t attn_splits, attn_heads, attn_hs, reserved0;
} mi355_int8_args;

int mi355_linear_int8(const mi355_int8_args* a, mi355_stream_t stream);

/* The same linear for WIDE inputs (M >= 32 rows: prompt prefill, no-cache evaluation; csrc/int8_gemm.hip): the outlier column
 * set is determined over ALL M rows of the call, as MatMul8bitLt does (mi355_linear_int8 takes <= 16 rows per launch and a
 * caller that chunks a longer input gets a per-chunk set), the int8 product runs on the MFMA over 128-row blocks.  bias and
 * attn_partials must be NULL; waves / grid / prefetch are ignored.  workspace: mi355_linear_int8_gemm_workspace_bytes(M, K)
 * bytes, 16-B aligned, scratch (f16 operands, int8 operands, row absmax, column mask, outlier list). */
size_t mi355_linear_int8_gemm_workspace_bytes(int M, int K);
int mi355_linear_int8_gemm(const mi355_int8_args* a, void* workspace, size_t workspace_bytes, mi355_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * GPTQ weight quantisation (offline producer of the int4 checkpoint format).
 * One block (<= 128 columns) of GPTQQuantizer.quantize's inner loop, lit_llama/quantization.py:573-592,
 * all rows in parallel, the columns in sequence:
 *     q = scale * (clamp(rint(w / scale) + zero, 0, maxq) - zero);   e = (w - q) / Hinv1[i, i];
 *     w_j -= e * Hinv1[i, j]  (j >= i);   loss = (w - q)^2 / Hinv1[i, i]^2
 * in f32 without fused multiply-adds (bit-identical to the reference's CPU arithmetic).
 * W1 [N, count] is read only (row stride ldw; the updated copy is not needed afterwards, :596);
 * Hinv1 [count, count] is the block of the upper Cholesky factor of H^-1; scale / zero are addressed as
 * [row * sz_row_stride + col * sz_col_stride] (per-row parameters: strides 1, 0);
 * Q1 (dequantised levels), Err1 and Loss1 are [N, count] outputs with row stride ldo.
 * ---------------------------------------------------------------------------------------- */
/* Row parameters of W[:, 0:cols] (find_params_weight, lit_llama/quantization.py:472-513, perchannel): scale[n] =
 * (max(row, 0) - min(row, 0)) / maxq, zero[n] = rint(-min / scale) (sym != 0: the symmetric variant). */
int mi355_gptq_row_params(const float* W, int64_t ldw, int N, int cols, int maxq, int sym, float* scale, float* zero,
                          mi355_stream_t stream);

int mi355_gptq_block(const float* W1, int64_t ldw, int N, int count, const float* Hinv1, int64_t ldh,
                     const float* scale, const float* zero, int64_t sz_row_stride, int64_t sz_col_stride,
                     int maxq, float* Q1, float* Err1, float* Loss1, int64_t ldo, mi355_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Whole-forward entry: LLaMA.forward (lit_llama/model.py:76-122) for B = 1, T <= 16 tokens with
 * KV cache, entered once per call; the T = 1 step can be captured in a hipGraph and replayed
 * (positions / token ids live in device memory so the graph is static).
 * ---------------------------------------------------------------------------------------- */
typedef struct mi355_weight {
    int32_t fmt;        /* MI355_W_* */
    int32_t R;
    const void* w;
    int32_t N, K;
    const void* scales;
    const void* zeros;
    const void* scales2;
    const void* zeros2;
    const float* scb;   /* I8 */
    const float* scb2;
    int32_t sz_dtype;
    int32_t waves;
    int32_t grid;
    int32_t prefetch;
    int32_t flags;      /* as mi355_linear_args.flags */
    int32_t group_cols; /* Q4: as mi355_linear_args.group_cols (0: one scale / zero per output row) */
} mi355_weight;

typedef struct mi355_layer {
    const void* rms1;
    const void* rms2;
    mi355_weight attn;   /* c_attn   [3C, C] */
    mi355_weight proj;   /* c_proj   [C, C]  */
    mi355_weight fc;     /* c_fc1/c_fc2 interleaved, N = n_hidden */
    mi355_weight mproj;  /* mlp.c_proj [C, n_hidden] */
    void* kcache;
    void* vcache;
    /* LLaMA-Adapter (lit_llama/adapter.py:62-171): prefix keys / values / gate of this block as mi355_adapter_args takes
     * them, adapter_len = adapter_prompt_length; 0 / NULL for a block without adapter (and for every plain model) */
    const float* adapter_k;
    const float* adapter_v;
    const float* adapter_gate;
    int32_t adapter_len;
    int32_t reserved0;
} mi355_layer;

typedef struct mi355_model {
    int32_t n_layer, n_head, n_embd, hs, n_hidden, vocab, S, block_size;
    int32_t param_dtype;  /* dtype of wte, norm scales */
    int32_t cache_dtype;
    int32_t tp_world;     /* >1: the caller all-reduces between segments (see mi355_model_segment) */
    int32_t max_T;
    float eps;
    float int8_threshold;
    const void* wte;
    const void* ln_f;
    mi355_weight lm_head;
    const float* rope;
    const mi355_layer* layers; /* host pointer to n_layer entries */
    /* device scratch owned by the caller */
    float* x;             /* [max_T, C] residual stream (f32) */
    float* qkv;           /* [max_T, 3C] */
    void* att;            /* [max_T, C] bf16 */
    void* hbuf;           /* [max_T, n_hidden] bf16 */
    float* partial;       /* [max_T, C] row-parallel partial sums (tensor parallel only) */
    float* logits;        /* [max_T, vocab] */
    int32_t* tokens;      /* [max_T] */
    int32_t* pos;         /* [max_T] */
    int32_t* next_token;  /* [1] */
    int32_t* out_tokens;  /* [block_size + 1] generated ids, indexed by pos + 1, or NULL */
    float* attn_part;     /* [n_head][attn_splits][hs + 4] partial records of the split decode attention, or NULL */
    int32_t attn_splits;  /* > 1: T == 1 steps run the attention over attn_splits workgroups per head */
    int32_t reserved0;
    void* gemm_ws;        /* scratch of mi355_linear_gemm_workspace_bytes(max_T, max K) bytes or NULL: with it, steps of
                             >= 32 tokens of an int4 model take the wide GEMM instead of chunks through the skinny kernel */
    uint64_t gemm_ws_bytes;
} mi355_model;

/* The tail of generate.py:68-85 on the device: logits / temperature, exact top-k threshold (values below the k-th
 * largest are dropped, ties kept; top_k <= 0 or >= V: none), softmax, and the inverse-CDF draw
 *     token = min { i : sum_{j <= i} p_j > u },   u = uniforms[pos[0]]  in [0, 1)  (caller's generator)
 * Writes next_token[0], out_tokens[pos[0] + 1] and, with `advance`, tokens[0] and pos[0] + 1 (so it can end a chained
 * decode step); probs_out (optional, [V]) receives the probabilities. */
int mi355_sample(const float* logits, int V, float temperature, int top_k, const float* uniforms, int32_t* next_token,
                 int32_t* out_tokens, int32_t* tokens, int32_t* pos, int advance, float* probs_out,
                 mi355_stream_t stream);

/* copy token ids / positions into the model's device slots (tiny kernel; arguments travel by value) */
int mi355_set_step(const mi355_model* m, const void* idx, int idx_is_i64, int T, int pos0, int from_next_token,
                   mi355_stream_t stream);

/* enqueue one forward over T tokens already placed by mi355_set_step.
 * logits_mode: 0 none, 1 last token only (row 0 of m->logits), 2 all T rows.
 * argmax: bit 0 appends greedy sampling (next_token[0], out_tokens[pos + 1]); bit 1 ("chained", T = 1 only, with
 * bit 0) additionally skips the embedding at the start -- m->x must already hold the embedding of tokens[0], e.g.
 * from mi355_forward_embed -- and ends the step by writing tokens[0] = argmax, pos[0] += 1 and the new token's
 * embedding row into m->x, so consecutive chained steps (or replays of a graph captured with argmax = 3) run
 * the greedy loop of generate.py:63-91 with no launch between them. */
int mi355_forward(const mi355_model* m, int T, int logits_mode, int argmax, mi355_stream_t stream);

/* The same forward cut into pieces, for callers that must interleave collectives (tensor parallel):
 *   mi355_forward_embed: x = wte[tokens]
 *   mi355_forward_segment(layer, seg_begin, seg_end): segments 0 = RMSNorm + c_attn + RoPE/KV/attention,
 *     1 = attn.c_proj, 2 = RMSNorm + c_fc1/c_fc2 + SwiGLU, 3 = mlp.c_proj.  With tp_world > 1 segments 1 and 3
 *     store the rank's partial sums in m->partial (row-parallel linears, shard_dims of
 *     scripts/convert_checkpoint.py:57-65); the caller all-reduces `partial` and calls mi355_residual_add.
 *   mi355_forward_head: ln_f + lm_head (+ greedy argmax when not sharded). */
int mi355_forward_embed(const mi355_model* m, int T, mi355_stream_t stream);
int mi355_forward_segment(const mi355_model* m, int T, int layer, int seg_begin, int seg_end, mi355_stream_t stream);
int mi355_residual_add(const mi355_model* m, int T, mi355_stream_t stream);
int mi355_forward_head(const mi355_model* m, int T, int logits_mode, int argmax, mi355_stream_t stream);

/* hipGraph of the T = 1 step (logits_mode 1).  capture/destroy synchronise the stream. */
typedef struct mi355_graph mi355_graph;
int mi355_graph_capture(const mi355_model* m, int argmax, mi355_stream_t stream, mi355_graph** out);
/* capture any sequence of this library's launches enqueued on `stream` between the two calls */
int mi355_graph_begin(mi355_stream_t stream);
int mi355_graph_end(mi355_stream_t stream, mi355_graph** out);
int mi355_graph_launch(mi355_graph* g, mi355_stream_t stream);
int mi355_graph_destroy(mi355_graph* g);


/* ------------------------------------------------------------------------------------------
 * The whole T = 1 decode step (LLaMA.forward for one token + greedy sampling, lit_llama/model.py:76-122,
 * generate.py:68-85) as ONE persistent launch: 7B-class models on a 256-CU device (kernel csrc/fused_step_ring.hip) and, per-row
 * gptq.int4 only, the 13B / 30B / 65B shapes (csrc/fused_step_wide.hip, weight_fmt 4 / 5); host entry csrc/fused_step.hip; mi355_fused_step_supported tells.  Everything the launch touches is laid out in
 * arenas so that a layer is addressed by a stride:
 *   w        Q4 streams (mi355_q4_repack) of layer l at w + l * layer_stride: c_attn (R = 1) at off_attn, attn.c_proj
 *            (R = 1) at off_proj, the interleaved c_fc1 / c_fc2 pair (R = 2) at off_fc, mlp.c_proj (R = 1) at off_mproj;
 *            layer_bytes = bytes of one layer; w_head / head_bytes: the lm_head stream (R = 1)
 *   sz       bf16 per-row scales / zeros, per layer (stride 10 C + 4 H elements):
 *            s_attn[3C] z_attn[3C] s_proj[C] z_proj[C] s_fc1[H] z_fc1[H] s_fc2[H] z_fc2[H] s_mproj[C] z_mproj[C];
 *            sz_head = s[V] z[V]
 *   norms    bf16 [n_layer][2][C] (rms_1, rms_2), then ln_f[C]
 *   kv       bf16 [n_layer][2][n_head][S][hs], rows < pos[0] valid; row pos[0] is written
 *   tokens / pos   device int32: the step's token id and position (pos[0] < S)
 *   workspace      mi355_fused_step_workspace_bytes(n_hidden) bytes, zeroed ONCE by the caller, then owned by the
 *            library (any device memory works; UNCACHED device memory — mi355_tp_buffer_alloc below — makes every hand-off
 *            3-7 % faster, the 7B int4 step 1.2 %: what lit_llama_amd's engine allocates): word 0 = abort code (0 = fine; a non-zero value after the launch means a hand-off timed
 *            out and the outputs are garbage), word 1 = step counter, word 2 = fp16 activation pairs that had to be
 *            clipped at +-65504 on the attention-output / SwiGLU edges since the caller last zeroed the word (non-zero:
 *            the steps computed with saturated activations, not what lit_llama/model.py computes; weight_fmt 3: pairs past the
 *            E4M3 limbs' range, +-448 x the edge's pre-scale), word 3 = 0x7FFFFFFF - the LOWEST pos[0] of a step that clipped (or
 *            overflowed the LLM.int8 outlier list, abort code 0x7xx) since the caller last zeroed it (0: none): everything
 *            generated from that position on has to be recomputed in a wider hand-off format (weight_fmt 3 -> 0) or by mi355_forward
 *   mode     0: logits only; 1: + greedy arg-max into next_token[0] / out_tokens[pos + 1]; 3: + chaining
 *            (tokens[0] = arg-max, pos[0] += 1), so a captured launch replays the loop of generate.py:63-91
 *   logits   f32 [vocab]
 * ---------------------------------------------------------------------------------------- */
typedef struct mi355_fused_step_args {
    const void* w;
    uint64_t layer_stride;
    uint32_t off_attn, off_proj, off_fc, off_mproj;
    uint32_t layer_bytes, head_bytes;
    const void* w_head;
    const void* sz;
    const void* sz_head;
    const void* norms;
    const void* wte;
    const float* rope;
    void* kv;
    int32_t* tokens;
    int32_t* pos;
    int32_t* next_token;
    int32_t* out_tokens;
    float* logits;
    void* workspace;
    uint64_t* debug_stamps; /* optional: uint64 [256][64] wall-clock stamps (100 MHz) of layer `reserved0`:
                               0 entry, 1 exit; gatherer: 2 x gathered, 3 q/k/v published, 4 head's q gathered, 5 attention
                               partials ready, 6 attention output published, 7 gathered, 8 c_proj published, 9 gathered,
                               10 hidden published, 11 gathered, 12 mlp.c_proj published; streamer wave 0: 20/21 c_attn
                               start / done, 23-25 attention (q staged, scores done, output done), 26/27 c_proj,
                               28/29 fc, 30/31 mlp.c_proj */
    int32_t n_layer, n_head, n_embd, hs, n_hidden, vocab, S, mode;
    float eps;
    int32_t reserved0;      /* layer to stamp when debug_stamps is given, else 0 */
    /* Grouped scales (GPTQ "groupsize", quantization.py:284-333 with tile_cols > 0): group_cols = 128 * 2^n input columns
     * per (scale, zero) pair, dividing n_embd and n_hidden (0: one pair per output row, `sz` / `sz_head` above).  Then `sz` and
     * `sz_head` are not read; `gt` holds per layer (stride gt_layer_stride bytes) the tables of c_attn, attn.c_proj, c_fc1,
     * c_fc2, mlp.c_proj in this order, `gt_head` lm_head's, each [N / 16 tiles][K / group_cols groups][16 rows] uint32 =
     * bf16 scale | bf16 zero << 16.  Register-ring implementation only; weight_fmt 0 or 3. */
    int32_t group_cols;
    /* 0: `w` / `w_head` hold int4 streams (mi355_q4_repack) as described above; 1 (round 4): BF16 streams of an unquantised model
     * (mi355_bf16_repack, same R / pair arguments: lit_llama/model.py with plain nn.Linear, BASELINE configs[1]) — `sz`, `sz_head`
     * and the group tables are not read, the hand-offs carry bf16 pairs; 2 (round 4): LLM.int8 streams (mi355_i8_repack) of a
     * Linear8bitLt model (lit_llama/quantization.py:38-77, BASELINE configs[3], threshold 6.0) — `sz` then holds per layer (stride
     * 5 C + 2 H floats) the f32 row scales SCB of c_attn[3C] attn.c_proj[C] c_fc1[H] c_fc2[H] mlp.c_proj[C], `sz_head` lm_head's [V];
     * at most 1024 outlier columns per gathered vector (more raise the abort word: such a step belongs on mi355_forward).
     * 3 (round 4; what lit_llama_amd's engine selects for per-row int4 models unless MI355_FUSED_F8=0): the streams, scales and zeros
     * of 0, computed through fp8 operands: one v_mfma_scale_f32_16x16x128_f8f6f4 per 1-KiB piece (an int4 level in a byte is the E4M3
     * code of q * 2^-9), the hand-offs carry three E4M3 limbs per activation under 16-bit tags; n_hidden <= 11776; per-row scales or (round 6) group
     * tables with at most 15 groups per streamer wave (group_cols >= 128: three MFMA columns per group).
     * A workspace that has carried hand-offs of another weight_fmt must be zeroed (all but its first 256 bytes) before the first step.
     * Register-ring implementation only.
     * 4 (round 6): the streams, scales and zeros of 0 through the WIDE-SHAPE kernel (csrc/fused_step_wide.hip): n_embd = 128 n_head with
     * 64 heads (LLaMA-65B, lit_llama/model.py:47: BASELINE configs[4] on one GPU) or 32 heads (the 7B shape, as a cross-check of the ring
     * kernel); fp16 operands and hand-offs, per-row scales only, n_hidden <= 22528.  mi355_fused_step_supported returns 2 for the shapes
     * only this format and 5 serve (1: the 7B shape, every format).
     * 5 (round 6; what lit_llama_amd's engine selects on those shapes unless MI355_FUSED_F8=0): the wide-shape kernel with the operands of 3
     * — one scaled fp8 MFMA per 1-KiB piece, three E4M3 limbs per activation under 16-bit tags; same shapes and limits as 4, the workspace
     * rule of 3 (zero it when the format changes).
     * 6 (round 6; what lit_llama_amd's engine selects for `gptq.int8` models unless MI355_FUSED_U8=0): 8-bit ColBlockQuantizedLinear levels
     * (lit_llama/quantization.py:340-423 with bits = 8, one (scale, zero) pair per row: `sz` / `sz_head` as for 0) at the 7B shape, streamed as they
     * are: `w` / `w_head` hold, per linear, [tile of 16 rows][unit of 128 columns][r][piece e = 0, 1][lane = 16 g + row][16 bytes] with byte b of
     * lane (g, row) of piece e = the level of column 128 u + 32 g + 16 e + 8 (b >> 3) + (0 4 1 5 2 6 3 7)[b & 7] (r = 0, 1: c_fc1, c_fc2 of the pair
     * stream; a linear takes N x K bytes; mi355_u8_repack builds it).  A byte is two int4 levels: the low nibbles of a unit's two pieces are one fp8 A operand (block scale
     * 2^9), the high nibbles another (2^13), both against the unit's three E4M3 limb planes; y = scale (acc - zero S) in f32.  Hand-offs,
     * tags, workspace rule and n_hidden limit of 3. */
    int32_t weight_fmt;
    const void* gt;
    const void* gt_head;
    uint64_t gt_layer_stride;
} mi355_fused_step_args;

size_t mi355_fused_step_workspace_bytes(int n_hidden);
int mi355_fused_step_supported(int n_embd, int n_head, int hs, int n_hidden, int vocab, int S);
int mi355_fused_step(const mi355_fused_step_args* a, mi355_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Tensor-parallel all-reduce of the decode step (SURVEY.md 8b `tp_allreduce(buf, n, comm)`; the partition is
 * scripts/convert_checkpoint.py:57-65): one-shot peer-write over xGMI (csrc/tp_comm.hip).  Every rank owns a
 * receive buffer of mi355_tp_comm_bytes(world, slot_floats) bytes from mi355_tp_buffer_alloc (fine-grained
 * device memory), exports it with mi355_ipc_export, and maps every peer's with mi355_ipc_open; `peer_buf[r]` is
 * rank r's buffer as seen from this process (own rank: the local pointer).  `state` is 2 zeroed uint32 of local
 * device memory: [0] step counter (mi355_tp_step_begin increments it once per forward, so tags never repeat and
 * a captured step can be replayed), [1] abort code (non-zero after a launch = a peer did not deliver in time).
 * mi355_tp_allreduce: x[i] (+)= sum over ranks, in rank order, of every rank's partial[i]; `call_index` numbers
 * the all-reduces of one forward (0, 1, 2, ...; the same sequence on every rank; consecutive calls alternate
 * buffer halves).  Returns at enqueue; the kernel completes when every peer's contribution has arrived.
 * ---------------------------------------------------------------------------------------- */
typedef struct mi355_tp_comm {
    int32_t world, rank;
    int32_t slot_floats; /* capacity of one rank's slot (>= the largest n) */
    int32_t reserved0;
    void* peer_buf[8];
    uint32_t* state;
} mi355_tp_comm;

size_t mi355_tp_comm_bytes(int world, int slot_floats);
int mi355_tp_buffer_alloc(size_t bytes, void** out);
int mi355_tp_buffer_free(void* p);
int mi355_ipc_export(void* dev_ptr, void* handle64);      /* hipIpcGetMemHandle: 64-byte handle */
int mi355_ipc_open(const void* handle64, void** out);     /* hipIpcOpenMemHandle in ANOTHER process */
int mi355_ipc_close(void* p);
int mi355_tp_step_begin(const mi355_tp_comm* c, mi355_stream_t stream);
int mi355_tp_allreduce(const mi355_tp_comm* c, const float* partial, float* x, int n, int call_index, int accumulate,
                       mi355_stream_t stream);
/* greedy sampling over vocabulary shards (lm_head split on dim 0): arg-max of the rank's [v_local] logits, exchange of
 * the `world` (value, global index) pairs, same winner on every rank (lowest index on ties); writes next_token[0],
 * out_tokens[pos[0] + 1] and, with `advance`, tokens[0] and pos[0] + 1 (the chained step of mi355_forward) */
int mi355_tp_argmax(const mi355_tp_comm* c, const float* logits_local, int v_local, int call_index, int32_t* next_token,
                    int32_t* out_tokens, int32_t* tokens, int32_t* pos, int advance, mi355_stream_t stream);

/* sizeof() of the ABI structs, for binding self-checks: 0 linear_args, 1 attn_args, 2 int8_args, 3 weight,
 * 4 layer, 5 model, 6 fused_step_args, 7 tp_comm; -1 for an unknown index */
int mi355_sizeof(int which);

#ifdef __cplusplus
}
#endif
#endif /* MI355_LLAMA_H */

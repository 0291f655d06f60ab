// Whole-forward entry of libmi355llama: LLaMA.forward (/root/reference lit_llama/model.py:76-122) for
// B = 1 and T <= max_T tokens with KV cache, entered ONCE per call from the host and, for the T = 1
// decode step, captured in a hipGraph and replayed.
//
// The reference walks 32 Blocks from Python with ~20 ATen launches each and one device->host sync per
// layer (model.py:214).  Here a layer is five launches:
//     [RMSNorm -> c_attn]  ->  [RoPE + KV write + attention]  ->  [attn.c_proj + residual]
//     -> [RMSNorm -> c_fc1/c_fc2 -> SwiGLU]  ->  [mlp.c_proj + residual]
// and token ids / positions live in device memory (mi355_set_step writes them), so the captured graph
// is static and nothing on the host depends on device results.
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "gemm_fuse.h"

// implemented in int8.hip: the LLM.int8 linear driven from a mi355_weight descriptor
int mi355_linear_int8_from_weight(const mi355_weight* w, const mi355_model* m, const void* x, int x_dtype, int M,
                                  int64_t ldx, const void* norm_scale, int epi, void* y, int y_dtype, int64_t ldy,
                                  hipStream_t stream, const float* attn_partials);

namespace {

__global__ void set_step_kernel(int32_t* tokens, int32_t* pos, const void* idx, int idx_is_i64, int T, int pos0,
                                const int32_t* next_token, int from_next) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) {
        int32_t tok;
        if (from_next)
            tok = next_token[0];
        else
            tok = idx_is_i64 ? (int32_t)((const int64_t*)idx)[t] : ((const int32_t*)idx)[t];
        tokens[t] = tok;
        pos[t] = pos0 + t;
    }
}

// Greedy chaining (generate.py:79-85 with top_k = 1, on the device): the step's argmax becomes the next step's
// token, the position advances, and the next step's embedding row (model.py:102) is written to the residual
// stream, so a chained graph needs neither a set_step nor an embedding launch between tokens.
__global__ void argmax_advance_kernel(const float* logits, int V, int32_t* next_token, int32_t* out_tokens,
                                      int32_t* tokens, int32_t* pos, const void* wte, int w_dtype, float* x, int C) {
    const int bi = block_argmax_first(logits, V);
    const int p = pos[0];
    if (w_dtype == MI355_BF16 && (C & 7) == 0) {
        const u32x4* src = (const u32x4*)((const bf16_t*)wte + (int64_t)bi * C);
        f32x4* dst = (f32x4*)x;
        for (int v = threadIdx.x; v < (C >> 3); v += blockDim.x) {
            const u32x4 r = src[v];
            f32x4 a, b;
            a[0] = __uint_as_float(r[0] << 16);
            a[1] = __uint_as_float(r[0] & 0xffff0000u);
            a[2] = __uint_as_float(r[1] << 16);
            a[3] = __uint_as_float(r[1] & 0xffff0000u);
            b[0] = __uint_as_float(r[2] << 16);
            b[1] = __uint_as_float(r[2] & 0xffff0000u);
            b[2] = __uint_as_float(r[3] << 16);
            b[3] = __uint_as_float(r[3] & 0xffff0000u);
            dst[2 * v] = a;
            dst[2 * v + 1] = b;
        }
    } else {
        for (int k = threadIdx.x; k < C; k += blockDim.x) x[k] = ld_as_f32(wte, (int64_t)bi * C + k, w_dtype);
    }
    __syncthreads();  // every thread has read pos[0] before it moves
    if (threadIdx.x == 0) {
        next_token[0] = bi;
        if (out_tokens != nullptr) out_tokens[p + 1] = bi;
        tokens[0] = bi;
        pos[0] = p + 1;
    }
}

__global__ void add_f32_kernel(float* x, const float* p, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] += p[i];
}

int run_linear_rows(const mi355_model* m, const mi355_weight& w, const void* x, int x_dtype, int M, int64_t ldx,
                    const void* norm_scale, int epi, void* y, int y_dtype, int64_t ldy, hipStream_t s,
                    const float* attn_partials);

// The wide-GEMM chain of one prompt pass with the staging passes and the K / V cache write folded into the producers'
// epilogues (gemm_fuse.h).  Decided once per mi355_forward call: every layer linear on the wide path over per-row scales,
// widths of whole 128-column units, a bf16 cache at head size 128, no adapter prefix, one GPU.
struct FuseEdge {      // partial sums ("shares") a producer leaves per row, [share][T] f32
    float* ss;         // sums of squares of the f32 residual row (residual epilogues only)
    float* sx;         // operand sums
    int n, ppu;        // shares, shares per 128-column unit (mi355_linear_gemm_plan of the producer's launch)
};
struct FuseCtx {
    bool on;
    bf16_t* xb;        // [T][n_embd]: the operand the last residual epilogue emitted (start of the GEMM workspace)
    FuseEdge proj;     // attn.c_proj -> c_fc1 / c_fc2
    FuseEdge mproj;    // mlp.c_proj -> the next layer's c_attn
    FuseEdge fc;       // SwiGLU output -> mlp.c_proj
    FuseEdge att;      // attention output -> attn.c_proj (one share per head)
    bool have_x;       // xb and the mproj edge hold the input of the next c_attn
};

void fill_linear_args(mi355_linear_args& a, const mi355_model* m, const mi355_weight& w, const void* x, int x_dtype, int M,
                      int64_t ldx, const void* norm_scale, int epi, void* y, int y_dtype, int64_t ldy) {
    memset(&a, 0, sizeof(a));
    a.fmt = w.fmt;
    a.R = w.R;
    a.w = w.w;
    a.N = w.N;
    a.K = w.K;
    a.x = x;
    a.x_dtype = x_dtype;
    a.M = M;
    a.ldx = ldx;
    a.norm_scale = norm_scale;
    a.norm_dtype = m->param_dtype;
    a.eps = m->eps;
    a.scales = w.scales;
    a.zeros = w.zeros;
    a.scales2 = w.scales2;
    a.zeros2 = w.zeros2;
    a.sz_dtype = w.sz_dtype;
    a.group_cols = w.group_cols;
    a.epi = epi;
    a.y = y;
    a.y_dtype = y_dtype;
    a.ldy = ldy;
}

bool wide_plain(const mi355_weight& w) {  // a linear the fused chain can run: int4 / bf16 stream, one scale per row
    return (w.fmt == MI355_W_Q4 || w.fmt == MI355_W_BF16) && (w.group_cols == 0 || w.group_cols >= w.K) && w.N % 4 == 0 &&
           w.K % 128 == 0;
}

FuseCtx plan_fusion(const mi355_model* m, int T) {
    FuseCtx z;
    memset(&z, 0, sizeof(z));
    if (T < 32) return z;  // (decode steps and short chunks: the streaming kernels, nothing to plan)
    // (read once per mi355_forward call, here, and nowhere on the launch path: the plan this returns is what every launch of the
    // call follows; tests/test_prefill_gpu.py toggles the variable between two calls of one process)
    const char* env = getenv("MI355_GEMM_FUSE");
    if (env != nullptr && env[0] == '0') return z;
    if (m->tp_world > 1 || m->gemm_ws == nullptr || m->hs != 128 || m->cache_dtype != MI355_BF16 ||
        (int64_t)m->S * 256 >= 0x7fffffffLL || m->n_head * m->hs != m->n_embd)
        return z;
    const int C = m->n_embd, H = m->n_hidden;
    for (int l = 0; l < m->n_layer; ++l) {
        const mi355_layer& L = m->layers[l];
        if (L.adapter_len > 0 || !wide_plain(L.attn) || !wide_plain(L.proj) || !wide_plain(L.fc) || !wide_plain(L.mproj)) return z;
        if (L.attn.N != 3 * C || L.attn.K != C || L.proj.N != C || L.proj.K != C || L.fc.N != H || L.fc.K != C || L.fc.R != 2 ||
            L.mproj.N != C || L.mproj.K != H || L.kcache == nullptr || L.vcache == nullptr)
            return z;
    }
    if (C % 128 != 0 || H % 128 != 0) return z;
    int ks, n, ppu;
    const size_t Ts = (size_t)T;
    // scratch: the tail of the workspace (mi355_linear_gemm_workspace_bytes reserves it behind the split-K partials)
    const int kmax = H > C ? H : C;
    const size_t tail = mi355_linear_gemm_fuse_scratch_bytes();
    if ((size_t)m->gemm_ws_bytes < mi355_linear_gemm_workspace_bytes(T, kmax)) return z;
    float* cur = (float*)((char*)m->gemm_ws + (((size_t)m->gemm_ws_bytes - tail) & ~(size_t)15));
    float* const end = (float*)((char*)m->gemm_ws + (size_t)m->gemm_ws_bytes);
    auto edge = [&](FuseEdge& e, int N, int K, int R, bool with_ss) {
        mi355_linear_gemm_plan(T, N, K, R, &ks, &n, &ppu);
        e.n = n;
        e.ppu = ppu;
        e.sx = cur;
        cur += (size_t)n * Ts;
        e.ss = with_ss ? cur : nullptr;
        if (with_ss) cur += (size_t)n * Ts;
    };
    edge(z.proj, C, C, 1, true);
    edge(z.mproj, C, H, 1, true);
    edge(z.fc, H, C, 2, false);
    z.att.n = m->n_head;
    z.att.ppu = 1;
    z.att.sx = cur;
    z.att.ss = nullptr;
    cur += (size_t)m->n_head * Ts;
    if (cur > end) return z;
    z.on = true;
    z.xb = (bf16_t*)m->gemm_ws;
    return z;
}

// One linear over M <= max_T rows.  The rows of one launch must fit the workgroup's LDS next to the combine
// buffers (7B: 13 rows at K = 4096, 5 at K = 11008), so wide inputs are fed in equal sub-chunks of rows: the
// prompt chunk size is set by the NARROW linears and only mlp.c_proj pays extra launches.
int run_linear(const mi355_model* m, const mi355_weight& w, const void* x, int x_dtype, int M, int64_t ldx,
               const void* norm_scale, int epi, void* y, int y_dtype, int64_t ldy, hipStream_t s,
               const float* attn_partials = nullptr) {
    // wide inputs (prompt chunks of >= 32 tokens) of an int4 model: the LDS-tiled MFMA GEMM over the same stream
    // (per-row or grouped scales)
    if ((w.fmt == MI355_W_Q4 || w.fmt == MI355_W_BF16) && M >= 32 && m->gemm_ws != nullptr &&
        attn_partials == nullptr && w.N % 4 == 0 && ldy % 4 == 0 && (w.group_cols == 0 || w.K % 128 == 0)) {
        mi355_linear_args a;
        fill_linear_args(a, m, w, x, x_dtype, M, ldx, norm_scale, epi, y, y_dtype, ldy);
        return mi355_linear_gemm(&a, m->gemm_ws, (size_t)m->gemm_ws_bytes, s);
    }
    if (w.fmt == MI355_W_I8 && M >= 32 && m->gemm_ws != nullptr && attn_partials == nullptr &&
        (size_t)m->gemm_ws_bytes >= mi355_linear_int8_gemm_workspace_bytes(M, w.K)) {
        // LLM.int8 prompt chunks: outlier columns over the whole chunk, int8 product on the MFMA (csrc/int8_gemm.hip)
        mi355_int8_args a;
        memset(&a, 0, sizeof(a));
        a.w = (const int8_t*)w.w;
        a.scb = w.scb;
        a.scb2 = w.scb2;
        a.N = w.N;
        a.K = w.K;
        a.x = x;
        a.x_dtype = x_dtype;
        a.M = M;
        a.ldx = ldx;
        a.norm_scale = norm_scale;
        a.norm_dtype = m->param_dtype;
        a.eps = m->eps;
        a.threshold = m->int8_threshold;
        a.R = w.R;
        a.epi = epi;
        a.y = y;
        a.y_dtype = y_dtype;
        a.ldy = ldy;
        return mi355_linear_int8_gemm(&a, m->gemm_ws, (size_t)m->gemm_ws_bytes, s);
    }
    int cap = mi355_linear_max_rows(w.fmt, w.K, w.R, w.waves);
    if (w.fmt == MI355_W_Q4 && w.group_cols > 0 && w.group_cols < w.K) {
        // the tile's (scale, zero) table shares the LDS with the activation rows: 2 x 16 R x groups dwords
        const int groups = (w.K + w.group_cols - 1) / w.group_cols;
        const int row_bytes = ((w.K + 127) / 128 + 1) * 256 + 16;
        cap -= (2 * 16 * w.R * groups * 4 + 16 + row_bytes - 1) / row_bytes;
        if (cap < 1) cap = 1;
    }
    MI355_CHECK_ARG(cap >= 1, MI355_E_SHAPE, "forward: a row of K=%d does not fit LDS", w.K);
    if (M <= cap) return run_linear_rows(m, w, x, x_dtype, M, ldx, norm_scale, epi, y, y_dtype, ldy, s, attn_partials);
    const int n_sub = (M + cap - 1) / cap, step = (M + n_sub - 1) / n_sub;
    const size_t xe = x_dtype == MI355_F32 ? 4 : 2, ye = y_dtype == MI355_F32 ? 4 : 2;
    for (int m0 = 0; m0 < M; m0 += step) {
        const int rows = M - m0 < step ? M - m0 : step;
        if (int rc = run_linear_rows(m, w, (const char*)x + (size_t)m0 * ldx * xe, x_dtype, rows, ldx, norm_scale, epi,
                                     (char*)y + (size_t)m0 * ldy * ye, y_dtype, ldy, s, nullptr))
            return rc;
    }
    return 0;
}

int run_linear_rows(const mi355_model* m, const mi355_weight& w, const void* x, int x_dtype, int M, int64_t ldx,
                    const void* norm_scale, int epi, void* y, int y_dtype, int64_t ldy, hipStream_t s,
                    const float* attn_partials) {
    if (w.fmt == MI355_W_I8) {
        return mi355_linear_int8_from_weight(&w, m, x, x_dtype, M, ldx, norm_scale, epi, y, y_dtype, ldy, s, attn_partials);
    }
    mi355_linear_args a;
    memset(&a, 0, sizeof(a));
    a.fmt = w.fmt;
    a.R = w.R;
    a.w = w.w;
    a.N = w.N;
    a.K = w.K;
    a.x = x;
    a.x_dtype = x_dtype;
    a.M = M;
    a.ldx = ldx;
    a.norm_scale = norm_scale;
    a.norm_dtype = m->param_dtype;
    a.eps = m->eps;
    a.scales = w.scales;
    a.zeros = w.zeros;
    a.scales2 = w.scales2;
    a.zeros2 = w.zeros2;
    a.sz_dtype = w.sz_dtype;
    a.epi = epi;
    a.bias = nullptr;
    a.y = y;
    a.y_dtype = y_dtype;
    a.ldy = ldy;
    a.waves = w.waves;
    a.grid = w.grid;
    a.prefetch = w.prefetch;
    a.flags = w.flags;
    a.group_cols = w.group_cols;
    if (attn_partials != nullptr) {
        a.attn_partials = attn_partials;
        a.attn_splits = m->attn_splits;
        a.attn_heads = m->n_head;
        a.attn_hs = m->hs;
    }
    return mi355_linear_fast(&a, s);
}

}  // namespace

extern "C" int mi355_set_step(const mi355_model* m, const void* idx, int idx_is_i64, int T, int pos0,
                              int from_next_token, mi355_stream_t stream) {
    MI355_CHECK_ARG(m != nullptr && m->tokens && m->pos, MI355_E_ARG, "set_step: null model/slots");
    MI355_CHECK_ARG(T >= 1 && T <= m->max_T, MI355_E_SHAPE, "set_step: T=%d outside 1..%d", T, m->max_T);
    MI355_CHECK_ARG(from_next_token ? (m->next_token != nullptr && T == 1) : (idx != nullptr), MI355_E_ARG,
                    "set_step: no token source");
    MI355_CHECK_ARG(pos0 >= 0 && pos0 + T <= m->block_size, MI355_E_SHAPE,
                    "set_step: positions %d..%d exceed block_size %d (RoPE table)", pos0, pos0 + T - 1, m->block_size);
    hipLaunchKernelGGL(set_step_kernel, dim3((T + 63) / 64), dim3(64), 0, (hipStream_t)stream, m->tokens, m->pos, idx,
                       idx_is_i64, T, pos0, m->next_token, from_next_token);
    MI355_LAUNCH_CHECK();
    return 0;
}

static int forward_segment_impl(const mi355_model* m, int T, int layer, int seg_begin, int seg_end, mi355_stream_t stream,
                                FuseCtx* fz);

extern "C" int mi355_forward(const mi355_model* m, int T, int logits_mode, int argmax, mi355_stream_t stream) {
    MI355_CHECK_ARG(m != nullptr && m->layers != nullptr, MI355_E_ARG, "forward: null model");
    MI355_CHECK_ARG(T >= 1 && T <= m->max_T, MI355_E_SHAPE, "forward: T=%d outside 1..%d", T, m->max_T);
    MI355_CHECK_ARG(logits_mode >= 0 && logits_mode <= 2, MI355_E_ARG, "forward: bad logits_mode");
    MI355_CHECK_ARG(!argmax || logits_mode != 0, MI355_E_ARG, "forward: argmax needs logits");
    MI355_CHECK_ARG(m->x && m->qkv && m->att && m->hbuf && m->tokens && m->pos, MI355_E_ARG, "forward: null scratch");
    MI355_CHECK_ARG(m->tp_world <= 1 || m->partial != nullptr, MI355_E_ARG, "forward: tensor parallel needs `partial`");
    MI355_CHECK_ARG(m->tp_world <= 1, MI355_E_STATE,
                    "forward: tensor-parallel models are driven segment by segment (mi355_forward_segment)");
    hipStream_t s = (hipStream_t)stream;
    const int C = m->n_embd;

    // argmax bit 1 (chained greedy step): x already holds the embedding of tokens[0], written by the previous
    // step's argmax_advance_kernel (or by mi355_forward_embed before the first chained step)
    MI355_CHECK_ARG(!(argmax & 2) || (T == 1 && logits_mode == 1 && (argmax & 1)), MI355_E_ARG,
                    "forward: a chained step is T = 1 with last-token logits and argmax");
    if (!(argmax & 2)) {
        if (int rc = mi355_embedding(m->tokens, 0, m->wte, m->param_dtype, m->x, MI355_F32, T, C, m->vocab, s)) return rc;
    }

    FuseCtx fz = plan_fusion(m, T);
    for (int l = 0; l < m->n_layer; ++l) {
        if (int rc = forward_segment_impl(m, T, l, 0, 4, s, fz.on ? &fz : nullptr)) return rc;
    }
    return mi355_forward_head(m, T, logits_mode, argmax, s);
}

// segments of layer l: 0 = RMSNorm + c_attn + attention, 1 = attn.c_proj, 2 = RMSNorm + fc + SwiGLU, 3 = mlp.c_proj.
// With tp_world > 1 segments 1 and 3 write the rank's partial sum to m->partial (the caller all-reduces it and
// calls mi355_residual_add); with tp_world == 1 they accumulate into the residual stream directly.
extern "C" int mi355_forward_segment(const mi355_model* m, int T, int layer, int seg_begin, int seg_end,
                                     mi355_stream_t stream) {
    return forward_segment_impl(m, T, layer, seg_begin, seg_end, stream, nullptr);
}

// the fused chain of a prompt pass (FuseCtx above): one segment with its producer / consumer roles
static int fused_segment(const mi355_model* m, int T, int layer, int seg, hipStream_t s, FuseCtx* fz) {
    const mi355_layer& L = m->layers[layer];
    const int C = m->n_embd, H = m->n_hidden;
    mi355_linear_args a;
    mi355_gemm_fuse f;
    memset(&f, 0, sizeof(f));
    switch (seg) {
        case 0: {
            // RMSNorm + c_attn; the epilogue rotates k and writes the K / V cache rows, q goes to qkv; then causal attention
            if (fz->have_x) {
                fill_linear_args(a, m, L.attn, fz->xb, MI355_BF16, T, C, nullptr, MI355_EPI_STORE, m->qkv, MI355_F32, 3 * C);
                f.prestaged = 1;
                f.in_sx = fz->mproj.sx;
                f.in_ss = fz->mproj.ss;
                f.in_sx_n = f.in_ss_n = fz->mproj.n;
                f.in_ppu = fz->mproj.ppu;
            } else {
                fill_linear_args(a, m, L.attn, m->x, MI355_F32, T, C, L.rms1, MI355_EPI_STORE, m->qkv, MI355_F32, 3 * C);
            }
            f.rope = m->rope;
            f.pos = m->pos;
            f.kcache = (bf16_t*)L.kcache;
            f.vcache = (bf16_t*)L.vcache;
            f.S = m->S;
            f.n_head = m->n_head;
            f.hs = m->hs;
            // q leaves the epilogue as the attention kernel's operand (rotated, scaled by softmax scale x log2 e, bf16): the kernel's
            // prologue then reads 8 MB instead of 32 MB of f32 rows + the RoPE table at T = 2048
            f.q_scale = (1.0f / sqrtf((float)m->hs)) * 1.44269504088896340736f;
            if (int rc = mi355_linear_gemm_fused(&a, &f, m->gemm_ws, (size_t)m->gemm_ws_bytes, s)) return rc;
            fz->have_x = false;
            return mi355_flash_prefill(m->qkv, MI355_Q_READY, 6 * C, m->rope, 0, m->pos, L.kcache, L.vcache, T, m->n_head, m->S, m->att,
                                       C, 1.0f / sqrtf((float)m->hs), fz->att.sx, s);
        }
        case 1:  // attn.c_proj over the attention output as it is; the residual epilogue emits the operand of c_fc1 / c_fc2
            fill_linear_args(a, m, L.proj, m->att, MI355_BF16, T, C, nullptr, MI355_EPI_ACCUM, m->x, MI355_F32, C);
            f.prestaged = 1;
            f.in_sx = fz->att.sx;
            f.in_sx_n = fz->att.n;
            f.in_ppu = fz->att.ppu;
            f.out_xb = fz->xb;
            f.out_ld = C;
            f.next_norm = L.rms2;
            f.next_norm_dtype = m->param_dtype;
            f.out_ss = fz->proj.ss;
            f.out_sx = fz->proj.sx;
            fz->have_x = true;
            return mi355_linear_gemm_fused(&a, &f, m->gemm_ws, (size_t)m->gemm_ws_bytes, s);
        case 2:  // c_fc1 / c_fc2 + SwiGLU over the emitted operand; partial operand sums of the hidden vector
            MI355_CHECK_ARG(fz->have_x, MI355_E_STATE, "forward: the fused chain lost its operand before layer %d's MLP", layer);
            fill_linear_args(a, m, L.fc, fz->xb, MI355_BF16, T, C, nullptr, MI355_EPI_SWIGLU, m->hbuf, MI355_BF16, H);
            f.prestaged = 1;
            f.in_sx = fz->proj.sx;
            f.in_ss = fz->proj.ss;
            f.in_sx_n = f.in_ss_n = fz->proj.n;
            f.in_ppu = fz->proj.ppu;
            f.out_sx = fz->fc.sx;
            fz->have_x = false;
            return mi355_linear_gemm_fused(&a, &f, m->gemm_ws, (size_t)m->gemm_ws_bytes, s);
        default:  // mlp.c_proj; the residual epilogue emits the next layer's c_attn operand
            fill_linear_args(a, m, L.mproj, m->hbuf, MI355_BF16, T, H, nullptr, MI355_EPI_ACCUM, m->x, MI355_F32, C);
            f.prestaged = 1;
            f.in_sx = fz->fc.sx;
            f.in_sx_n = fz->fc.n;
            f.in_ppu = fz->fc.ppu;
            if (layer + 1 < m->n_layer) {
                f.out_xb = fz->xb;
                f.out_ld = C;
                f.next_norm = m->layers[layer + 1].rms1;
                f.next_norm_dtype = m->param_dtype;
                f.out_ss = fz->mproj.ss;
                f.out_sx = fz->mproj.sx;
                fz->have_x = true;
            }
            return mi355_linear_gemm_fused(&a, &f, m->gemm_ws, (size_t)m->gemm_ws_bytes, s);
    }
}

static int forward_segment_impl(const mi355_model* m, int T, int layer, int seg_begin, int seg_end, mi355_stream_t stream,
                                FuseCtx* fz) {
    MI355_CHECK_ARG(m != nullptr && m->layers != nullptr, MI355_E_ARG, "forward_segment: null model");
    MI355_CHECK_ARG(layer >= 0 && layer < m->n_layer, MI355_E_ARG, "forward_segment: bad layer %d", layer);
    MI355_CHECK_ARG(seg_begin >= 0 && seg_end <= 4 && seg_begin < seg_end, MI355_E_ARG, "forward_segment: bad range");
    MI355_CHECK_ARG(T >= 1 && T <= m->max_T, MI355_E_SHAPE, "forward_segment: T=%d outside 1..%d", T, m->max_T);
    hipStream_t s = (hipStream_t)stream;
    const mi355_layer& L = m->layers[layer];
    const int C = m->n_embd;
    const int Cl = m->n_head * m->hs;  // local attention width (== C unless tensor parallel)
    const bool tp = m->tp_world > 1;
    // decode steps spread each head's K/V over attn_splits workgroups; the c_proj prologue combines the partials
    const bool split = T == 1 && m->attn_splits > 1 && m->attn_part != nullptr;
    for (int seg = seg_begin; seg < seg_end; ++seg) {
        if (fz != nullptr) {
            if (int rc = fused_segment(m, T, layer, seg, s, fz)) return rc;
            continue;
        }
        switch (seg) {
            case 0: {
                if (int rc = run_linear(m, L.attn, m->x, MI355_F32, T, C, L.rms1, MI355_EPI_STORE, m->qkv, MI355_F32,
                                        3 * Cl, s))
                    return rc;
                mi355_attn_args a;
                memset(&a, 0, sizeof(a));
                a.qkv = m->qkv;
                a.qkv_dtype = MI355_F32;
                a.B = 1;
                a.ld_qkv = 3 * Cl;
                a.rope = m->rope;
                a.pos = m->pos;
                a.kcache = L.kcache;
                a.vcache = L.vcache;
                a.cache_dtype = m->cache_dtype;
                a.T = T;
                a.n_head = m->n_head;
                a.hs = m->hs;
                a.S = m->S;
                a.y_dtype = MI355_BF16;
                a.y = m->att;
                a.ldy = Cl;
                if (split) {
                    a.n_split = m->attn_splits;
                    a.partials = m->attn_part;
                }
                if (L.adapter_len > 0) {  // LLaMA-Adapter: the gated prefix term rides along (partial records included)
                    a.adapter_k = L.adapter_k;
                    a.adapter_v = L.adapter_v;
                    a.adapter_gate = L.adapter_gate;
                    a.adapter_len = L.adapter_len;
                }
                if (int rc = mi355_attention(&a, s)) return rc;
                break;
            }
            case 1: {
                // the c_proj prologue can combine <= 4 splits of a row that fits one 16-B vector per thread
                // (mi355_linear_fast); wider shards (13B..65B single GPU) take the stand-alone combine first
                const int proj_threads = 64 * (L.proj.waves >= 4 && L.proj.waves < 8 ? L.proj.waves : 8);  // lower bound of
                                                                                             // what the launcher picks
                const bool fused = split && m->attn_splits <= 4 && Cl / 8 <= proj_threads && Cl % 8 == 0;
                if (split && !fused) {
                    if (int rc = mi355_attn_combine(m->attn_part, m->attn_splits, 1, m->n_head, m->hs, m->att, MI355_BF16,
                                                    Cl, s))
                        return rc;
                }
                if (int rc = run_linear(m, L.proj, m->att, MI355_BF16, T, Cl, nullptr,
                                        tp ? MI355_EPI_STORE : MI355_EPI_ACCUM, tp ? m->partial : m->x, MI355_F32, C, s,
                                        fused ? m->attn_part : nullptr))
                    return rc;
                break;
            }
            case 2:
                if (int rc = run_linear(m, L.fc, m->x, MI355_F32, T, C, L.rms2, MI355_EPI_SWIGLU, m->hbuf, MI355_BF16,
                                        m->n_hidden, s))
                    return rc;
                break;
            case 3:
                if (int rc = run_linear(m, L.mproj, m->hbuf, MI355_BF16, T, m->n_hidden, nullptr,
                                        tp ? MI355_EPI_STORE : MI355_EPI_ACCUM, tp ? m->partial : m->x, MI355_F32, C, s))
                    return rc;
                break;
        }
    }
    return 0;
}

extern "C" int mi355_forward_embed(const mi355_model* m, int T, mi355_stream_t stream) {
    MI355_CHECK_ARG(m != nullptr && m->x && m->tokens, MI355_E_ARG, "forward_embed: null model");
    MI355_CHECK_ARG(T >= 1 && T <= m->max_T, MI355_E_SHAPE, "forward_embed: T=%d outside 1..%d", T, m->max_T);
    return mi355_embedding(m->tokens, 0, m->wte, m->param_dtype, m->x, MI355_F32, T, m->n_embd, m->vocab,
                           (hipStream_t)stream);
}

extern "C" int mi355_residual_add(const mi355_model* m, int T, mi355_stream_t stream) {
    MI355_CHECK_ARG(m != nullptr && m->x && m->partial, MI355_E_ARG, "residual_add: null model/partial");
    const int n = T * m->n_embd;
    hipLaunchKernelGGL(add_f32_kernel, dim3((n + 255) / 256 > 64 ? 64 : (n + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, m->x, m->partial, n);
    MI355_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi355_forward_head(const mi355_model* m, int T, int logits_mode, int argmax, mi355_stream_t stream) {
    MI355_CHECK_ARG(m != nullptr, MI355_E_ARG, "forward_head: null model");
    MI355_CHECK_ARG(T >= 1 && T <= m->max_T, MI355_E_SHAPE, "forward_head: T=%d outside 1..%d", T, m->max_T);
    if (logits_mode == 0) return 0;
    MI355_CHECK_ARG(m->logits != nullptr, MI355_E_ARG, "forward_head: null logits buffer");
    hipStream_t s = (hipStream_t)stream;
    const int C = m->n_embd;
    const int V = m->lm_head.N;  // local vocab rows (== vocab unless tensor parallel)
    // ln_f is fused into the lm_head prologue (model.py:118-120)
    if (logits_mode == 1) {
        if (int rc = run_linear(m, m->lm_head, m->x + (size_t)(T - 1) * C, MI355_F32, 1, C, m->ln_f, MI355_EPI_STORE,
                                m->logits, MI355_F32, V, s))
            return rc;
    } else {
        if (int rc = run_linear(m, m->lm_head, m->x, MI355_F32, T, C, m->ln_f, MI355_EPI_STORE, m->logits, MI355_F32, V,
                                s))
            return rc;
    }
    if (argmax) {
        MI355_CHECK_ARG(m->next_token != nullptr, MI355_E_ARG, "forward_head: argmax without next_token slot");
        MI355_CHECK_ARG(m->tp_world <= 1, MI355_E_STATE, "forward_head: argmax over sharded logits is done by the caller");
        const float* row = logits_mode == 1 ? m->logits : m->logits + (size_t)(T - 1) * V;
        if (argmax & 2) {
            MI355_CHECK_ARG(T == 1 && m->tokens && m->pos && m->x, MI355_E_ARG, "forward_head: chained step needs T = 1");
            hipLaunchKernelGGL(argmax_advance_kernel, dim3(1), dim3(1024), 0, s, row, V, m->next_token, m->out_tokens,
                               m->tokens, m->pos, m->wte, m->param_dtype, m->x, C);
            MI355_LAUNCH_CHECK();
            return 0;
        }
        // generate.py:79,85: the new id lands at position input_pos[-1] + 1
        if (int rc = mi355_argmax(row, V, m->next_token, m->out_tokens, m->out_tokens ? m->pos + (T - 1) : nullptr, s))
            return rc;
    }
    return 0;
}

// ------------------------------------------------------------------------------------ hipGraph of the T = 1 step
struct mi355_graph {
    hipGraphExec_t exec;
};

extern "C" int mi355_graph_capture(const mi355_model* m, int argmax, mi355_stream_t stream, mi355_graph** out) {
    MI355_CHECK_ARG(m != nullptr && out != nullptr, MI355_E_ARG, "graph_capture: null argument");
    MI355_CHECK_ARG(stream != nullptr, MI355_E_ARG,
                    "graph_capture: the legacy default stream cannot be captured; pass a created stream");
    hipStream_t s = (hipStream_t)stream;
    MI355_HIP(hipStreamSynchronize(s));
    MI355_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = mi355_forward(m, 1, 1, argmax, s);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(s, &graph);
    if (rc != 0) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
    }
    if (e != hipSuccess || graph == nullptr) {
        mi355_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e));
        return e != hipSuccess ? (int)e : MI355_E_STATE;
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t e2 = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e2 != hipSuccess) {
        mi355_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e2));
        return (int)e2;
    }
    mi355_graph* g = new mi355_graph;
    g->exec = exec;
    *out = g;
    return 0;
}

// Capture of an arbitrary sequence of this library's launches (e.g. a tensor-parallel step: segments + peer-write
// all-reduces + sharded arg-max, lit_llama_amd/tp.py): begin, enqueue on `stream`, end.
extern "C" int mi355_graph_begin(mi355_stream_t stream) {
    MI355_CHECK_ARG(stream != nullptr, MI355_E_ARG,
                    "graph_begin: the legacy default stream cannot be captured; pass a created stream");
    hipStream_t s = (hipStream_t)stream;
    MI355_HIP(hipStreamSynchronize(s));
    MI355_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    return 0;
}

extern "C" int mi355_graph_end(mi355_stream_t stream, mi355_graph** out) {
    MI355_CHECK_ARG(out != nullptr, MI355_E_ARG, "graph_end: null argument");
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture((hipStream_t)stream, &graph);
    if (e != hipSuccess || graph == nullptr) {
        mi355_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e));
        return e != hipSuccess ? (int)e : MI355_E_STATE;
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t e2 = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e2 != hipSuccess) {
        mi355_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e2));
        return (int)e2;
    }
    mi355_graph* g = new mi355_graph;
    g->exec = exec;
    *out = g;
    return 0;
}

extern "C" int mi355_graph_launch(mi355_graph* g, mi355_stream_t stream) {
    MI355_CHECK_ARG(g != nullptr && g->exec != nullptr, MI355_E_ARG, "graph_launch: null graph");
    MI355_HIP(hipGraphLaunch(g->exec, (hipStream_t)stream));
    return 0;
}

extern "C" int mi355_graph_destroy(mi355_graph* g) {
    if (g == nullptr) return 0;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    delete g;
    return 0;
}

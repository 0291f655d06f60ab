// Tensor-parallel all-reduce for the decode step: one-shot peer-write over xGMI (SURVEY.md §8b `tp_allreduce`,
// BASELINE.json configs[4]: LLaMA-65B gptq.int4, TP = 8).
//
// The reference has no tensor parallelism (its 65B path is single-GPU, /root/reference generate.py:124-131); the
// partition is the one of scripts/convert_checkpoint.py:57-65 (lit_llama_amd/tp.py).  Per layer two row-parallel
// linears leave a partial sum of [n_embd] f32 per rank (32 KiB for 65B): a message that small is latency-bound, so a
// ring / tree (what RCCL's all-reduce runs) pays several hops of ~10 us each, 160 times per token.  Here every rank
// WRITES its partial vector straight into a receive slot of every peer (7 xGMI links, all used at once, one hop) as
// 8-byte {tag, value} granules, then sums the `world` slots of its own buffer in rank order (bit-identical on every
// rank) as soon as every tag carries the call's epoch — the data is the flag: no barrier, no second hop, and the
// residual add (x += sum, lit_llama/model.py:166-167) is fused.  The launch is an ordinary stream operation, so a whole
// TP step can be captured in a hipGraph; tags are (device step counter, call index), so replays need no resets.
//
// Buffers are fine-grained (uncached) device memory shared through HIP IPC; stores / loads are system-scope relaxed
// atomics (sc0 sc1).  Every spin is bounded by wall-clock time; a time-out raises the abort word.
#include "common.h"

namespace {

typedef unsigned long long u64;

constexpr int kMaxWorld = 8;
constexpr int kMaxCalls = 1024;          // call indices per step (2 per layer + a few)
constexpr u64 kTimeoutTicks = 400000000ull;  // 4 s of the 100 MHz wall clock (two processes may time-slice one GPU)

struct ArParams {
    u64* peer[kMaxWorld];  // receive buffer of every rank: [2 parities + arg-max plane][world][slot_floats] granules
    const float* partial;
    float* x;
    unsigned* state;       // local: [0] step counter, [1] abort code
    int world, rank, n, slot_floats, call_index, accumulate;
};

__global__ void tp_step_begin_kernel(unsigned* state) { state[0] = state[0] + 1u; }

__global__ __launch_bounds__(256) void tp_allreduce_kernel(const ArParams p) {
    const unsigned epoch = p.state[0] * (unsigned)kMaxCalls + (unsigned)p.call_index + 1u;
    const int parity = p.call_index & 1;
    const size_t slot0 = (size_t)parity * p.world * p.slot_floats;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += gridDim.x * blockDim.x) {
        const u64 g = ((u64)epoch << 32) | (u64)__float_as_uint(p.partial[i]);
        // this rank's slot in every receive buffer (its own included: the sum below reads one uniform layout)
        for (int r = 0; r < p.world; ++r)
            __hip_atomic_store(p.peer[r] + slot0 + (size_t)p.rank * p.slot_floats + i, g, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
    u64* mine = p.peer[p.rank] + slot0;
    const u64 t0 = wall_clock64();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += gridDim.x * blockDim.x) {
        float sum = 0.f;
        for (int r = 0; r < p.world; ++r) {  // rank order: every rank computes the same f32 sum
            u64 g;
            for (;;) {
                g = __hip_atomic_load(mine + (size_t)r * p.slot_floats + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((unsigned)(g >> 32) == epoch) break;
                if (wall_clock64() - t0 > kTimeoutTicks ||
                    __hip_atomic_load(p.state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    __hip_atomic_store(p.state + 1, 0x700u + (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    g = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            sum += __uint_as_float((unsigned)g);
        }
        p.x[i] = p.accumulate ? p.x[i] + sum : sum;
    }
}

// Greedy sampling over vocabulary shards (generate.py:68-85 with top_k = 1; lm_head rows are split by
// scripts/convert_checkpoint.py:57-65's dim 0): every rank takes the arg-max of its own [v_local] logits, the
// (value, global index) pairs travel as two granules per rank, and every rank picks the same winner (highest value,
// lowest index on ties, as torch.argmax on the gathered row would).  Ends the step like argmax_advance_kernel: next
// token id, position + 1 — so a captured TP step replays with no host work and no RCCL call.
struct AmParams {
    u64* peer[kMaxWorld];
    const float* logits;
    unsigned* state;
    int32_t* next_token;
    int32_t* out_tokens;
    int32_t* tokens;
    int32_t* pos;
    int world, rank, v_local, slot_floats, call_index, advance;
};

__global__ __launch_bounds__(1024) void tp_argmax_kernel(const AmParams p) {
    const int bi = block_argmax_first(p.logits, p.v_local);
    const unsigned epoch = p.state[0] * (unsigned)kMaxCalls + (unsigned)p.call_index + 1u;
    // the arg-max exchange has a plane of its own: its call index (2 n_layer) has the parity of the NEXT step's first
    // all-reduce and the plane of the all-reduce just before it is still being summed by slow peers, so in either
    // parity plane a fast rank's granules 0 / 1 could overwrite a pair a peer has not read yet (spin to the time-out)
    const size_t slot0 = (size_t)2 * p.world * p.slot_floats;
    if ((int)threadIdx.x < p.world) {
        const int r = threadIdx.x;
        u64* dst = p.peer[r] + slot0 + (size_t)p.rank * p.slot_floats;
        __hip_atomic_store(dst, ((u64)epoch << 32) | __float_as_uint(p.logits[bi]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dst + 1, ((u64)epoch << 32) | (unsigned)(bi + p.rank * p.v_local), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __shared__ float sv[kMaxWorld];
    __shared__ int si[kMaxWorld];
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    if ((int)threadIdx.x < p.world) {
        const int r = threadIdx.x;
        const u64* src = p.peer[p.rank] + slot0 + (size_t)r * p.slot_floats;
        const u64 t0 = wall_clock64();
        u64 g0 = 0, g1 = 0;
        for (;;) {
            g0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            g1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned)(g0 >> 32) == epoch && (unsigned)(g1 >> 32) == epoch) break;
            if (wall_clock64() - t0 > kTimeoutTicks) {
                __hip_atomic_store(p.state + 1, 0x780u + (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bad = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        sv[r] = __uint_as_float((unsigned)g0);
        si[r] = (int)(unsigned)g1;
    }
    __syncthreads();
    if (threadIdx.x == 0 && !bad) {
        float bv = sv[0];
        int bx = si[0];
        for (int r = 1; r < p.world; ++r)
            if (sv[r] > bv || (sv[r] == bv && si[r] < bx)) {
                bv = sv[r];
                bx = si[r];
            }
        const int ps = p.pos[0];
        p.next_token[0] = bx;
        if (p.out_tokens != nullptr) p.out_tokens[ps + 1] = bx;
        if (p.advance) {
            p.tokens[0] = bx;
            p.pos[0] = ps + 1;
        }
    }
}

}  // namespace

extern "C" size_t mi355_tp_comm_bytes(int world, int slot_floats) {
    if (world < 1 || world > kMaxWorld || slot_floats < 1) return 0;
    return (size_t)3 * world * slot_floats * sizeof(u64);  // two parity planes for the all-reduces + the arg-max plane
}

extern "C" int mi355_tp_buffer_alloc(size_t bytes, void** out) {
    MI355_CHECK_ARG(out != nullptr && bytes > 0, MI355_E_ARG, "tp_buffer_alloc: bad argument");
    void* p = nullptr;
    // fine-grained (uncached) device memory: what a peer writes over xGMI must be seen by a kernel that is already
    // running here (coarse-grained allocations are only coherent at kernel boundaries)
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    }
    MI355_CHECK_ARG(e == hipSuccess, (int)e, "tp_buffer_alloc: hipExtMallocWithFlags(%zu) failed: %s", bytes,
                    hipGetErrorString(e));
    MI355_HIP(hipMemset(p, 0, bytes));
    MI355_HIP(hipDeviceSynchronize());
    *out = p;
    return 0;
}

extern "C" int mi355_tp_buffer_free(void* p) {
    if (p != nullptr) MI355_HIP(hipFree(p));
    return 0;
}

extern "C" int mi355_ipc_export(void* dev_ptr, void* handle64) {
    MI355_CHECK_ARG(dev_ptr && handle64, MI355_E_ARG, "ipc_export: null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    hipIpcMemHandle_t h;
    MI355_HIP(hipIpcGetMemHandle(&h, dev_ptr));
    memcpy(handle64, &h, sizeof(h));
    return 0;
}

extern "C" int mi355_ipc_open(const void* handle64, void** out) {
    MI355_CHECK_ARG(handle64 && out, MI355_E_ARG, "ipc_open: null argument");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    void* p = nullptr;
    MI355_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    *out = p;
    return 0;
}

extern "C" int mi355_ipc_close(void* p) {
    if (p != nullptr) MI355_HIP(hipIpcCloseMemHandle(p));
    return 0;
}

extern "C" int mi355_tp_step_begin(const mi355_tp_comm* c, mi355_stream_t stream) {
    MI355_CHECK_ARG(c != nullptr && c->state != nullptr, MI355_E_ARG, "tp_step_begin: null comm");
    hipLaunchKernelGGL(tp_step_begin_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, c->state);
    MI355_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi355_tp_allreduce(const mi355_tp_comm* c, const float* partial, float* x, int n, int call_index,
                                  int accumulate, mi355_stream_t stream) {
    MI355_CHECK_ARG(c != nullptr && partial != nullptr && x != nullptr, MI355_E_ARG, "tp_allreduce: null argument");
    MI355_CHECK_ARG(c->world >= 1 && c->world <= kMaxWorld && c->rank >= 0 && c->rank < c->world, MI355_E_ARG,
                    "tp_allreduce: world %d rank %d", c->world, c->rank);
    MI355_CHECK_ARG(n >= 1 && n <= c->slot_floats, MI355_E_SHAPE, "tp_allreduce: n=%d exceeds the slot of %d floats", n,
                    c->slot_floats);
    MI355_CHECK_ARG(call_index >= 0 && call_index < kMaxCalls, MI355_E_ARG, "tp_allreduce: call_index %d", call_index);
    MI355_CHECK_ARG(c->state != nullptr, MI355_E_ARG, "tp_allreduce: null state");
    ArParams p;
    memset(&p, 0, sizeof(p));
    for (int r = 0; r < c->world; ++r) {
        MI355_CHECK_ARG(c->peer_buf[r] != nullptr, MI355_E_ARG, "tp_allreduce: peer buffer %d not mapped", r);
        p.peer[r] = (u64*)c->peer_buf[r];
    }
    p.partial = partial;
    p.x = x;
    p.state = c->state;
    p.world = c->world;
    p.rank = c->rank;
    p.n = n;
    p.slot_floats = c->slot_floats;
    p.call_index = call_index;
    p.accumulate = accumulate;
    const int grid = (n + 255) / 256 > 32 ? 32 : (n + 255) / 256;
    hipLaunchKernelGGL(tp_allreduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    MI355_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi355_tp_argmax(const mi355_tp_comm* c, const float* logits_local, int v_local, int call_index,
                               int32_t* next_token, int32_t* out_tokens, int32_t* tokens, int32_t* pos, int advance,
                               mi355_stream_t stream) {
    MI355_CHECK_ARG(c != nullptr && logits_local && next_token && pos, MI355_E_ARG, "tp_argmax: null argument");
    MI355_CHECK_ARG(c->world >= 1 && c->world <= kMaxWorld && c->rank >= 0 && c->rank < c->world, MI355_E_ARG,
                    "tp_argmax: world %d rank %d", c->world, c->rank);
    MI355_CHECK_ARG(v_local >= 1 && c->slot_floats >= 2, MI355_E_SHAPE, "tp_argmax: bad sizes");
    MI355_CHECK_ARG(call_index >= 0 && call_index < kMaxCalls, MI355_E_ARG, "tp_argmax: call_index %d", call_index);
    MI355_CHECK_ARG(!advance || tokens != nullptr, MI355_E_ARG, "tp_argmax: advance needs the token slot");
    AmParams p;
    memset(&p, 0, sizeof(p));
    for (int r = 0; r < c->world; ++r) {
        MI355_CHECK_ARG(c->peer_buf[r] != nullptr, MI355_E_ARG, "tp_argmax: peer buffer %d not mapped", r);
        p.peer[r] = (u64*)c->peer_buf[r];
    }
    p.logits = logits_local;
    p.state = c->state;
    p.next_token = next_token;
    p.out_tokens = out_tokens;
    p.tokens = tokens;
    p.pos = pos;
    p.world = c->world;
    p.rank = c->rank;
    p.v_local = v_local;
    p.slot_floats = c->slot_floats;
    p.call_index = call_index;
    p.advance = advance;
    hipLaunchKernelGGL(tp_argmax_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, p);
    MI355_LAUNCH_CHECK();
    return 0;
}

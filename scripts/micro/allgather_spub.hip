// (round 6 variant of allgather.hip: PUB 1 publishes through the SCALAR memory path — s_store_dwordx2 glc on uncached memory — which does not share
// the CU's in-order vector memory pipeline with the weight stream; does a publish behind a ring refill ("order 0") still cost what it did?)
// Microbenchmark for the persistent decode step: how long does ONE phase of
//   [all-gather of a 4096-value activation vector from 256 resident workgroups]  ->  [consume a weight slice that
//   was requested BEFORE the gather]  ->  [combine, publish this workgroup's 16 outputs]
// take on this box, as a function of the bytes streamed per phase?  This is the dependency chain of a fused
// decode layer: 8 streamer waves per workgroup own the weight loads (deep register ring, nt),
// one gatherer wave owns every hand-off access (8-byte {tag, value} granules, sc1 store / sc1 loads, no fences).
// Every granule carries a checkable value, so the run also verifies the protocol under load (errors, timeouts).
//   hipcc --offload-arch=gfx950 -O3 allgather.hip -o allgather && timeout 120 ./allgather
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

#ifndef SWEEP_AUX
#define SWEEP_AUX 16
#endif
#ifndef GPW
#define GPW 8  // granules per workgroup and phase: 8 = a 4096-value vector (x / attention-output edges, 16 KB of granules); round 5: -DGPW=22 =
               // the 11008-value MLP hidden vector (5632 granules, 44 KB: the hidden edge of the fused step); then only the list at the end of main()
#endif
constexpr int kGranPerWg = GPW;
constexpr int kNGran = 256 * GPW;        // granules of the vector (256 workgroups)
constexpr unsigned kSpinLimit = 1u << 18;

__device__ __forceinline__ unsigned payload(int wg, int i, int it) { return (unsigned)(wg * 131 + i * 7 + it * 2654435) ^ 0x5bd1e995u; }

// PIECES: 1-KiB wave loads per streamer wave and phase (ring held in registers), NGW: gatherer waves,
// ORD 0: the ring is refilled while it is consumed (the refill queues in front of the publish store in the CU's
// memory pipeline), ORD 1: the refill waits (third barrier) until the gatherer has published
// REP: copies of every granule (copy c read by the workgroups of XCD c % REP): does the all-gather hot-spot the
// memory channels that hold the 16-KiB vector (every one of the 256 CUs reads the same lines, past its L2)?
template <int PIECES, int NGW, int ORD, int REP = 1, int PUB = 0>
__global__ __launch_bounds__(512 + 64 * NGW) void k_phases(const uint8_t* w, unsigned w_bytes, u64* gran /*[2][G*8]*/,
                                                         unsigned* flags /*[0]=abort [1]=errors [2]=timeouts*/,
                                                         int iters, unsigned epoch0, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* xs = (unsigned*)smem;                 // [G * 8] gathered dwords
    unsigned* part = xs + 2 * kNGran;               // [8] per-wave results
    const int G = gridDim.x, bid = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_gran = G * kGranPerWg;
    if (wave < 8) {
        // ------------------------------------------------ streamer: weight loads only
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)w_bytes, 0x00020000);
        u4 ring[PIECES > 0 ? PIECES : 1];
        const unsigned cursor0 = ((unsigned)bid * 8u + wave) * (unsigned)(PIECES > 0 ? PIECES : 1) * 1024u + lane * 16u;
        const unsigned stride = (unsigned)G * 8u * (PIECES > 0 ? PIECES : 1) * 1024u;
        const int n_slots = (int)(w_bytes / stride);  // phases before the buffer is re-read (3 GiB apart: not cached)
        int phase = 0;
        u4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < PIECES; ++j)
            ring[j] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, cursor0 + j * 1024, 0, 2));
        for (int it = 0; it < iters; ++it) {
            __syncthreads();  // x gathered
            const unsigned xv = xs[(unsigned)(lane * 61 + wave * 8 + it) % (unsigned)n_gran];
            phase = phase + 1 == n_slots ? 0 : phase + 1;
            const unsigned cursor = cursor0 + (unsigned)phase * stride;
            if constexpr (ORD == 0) {
#pragma unroll
                for (int j = 0; j < PIECES; ++j) {
                    acc ^= ring[j];
                    ring[j] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, cursor + j * 1024, 0, 2));
                }
            } else {
#pragma unroll
                for (int j = 0; j < PIECES; ++j) acc ^= ring[j];
            }
            acc[0] += xv;
            if (lane == 0) part[wave] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
            __syncthreads();  // partials published
            if constexpr (ORD == 1) {
                __syncthreads();  // the gatherer has issued its publish stores
#pragma unroll
                for (int j = 0; j < PIECES; ++j)
                    ring[j] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, cursor + j * 1024, 0, 2));
            }
        }
        if ((acc[0] ^ acc[1]) == 0x12345u) sink[0] = 1.f;
    } else {
        // ------------------------------------------------ gatherer: every hand-off access
        const int gw = wave - 8;
        const __amdgpu_buffer_rsrc_t grs =
            __builtin_amdgcn_make_buffer_rsrc((void*)gran, 0, (int)(REP * 2 * n_gran * 8), 0x00020000);
        unsigned carry = 0;
        for (int it = 0; it < iters; ++it) {
            const unsigned epoch = epoch0 + (unsigned)it;
            u64* buf = gran + (size_t)(it & 1) * n_gran;
            // publish (depends on the previous phase's combine through `carry`)
            if constexpr (PUB == 1) {
                static_assert(REP == 1, "scalar publish: one copy");
                if (gw == 0) {
                    const unsigned c0 = __builtin_amdgcn_readfirstlane(carry) & 0u;
                    u64* dst = buf + (size_t)bid * kGranPerWg;
#pragma unroll
                    for (int i = 0; i < kGranPerWg; i += 2) {
                        const unsigned p0 = __builtin_amdgcn_readfirstlane(payload(bid, i, it) + c0);
                        const unsigned p1 = __builtin_amdgcn_readfirstlane(payload(bid, i + 1, it) + c0);
                        u4 v = {p0, epoch, p1, epoch};
                        asm volatile("s_store_dwordx4 %0, %1, %2 glc" ::"s"(v), "s"(dst), "i"(i * 8) : "memory");
                    }
                }
            } else if (gw == 0 && lane < kGranPerWg * REP)
                __hip_atomic_store(buf + (size_t)(lane / kGranPerWg) * 2 * n_gran + bid * kGranPerWg + lane % kGranPerWg,
                                   ((u64)epoch << 32) | (payload(bid, lane % kGranPerWg, it) + (carry & 0u)),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ORD == 1 && it > 0) __syncthreads();  // third barrier of the previous phase: publish issued, refill may go
            // sweep: this wave's share of the granules, 2 per 16-B load
            constexpr int kLoads = kNGran / NGW / 2 / 64;    // 16-B loads per lane (G = 256)
            constexpr int per_wave = kNGran / NGW;           // granules
            static_assert(kLoads * NGW * 2 * 64 == kNGran, "granules must divide over the gatherer waves");
            constexpr int loads = kLoads;
            const unsigned base = ((unsigned)((bid & 7) % REP) * 2 * n_gran + (unsigned)(it & 1) * n_gran + gw * per_wave) * 8u;
            bool done = false;
            unsigned spins = 0;
            while (!done) {
                bool ok = true;
                u4 v[kLoads];
#pragma unroll
                for (int k = 0; k < kLoads; ++k)
                    v[k] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(
                                                      grs, base + (unsigned)(k * 64 + lane) * 16u, 0, SWEEP_AUX));  // 16 = sc1 (the protocol); 1 = sc0, 17 = sc0 sc1 (uncached granules only)
#pragma unroll
                for (int k = 0; k < kLoads; ++k) {
                    ok &= v[k][1] == epoch && v[k][3] == epoch;
                    const int gi = gw * per_wave + (k * 64 + lane) * 2;
                    *(unsigned long long*)(xs + gi) = ((unsigned long long)v[k][2] << 32) | v[k][0];
                }
                done = __all(ok);
                if (!done) {
                    if (++spins > kSpinLimit ||
                        __hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                        if (lane == 0) {
                            __hip_atomic_store(flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            atomicAdd(flags + 2, 1u);
                        }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            // verify every word (all lanes of this wave wrote their share to LDS)
            if (done) {
                unsigned bad = 0;
                for (int k = 0; k < loads; ++k) {
                    const int gi = gw * per_wave + (k * 64 + lane) * 2;
                    bad += xs[gi] != payload(gi / kGranPerWg, gi % kGranPerWg, it);
                    bad += xs[gi + 1] != payload((gi + 1) / kGranPerWg, (gi + 1) % kGranPerWg, it);
                }
                if (bad) atomicAdd(flags + 1, bad);
            }
            __syncthreads();  // x gathered
            __syncthreads();  // partials published
            unsigned c = 0;
            for (int i = 0; i < 8; ++i) c ^= part[i];
            carry = c;
            if (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                // keep the barrier count balanced: the streamers run all `iters` phases
                if (ORD == 1) __syncthreads();
                for (int r = it + 1; r < iters; ++r) {
                    __syncthreads();
                    __syncthreads();
                    if (ORD == 1) __syncthreads();
                }
                carry = 0xdeadu;
                break;
            }
        }
        if (ORD == 1 && carry != 0xdeadu) __syncthreads();  // the last phase's third barrier
        if (carry == 0x9999u) sink[1] = 1.f;
    }
}

template <int PIECES, int NGW, int ORD, int REP = 1, int PUB = 0>
int run(const uint8_t* w, unsigned w_bytes, u64* gran, unsigned* flags, float* sink, int G, int iters, unsigned& epoch) {
    hipStream_t s = 0;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const size_t lds = 100 * 1024;  // one workgroup per CU, as in the real kernel
    CK(hipFuncSetAttribute((const void*)k_phases<PIECES, NGW, ORD, REP, PUB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(flags, 0, 16));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a, s));
        hipLaunchKernelGGL((k_phases<PIECES, NGW, ORD, REP, PUB>), dim3(G), dim3(512 + 64 * NGW), lds, s, w, w_bytes, gran, flags, iters,
                           epoch, sink);
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        epoch += (unsigned)iters;
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        unsigned h[4];
        CK(hipMemcpy(h, flags, 16, hipMemcpyDeviceToHost));
        if (rep == 2 || h[0] || h[1])
            printf("pieces/wave %2d (%3d KiB/CU/phase, %5.1f MB/phase) gatherers %d order %d copies %d pub %d: %6.2f us/phase  (%4.0f GB/s)  abort %u errors %u timeouts %u\n",
                   PIECES, PIECES * 8, PIECES * 8.0 * 1024 * G / 1e6, NGW, ORD, REP, PUB, ms * 1e3 / iters,
                   PIECES * 8.0 * 1024 * G / (ms * 1e3 / iters) / 1e3, h[0], h[1], h[2]);
        if (h[0]) return 1;
    }
    return 0;
}

int main(int argc, char** argv) {
    int dev = 0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    const int G = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, G);
    if (G != 256) {
        printf("written for 256 CUs\n");
        return 0;
    }
    const unsigned w_bytes = 0xC0000000u;  // 3 GiB: every phase reads fresh lines (Infinity Cache is 256 MiB)
    uint8_t* w;
    const int wuc = argc > 2 ? atoi(argv[2]) : 0;  // round 6: the weight stream from uncached memory too?
    if (wuc) CK(hipExtMallocWithFlags((void**)&w, w_bytes, hipDeviceMallocUncached));
    else CK(hipMalloc(&w, w_bytes));
    printf("weights in %s memory\n", wuc ? "UNCACHED" : "plain hipMalloc");
    CK(hipMemset(w, 1, w_bytes));
    u64* gran;
    const int uc = argc > 1 ? atoi(argv[1]) : 1;
    if (uc == 2) CK(hipExtMallocWithFlags((void**)&gran, 8 * 2 * kNGran * 8, hipDeviceMallocFinegrained));
    else if (uc == 3) CK(hipExtMallocWithFlags((void**)&gran, 8 * 2 * kNGran * 8, hipMallocSignalMemory));
    else if (uc) CK(hipExtMallocWithFlags((void**)&gran, 8 * 2 * kNGran * 8, hipDeviceMallocUncached));
    else CK(hipMalloc(&gran, 8 * 2 * kNGran * 8));
    printf("granules in %s memory\n", uc == 2 ? "FINE-GRAINED" : uc == 3 ? "SIGNAL" : uc ? "UNCACHED (hipDeviceMallocUncached)" : "plain hipMalloc");
    CK(hipMemset(gran, 0, 8 * 2 * kNGran * 8));
    unsigned* flags;
    CK(hipMalloc(&flags, 16));
    float* sink;
    CK(hipMalloc(&sink, 16));
    unsigned epoch = 1;
    const int iters = 400;
    int rc = 0;
    // idle protocol
    rc |= run<0, 2, 1, 1, 0>(w, w_bytes, gran, flags, sink, G, iters, epoch);
    rc |= run<0, 2, 1, 1, 1>(w, w_bytes, gran, flags, sink, G, iters, epoch);
    // 96 KiB per CU and phase: publish-then-refill (order 1) and refill-in-front-of-the-publish (order 0), vector vs scalar publish
    rc |= run<12, 2, 1, 1, 0>(w, w_bytes, gran, flags, sink, G, iters, epoch);
    rc |= run<12, 2, 1, 1, 1>(w, w_bytes, gran, flags, sink, G, iters, epoch);
    rc |= run<12, 2, 0, 1, 0>(w, w_bytes, gran, flags, sink, G, iters, epoch);
    rc |= run<12, 2, 0, 1, 1>(w, w_bytes, gran, flags, sink, G, iters, epoch);
    // 32 KiB per CU and phase (attn.c_proj)
    rc |= run<4, 2, 1, 1, 0>(w, w_bytes, gran, flags, sink, G, iters, epoch);
    rc |= run<4, 2, 1, 1, 1>(w, w_bytes, gran, flags, sink, G, iters, epoch);
    rc |= run<4, 2, 0, 1, 0>(w, w_bytes, gran, flags, sink, G, iters, epoch);
    rc |= run<4, 2, 0, 1, 1>(w, w_bytes, gran, flags, sink, G, iters, epoch);
    // 192 KiB per CU and phase (c_fc1/c_fc2)
    rc |= run<24, 2, 1, 1, 0>(w, w_bytes, gran, flags, sink, G, iters, epoch);
    rc |= run<24, 2, 0, 1, 0>(w, w_bytes, gran, flags, sink, G, iters, epoch);
    rc |= run<24, 2, 0, 1, 1>(w, w_bytes, gran, flags, sink, G, iters, epoch);
    printf(rc ? "FAILED\n" : "done\n");
    return rc;
}

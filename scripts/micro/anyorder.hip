// Microbenchmark: can consecutive kernels of ONE stream overlap on gfx950 (hipExtAnyOrderLaunch clears the AQL
// barrier bit) and hand data over through device-scope flags?  Emulates the decode chain: every kernel streams its
// own weights (independent of its predecessor), then needs the predecessor's 16 KiB activation vector.
//   hipcc --offload-arch=gfx950 -O3 anyorder.hip -o anyorder && timeout 120 ./anyorder
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ void k_spin(long long ticks, unsigned* out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0 && out) atomicAdd(out, 1u);
}

// One "layer": block b streams chunk b of w (n16 x 16 B per block), then reads the whole x_in (nx floats) and writes
// x_out[b*16 .. b*16+15] = f(x_in, w).  wait != nullptr: spin until *wait >= expected before touching x_in.
__global__ __launch_bounds__(512) void k_layer(const u4* w, size_t n16, const float* x_in, float* x_out, int nx,
                                               const unsigned* wait, unsigned expected, unsigned* done,
                                               unsigned* timeouts) {
    __shared__ float red[8];
    __shared__ unsigned wsum;
    const int tid = threadIdx.x;
    const u4* mine = w + (size_t)blockIdx.x * n16;
    u4 acc = {0, 0, 0, 0};
    // weight stream first: independent of the predecessor
    for (size_t i = tid; i < n16; i += 512 * 4) {
        u4 a = __builtin_nontemporal_load(mine + i);
        u4 b = i + 512 < n16 ? __builtin_nontemporal_load(mine + i + 512) : u4{0, 0, 0, 0};
        u4 c = i + 1024 < n16 ? __builtin_nontemporal_load(mine + i + 1024) : u4{0, 0, 0, 0};
        u4 d = i + 1536 < n16 ? __builtin_nontemporal_load(mine + i + 1536) : u4{0, 0, 0, 0};
        acc ^= a ^ b ^ c ^ d;
    }
    if (wait != nullptr) {
        if (tid == 0) {
            int n = 0;
            while (__hip_atomic_load(wait, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < expected) {
                if (++n > 2000000) {  // bounded: never hang the box
                    atomicAdd(timeouts, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        __atomic_thread_fence(__ATOMIC_ACQUIRE);  // agent scope by default for HIP device code (seq: workgroup?)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    float s = 0.f;
    for (int i = tid; i < nx; i += 512) s += __builtin_nontemporal_load(x_in + i);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    if (tid == 0) wsum = 0;
    __syncthreads();
    atomicXor(&wsum, acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
    __syncthreads();
    if (tid < 16) {
        float t = 0;
        for (int i = 0; i < 8; ++i) t += red[i];
        // bounded dynamics: new value depends on the whole previous vector and (weakly) on the weights
        x_out[blockIdx.x * 16 + tid] = 0.5f * t / nx + 0.25f + 1e-3f * (float)((wsum >> tid) & 1u) + 1e-2f * tid;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    }
    __syncthreads();
    if (tid == 0 && done != nullptr) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

int main() {
    hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    unsigned* cnt; CK(hipMalloc(&cnt, 4096)); CK(hipMemset(cnt, 0, 4096));
    // ---- 1. do any-order launches overlap at all?
    for (int flags : {0, (int)hipExtAnyOrderLaunch}) {
        long long ticks = 2000;  // 20 us at 100 MHz
        void* args[] = {&ticks, &cnt};
        for (int i = 0; i < 4; ++i) CK(hipExtLaunchKernel((const void*)k_spin, dim3(1), dim3(64), args, 0, s, nullptr, nullptr, flags));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 20; ++i) CK(hipExtLaunchKernel((const void*)k_spin, dim3(1), dim3(64), args, 0, s, nullptr, nullptr, flags));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("20 x spin(20us), flags=%d: %.1f us total (%s)\n", flags, ms * 1e3, ms * 1e3 < 200 ? "OVERLAPPED" : "serial");
    }
    // ---- 2. chained layers: barrier launches vs any-order + flags
    const int L = 160, G = 256, nx = G * 16;
    for (size_t mb : {8, 24}) {
        size_t bytes = mb << 20, n16 = bytes / 16 / G;
        int nbuf = (int)((600ull << 20) / bytes) + 1;
        std::vector<u4*> w(nbuf);
        for (int i = 0; i < nbuf; ++i) { CK(hipMalloc(&w[i], bytes)); CK(hipMemset(w[i], 17 * i + 3, bytes)); }
        float *xa, *xb; CK(hipMalloc(&xa, nx * 4)); CK(hipMalloc(&xb, nx * 4));
        unsigned *done, *to; CK(hipMalloc(&done, (L + 1) * 4)); CK(hipMalloc(&to, 4));
        std::vector<float> ref(nx), got(nx), init(nx, 1.0f);
        for (int mode = 0; mode < 3; ++mode) {  // 0 one stream, 1 one stream + flags, 2 two streams alternating + flags
            double best = 1e30;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemcpy(xa, init.data(), nx * 4, hipMemcpyHostToDevice));
                CK(hipMemset(done, 0, (L + 1) * 4)); CK(hipMemset(to, 0, 4));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, s));
                if (mode == 2) CK(hipStreamWaitEvent(s2, e0, 0));
                for (int l = 0; l < L; ++l) {
                    const u4* wp = w[l % nbuf];
                    const float* xi = (l & 1) ? xb : xa; float* xo = (l & 1) ? xa : xb;
                    int nxv = nx;
                    const unsigned* wait = (mode && l > 0) ? done + (l - 1) : nullptr;
                    unsigned expected = G; unsigned* dn = done + l;
                    void* args[] = {&wp, &n16, &xi, &xo, &nxv, &wait, &expected, &dn, &to};
                    CK(hipExtLaunchKernel((const void*)k_layer, dim3(G), dim3(512), args, 0,
                                          (mode == 2 && (l & 1)) ? s2 : s, nullptr, nullptr, 0));
                }
                if (mode == 2) { CK(hipEventRecord(e2, s2)); CK(hipStreamWaitEvent(s, e2, 0)); }
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms * 1e3 / L < best) best = ms * 1e3 / L;
            }
            unsigned nto = 0; CK(hipMemcpy(&nto, to, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(got.data(), (L & 1) ? xb : xa, nx * 4, hipMemcpyDeviceToHost));
            if (mode == 0) ref = got;
            int bad = 0; for (int i = 0; i < nx; ++i) bad += got[i] != ref[i];
            printf("%2zu MiB x %d layers, mode %d: %.2f us per layer, %u spin timeouts, %d / %d outputs differ from barrier run (x[0]=%f)\n",
                   mb, L, mode, best, nto, bad, nx, got[0]);
        }
        for (auto p : w) hipFree(p);
        hipFree(xa); hipFree(xb); hipFree(done); hipFree(to);
    }
    return 0;
}

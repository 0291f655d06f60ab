// Microbenchmark: cost of a dependent kernel boundary on this box (eager and hipGraph), and of minimal
// "stage + barrier" kernels.   hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor && ./launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void k_touch(float* x, float* y, int n) {  // every block reads n floats (L2), block-reduces, one store
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { float t = 0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w]; y[blockIdx.x] = t; }
}
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k_stream(const u4* w, float* y, size_t n16) {  // pure streaming read, 16 B per lane per step
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u4 acc = {0, 0, 0, 0};
    for (; i < n16; i += (size_t)gridDim.x * blockDim.x) acc ^= __builtin_nontemporal_load(w + i);
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) y[0] = 1.f;
}

template <class F> double time_loop(hipStream_t s, int iters, F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) f();
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3 / iters;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    float *x, *y; CK(hipMalloc(&x, 1 << 20)); CK(hipMalloc(&y, 1 << 20)); CK(hipMemset(x, 0, 1 << 20));
    int iters = 2000;
    printf("eager  empty<<<1,64>>>      %.2f us\n", time_loop(s, iters, [&] { k_empty<<<1, 64, 0, s>>>(nullptr); }));
    printf("eager  empty<<<256,512>>>   %.2f us\n", time_loop(s, iters, [&] { k_empty<<<256, 512, 0, s>>>(nullptr); }));
    printf("eager  empty<<<1024,512>>>  %.2f us\n", time_loop(s, iters, [&] { k_empty<<<1024, 512, 0, s>>>(nullptr); }));
    printf("eager  touch 16KB<<<256,512>>> %.2f us\n", time_loop(s, iters, [&] { k_touch<<<256, 512, 0, s>>>(x, y, 4096); }));
    printf("eager  touch 16KB<<<512,512>>> %.2f us\n", time_loop(s, iters, [&] { k_touch<<<512, 512, 0, s>>>(x, y, 4096); }));
    // graph of 160 kernels
    for (int variant = 0; variant < 3; ++variant) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 160; ++i) {
            if (variant == 0) k_empty<<<1, 64, 0, s>>>(nullptr);
            else if (variant == 1) k_empty<<<256, 512, 0, s>>>(nullptr);
            else k_touch<<<256, 512, 0, s>>>(x, y, 4096);
        }
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        double us = time_loop(s, 200, [&] { hipGraphLaunch(ge, s); });
        printf("graph  160 x %s: %.2f us per replay = %.2f us per kernel\n",
               variant == 0 ? "empty<<<1,64>>>" : variant == 1 ? "empty<<<256,512>>>" : "touch16KB<<<256,512>>>", us, us / 160);
    }
    // streaming bandwidth: 64 MiB buffers, rotate 16 of them (1 GiB > Infinity Cache)
    std::vector<u4*> bufs;
    size_t bytes = 64ull << 20;
    for (int i = 0; i < 16; ++i) { u4* p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, i + 1, bytes)); bufs.push_back(p); }
    for (int grid : {256, 512, 1024, 2048, 4096}) {
        for (int bs : {256, 512}) {
            int it = 0;
            double us = time_loop(s, 160, [&] { k_stream<<<grid, bs, 0, s>>>(bufs[it++ % 16], y, bytes / 16); });
            printf("stream 64MiB grid %4d x %3d: %.2f us -> %.0f GB/s\n", grid, bs, us, bytes / us / 1e3);
        }
    }
    // small streams (8 MiB like attn.c_proj): rotate across the 1 GiB
    for (size_t mb : {8, 24, 45}) {
        size_t b2 = mb << 20; int it = 0;
        double us = time_loop(s, 320, [&] { k_stream<<<1024, 512, 0, s>>>((u4*)((char*)bufs[it % 16] + ((it / 16) % 1) * b2), y, b2 / 16); ++it; });
        printf("stream %zu MiB grid 1024x512: %.2f us -> %.0f GB/s\n", mb, us, b2 / us / 1e3);
    }
    return 0;
}

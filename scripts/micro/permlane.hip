// Semantics check of the gfx950 row-swap instructions used as xor-16 / xor-32 lane exchanges (VALU, no LDS trip).
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ unsigned xchg16(unsigned v) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return ((threadIdx.x >> 4) & 1) ? r[0] : r[1];
}
__device__ __forceinline__ unsigned xchg32(unsigned v) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (threadIdx.x & 32) ? r[0] : r[1];
}
__global__ void k(unsigned* y) {
    y[threadIdx.x] = xchg16(threadIdx.x);
    y[64 + threadIdx.x] = xchg32(threadIdx.x);
    const auto r = __builtin_amdgcn_permlane16_swap(threadIdx.x, threadIdx.x + 100, false, false);
    y[128 + threadIdx.x] = r[0];
    y[192 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 1024); k<<<1, 64>>>(d);
    unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    int bad16 = 0, bad32 = 0;
    for (int i = 0; i < 64; ++i) { bad16 += h[i] != (unsigned)(i ^ 16); bad32 += h[64 + i] != (unsigned)(i ^ 32); }
    printf("xor16 mismatches %d, xor32 mismatches %d\n", bad16, bad32);
    printf("permlane16_swap(lane, lane+100): r0:"); for (int i = 0; i < 64; i += 8) printf(" %u", h[128 + i]);
    printf("\n                                   r1:"); for (int i = 0; i < 64; i += 8) printf(" %u", h[192 + i]);
    printf("\nxchg16:"); for (int i = 0; i < 64; i += 4) printf(" %u", h[i]);
    printf("\nxchg32:"); for (int i = 0; i < 64; i += 4) printf(" %u", h[64 + i]);
    printf("\n");
    return 0;
}

// Microbenchmark behind the fp8-limb operand path of the persistent decode step (csrc/fused_step_ring.hip FMT 3; DESIGN.md section 5;
// output of a run: profiles/r04_mx_fp8_microbench.txt):
// does v_mfma_scale_f32_16x16x128_f8f6f4 do what that path needs, and what does it cost next to the four 16x16x32 f16 MFMAs a 1-KiB
// int4 piece takes today?
//
//  (1) an int4 level q in a byte IS the OCP E4M3 encoding of q * 2^-9 (codes 0..7 are subnormals m * 2^-9, codes 8..15 the first
//      binade (8 + m) * 2^-9): with the A block scale 2^9 (E8M0 byte 136) the matrix pipe multiplies by q itself — IF it honours
//      fp8 subnormals.  Checked: D = sum_k q[i][k] * B[k][j] exactly, q over all 16 levels.
//  (2) the block scales are per LANE (lane (g, n) scales the 32 k it holds): B columns 0 / 1 / 2 with scales 2^0 / 2^-4 / 2^-8 are
//      three limbs of one activation vector riding in the token columns that are copies at M = 1.
//  (3) A and B share the map (lane group g, byte position p) -> k, whatever it is: the host reference below never names k.
//  (4) cycles per instruction, four independent accumulators, one wave per SIMD and four: scaled fp8 K = 128 against f16 K = 32.
//
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o /tmp/mx_fp8 scripts/micro/mx_fp8.hip && /tmp/mx_fp8
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// a: [64 lanes][32 bytes], b: [64 lanes][32 bytes], sb: [64] E8M0 byte per lane, d: [64 lanes][4]
__global__ void check_kernel(const uint8_t* a, const uint8_t* b, const int* sa, const int* sb, float* d) {
    const int l = threadIdx.x;
    const i32x8 av = *(const i32x8*)(a + l * 32);
    const i32x8 bv = *(const i32x8*)(b + l * 32);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    // cbsz = 0 (A: fp8 E4M3), blgp = 0 (B: fp8 E4M3); scales in byte 0 of the scale registers
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, sa[l], 0, sb[l]);
    *(f32x4*)(d + l * 4) = c;
}

template <int MODE>
__global__ void rate_kernel(long long* cycles, float* sink, int iters) {
    i32x8 av, bv;
    for (int i = 0; i < 8; ++i) {
        av[i] = 0x01020304 * (threadIdx.x & 3);
        bv[i] = 0x38383838;
    }
    f16x8 ah, bh;
    for (int i = 0; i < 8; ++i) {
        ah[i] = (_Float16)(float)(threadIdx.x & 7);
        bh[i] = (_Float16)1.0f;
    }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c0, 0, 0, 0, 136, 0, 127);
            c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c1, 0, 0, 0, 136, 0, 127);
            c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c2, 0, 0, 0, 136, 0, 127);
            c3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c3, 0, 0, 0, 136, 0, 127);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c3, 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}


// (5) the publishing side: x -> three E4M3 limbs l0 + l1 / 16 + l2 / 256 (residual splitting; v_cvt_pk_fp8_f32 / v_cvt_pk_f32_fp8 are
// OCP conversions on gfx950?  do they round to nearest even?  what happens past +-448?)
__device__ __forceinline__ void f8_limbs(float a, float b, unsigned& lo32, unsigned& hi16) {
    auto cl = [](float v) { return __builtin_amdgcn_fmed3f(v, -448.f, 448.f); };
    const int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(a), cl(b), 0, false);
    const auto f0 = __builtin_amdgcn_cvt_pk_f32_fp8(w0, false);
    float ra = a - f0[0], rb = b - f0[1];
    const int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(ra * 16.f), cl(rb * 16.f), 0, false);
    const auto f1 = __builtin_amdgcn_cvt_pk_f32_fp8(w1, false);
    ra -= f1[0] * 0.0625f;
    rb -= f1[1] * 0.0625f;
    const int w2 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(ra * 256.f), cl(rb * 256.f), 0, false);
    lo32 = ((unsigned)w0 & 0xFFFFu) | ((unsigned)w1 << 16);
    hi16 = (unsigned)w2 & 0xFFFFu;
}
__global__ void limb_kernel(const float* x, unsigned* out, int n, int raw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n + 1) return;
    unsigned lo, hi;
    if (raw) {  // unclamped conversion of the value itself: what does the instruction do past the format's range?
        lo = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false) & 0xFFFFu;
        hi = 0u;
    } else {
        f8_limbs(x[2 * i], x[2 * i + 1], lo, hi);
    }
    out[2 * i] = lo;
    out[2 * i + 1] = hi;
}

static double e4m3(uint8_t v) {  // OCP E4M3 (bias 7; 0x7F / 0xFF = NaN, not generated here)
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    const double mag = e == 0 ? ldexp(m / 8.0, -6) : ldexp(1.0 + m / 8.0, e - 7);
    return s ? -mag : mag;
}

int main() {
    uint8_t ha[64 * 32], hb[64 * 32];
    int hsa[64], hsb[64];
    srand(7);
    for (int l = 0; l < 64; ++l) {
        const int n = l & 15;
        for (int p = 0; p < 32; ++p) {
            ha[l * 32 + p] = (uint8_t)((n + 3 * p + 5 * (l >> 4)) & 15);  // every int4 level, subnormal codes included
            uint8_t v;
            do v = (uint8_t)(rand() & 0xFF); while ((v & 0x7F) == 0x7F);
            hb[l * 32 + p] = n < 3 ? v : (n == 3 ? 0x38 /* 1.0 */ : 0);
        }
        hsa[l] = 136;                                         // 2^9: the A operand is q itself
        hsb[l] = n == 0 ? 127 : n == 1 ? 123 : n == 2 ? 119 : 127;  // limb scales 2^0, 2^-4, 2^-8; column 3: ones (row sums of q)
    }
    uint8_t *da, *db;
    int *dsa, *dsb;
    float* dd;
    hipMalloc(&da, sizeof(ha));
    hipMalloc(&db, sizeof(hb));
    hipMalloc(&dsa, sizeof(hsa));
    hipMalloc(&dsb, sizeof(hsb));
    hipMalloc(&dd, 64 * 4 * sizeof(float));
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice);
    hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa, sizeof(hsa), hipMemcpyHostToDevice);
    hipMemcpy(dsb, hsb, sizeof(hsb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
    float hd[64 * 4];
    hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
    // reference: D[i][j] = sum over (g, p) of A[lane (g, i)][p] * B[lane (g, j)][p] * 2^(sa - 127) * 2^(sb_j - 127)
    double worst = 0.0;
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int j = l & 15;
        for (int r = 0; r < 4; ++r) {
            const int i = (l >> 4) * 4 + r;
            double ref = 0.0, tmax = 1e-30;
            for (int g = 0; g < 4; ++g)
                for (int p = 0; p < 32; ++p) {
                    const double t = e4m3(ha[(g * 16 + i) * 32 + p]) * ldexp(1.0, hsa[g * 16 + i] - 127) * e4m3(hb[(g * 16 + j) * 32 + p]) *
                                     ldexp(1.0, hsb[g * 16 + j] - 127);
                    ref += t;
                    tmax = fmax(tmax, fabs(t));
                }
            // the pipe does NOT accumulate the 128 products in full f32 precision (measured: errors of 1e-4..3.8e-4 of the largest
            // product, either sign): the bar is 2^-11 of the largest |a b| of the dot product
            const double err = fabs((double)hd[l * 4 + r] - ref);
            if (err > tmax / 2048.0) {
                if (bad < 8) printf("MISMATCH D[%d][%d] = %.6f, expected %.6f (largest product %.1f)\n", i, j, hd[l * 4 + r], ref, tmax);
                ++bad;
            }
            worst = fmax(worst, err / tmax);
        }
    }
    printf("check: %d of 256 results off by more than 2^-11 of the largest product (worst: %.3g of it) -> %s\n", bad, worst,
           bad == 0 ? "subnormal int4 bytes, per-lane block scales and the shared (g, p) -> k map all hold" : "FAILED");
    printf("sample: D[0][0..3] = %.4f %.4f %.4f %.4f (column 3 = row sum of q)\n", hd[0], hd[4], hd[8], hd[12]);


    {
        // (5) limb split on the device, decoded here as OCP E4M3
        const int n = 4096;
        static float hx[4096];
        static unsigned ho[4096];
        for (int i = 0; i < n; ++i) {
            const double mag = ldexp(1.0 + (rand() % 4096) / 4096.0, (i % 24) - 14);  // 2^-14 .. 2^10
            hx[i] = (float)((rand() & 1) ? -mag : mag);
        }
        float* dx;
        unsigned* dout;
        hipMalloc(&dx, sizeof(hx));
        hipMalloc(&dout, sizeof(ho));
        hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(limb_kernel, dim3(n / 2 / 64), dim3(64), 0, 0, dx, dout, n, 0);
        hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
        double worst_rel[24] = {0};
        for (int i = 0; i < n; ++i) {
            const unsigned lo = ho[(i / 2) * 2], hi = ho[(i / 2) * 2 + 1];
            const int sh = (i & 1) * 8;
            const double rec = e4m3((lo >> sh) & 0xFF) + e4m3((lo >> (16 + sh)) & 0xFF) / 16.0 + e4m3((hi >> sh) & 0xFF) / 256.0;
            const double rel = fabs(rec - hx[i]) / fabs(hx[i]);
            if (rel > worst_rel[i % 24]) worst_rel[i % 24] = rel;
        }
        printf("limbs (decoded as OCP E4M3): worst |x~ - x| / |x| per binade 2^e <= |x| < 2^(e+1):\n ");
        for (int e = 0; e < 24; ++e) printf(" e=%d:%.1e", e - 14, worst_rel[e]);
        printf("\n");
        // raw conversions: rounding and range behaviour
        const float probe[8] = {0.0009765625f /* 2^-10: half the smallest subnormal */, 0.0029296875f /* 1.5 * 2^-9 */, 17.0f /* tie 16 | 18 */,
                                19.0f /* tie 18 | 20 */, 448.0f, 464.1f, 1.0e6f, -1.0e6f};
        hipMemcpy(dx, probe, sizeof(probe), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(limb_kernel, dim3(1), dim3(64), 0, 0, dx, dout, 8, 1);
        hipMemcpy(ho, dout, 8 * sizeof(unsigned), hipMemcpyDeviceToHost);
        printf("raw v_cvt_pk_fp8_f32:");
        for (int i = 0; i < 8; ++i) {
            const unsigned byte = (ho[(i / 2) * 2] >> ((i & 1) * 8)) & 0xFF;
            printf("  %g -> 0x%02X (= %g as OCP E4M3)", probe[i], byte, (byte & 0x7F) == 0x7F ? NAN : e4m3((uint8_t)byte));
        }
        printf("\n");
    }

    long long* dc;
    float* sink;
    hipMalloc(&dc, 1024 * sizeof(long long));
    hipMalloc(&sink, 1024 * 1024 * sizeof(float));
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int waves = 1; waves <= 8; waves *= 2) {  // waves per workgroup of one CU: 4 = one per SIMD, 8 = two per SIMD
        long long hc[2] = {0, 0};
        float ms[2] = {0.f, 0.f};
        for (int mode = 0; mode < 2; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(256), dim3(64 * waves), 0, 0, dc, sink, iters);
                else hipLaunchKernelGGL(rate_kernel<1>, dim3(256), dim3(64 * waves), 0, 0, dc, sink, iters);
                hipEventRecord(e1, 0);
            }
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms[mode], e0, e1);
            hipMemcpy(&hc[mode], dc, sizeof(long long), hipMemcpyDeviceToHost);
        }
        const double per_simd = waves <= 4 ? 1.0 : waves / 4.0;  // instructions of the other waves on the same SIMD share the pipe
        const double n = 4.0 * iters * per_simd;
        printf("%d waves per CU: scaled fp8 16x16x128 %.2f ns per instruction and SIMD (%.1f clock64 ticks), f16 16x16x32 %.2f ns (%.1f) "
               "-> a 1-KiB int4 piece: 1 x %.2f = %.2f ns against 4 x %.2f = %.2f ns\n",
               waves, ms[0] * 1e6 / n, hc[0] / n, ms[1] * 1e6 / n, hc[1] / n, ms[0] * 1e6 / n, ms[0] * 1e6 / n, ms[1] * 1e6 / n,
               4.0 * ms[1] * 1e6 / n);
    }
    printf("(ns from hipEvents around the whole launch, launch overhead included: ~5 us of %d x 4 instructions)\n", iters);
    return bad != 0;
}

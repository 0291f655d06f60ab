// hipcc (ROCm 7.2, clang 20) pitfall met in csrc/fused_step_ring.hip (NOTES.md item 35): __builtin_bit_cast applied to an ELEMENT of an
// ext_vector_type value reads element 0 whatever the index.  Compile to ISA and look:
//   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only bitcast_vector_element.hip -o - | grep -E 'global_load|v_dot2'
// prints ONE global_load_dword and the same register in every v_dot2c; with the elements copied into scalars first
// (const unsigned e1 = w[1]; ... bit_cast(h2, e1)) it is a global_load_dwordx4 and four registers.
#include <hip/hip_runtime.h>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(const u4* p, float* o) {
    const u4 w = p[threadIdx.x];
    const h2 one = {(_Float16)1.0f, (_Float16)1.0f};
    float a = 0.f, b = 0.f;
    a = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w[0]), one, a, false);
    b = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w[1]), one, b, false);
    a = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w[2]), one, a, false);
    b = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w[3]), one, b, false);
    o[2 * threadIdx.x] = a;
    o[2 * threadIdx.x + 1] = b;
}

// probe: which lane's scale register (and which byte of it) scales which (row, K-block) of v_mfma_scale_f32_16x16x128_f8f6f4
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
// per-lane raw scale dwords for the first (xa) and second (xb) operand
__global__ void k_mfma(const uint8_t* A, const uint8_t* B, const int* xa, const int* xb, float* C) {
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    v8i a, b;
    const int* pa = reinterpret_cast<const int*>(A + r * 128 + q * 32);
    const int* pb = reinterpret_cast<const int*>(B + r * 128 + q * 32);
    for (int i = 0; i < 8; ++i) { a[i] = pa[i]; b[i] = pb[i]; }
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, xa[lane], 0, xb[lane]);
    for (int i = 0; i < 4; ++i) C[(4 * q + i) * 16 + r] = c[i];
}
extern "C" int probe_mfma(const void* A, const void* B, const void* xa, const void* xb, void* C) {
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, (const uint8_t*)A, (const uint8_t*)B, (const int*)xa, (const int*)xb, (float*)C);
    return (int)hipDeviceSynchronize();
}

// probe: operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3, unit scales) and of v_cvt_pk_fp8_f32 on gfx950
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k_mfma(const uint8_t* A, const uint8_t* B, float* C) {
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    v8i a, b;
    const int* pa = reinterpret_cast<const int*>(A + r * 128 + q * 32);
    const int* pb = reinterpret_cast<const int*>(B + r * 128 + q * 32);
    for (int i = 0; i < 8; ++i) { a[i] = pa[i]; b[i] = pb[i]; }
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
    for (int i = 0; i < 4; ++i) C[(4 * q + i) * 16 + r] = c[i];
}
__global__ void k_cvt(const float* x, uint8_t* y, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 < n) {
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false);
        y[2 * i] = w & 0xff; y[2 * i + 1] = (w >> 8) & 0xff;
    }
}
extern "C" int probe_mfma(const void* A, const void* B, void* C) { hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, (const uint8_t*)A, (const uint8_t*)B, (float*)C); return (int)hipDeviceSynchronize(); }
extern "C" int probe_cvt(const void* x, void* y, int n) { hipLaunchKernelGGL(k_cvt, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, (const float*)x, (uint8_t*)y, n); return (int)hipDeviceSynchronize(); }

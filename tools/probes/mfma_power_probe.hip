// What the chip sustains at its 1,400 W package-power cap: a bare MFMA loop (no LDS, no memory, no vector work) on every SIMD for ~8 s per
// shape, while tools/probes/power_probe.sh samples rocm-smi.  Prints TFLOP/s per shape; the clock it ran at is in the rocm-smi samples.
//   hipcc --offload-arch=gfx950 -O3 mfma_power_probe.hip -o mfma_power_probe;   ./mfma_power_probe [seconds]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(512) void burn(float* out, float seed, int iters)
{
    f32x4 a4[8]; f32x16 a16[4];
    for (int i = 0; i < 8; ++i) a4[i] = f32x4{seed, seed, seed, seed};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) a16[i][j] = seed;
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(seed + 0.001f * (threadIdx.x + i)); fb[i] = (__bf16)(seed - 0.002f * (threadIdx.x + 3 * i)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if constexpr (SHAPE == 16) a4[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, a4[r], 0, 0, 0);
            else                       a16[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, a16[r & 3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a4[i][0];
    for (int i = 0; i < 4; ++i) s += a16[i][0];
    if (s == 12345.678f) out[0] = s;
}

template <int SHAPE>
void run(float* d, double seconds)
{
    const int iters = 20000;
    const double flop_per_launch = 256.0 * 8 * 64 / 64 * iters * 8 * (SHAPE == 16 ? 16.0 * 16 * 32 * 2 : 32.0 * 32 * 16 * 2);   // 256 blocks x 8 waves
    burn<SHAPE><<<256, 512>>>(d, 0.5f, iters);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    double dt = 0;
    do {
        for (int i = 0; i < 8; ++i) burn<SHAPE><<<256, 512>>>(d, 0.5f, iters);
        hipDeviceSynchronize();
        n += 8;
        dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (dt < seconds);
    printf("mfma %dx%d: %.0f TFLOP/s sustained over %.1f s (non-zero operands)\n", SHAPE, SHAPE, flop_per_launch * n / dt / 1e12, dt);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 8.0;
    float* d; hipMalloc(&d, 64);
    run<16>(d, seconds);
    run<32>(d, seconds);
    return 0;
}

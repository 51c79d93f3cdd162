// How many VALU / transcendental instructions fit in the shadow of one MFMA on gfx950, for the 16x16x32 and the 32x32x16 bf16 shapes.
// One wave per SIMD (grid = 256 CUs x 4 waves), REP MFMAs on independent accumulators each followed by K VALU ops; prints the
// shader clocks (s_memtime) per MFMA.   hipcc --offload-arch=gfx950 -O3 mfma_issue_probe.hip -o mfma_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE, int K, int TRANS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe(unsigned long long* out, float seed)
{
    f32x4 a4[8]; f32x16 a16[4];
    for (int i = 0; i < 8; ++i) a4[i] = f32x4{seed, seed, seed, seed};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) a16[i][j] = seed;
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(seed + i); fb[i] = (__bf16)(seed - i); }
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = seed * (i + 1);
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if constexpr (SHAPE == 16) a4[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, a4[r], 0, 0, 0);
            else                       a16[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, a16[r & 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if constexpr (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k]));
                else                 asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[k]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a4[i][0];
    for (int i = 0; i < 4; ++i) s += a16[i][0];
    for (int i = 0; i < 12; ++i) s += v[i];
    if (s == 12345.678f) out[1] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}

template <int SHAPE, int K, int TRANS, int WAVES>
void run(unsigned long long* d)
{
    probe<SHAPE, K, TRANS, WAVES><<<256, 64 * WAVES>>>(d, 0.f);
    probe<SHAPE, K, TRANS, WAVES><<<256, 64 * WAVES>>>(d, 0.f);
    unsigned long long h = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("mfma %2d  waves/SIMD %d  %d %s per MFMA: %6.1f clocks per MFMA (per wave)\n", SHAPE, WAVES / 4, K, TRANS ? "v_exp" : "v_fma", (double)h / (64 * 8));
}
template <int SHAPE, int TRANS, int WAVES, int... Ks> void sweep(unsigned long long* d) { (run<SHAPE, Ks, TRANS, WAVES>(d), ...); }

int main()
{
    unsigned long long* d; hipMalloc(&d, 64);
    sweep<16, 0, 4, 0, 1, 2, 3, 4, 6>(d);
    sweep<32, 0, 4, 0, 2, 4, 5, 6, 7, 8, 10>(d);
    sweep<16, 1, 4, 0, 1, 2>(d);
    sweep<32, 1, 4, 0, 1, 2, 3, 4>(d);
    sweep<16, 0, 8, 0, 2, 4, 6>(d);
    sweep<32, 0, 8, 0, 4, 6, 8, 12>(d);
    sweep<16, 1, 8, 0, 1, 2>(d);
    sweep<32, 1, 8, 0, 1, 2, 3, 4>(d);
    return 0;
}

// How many VALU / transcendental / LDS-read / global-load instructions fit in the shadow of one MFMA on gfx950, for the 16x16x32 and
// the 32x32x16 bf16 shapes, with one and with two waves per SIMD (256 blocks of 4 or 8 waves).  REP MFMAs on independent accumulators
// each followed by K fillers; prints shader clocks (s_memtime) per MFMA: the SLOWEST wave of block 0 (with two waves per SIMD the older
// wave wins the arbitration, so the first wave alone looks undisturbed).
//   hipcc --offload-arch=gfx950 -O3 mfma_issue_probe.hip -o mfma_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// FILL: 0 v_fma_f32, 1 v_exp_f32, 2 ds_read_b128, 3 global_load_dwordx4 (L2 / L1 hits: one 1-KiB block)
template <int SHAPE, int K, int FILL, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe(unsigned long long* out, float seed, const unsigned char* gsrc)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[8192];
    f32x4 a4[8]; f32x16 a16[4];
    for (int i = 0; i < 8; ++i) a4[i] = f32x4{seed, seed, seed, seed};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) a16[i][j] = seed;
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(seed + i); fb[i] = (__bf16)(seed - i); }
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = seed * (i + 1);
    u32x4 ld[4] = {};
    const unsigned laddr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)lds) + (threadIdx.x & 63) * 16;
    const unsigned goff = (threadIdx.x & 63) * 16;
    reinterpret_cast<float*>(lds)[threadIdx.x] = seed;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if constexpr (SHAPE == 16) a4[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, a4[r], 0, 0, 0);
            else                       a16[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, a16[r & 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if constexpr (FILL == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[k]));
                else if constexpr (FILL == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k]));
                else if constexpr (FILL == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[k & 3]) : "v"(laddr), "n"(1024 * (k & 3)) : "memory");
                else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ld[k & 3]) : "v"(goff), "s"(gsrc) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (FILL == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (FILL == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a4[i][0];
    for (int i = 0; i < 4; ++i) s += a16[i][0];
    for (int i = 0; i < 12; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += __builtin_bit_cast(float, ld[i][0]);
    if (s == 12345.678f) out[1] = 1;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicMax(out, t1 - t0);
}

template <int SHAPE, int K, int FILL, int WAVES>
void run(unsigned long long* d, const unsigned char* g)
{
    static const char* names[] = {"v_fma_f32", "v_exp_f32", "ds_read_b128", "global_load_dwordx4"};
    probe<SHAPE, K, FILL, WAVES><<<256, 64 * WAVES>>>(d, 0.f, g);
    hipMemset(d, 0, 16);
    probe<SHAPE, K, FILL, WAVES><<<256, 64 * WAVES>>>(d, 0.f, g);
    unsigned long long h = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("mfma %2d  waves/SIMD %d  %2d %-20s per MFMA: %6.1f clocks per MFMA (slowest wave)\n", SHAPE, WAVES / 4, K, names[FILL], (double)h / (64 * 8));
}
template <int SHAPE, int FILL, int WAVES, int... Ks> void sweep(unsigned long long* d, const unsigned char* g) { (run<SHAPE, Ks, FILL, WAVES>(d, g), ...); }

int main()
{
    unsigned long long* d; hipMalloc(&d, 64);
    unsigned char* g; hipMalloc(&g, 4096); hipMemset(g, 0, 4096);
    sweep<16, 0, 4, 0, 1, 2, 3, 4>(d, g);
    sweep<32, 0, 4, 0, 4, 5, 6, 8>(d, g);
    sweep<16, 0, 8, 0, 1, 2, 3, 4>(d, g);
    sweep<32, 0, 8, 0, 2, 4, 5, 6, 8>(d, g);
    sweep<16, 1, 8, 1, 2>(d, g);
    sweep<32, 1, 8, 1, 2, 3, 4>(d, g);
    sweep<16, 2, 4, 1, 2, 3>(d, g);
    sweep<16, 2, 8, 1, 2, 3>(d, g);
    sweep<32, 2, 8, 1, 2, 3, 4>(d, g);
    sweep<16, 3, 4, 1, 2>(d, g);
    sweep<16, 3, 8, 1, 2>(d, g);
    sweep<32, 3, 8, 1, 2, 4>(d, g);
    return 0;
}

// gemm_8phase.h -- 256x256x64 LDS-DMA GEMM with a phase-interleaved schedule (k_gemm_8ph).
//
// Same block tile, LDS image (two 64-KiB stages: A [256][64] then B [256][64], 128-byte rows, chunk ^ ((row>>1)&7))
// and epilogue as k_gemm_dma<2,4,8,4>; what changes is WHEN things happen.  The two-stage kernel requests all of
// K-tile k+1 after the barrier and waits for all of it (vmcnt(0)) one K-tile later: one tile of prefetch distance, a
// full drain per tile, every wave reading fragments and multiplying in lock step.  Here
//   * a K-tile is worked in 4 PHASES, one 128x128 quadrant of the block tile each, in the order (A-half, B-half) =
//     (0,0) (0,1) (1,1) (1,0); a wave's 128x64 output spans all four quadrants (64x32 of each = 16 MFMAs per phase),
//     so in any phase ALL waves read the same A half and B half:  half-tiles die one by one -- A0 and B0 after phase 0
//     (B0 stays in registers until phase 3), B1 after phase 1, A1 after phase 2 -- and are re-staged one per phase,
//     two phases after their last read:  phase 0: B1(t+1), 1: A1(t+1), 2: A0(t+2), 3: B0(t+2);
//   * every half-tile is requested 5 or 6 phases before its first read; the wait is COUNTED -- `s_waitcnt vmcnt(6)` at
//     the head of each phase retires everything requested four or more phases ago and leaves three half-tiles in
//     flight -- and the phase's barrier publishes it;  "half" = the rows the waves read together: A half h = tile rows
//     {wr*128 + h*64 ..+64}, B half h = {wc*64 + h*32 ..+32}, so a wave keeps a contiguous 128x64 output;
//   * the two wave groups (wr = 0 / 1; a SIMD hosts one wave of each) run half a phase apart: while one group issues
//     its DMA requests and ds_reads, the other runs its 16 MFMAs at raised priority, then they swap -- the matrix pipe
//     sees back-to-back MFMA clusters and fragment-read latency sits under the other group's cluster.
// Hazards (P = global phase index, a phase = LOAD step | barrier | MFMA step | barrier, group 1 one step late):
//   RAW  a half-tile requested in phase I is retired by the vmcnt(6) at the head of phase I+4 in every wave, published
//        by that phase's barrier for the late group too, hence readable from phase I+5: all four distances are >= 5.
//   WAR  reads issued in LOAD(P') are in registers after the lgkmcnt(0) heading MFMA(P') -- for the late group one
//        step later -- so the region may be re-requested from LOAD(P'+2) on: all four re-stagings are >= 2 phases late.
// Preconditions beyond k_gemm_dma's: M and N multiples of 256 (no edge clamping), a1 (if any) row-linear.
#pragma once
#include "gemm_dma.h"

namespace ncsn {

struct Cfg8ph {
    static constexpr int BM_ = 256, BN_ = 256, THREADS = 512, STAGE_BYTES = 65536;
    using Epi = EpiCfg<2, 4, 8, 4, 2 * STAGE_BYTES + 4096>;
    static constexpr int LDS_BYTES = Epi::SLAB_BYTES > 2 * STAGE_BYTES ? Epi::SLAB_BYTES : 2 * STAGE_BYTES;
};

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void phase_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

template <int FLAGS>      // tuning: bit 0 = no s_setprio around the MFMA cluster, bit 1 = fragment reads before the DMA requests
__global__ __launch_bounds__(512) void k_gemm_8ph(const GemmArgs g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_poison();
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int nN = g.N / 256, nM = g.M / 256;
    const int tile = xcd_remap(blockIdx.x, nM * nN);
    const int m0 = (tile / nN) * 256, n0 = (tile % nN) * 256;
    const int z = blockIdx.z;

    const bf16* a0 = g.a0 + (int64_t)z * g.a_bs;
    const bf16* a1 = g.a1 ? g.a1 + (int64_t)z * g.a_bs : nullptr;
    const bf16* bp = g.b + (int64_t)z * g.b_bs;
    const int K0 = g.taps * g.a0_C, K1 = g.a1 ? g.a1_C : 0;
    const int nk0 = K0 / BK, nk = nk0 + K1 / BK;
    const int Wp = (1 << g.logW) + 2, Hp = (1 << (g.logHW - g.logW)) + 2;

    // ---- DMA source addresses.  Wave w stages pieces 2w, 2w+1 of every half-tile: A rows wr*128 + h*64 + (2(w&3)+j)*8 + ..,
    // ---- B rows (w>>1)*64 + h*32 + (2(w&1)+j)*8 + ..
    const int prow = lane >> 3;
    uint64_t a_adr[2][2], a1_adr[2], b_adr[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lchunk = ((lane & 7) ^ ((j * 4 + (lane >> 4)) & 7)) << 3;          // (row>>1)&7 of these rows is j*4 + lane>>4
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m0 + wr * 128 + h * 64 + (2 * (wave & 3) + j) * 8 + prow;
            int64_t off0;
            if (g.taps == 9) {
                const int b = m >> g.logHW, p = m & ((1 << g.logHW) - 1), y = p >> g.logW, x = p & ((1 << g.logW) - 1);
                off0 = ((int64_t)(b * Hp + y + 1) * Wp + x + 1) * g.a0_ld;
            } else {
                off0 = (int64_t)m * g.a0_ld;
            }
            a_adr[h][j] = reinterpret_cast<uint64_t>(a0 + off0 + lchunk);
        }
        a1_adr[j] = a1 ? reinterpret_cast<uint64_t>(a1 + (int64_t)(m0 + wr * 128 + (2 * (wave & 3) + j) * 8 + prow) * g.a1_ld + lchunk) : 0;
        b_adr[j] = reinterpret_cast<uint64_t>(bp + (int64_t)(n0 + (wave >> 1) * 64 + (2 * (wave & 1) + j) * 8 + prow) * g.b_ld + lchunk);
    }
    const int64_t a1_half = (int64_t)64 * g.a1_ld * 2, b_half = (int64_t)32 * g.b_ld * 2;       // byte offsets of half 1
    unsigned char* const dA = smem + (wr * 128 + 2 * (wave & 3) * 8) * 128;                     // + buf*65536 + h*8192 + j*1024
    unsigned char* const dB = smem + 32768 + ((wave >> 1) * 64 + 2 * (wave & 1) * 8) * 128;     // + buf*65536 + h*4096 + j*1024

    auto stage_a = [&](int h, int kt) __attribute__((always_inline)) {
        unsigned char* d = dA + (kt & 1) * 65536 + h * 8192;
        if (kt < nk0) {
            int tap = 0, c0 = kt * BK;
            if (g.taps == 9) { const int cch = kt / 9; tap = kt - 9 * cch; c0 = cch * BK; }
            const int dy = g.taps == 9 ? tap / 3 - 1 : 0, dx = g.taps == 9 ? tap % 3 - 1 : 0;
            const int64_t sh = ((int64_t)(dy * Wp + dx) * g.a0_ld + c0) * 2;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(a_adr[h][j] + (uint64_t)sh), (lds_void*)(d + j * 1024), 16, 0, 0);
        } else {
            const int64_t sh = (int64_t)(kt - nk0) * BK * 2 + (h ? a1_half : 0);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(a1_adr[j] + (uint64_t)sh), (lds_void*)(d + j * 1024), 16, 0, 0);
        }
    };
    auto stage_b = [&](int h, int kt) __attribute__((always_inline)) {
        unsigned char* d = dB + (kt & 1) * 65536 + h * 4096;
        const int64_t sh = (int64_t)(kt < nk0 ? kt * BK : K0 + (kt - nk0) * BK) * 2 + (h ? b_half : 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(b_adr[j] + (uint64_t)sh), (lds_void*)(d + j * 1024), 16, 0, 0);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- fragment read addresses
    const int frow = lane & 15, fq = lane >> 4, fswz = (frow >> 1) & 7;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_u8*)smem);
    const unsigned ra = lds0 + (wr * 128 + frow) * 128, rb = lds0 + 32768 + (wc * 64 + frow) * 128;
    const unsigned ko0 = ((0 | fq) ^ fswz) << 4, ko1 = ((4 | fq) ^ fswz) << 4;
    u32x4 fa[2][4], fb[2][2][2];                                   // fa[ks][i]; fb[half][ks][j]

    // ---- prologue: K-tile 0 complete, and the part of K-tile 1 the steady state would have requested by now
    stage_a(0, 0); stage_b(0, 0); stage_b(1, 0); stage_a(1, 0);
    if (nk > 1) { stage_a(0, 1); stage_b(0, 1); }
    wait_vmcnt<0>();
    phase_barrier();
    if (wr == 1) phase_barrier();                                   // group 1 runs one step behind group 0

#define NATINF_8PH_MFMA(IH, JH)                                                                                         \
    {                                                                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if (!(FLAGS & 1)) __builtin_amdgcn_s_setprio(1);                                                                \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                               \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
                    acc[IH * 4 + i][JH * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                              \
                        __builtin_bit_cast(bf16x8, fb[JH][ks][j]), __builtin_bit_cast(bf16x8, fa[ks][i]), acc[IH * 4 + i][JH * 2 + j], 0, 0, 0); \
        if (!(FLAGS & 1)) __builtin_amdgcn_s_setprio(0);                                                                \
        phase_barrier();                                                                                                \
    }
#define NATINF_8PH_READ_A(IH)                                                                                           \
    {                                                                                                                   \
        fa[0][0] = lds_read16<IH * 8192 + 0>(pa0); fa[0][1] = lds_read16<IH * 8192 + 2048>(pa0);                        \
        fa[0][2] = lds_read16<IH * 8192 + 4096>(pa0); fa[0][3] = lds_read16<IH * 8192 + 6144>(pa0);                     \
        fa[1][0] = lds_read16<IH * 8192 + 0>(pa1); fa[1][1] = lds_read16<IH * 8192 + 2048>(pa1);                        \
        fa[1][2] = lds_read16<IH * 8192 + 4096>(pa1); fa[1][3] = lds_read16<IH * 8192 + 6144>(pa1);                     \
    }
#define NATINF_8PH_READ_B(JH)                                                                                           \
    {                                                                                                                   \
        fb[JH][0][0] = lds_read16<JH * 4096 + 0>(pb0); fb[JH][0][1] = lds_read16<JH * 4096 + 2048>(pb0);                \
        fb[JH][1][0] = lds_read16<JH * 4096 + 0>(pb1); fb[JH][1][1] = lds_read16<JH * 4096 + 2048>(pb1);                \
    }

    for (int kt = 0; kt < nk; ++kt) {
        const unsigned bo = (kt & 1) * 65536;
        const unsigned pa0 = ra + bo + ko0, pa1 = ra + bo + ko1, pb0 = rb + bo + ko0, pb1 = rb + bo + ko1;
        const bool steady = kt + 2 < nk, has1 = kt + 1 < nk;
        // phase 0: quadrant (A0, B0)
        if (steady) wait_vmcnt<6>(); else wait_vmcnt<0>();
        if (!(FLAGS & 2) && has1) stage_b(1, kt + 1);
        NATINF_8PH_READ_B(0)
        NATINF_8PH_READ_A(0)
        if ((FLAGS & 2) && has1) stage_b(1, kt + 1);
        phase_barrier();
        NATINF_8PH_MFMA(0, 0)
        // phase 1: (A0, B1)
        if (steady) wait_vmcnt<6>(); else wait_vmcnt<0>();
        if (!(FLAGS & 2) && has1) stage_a(1, kt + 1);
        NATINF_8PH_READ_B(1)
        if ((FLAGS & 2) && has1) stage_a(1, kt + 1);
        phase_barrier();
        NATINF_8PH_MFMA(0, 1)
        // phase 2: (A1, B1)
        if (steady) wait_vmcnt<6>(); else wait_vmcnt<0>();
        if (!(FLAGS & 2) && steady) stage_a(0, kt + 2);
        NATINF_8PH_READ_A(1)
        if ((FLAGS & 2) && steady) stage_a(0, kt + 2);
        phase_barrier();
        NATINF_8PH_MFMA(1, 1)
        // phase 3: (A1, B0)
        if (steady) wait_vmcnt<6>(); else wait_vmcnt<0>();
        if (steady) stage_b(0, kt + 2);
        phase_barrier();
        NATINF_8PH_MFMA(1, 0)
    }
#undef NATINF_8PH_MFMA
#undef NATINF_8PH_READ_A
#undef NATINF_8PH_READ_B
    if (wr == 0) phase_barrier();
    __syncthreads();
    dma_tile_epilogue<2, 4, 8, 4, Cfg8ph::Epi>(g, smem, acc, m0, n0, z, tid, lane, wr, wc);
}

}  // namespace ncsn

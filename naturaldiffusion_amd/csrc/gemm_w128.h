// gemm_w128.h -- k_gemm_w128: the plain bf16 GEMM (taps = 1, one K segment) on 256 x 256 x 64 block tiles worked by FOUR waves -- ONE WAVE PER SIMD, wave tile
// 128 x 128, its 256 fp32 accumulators in AGPRs -- with both LDS stages (2 x 64 KB) filled by LDS-DMA TWO K-tiles ahead.
//
// Why (round 4, calibration against the vendor library on the transformer engines' shapes: 1,336-1,495 TFLOP/s against 1,026-1,259 for k_gemm_dma<2,4,8,4,6> on the same
// box): k_gemm_dma / k_gemm_ring run two waves per SIMD with 128 x 64 wave tiles -- 12 fragment reads per 32 MFMAs, twice the waves issuing LDS-DMA, barriers and
// waits through the one vector-issue port a SIMD has, and a 256-register budget that leaves no room for a second fragment set.  Here
//   * a 128 x 128 wave tile needs 16 fragment reads per 64 MFMAs (0.25 per MFMA instead of 0.375) and the block 16 DMA pieces per wave and 128 MFMAs;
//   * the accumulators live in a[0:255] (inline-asm "+a" operands), which leaves the 256 VGPRs for TWO complete fragment sets (8 A + 8 B fragments per 32-wide
//     K step): the reads of K step s + 1 are issued while step s multiplies, so an MFMA never waits for an LDS read issued just before it;
//   * with a single wave per SIMD nothing else competes for the issue port: the loop is ONE instruction stream, written out slot by slot (W128Step below: every
//     instruction of the K loop is a volatile asm statement, so hipcc keeps the order; it only allocates the registers);
//   * LDS-DMA two tiles ahead into a two-stage ring: the A half of stage `cur` is released as soon as every wave holds its K-step-1 A fragments (barrier 1), the B half
//     after the B fragments (barrier 2); tile kt + 2 is then requested into the half just released, and tile kt + 1 -- requested a whole iteration earlier -- is
//     awaited with a COUNTED vmcnt at two thirds of the iteration (barrier 3).  A request has ~1.3 iterations (~2,700 clocks) to land.
// The tile is multiplied as two 256 x 128 halves (wave column wn owns columns wn * 64 .. + 63 of EACH half), so the epilogue is tile_epilogue<2, 2, 8, 4> of
// gemm_dma.h called once per half: every fused epilogue of the 4-wave 256 x 128 tile works unchanged.
// Limits (launch_gemm checks): taps == 1, no second K segment, K % 64 == 0, K >= 128, M % 8 == 0, N % 8 == 0, operand matrices < 4 GiB (32-bit offsets).
#pragma once
#include "gemm_dma.h"

namespace ncsn {

struct W128Cfg {
    static constexpr int WM = 2, WN = 2, TM = 8, TN = 8, NW = 4, THREADS = 256, BM_ = 256, BN_ = 256;
    static constexpr int HALF_BYTES = 256 * BK * 2, STAGE_BYTES = 2 * HALF_BYTES;      // A half, B half of a stage: 32 KB each
    using Epi = EpiCfg<2, 2, 8, 4, 2 * STAGE_BYTES + 8192>;                            // the 256 x 128 half tile
    static constexpr int SLAB2_OFF = (Epi::PACK_BYTES + 1023) / 1024 * 1024;           // the second half tile's packed slab: its own LDS, so no barrier between the halves
    static constexpr int LDS_BYTES = Epi::NEED > SLAB2_OFF + Epi::PACK_BYTES ? Epi::NEED : SLAB2_OFF + Epi::PACK_BYTES;
    static_assert(BK == 64 && LDS_BYTES <= 163840 && LDS_BYTES >= 2 * STAGE_BYTES, "two 64-KB stages");
};

typedef int w128_rsrc __attribute__((ext_vector_type(4)));
struct W128Dma {
    unsigned voa, vob;                        // the lane's byte offset inside an 8-row piece of A / B (row lane >> 3, its swizzled 16-byte chunk)
    unsigned soa[8], sob[8];                  // byte offset of this wave's piece n from the operand's base (wave-uniform)
    w128_rsrc ra, rb;                         // buffer descriptors at the operand bases advanced to the K-tile being requested
    unsigned pfa, pfb, junk;                  // L2 prefetch: the lane's row of A / B (one 128-byte line per lane and K-tile); the loads' common destination
};
struct W128Addr {
    unsigned a_cur, b_cur, a_nxt, b_nxt;      // per-lane LDS byte address of fragment 0, K step 0 (K step 1: ^ 64), in stage cur / cur ^ 1
    unsigned da, db;                          // LDS-DMA destinations of this wave's piece 0 in stage cur (wave-uniform)
};

__device__ __forceinline__ void w128_mfma(f32x4& acc, const u32x4& b, const u32x4& a) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(b), "v"(a));
}
__device__ __forceinline__ void w128_glds(unsigned vo, const void* sbase, unsigned dst) {       // prologue form: destination set right in front
    NATINF_M0_ASM_BEGIN
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(vo), "s"(sbase), "s"(dst) : "memory", "m0");
    NATINF_M0_ASM_END
}
// K-loop form: M0 is set ONE MFMA EARLIER (w128_m0_set / w128_m0_next: the MFMA in between is the wait state the hardware asks for between a write of M0 and an
// LDS-DMA instruction), so that a request costs its gap one instruction, not four.  Nothing else in the loop touches M0 (the loop holds no compiler-generated code).
NATINF_M0_ASM_BEGIN
__device__ __forceinline__ void w128_m0_set(unsigned dst) { asm volatile("s_mov_b32 m0, %0" :: "s"(dst) : "memory", "m0"); }
NATINF_M0_ASM_END
NATINF_M0_ASM_BEGIN
__device__ __forceinline__ void w128_m0_next() { asm volatile("s_add_u32 m0, m0, 0x1000" ::: "memory", "m0", "scc"); }
NATINF_M0_ASM_END
// The requests are BUFFER loads: address = descriptor base + the lane's offset inside an 8-row piece (ONE register per operand) + the piece's scalar offset -- fourteen
// lane registers fewer than global_load_lds with an offset per piece (the fp8 kernel spilled them, and hipcc put a vmcnt(0) behind every reload).  A piece whose rows
// lie beyond the matrix is pointed at the last valid 8-row group (M % 8 == 0, N % 8 == 0): every address is in bounds, the epilogue masks the rows.
__device__ __forceinline__ w128_rsrc w128_make_rsrc(const void* base, unsigned bytes = 0xffffffffu) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    return w128_rsrc{(int)__builtin_amdgcn_readfirstlane((unsigned)a), (int)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu),
                     (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000};
}
// L2 prefetch (W128SchP, development builds; measured slower): one dword per 128-byte line of K-tile kt + 2 + PF, ahead of the LDS-DMA requests -- an HBM round trip
// under load is longer than the ~1.3 iterations a request has (tile timeline, tools/w128_timeline.py, at (32768, 1536, 6144), where a row panel of A is read by six
// tiles only: 3,282 clocks per K-tile against 2,222 at 8192^3).  The descriptor's num_records is the operand's real extent: a lane past the last row or K-tile reads nothing.
template <int OFF> __device__ __forceinline__ void w128_prefetch(unsigned& junk, unsigned vo, const w128_rsrc& rs) {
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:%3" : "+v"(junk) : "v"(vo), "s"(rs), "n"(OFF) : "memory");
}
__device__ __forceinline__ void w128_bufdma(unsigned vo, const w128_rsrc& rs, unsigned soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(vo), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void w128_bufdma_at(unsigned vo, const w128_rsrc& rs, unsigned soff, unsigned dst) {       // prologue form: destination set right in front
    NATINF_M0_ASM_BEGIN
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(vo), "s"(rs), "s"(soff), "s"(dst) : "memory", "m0");
    NATINF_M0_ASM_END
}

// The direct fp32 residual-stream epilogue (EPI 7 here, 3 in the fp8 kernel: out = resid + gate * (acc + bias), in place) is the slow one of this tile, measured in round 6
// (tools/w128_e7_timeline.py, profiles/r06/w128_e7_timeline.log): 43k shader clocks per tile against 12.5k for the packed bf16 epilogue -- each half tile is one round trip for
// 128 KB of residual per CU that ALL CUs ask for at the same moment (32 MB per half tile and round; 64 MB read + 64 MB written per round at ~3.5 TB/s).  In the MMDiT engine
// the attention output projection (4096 x 1536 x 1536 x 8) runs at 233 us = 662 TFLOP/s where q | k with the packed epilogue runs at 1,217.  Two remedies were built,
// measured and removed (profiles/r06/resid_warm_*.log, stagger_bf16.log; same bytes both):
//  * touching the residual tile once at kernel start (one dword per 128-byte line into a junk register, in front of the first operand requests): 233 -> 257 us, fc2
//    525 -> 554 us -- in-order return puts an HBM round trip under load in front of K-tile 0, and the lines are gone again when the epilogue asks (32 CUs x 256 KB per
//    XCD against 4 MB of L2);
//  * a phase stagger (the first round's blocks sleep 0 / 1 / 2 x 8 or 16 us by CU, so that a third of the chip is in its epilogue while two thirds multiply): the epilogues
//    do get shorter (-17 .. -24 us per launch), and the last group's sleep costs exactly that (+-0 at 8 us, +8 us at 16 us per launch; three rounds per launch).
// What helped is fewer bytes: the MMDiT engine keeps its image stream in IEEE half (natinf_set_mmdit_stream16, GemmArgs::stream_f16: gemm_dma.h direct_f32_epilogue, 16-byte accesses through v_permlane16_swap) -- -3.3 % bf16 /
// -5.2 % fp8 per forward; the residual under the K loop needs registers that are not there.
// The slot table of one iteration (slot S = the instructions issued in front of MFMA S; MFMA S = K step S / 64, A fragment (S / 8) % 8, B fragment S % 8).
// MODE 0: steady state; 1: second-to-last tile (nothing left to request; the last tile is awaited with vmcnt(0)); 2: last tile (K step 1's reads only).
// A schedule SCH names the slots: K step 1's fragment reads (rd1: 0..7 = A, 8..15 = B), the two "half is free" barriers, the sixteen requests (piece p: 0..7 = A,
// 8..15 = B; destinations 4 KB apart in that order; a schedule may give wave WV of the block its own slots -- unused: see below), the counted wait for tile kt + 1, K step 0's reads of tile kt + 1.
struct W128SchB {          // shipped: both halves released by ONE barrier behind the sixteen reads of K step 1 (all in the first seventeen slots); the requests four
    static constexpr bool NO_DMA = false;                        // MFMAs apart in gaps that carry no fragment read; tile kt + 1 awaited at slot 60
    static constexpr int WAIT1 = 24, WAIT2 = 24, WAIT3 = 60;
    static constexpr int rd1(int n) { return n < 8 ? 1 + 2 * n : 2 + 2 * (n - 8); }
    static constexpr int dma(int p, int) { return 27 + 4 * p; }
    static constexpr int rd0(int n) { return 90 + 2 * n; }                                      // 0..7 = B, 8..15 = A
    static constexpr int PF = 0, PF_A = 89, PF_B = 123;                                         // L2 prefetch of K-tile kt + 2 + PF (0 = none; W128SchP): behind the last request / read
};
#ifdef NATINF_DEV
// Measured beside it (tools/ab_w128_sched.py, one box, TFLOP/s at 8192^3 / (32768, 6144, 1536) / (32768, 1536, 1536) / (32768, 1536, 6144); SchB: 1,526-1,620 / 1,325-1,342 /
// 1,331-1,338 / 1,332-1,381; the two-waves-per-SIMD 256 x 256 tile: 1,243-1,396 / 1,084-1,225 / 1,254-1,262 / 1,109-1,228):
struct W128SchA {          // the first form: A half and B half released by two barriers, requests two MFMAs apart between the B fragment reads: 1,366-1,581 / 1,272-1,301 / 1,281-1,298 / 1,312-1,358
    static constexpr bool NO_DMA = false;
    static constexpr int WAIT1 = 21, WAIT2 = 50, WAIT3 = 88;
    static constexpr int rd1(int n) { return n < 8 ? 1 + 2 * n : 24 + 2 * (n - 8); }
    static constexpr int dma(int p, int) { return p < 8 ? 23 + 2 * p : (p < 13 ? 52 + 2 * (p - 8) : 91 + 4 * (p - 13)); }
    static constexpr int rd0(int n) { return n < 8 ? 90 + 2 * n : 106 + 2 * (n - 8); }
    static constexpr int PF = 0, PF_A = 0, PF_B = 0;
};
// (the reads of tile kt + 1 spread over the second half of the iteration, between the requests -- rd0(n) = 65 + 4 n --: 1,511-1,540 / 1,237 / 1,280-1,290 / 1,308-1,327.  Not kept.)
struct W128SchP : W128SchB { static constexpr int PF = 3; };              // + L2 prefetch three K-tiles ahead of the requests (w128_prefetch): 8192^3 1,354 against 1,570, (32768, 6144, 1536) 1,125 against
                                                                           // 1,260 -- a dword load of 64 different lines costs the L1 path eight requests' worth.  Not kept.
struct W128SchX : W128SchB { static constexpr bool NO_DMA = true; };      // ablation: nothing requested after the prologue (wrong results): 1,806-1,842 / 1,467-1,483 / 1,308-1,408 /
                                                                           // 1,765-1,784 -- what the sixteen requests of an iteration cost (~19 clocks of matrix pipe each)
// (the four waves' requests one MFMA apart from each other -- four copies of the loop behind a branch on the wave index -- made hipcc spill: 1,630 scratch
// accesses, fragment registers spilled with their loads in flight: wrong results at 85 TFLOP/s.  Not kept.)
#endif
template <class SCH, int WV> constexpr int w128_early() { int c = SCH::PF ? 2 : 0; for (int p = 0; p < 16; ++p) c += SCH::dma(p, WV) < SCH::WAIT3; return c; }      // + the two prefetch loads of the iteration before

template <class SCH, int MODE, int S, int WV>
struct W128Step {
    template <int P>
    static __device__ __forceinline__ void dma(const W128Addr& ad, const W128Dma& dm) {
        if constexpr (P < 16) {
            constexpr int at = SCH::dma(P, WV);
            static_assert(P == 0 || SCH::dma(P - 1, WV) < at, "requests in piece order, at most one per slot");
            static_assert(at > (P < 8 ? SCH::WAIT1 + 1 : SCH::WAIT2 + 1) && at >= 2 && at < 128, "behind the barrier that frees its half");
            if constexpr (S == at - 1) { if constexpr (P == 0) w128_m0_set(ad.da); else w128_m0_next(); }
            if constexpr (S == at) w128_bufdma(P < 8 ? dm.voa : dm.vob, P < 8 ? dm.ra : dm.rb, (P < 8 ? dm.soa : dm.sob)[P & 7]);
            dma<P + 1>(ad, dm);
        }
    }
    template <int N>
    static __device__ __forceinline__ void reads(u32x4 (&fa)[2][8], u32x4 (&fb)[2][8], const W128Addr& ad) {
        if constexpr (N < 16) {
            if constexpr (S == SCH::rd1(N)) {
                if constexpr (N < 8) fa[1][N] = lds_read16<N * 2048>(ad.a_cur ^ 64u);
                else fb[1][N - 8] = lds_read16<((N - 8) >> 2) * 16384 + ((N - 8) & 3) * 2048>(ad.b_cur ^ 64u);
            }
            if constexpr (MODE < 2 && S == SCH::rd0(N)) {
                if constexpr (N < 8) fb[0][N] = lds_read16<(N >> 2) * 16384 + (N & 3) * 2048>(ad.b_nxt);
                else fa[0][N - 8] = lds_read16<(N - 8) * 2048>(ad.a_nxt);
            }
            reads<N + 1>(fa, fb, ad);
        }
    }
    static __device__ __forceinline__ void run(f32x4 (&accL)[8][4], f32x4 (&accH)[8][4], u32x4 (&fa)[2][8], u32x4 (&fb)[2][8], const W128Addr& ad,
                                               W128Dma& dm)
    {
        static_assert(SCH::rd1(7) < SCH::WAIT1 && SCH::rd1(15) < SCH::WAIT2 && SCH::rd0(0) > SCH::WAIT3 + 1 && SCH::rd0(15) < 127, "reads in front of their waits");
        if constexpr (S == SCH::WAIT1 || S == SCH::WAIT2 || (S == 127 && MODE < 2)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr ((S == SCH::WAIT1 + 1 || S == SCH::WAIT2 + 1) && MODE == 0) asm volatile("s_barrier" ::: "memory");          // the A / B half of stage cur is free
        if constexpr (MODE < 2) {
            if constexpr (S == SCH::WAIT3 && !(SCH::NO_DMA && MODE == 0)) {          // tile kt + 1 has landed: MODE 0 retires exactly the previous iteration's sixteen requests
                if constexpr (MODE == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(w128_early<SCH, WV>()) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if constexpr (S == SCH::WAIT3 + 1) asm volatile("s_barrier" ::: "memory");
        }
        reads<0>(fa, fb, ad);
        if constexpr (MODE == 0 && !SCH::NO_DMA) dma<0>(ad, dm);
        if constexpr (MODE == 0 && SCH::PF > 0) {
            static_assert(SCH::PF == 0 || (SCH::PF_A > SCH::dma(15, WV) && SCH::PF_B > SCH::PF_A && SCH::PF_A > SCH::WAIT3), "behind this iteration's requests: the next counted wait leaves them in flight");
            if constexpr (S == SCH::PF_A) w128_prefetch<SCH::PF * 128>(dm.junk, dm.pfa, dm.ra);
            if constexpr (S == SCH::PF_B) w128_prefetch<SCH::PF * 128>(dm.junk, dm.pfb, dm.rb);
        }
        constexpr int ks = S >> 6, i = (S >> 3) & 7, j = S & 7;
        if constexpr (j < 4) w128_mfma(accL[i][j], fb[ks][j], fa[ks][i]);
        else w128_mfma(accH[i][j - 4], fb[ks][j], fa[ks][i]);
        if constexpr (S + 1 < 128) W128Step<SCH, MODE, S + 1, WV>::run(accL, accH, fa, fb, ad, dm);
    }
};

template <int EPI, class SCH = W128SchB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_gemm_w128(const GemmArgs g)
{
    using Cfg = W128Cfg;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_poison();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nN = (g.N + 255) / 256, nM = (g.M + 255) / 256;
    const int tile = xcd_remap(blockIdx.x, nM * nN);
    int mt_, nt_;
    tile_coords(tile, nM, nN, g.raster_g, mt_, nt_);
    const int m0 = mt_ * 256, n0 = nt_ * 256;
    const int z = blockIdx.z;
    // EPI 9 (split-K, launch_gemm): blockIdx.y = K slice; the tile's fp32 partial sums go to g.c [slice][batch][M][N], k_splitk_reduce_f32 sums the slices in order
    const int nk_all = g.a0_C / BK;
    const int kt_lo = EPI == 9 ? (int)((int64_t)nk_all * blockIdx.y / g.splitk) : 0;
    const int nk = EPI == 9 ? (int)((int64_t)nk_all * (blockIdx.y + 1) / g.splitk) - kt_lo : nk_all;      // >= 2
    const bf16* abase = g.a0 + (int64_t)z * g.a_bs + (int64_t)kt_lo * BK;
    const bf16* bbase = g.b + (int64_t)z * g.b_bs + (int64_t)kt_lo * BK;
    NATINF_TS(0);

    // LDS-DMA: piece n of this wave = rows n * 32 + wave * 8 .. + 7 of the A (B) half, one 128-byte row per eight lanes, the row's 16-byte chunks stored at
    // chunk ^ ((row >> 1) & 7) (the image k_gemm_dma's fragment reads are conflict-free on)
    W128Dma dm;
    {
        const int chunk = (lane & 7) ^ ((wave & 1) * 4 + (lane >> 4));
        dm.voa = (unsigned)(((lane >> 3) * g.a0_ld + chunk * 8) * 2);
        dm.vob = (unsigned)(((lane >> 3) * g.b_ld + chunk * 8) * 2);
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const int r = n * 32 + wave * 8;
            dm.soa[n] = (unsigned)((int64_t)min(m0 + r, g.M - 8) * g.a0_ld * 2);
            dm.sob[n] = (unsigned)((int64_t)min(n0 + r, g.N - 8) * g.b_ld * 2);
        }
        dm.pfa = dm.pfb = dm.junk = 0;
        if constexpr (SCH::PF > 0) {                                  // (development schedules only)
            dm.pfa = (unsigned)((int64_t)min(m0 + wave * 64 + lane, g.M - 1) * g.a0_ld * 2);
            dm.pfb = (unsigned)((int64_t)min(n0 + wave * 64 + lane, g.N - 1) * g.b_ld * 2);
        }
    }
    // the operands' extents behind their bases (the descriptors' num_records: what keeps a prefetch past the last K-tile from touching memory)
    const unsigned a_bytes = (unsigned)((((int64_t)g.M - 1) * g.a0_ld + g.a0_C - (int64_t)kt_lo * BK) * 2), b_bytes = (unsigned)((((int64_t)g.N - 1) * g.b_ld + g.a0_C - (int64_t)kt_lo * BK) * 2);
    auto rsrc_at = [&](int kt) __attribute__((always_inline)) {
        dm.ra = w128_make_rsrc(abase + (int64_t)kt * BK, a_bytes - (unsigned)kt * (BK * 2)); dm.rb = w128_make_rsrc(bbase + (int64_t)kt * BK, b_bytes - (unsigned)kt * (BK * 2));
    };
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_u8*)smem);
    const unsigned dw = lds0 + wave * 1024;
#pragma unroll
    for (int t = 0; t < 2; ++t) {                                     // tiles 0 and 1 -> stages 0 and 1
        rsrc_at(t);
#pragma unroll
        for (int n = 0; n < 8; ++n) w128_bufdma_at(dm.voa, dm.ra, dm.soa[n], dw + t * Cfg::STAGE_BYTES + n * 4096);
#pragma unroll
        for (int n = 0; n < 8; ++n) w128_bufdma_at(dm.vob, dm.rb, dm.sob[n], dw + t * Cfg::STAGE_BYTES + Cfg::HALF_BYTES + n * 4096);
    }
    if constexpr (SCH::PF > 0) {                                      // (every iteration's counted wait assumes two prefetch loads behind the previous iteration's requests)
        w128_prefetch<SCH::PF * 128>(dm.junk, dm.pfa, dm.ra);
        w128_prefetch<SCH::PF * 128>(dm.junk, dm.pfb, dm.rb);
    }

    f32x4 accL[8][4], accH[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { accL[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; accH[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int frow = lane & 15, fq = lane >> 4, fswz = (frow >> 1) & 7;
    const unsigned a_off = lds0 + (wm * 128 + frow) * 128 + ((fq ^ fswz) << 4);
    const unsigned b_off = lds0 + Cfg::HALF_BYTES + (wn * 64 + frow) * 128 + ((fq ^ fswz) << 4);
    u32x4 fa[2][8], fb[2][8];
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(SCH::PF > 0 ? 18 : 16) : "memory");    // tile 0 is in stage 0
    NATINF_TS(1);
#pragma unroll
    for (int n = 0; n < 8; ++n) fb[0][n] = lds_read16<0>(b_off + (n >> 2) * 16384 + (n & 3) * 2048);
#pragma unroll
    for (int n = 0; n < 8; ++n) fa[0][n] = lds_read16<0>(a_off + n * 2048);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    W128Addr ad;
    auto iter = [&](auto mode) __attribute__((always_inline)) { W128Step<SCH, decltype(mode)::value, 0, 0>::run(accL, accH, fa, fb, ad, dm); };
    for (int kt = 0; kt < nk - 2; ++kt) {
        const unsigned cur = (kt & 1) * Cfg::STAGE_BYTES, nxt = cur ^ Cfg::STAGE_BYTES;
        ad.a_cur = a_off + cur; ad.b_cur = b_off + cur; ad.a_nxt = a_off + nxt; ad.b_nxt = b_off + nxt;
        ad.da = dw + cur; ad.db = dw + cur + Cfg::HALF_BYTES;
        rsrc_at(kt + 2);
        iter(std::integral_constant<int, 0>{});
    }
    {
        const unsigned cur = ((nk - 2) & 1) * Cfg::STAGE_BYTES, nxt = cur ^ Cfg::STAGE_BYTES;
        ad.a_cur = a_off + cur; ad.b_cur = b_off + cur; ad.a_nxt = a_off + nxt; ad.b_nxt = b_off + nxt;
        ad.da = 0; ad.db = 0;
        iter(std::integral_constant<int, 1>{});
        ad.a_cur = a_off + nxt; ad.b_cur = b_off + nxt;
        iter(std::integral_constant<int, 2>{});
    }
    // the last MFMAs' results (inline asm: hipcc does not see the writes it would pad for) before anything reads an accumulator
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    NATINF_TS(5);
    if constexpr (SCH::PF > 0) asm volatile("" :: "v"(dm.junk));
    if constexpr (EPI == 9) {                                            // split-K partial sums, straight from the registers (16-byte stores)
        float* out = reinterpret_cast<float*>(g.c) + ((int64_t)blockIdx.y * g.batch + z) * g.M * g.N;
        const int r = lane & 15, q = lane >> 4;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            if (n0 + hh * 128 >= g.N) break;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = m0 + wm * 128 + i * 16 + r;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + hh * 128 + wn * 64 + j * 16 + q * 4;
                    if (m < g.M && n < g.N) *reinterpret_cast<f32x4*>(out + (int64_t)m * g.N + n) = hh ? accH[i][j] : accL[i][j];
                }
            }
        }
        return;
    }
    if constexpr (EPI == 7) {                                            // direct fp32 residual stream: no LDS, the whole half tile's residual in one round trip
        direct_f32_epilogue<2, 2, 8, 4, false, 8>(g, accL, m0, n0, z, lane, wm, wn);
        NATINF_TS(6);
        if (n0 + 128 < g.N) direct_f32_epilogue<2, 2, 8, 4, false, 8>(g, accH, m0, n0 + 128, z, lane, wm, wn);
#ifdef NATINF_DEV
        if (g.dbg_ts) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (timeline runs: stamp 7 is "the tile's stores have been acknowledged")
#endif
        NATINF_TS(7);
        return;
    }
    __syncthreads();
    tile_epilogue<2, 2, 8, 4, typename Cfg::Epi, EPI>(g, smem, accL, m0, n0, z, tid, lane, wm, wn);
    NATINF_TS(6);
    if (n0 + 128 < g.N) {
        // packed epilogues: the second half has a slab of its own (a wave goes from the first half's copy-out straight to the second half's register phase);
        // the fp32-slab epilogue reuses the one slab behind a barrier; the direct epilogue (7) has none
        if constexpr (EPI == 0) __syncthreads();
        tile_epilogue<2, 2, 8, 4, typename Cfg::Epi, EPI>(g, smem + (EPI == 0 || EPI == 7 ? 0 : Cfg::SLAB2_OFF), accH, m0, n0 + 128, z, tid, lane, wm, wn);
    }
    NATINF_TS(7);
}

// Built, measured and not kept (round 4): the PERSISTENT form -- one block per CU walking output tiles, the sixteen requests for the next tile's first K-tile issued
// inside the last iteration of the current one (stage 0 is free by then when the K-tile count is even), the epilogue slab behind stage 0, the accumulators zeroed while
// the requests land.  Parity green; (32768, 1536, 1536) +8 %, every other shape of tools/bench_w128.py +-1 % (the epilogue's stores sit in front of the next tile's
// requests in vmcnt order, and what a block-per-tile launch loses between tiles is ~3-6k clocks of a 70-330k-clock tile).
// Also not kept: bf16 outputs straight from the accumulator registers (8-byte stores, no slab, no barrier -- the slab round trip is ~6.6k clocks per half tile,
// tools/w128_timeline.py): 4-8 % SLOWER per GEMM at K <= 1,536 (bf16 and fp8 operands alike), SD3 forward +0.5 / +0.7 ms -- a store instruction that covers sixteen
// 32-byte pieces costs the store path what one covering eight full lines does.

// Split-K of an under-filled long-K GEMM whose epilogue is the gated fp32 residual stream (DiT-XL/2's fc2 at Validate's size: (4096, 1152, 4608) is 80 tiles of
// 256 x 256 -- or 288 of 128 x 128 on 512 slots -- for 72 K-tiles each): S slices of K as S x 80 blocks of k_gemm_w128<9>, then this pass sums the slices IN ORDER
// (deterministic) and applies out = (resid + gate[sample] * (sum + bias)) * scale, four columns per thread.  resid may be out (same element, same thread).
__global__ __launch_bounds__(256) void k_splitk_reduce_f32(const float* __restrict__ part, int S, int64_t slice_stride, int M, int N, const float* __restrict__ bias_n,
                                                            const float* __restrict__ gate, int gate_ld, int log_rows_per_sample, int z_samples,
                                                            const float* resid, int resid_ld, int64_t c_bs, float scale, float* c, int c_ld, int stream_f16 = 0)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, per = (int64_t)M * (N >> 2);
    const int z = blockIdx.y;
    if (idx >= per) return;
    const int m = (int)(idx / (N >> 2)), n = (int)(idx - (int64_t)m * (N >> 2)) * 4;
    const float* p = part + ((int64_t)z * M + m) * N + n;
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    for (int sl = 1; sl < S; ++sl) { const f32x4 u = *reinterpret_cast<const f32x4*>(p + sl * slice_stride); v += u; }
    if (bias_n) { const f32x4 b = *reinterpret_cast<const f32x4*>(bias_n + n); v += b; }
    if (gate) { const f32x4 gt = *reinterpret_cast<const f32x4*>(gate + (int64_t)((m >> log_rows_per_sample) + z * z_samples) * gate_ld + n); v *= gt; }
    typedef _Float16 f16x4_sk __attribute__((ext_vector_type(4)));
    if (stream_f16) {                                                 // (GemmArgs::stream_f16: resid and c are IEEE-half rows -- the transformer engines' 16-bit residual stream)
        if (resid) {
            const f16x4_sk rs = *reinterpret_cast<const f16x4_sk*>(reinterpret_cast<const _Float16*>(resid) + (int64_t)z * c_bs + (int64_t)m * resid_ld + n);
            v += f32x4{(float)rs[0], (float)rs[1], (float)rs[2], (float)rs[3]};
        }
        v *= scale;
        const f16x4_sk o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        *reinterpret_cast<f16x4_sk*>(reinterpret_cast<_Float16*>(c) + (int64_t)z * c_bs + (int64_t)m * c_ld + n) = o;
        return;
    }
    if (resid) { const f32x4 rs = *reinterpret_cast<const f32x4*>(resid + (int64_t)z * c_bs + (int64_t)m * resid_ld + n); v += rs; }
    v *= scale;
    *reinterpret_cast<f32x4*>(c + (int64_t)z * c_bs + (int64_t)m * c_ld + n) = v;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------------
// k_gemm_w128_fp8: the same tile for e4m3 operands (gemm_fp8.h's contract: one fp32 scale per row / column applied in the epilogue, optionally E8M0 block scales on
// the A operand).  A K-tile is 128 BYTES per row -- the LDS image, pieces and swizzle of the bf16 kernel -- and ONE K step of v_mfma_(scale_)f32_16x16x128_f8f6f4:
// 64 MFMAs of 32 clocks per wave and tile.  A fragment is eight registers (two 16-byte reads: chunks q and 4 + q of the row), so two complete fragment sets do not
// fit beside nothing; instead the B fragments are double-buffered as a set (2 x 64 registers) and the A fragments roll: A fragment i of tile kt + 1 is read as soon as
// row i of tile kt has been multiplied (A fragment 7 at the top of the next iteration).  Every read of stage `cur` is therefore complete a few slots into iteration
// kt + 1 -- where ONE barrier both releases the stage to the requests for tile kt + 2 and publishes tile kt + 1 (awaited with vmcnt(0): its sixteen requests were
// this wave's last).  One barrier and two waits per 64 MFMAs.
typedef int i32x8 __attribute__((ext_vector_type(8)));
struct W128F8Cfg {
    static constexpr int THREADS = 256, BM_ = 256, BN_ = 256, HALF_BYTES = 32768, STAGE_BYTES = 65536, MX_BASE = 2 * STAGE_BYTES;     // + 2 x 1 KB of E8M0 scales
    using Epi = EpiCfg<2, 2, 8, 4, 2 * STAGE_BYTES + 8192>;
    static constexpr int SLAB2_OFF = (Epi::PACK_BYTES + 1023) / 1024 * 1024;
    static constexpr int LDS_A = Epi::NEED > MX_BASE + 2048 ? Epi::NEED : MX_BASE + 2048;
    static constexpr int LDS_BYTES = LDS_A > SLAB2_OFF + Epi::PACK_BYTES ? LDS_A : SLAB2_OFF + Epi::PACK_BYTES;
    static_assert(LDS_BYTES <= 163840, "two 64-KB stages");
};
template <bool MXA>
__device__ __forceinline__ void w128_mfma8(f32x4& acc, const u32x4& bl, const u32x4& bh, const u32x4& al, const u32x4& ah, unsigned unit, unsigned sc) {
    const i32x8 b = i32x8{(int)bl[0], (int)bl[1], (int)bl[2], (int)bl[3], (int)bh[0], (int)bh[1], (int)bh[2], (int)bh[3]};
    const i32x8 a = i32x8{(int)al[0], (int)al[1], (int)al[2], (int)al[3], (int)ah[0], (int)ah[1], (int)ah[2], (int)ah[3]};
    if constexpr (MXA) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+a"(acc) : "v"(b), "v"(a), "v"(unit), "v"(sc));
    else asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+a"(acc) : "v"(b), "v"(a));
}
template <int OFF> __device__ __forceinline__ unsigned lds_read_u8(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
struct W128F8Addr {
    unsigned a_rd[2], b_rd[2], mx_rd;         // per-lane LDS byte addresses (chunk q; chunk 4 + q: ^ 64) of fragment 0 in stage 0 / 1; of the lane's first scale byte
    unsigned d[2], mxd[2], mx_vo;             // LDS-DMA destination of this wave's piece 0 in stage 0 / 1; of the scale piece; the lane's offset into the scale piece
    const uint8_t* pmx;                       // scale bytes of tile kt + 2
    int wave;
};
// slots of an iteration (in front of MFMA S = A fragment S / 8, B fragment S % 8).  PAR = kt & 1 = the stage of tile kt = the B fragment set it multiplies.
namespace w128f8 {
constexpr int WAIT = 5, BAR = 6, MXDMA = 7;
constexpr int dma(int p) { return 8 + 3 * p; }
constexpr int rd_b(int j) { return 9 + 3 * j; }                                   // low half; the high half one slot later
constexpr int rd_a(int i) { return i < 6 ? 33 + 3 * i : 57; }                    // A fragment i of tile kt + 1 (i <= 6): behind row i of tile kt (slot 8 i + 7)
}
template <bool MXA, int MODE, int PAR, int S>
struct W128F8Step {
    template <int P>
    static __device__ __forceinline__ void dma(const W128F8Addr& ad, const W128Dma& dm) {
        if constexpr (P < 16) {
            constexpr int at = w128f8::dma(P);
            if constexpr (S == at - 1) { if constexpr (P == 0) w128_m0_set(ad.d[PAR]); else w128_m0_next(); }
            if constexpr (S == at) w128_bufdma(P < 8 ? dm.voa : dm.vob, P < 8 ? dm.ra : dm.rb, (P < 8 ? dm.soa : dm.sob)[P & 7]);
            dma<P + 1>(ad, dm);
        }
    }
    template <int N>
    static __device__ __forceinline__ void reads(u32x4 (&fal)[8], u32x4 (&fah)[8], u32x4 (&fbl)[2][8], u32x4 (&fbh)[2][8], unsigned (&sc)[8], const W128F8Addr& ad) {
        using namespace w128f8;
        if constexpr (N < 8) {
            constexpr int NX = PAR ^ 1;
            if constexpr (S == rd_b(N)) fbl[NX][N] = lds_read16<(N >> 2) * 16384 + (N & 3) * 2048>(ad.b_rd[NX]);
            if constexpr (S == rd_b(N) + 1) fbh[NX][N] = lds_read16<(N >> 2) * 16384 + (N & 3) * 2048>(ad.b_rd[NX] ^ 64u);
            if constexpr (N < 7) {
                static_assert(rd_a(N) > 8 * N + 7 && rd_a(N) + 1 < 63 && rd_a(N) < 8 * N + 64, "behind the last use of the register, ahead of the next");
                if constexpr (S == rd_a(N)) {
                    fal[N] = lds_read16<N * 2048>(ad.a_rd[NX]);
                    if constexpr (MXA) sc[N] = lds_read_u8<NX * 1024 + N * 64>(ad.mx_rd);
                }
                if constexpr (S == rd_a(N) + 1) fah[N] = lds_read16<N * 2048>(ad.a_rd[NX] ^ 64u);
            }
            reads<N + 1>(fal, fah, fbl, fbh, sc, ad);
        }
    }
    static __device__ __forceinline__ void run(f32x4 (&accL)[8][4], f32x4 (&accH)[8][4], u32x4 (&fal)[8], u32x4 (&fah)[8], u32x4 (&fbl)[2][8], u32x4 (&fbh)[2][8],
                                               unsigned (&sc)[8], unsigned unit, const W128F8Addr& ad, const W128Dma& dm)
    {
        using namespace w128f8;
        // A fragment 7 of THIS tile: the last read of stage PAR
        if constexpr (S == 0) { fal[7] = lds_read16<7 * 2048>(ad.a_rd[PAR]); if constexpr (MXA) sc[7] = lds_read_u8<PAR * 1024 + 7 * 64>(ad.mx_rd); }
        if constexpr (S == 1) fah[7] = lds_read16<7 * 2048>(ad.a_rd[PAR] ^ 64u);
        if constexpr (S == WAIT) {
            if constexpr (MODE < 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // tile kt + 1 is in LDS (this wave's pieces); my reads of stage PAR are done
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if constexpr (S == BAR && MODE < 2) asm volatile("s_barrier" ::: "memory");
        if constexpr (MODE == 0) {
            if constexpr (MXA && S == MXDMA) { if (ad.wave == 0) w128_glds(ad.mx_vo, ad.pmx, ad.mxd[PAR]); }      // the 256 rows x 4 scale bytes of tile kt + 2: 1 KB
            dma<0>(ad, dm);
        }
        if constexpr (MODE < 2) {
            reads<0>(fal, fah, fbl, fbh, sc, ad);
            if constexpr (S == 63) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        constexpr int i = S >> 3, j = S & 7;
        if constexpr (j < 4) w128_mfma8<MXA>(accL[i][j], fbl[PAR][j], fbh[PAR][j], fal[i], fah[i], unit, sc[i]);
        else w128_mfma8<MXA>(accH[i][j - 4], fbl[PAR][j], fbh[PAR][j], fal[i], fah[i], unit, sc[i]);
        if constexpr (S + 1 < 64) W128F8Step<MXA, MODE, PAR, S + 1>::run(accL, accH, fal, fah, fbl, fbh, sc, unit, ad, dm);
    }
};

// prologue: tile 0's B fragments and A fragments 0..6 from stage T
template <bool MXA, int T>
__device__ __forceinline__ void w128f8_first(u32x4 (&fal)[8], u32x4 (&fah)[8], u32x4 (&fbl)[2][8], u32x4 (&fbh)[2][8], unsigned (&sc)[8], const W128F8Addr& ad) {
#define NATINF_W128F8_B(N) fbl[T][N] = lds_read16<((N) >> 2) * 16384 + ((N) & 3) * 2048>(ad.b_rd[T]); fbh[T][N] = lds_read16<((N) >> 2) * 16384 + ((N) & 3) * 2048>(ad.b_rd[T] ^ 64u);
#define NATINF_W128F8_A(N) fal[N] = lds_read16<(N) * 2048>(ad.a_rd[T]); fah[N] = lds_read16<(N) * 2048>(ad.a_rd[T] ^ 64u); if constexpr (MXA) sc[N] = lds_read_u8<T * 1024 + (N) * 64>(ad.mx_rd);
    NATINF_W128F8_B(0) NATINF_W128F8_B(1) NATINF_W128F8_B(2) NATINF_W128F8_B(3) NATINF_W128F8_B(4) NATINF_W128F8_B(5) NATINF_W128F8_B(6) NATINF_W128F8_B(7)
    NATINF_W128F8_A(0) NATINF_W128F8_A(1) NATINF_W128F8_A(2) NATINF_W128F8_A(3) NATINF_W128F8_A(4) NATINF_W128F8_A(5) NATINF_W128F8_A(6)
#undef NATINF_W128F8_A
#undef NATINF_W128F8_B
}

// EPI as k_gemm_fp8: 0 = general fp32-slab epilogue, 1 = packed bf16, 2 = packed e4m3 + E8M0 with tanh-GELU, 3 = direct fp32 residual stream.
template <bool MXA, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_gemm_w128_fp8(const GemmArgs g)
{
    using Cfg = W128F8Cfg;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_poison();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nN = (g.N + 255) / 256, nM = (g.M + 255) / 256;
    const int tile = xcd_remap(blockIdx.x, nM * nN);
    int mt_, nt_;
    tile_coords(tile, nM, nN, g.raster_g, mt_, nt_);
    const int m0 = mt_ * 256, n0 = nt_ * 256;
    const int z = blockIdx.z;
    const uint8_t* abase = reinterpret_cast<const uint8_t*>(g.a0) + (int64_t)z * g.a_bs;         // strides in bytes = elements
    const uint8_t* bbase = reinterpret_cast<const uint8_t*>(g.b) + (int64_t)z * g.b_bs;
    const uint8_t* mxbase = MXA ? g.a_mx + (int64_t)z * g.a_mx_bs + (int64_t)m0 * 4 : nullptr;  // K-tile major: the tile's 256 rows x 4 blocks are 1 KB of consecutive bytes
    const int64_t mx_step = MXA ? (int64_t)g.a_mx_ld * 4 : 0;
    const int nk = g.a0_C / 128;                                      // >= 2, even

    W128Dma dm;
    {
        const int chunk = (lane & 7) ^ ((wave & 1) * 4 + (lane >> 4));
        dm.voa = (unsigned)((lane >> 3) * g.a0_ld + chunk * 16);
        dm.vob = (unsigned)((lane >> 3) * g.b_ld + chunk * 16);
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const int r = n * 32 + wave * 8;
            dm.soa[n] = (unsigned)((int64_t)min(m0 + r, g.M - 8) * g.a0_ld);
            dm.sob[n] = (unsigned)((int64_t)min(n0 + r, g.N - 8) * g.b_ld);
        }
    }
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_u8*)smem);
    const int frow = lane & 15, fq = lane >> 4, fswz = (frow >> 1) & 7;
    W128F8Addr ad;
    ad.wave = wave; ad.mx_vo = lane * 16;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        ad.d[t] = lds0 + t * Cfg::STAGE_BYTES + wave * 1024; ad.mxd[t] = lds0 + Cfg::MX_BASE + t * 1024;
        ad.a_rd[t] = lds0 + t * Cfg::STAGE_BYTES + (wm * 128 + frow) * 128 + ((fq ^ fswz) << 4);
        ad.b_rd[t] = lds0 + t * Cfg::STAGE_BYTES + Cfg::HALF_BYTES + (wn * 64 + frow) * 128 + ((fq ^ fswz) << 4);
    }
    ad.mx_rd = lds0 + Cfg::MX_BASE + (wm * 128 + frow) * 4 + fq;
#pragma unroll
    for (int t = 0; t < 2; ++t) {                                     // tiles 0 and 1 -> stages 0 and 1
#pragma unroll
        for (int n = 0; n < 8; ++n) w128_bufdma_at(dm.voa, w128_make_rsrc(abase + t * 128), dm.soa[n], ad.d[t] + n * 4096);
#pragma unroll
        for (int n = 0; n < 8; ++n) w128_bufdma_at(dm.vob, w128_make_rsrc(bbase + t * 128), dm.sob[n], ad.d[t] + Cfg::HALF_BYTES + n * 4096);
        if constexpr (MXA) { if (wave == 0) w128_glds(ad.mx_vo, mxbase + t * mx_step, ad.mxd[t]); }
    }

    f32x4 accL[8][4], accH[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { accL[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; accH[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    u32x4 fal[8], fah[8], fbl[2][8], fbh[2][8];
    unsigned sc[8] = {127u, 127u, 127u, 127u, 127u, 127u, 127u, 127u};
    const unsigned unit = 127u;                                       // E8M0 2^0
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    w128f8_first<MXA, 0>(fal, fah, fbl, fbh, sc, ad);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    auto ptrs = [&](int kt) __attribute__((always_inline)) {
        dm.ra = w128_make_rsrc(abase + (int64_t)(kt + 2) * 128); dm.rb = w128_make_rsrc(bbase + (int64_t)(kt + 2) * 128); ad.pmx = mxbase + (int64_t)(kt + 2) * mx_step;
    };
    // nk is EVEN (launch_gemm_fp8 checks K % 256 == 0): the stage / B-set parity of an iteration is a template parameter, and a tail that chose between two
    // parities at run time made hipcc spill ~100 accumulator tiles where the branches join
    for (int kt = 0; kt < nk - 2; kt += 2) {
        ptrs(kt); W128F8Step<MXA, 0, 0, 0>::run(accL, accH, fal, fah, fbl, fbh, sc, unit, ad, dm);
        ptrs(kt + 1); W128F8Step<MXA, 0, 1, 0>::run(accL, accH, fal, fah, fbl, fbh, sc, unit, ad, dm);
    }
    dm.ra = w128_make_rsrc(abase); dm.rb = dm.ra; ad.pmx = mxbase;
    W128F8Step<MXA, 1, 0, 0>::run(accL, accH, fal, fah, fbl, fbh, sc, unit, ad, dm);
    W128F8Step<MXA, 2, 1, 0>::run(accL, accH, fal, fah, fbl, fbh, sc, unit, ad, dm);
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    auto epi = [&](f32x4 (&acc)[8][4], int nb, unsigned char* slab) __attribute__((always_inline)) {
        if constexpr (EPI == 1) packed_tile_epilogue<2, 2, 8, 4, typename Cfg::Epi, ACT_NONE, false, false, true, false>(g, slab, acc, m0, nb, z, tid, lane, wm, wn);
        else if constexpr (EPI == 2) packed_tile_epilogue<2, 2, 8, 4, typename Cfg::Epi, ACT_GELU_TANH, false, false, true, true>(g, slab, acc, m0, nb, z, tid, lane, wm, wn);
        else if constexpr (EPI == 3) direct_f32_epilogue<2, 2, 8, 4, true, 8>(g, acc, m0, nb, z, lane, wm, wn);      // (the whole half tile's residual in one round trip)
        else dma_tile_epilogue<2, 2, 8, 4, typename Cfg::Epi>(g, slab, acc, m0, nb, z, tid, lane, wm, wn);
    };
    epi(accL, n0, smem);
    if (n0 + 128 < g.N) {
        if constexpr (EPI == 0) __syncthreads();
        epi(accH, n0 + 128, smem + (EPI == 0 || EPI == 3 ? 0 : Cfg::SLAB2_OFF));      // packed epilogues: a slab of its own, no barrier between the halves
    }
}

}  // namespace ncsn

// conv_gn.h -- 3x3 convolution with GroupNorm-apply + SiLU fused into its operand path (k_conv_gn).
//
// Reference arithmetic: ResnetBlockBigGANpp.forward, layerspp.py:242-274:  h = Conv(act(GroupNorm(x))).
// Round 1 ran  act(GroupNorm(.))  as its own HBM pass (k_gn_apply: read the raw tensor, write a normalised, zero-bordered copy;
// 20 % of the device time of a sampling step) and the convolution as a tap-by-tap implicit GEMM that re-fetches its A tile from
// L2 for each of the nine taps (that L2->LDS fill bounds its K loop).  Here the convolution reads the RAW tensor:
//
//   * block tile 256 output pixels x 128 channels, 4 waves of 128x64, <= 80 KB of LDS: TWO blocks per CU, so one block's
//     prologue, epilogue, barriers and VALU work run beside the other block's MFMAs (nothing is hand-staggered);
//   * the K loop walks 32-channel HALF-CHUNKS.  Per half-chunk the block DMAs the PATCH of its pixels -- the pixels plus a
//     one-pixel halo: (8+2) x (32+2) pixels of a 32x32 image, (16+2) x (16+2) of a 16x16 one -- into LDS ONCE (64-byte rows),
//     normalises it IN PLACE (x * scale[b,c] + shift[b,c], SiLU; halo pixels outside the image -> 0) and all nine taps read
//     their A fragments from it: the A-side fill drops 9x and the separate pass disappears;
//   * patch rows are laid out with a row stride WS = 40 / 24 pixels (a multiple of 8), so that every tap shift and every
//     row-tile of a wave is a compile-time byte offset from THREE per-lane base addresses (one per dx): no address arithmetic
//     in the loop;
//   * the only per-tap stream is the weight tile (128 x 64 B) through a 3-slot ring with counted vmcnt waits; the patch of
//     half-chunk h+1 is requested at tap 0 of half-chunk h and normalised in six slices behind taps 3..8;
//   * the 1x1 shortcut segment (a1: Conv_2 of the res-block, raw input, no normalisation) runs after the half-chunks as plain
//     32-wide K-tiles through the same buffers.
// LDS swizzle (64-byte rows, four 16-byte slots): chunk k of row p sits at slot k ^ ((p >> 1) & 2).  It is conflict-free for
// ds_read_b128 on ANY window of 16 consecutive rows (the taps shift the window): a lane group holds the 16 rows once each, rows
// 0-3 / 12-15 with k-chunk q and rows 4-11 with q ^ 1; rows of equal p & 3 share a 64-byte quarter of a bank line, there are
// four of them in a window -- p0, p0+4, p0+8, p0+12 with chunks q, q^1, q^1, q -- and bit 2 of p alternates along them, so
// their slots q ^ 2b, q ^ 1 ^ 2(1-b), q ^ 1 ^ 2b, q ^ 2(1-b) are the four distinct ones.  A row stride that is a multiple of
// 8 rows keeps bit 2 of p, hence the swizzle, under dy shifts.
// K order: the packed weights keep round 1's order (64-channel chunk, tap, channel); K-tile (chunk c, half h, tap t) reads
// columns (c*9 + t)*64 + h*32 -- only the accumulation order differs from the unfused kernels.
#pragma once
#include "gemm_dma.h"

namespace ncsn {

template <int RES> struct PatchGeo;
template <> struct PatchGeo<32> { static constexpr int W = 32, WP = 34, WS = 40; };
template <> struct PatchGeo<16> { static constexpr int W = 16, WP = 18, WS = 24; };

// WIDE = false: block tile 256 pixels x 128 channels (wave tile 128 x 64: A row-tiles stream past four resident weight fragments);
// WIDE = true:  block tile 128 pixels x 256 channels (wave tile 64 x 128: weight col-tiles stream past four resident A fragments) --
//               for N = 256 layers: the patch, hence the normalisation work, per MFMA is 0.55x that of two 256 x 128 tiles.
template <int RES, bool WIDE_ = false>
struct ConvGnCfg {
    using Geo = PatchGeo<RES>;
    static constexpr bool WIDE = WIDE_;
    static constexpr int WM = 2, WN = 2, TM = WIDE ? 4 : 8, TN = WIDE ? 8 : 4, NW = 4, THREADS = 256, NSB = 3, KT = 32;
    static constexpr int BM_ = WM * TM * 16, BN_ = WN * TN * 16;
    static constexpr int PR = BM_ / Geo::W + 2;                         // image rows of a tile + the halo rows
    static constexpr int PPIX = PR * Geo::WS;                           // patch rows in LDS (pad columns included)
    static constexpr int NREAL = PR * Geo::WP;                          // pixels that are ever read
    static constexpr int NPIECE = (PPIX + 15) / 16;                     // 1-KiB DMA pieces (16 patch rows of 64 B)
    static constexpr int PPW = (NPIECE + NW - 1) / NW;                  // per wave (the tail repeats the last piece)
    static constexpr int PSW = BM_ / 16 / NW;                           // shortcut-tile pieces per wave
    static constexpr int PB = BN_ / 16 / NW;                            // weight-tile pieces per wave
    static constexpr int PATCH_BYTES = NPIECE * 1024, BT_BYTES = BN_ * 64, TAB_BYTES = 256;
    static constexpr int AUXP = 1 + PPW;                                // DMA instructions per wave of a (table, patch) request
    static constexpr int TILES_BYTES = 2 * PATCH_BYTES + NSB * BT_BYTES + 2 * TAB_BYTES;
    using Epi = EpiCfg<WM, WN, TM, TN, TILES_BYTES>;
    static constexpr int LDS_BYTES = TILES_BYTES;
    static constexpr int NROUND = (NREAL * 4 + THREADS - 1) / THREADS;  // in-place normalisation slices of a half-chunk
    static_assert(RES * RES % BM_ == 0, "a tile lies inside one image");
    static_assert(Geo::WS % 8 == 0 && Geo::WS >= Geo::WP, "row stride: a multiple of 8 pixels");
    static_assert(NROUND <= 6, "the slices run behind taps 3..8");
    static_assert(Epi::PACK_OK && LDS_BYTES <= 81920, "two blocks per CU");
};

#ifndef NATINF_CG_ABL
#define NATINF_CG_ABL 0            // development: 1 = no normalisation inside the K loop (timing ablation, wrong results)
#endif
// The tile-timeline stamps of k_conv_gn2 (tools/conv_gn_timeline.py) have their own switch since round 4: `make EXTRA="-DNATINF_DEV -DNATINF_CG_TIMELINE"`.
// With them in every -DNATINF_DEV build the 256-register instantiations spilled 2-4 vector registers (130 scalar spills parked in vector lanes on top of the
// round-3 kernel) -- among them destinations of asm loads in flight: the development library faulted in its first fused convolution.
#if defined(NATINF_DEV) && defined(NATINF_CG_TIMELINE)
// timeline builds: shader-clock stamps at the section boundaries of a K-tile (block 0 / wave 0), summed over the tile's K loop
__device__ __forceinline__ unsigned long long cg_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define NATINF_CG_STAMP(v) const unsigned long long v = cg_stamp();
#define NATINF_CG_ADD(acc, a, b) acc += (b) - (a);
#else
#define NATINF_CG_STAMP(v)
#define NATINF_CG_ADD(acc, a, b)
#endif
template <int N> __device__ __forceinline__ void wait_vm_lgkm_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(N) : "memory");
}

// EPI: the packed epilogues of gemm_dma.h (1 plain, 2 + GroupNorm partials, 5 + bf16 residual, 6 both)
template <int RES, bool WIDE, int EPI>
__global__ __launch_bounds__(256, 2) void k_conv_gn(const GemmArgs g)
{
    using Cfg = ConvGnCfg<RES, WIDE>;
    using Geo = typename Cfg::Geo;
    constexpr int BM_ = Cfg::BM_, BN_ = Cfg::BN_, NW = Cfg::NW, THREADS = Cfg::THREADS, TM = Cfg::TM, TN = Cfg::TN, NSB = Cfg::NSB, KT = Cfg::KT;
    constexpr int W = Geo::W, WP = Geo::WP, WS = Geo::WS, HW = RES * RES;
    constexpr int PB = Cfg::PB, PPW = Cfg::PPW, PSW = Cfg::PSW, AUXP = Cfg::AUXP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_poison();
    unsigned char* const sPatch = smem;                                   // [2][PATCH_BYTES]
    unsigned char* const sB = smem + 2 * Cfg::PATCH_BYTES;                // [NSB][BT_BYTES]
    unsigned char* const sTab = sB + NSB * Cfg::BT_BYTES;                 // [2][scale 32 | shift 32] fp32
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nN = (g.N + BN_ - 1) / BN_, nM = g.M / BM_;
    const int tile = xcd_remap(blockIdx.x, nM * nN);
    const int mt = tile / nN, nt = tile - mt * nN;                        // the N-tiles of a pixel tile are neighbours: the raw patch is an L2 hit
    const int m0 = mt * BM_, n0 = nt * BN_;
    const int b = m0 / HW, y0 = (m0 % HW) / W;                            // image and first image row of this tile
    const bf16* const img = g.a0 + (int64_t)b * HW * g.a0_ld;
    const float* const gsc = g.gn_scale + (int64_t)b * g.gn_ld;
    const float* const gsh = g.gn_shift + (int64_t)b * g.gn_ld;
    const int n_half = g.a0_C / KT, n_sc = g.a1 ? g.a1_C / KT : 0;
    const int nk = 9 * n_half, NT = nk + n_sc;
    const int K0 = 9 * g.a0_C;

    // Every LDS access and every LDS-DMA inside the K loop is inline asm.  (i) With a builtin LDS-DMA in flight hipcc puts
    // `s_waitcnt vmcnt(0)` in front of any LDS access IT can see (a pending LDS write to it), which drains the weight ring at every
    // tap.  (ii) The builtin takes a 64-bit per-lane address: a dozen VALU ops per 1-KiB piece, or -- hoisted -- two registers per
    // piece.  Here a piece is `global_load_lds_dwordx4 voffset, sbase`: a 32-bit per-lane byte offset (the weight rows: computed once per tile, two registers;
    // the patch rows: recomputed at tap 0 of every half-chunk) and a scalar base that carries everything that changes from tap to tap; no vector instruction per request.
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    auto lds_addr = [](const unsigned char* p) __attribute__((always_inline)) { return (unsigned)(uintptr_t)((lds_u8*)const_cast<unsigned char*>(p)); };
    auto glds16 = [](unsigned voff, const void* sbase, unsigned lds_dst) __attribute__((always_inline)) {
        NATINF_M0_ASM_BEGIN
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
        NATINF_M0_ASM_END
    };
    unsigned off_b[PB];
    {
        const int prow = lane >> 2, pslot = lane & 3;
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int r = (wave * PB + j) * 16 + prow;
            off_b[j] = (unsigned)(min(n0 + r, g.N - 1) * g.b_ld + ((pslot ^ ((r >> 1) & 2)) << 3)) * 2u;
        }
    }
    const unsigned lds_patch = lds_addr(sPatch), lds_b = lds_addr(sB), lds_tab = lds_addr(sTab);
    // (every request lambda takes a piece range [j0, j1): inside the K loop the pieces are issued one or two at a time behind
    // MFMA groups -- an LDS-DMA instruction costs its wave 60-185 issue cycles (MI355X_MICROARCH.md), which then pass while the
    // wave's own MFMAs execute instead of in front of them)
    auto issue_b = [&](int kt, int j0 = 0, int j1 = Cfg::PB) __attribute__((always_inline)) {            // weight K-tile kt -> ring slot kt % NSB
        int col;
        if (kt < nk) { const int hc = kt / 9, t = kt - 9 * hc; col = ((hc >> 1) * 9 + t) * 64 + (hc & 1) * KT; }
        else col = K0 + (kt - nk) * KT;
        const unsigned dst = lds_b + (kt % NSB) * Cfg::BT_BYTES + wave * (PB * 1024);
        const bf16* base = g.b + col;
#pragma unroll
        for (int j = 0; j < PB; ++j)
            if (j >= j0 && j < j1) glds16(off_b[j], base, dst + j * 1024);
    };
    // items: 0 = the (scale | shift) table, 1 .. PPW = the raw patch pieces of half-chunk hc; -> buffers hc & 1
    auto issue_patch = [&](int hc, int j0 = 0, int j1 = 1 + Cfg::PPW) __attribute__((always_inline)) {
        const int buf = hc & 1;
        if (j0 == 0) {
            const float* src = (lane < 32 ? gsc : gsh - 32) + (unsigned)(hc * KT + lane);
            const unsigned dst = lds_tab + buf * Cfg::TAB_BYTES;
            NATINF_M0_ASM_BEGIN
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(src), "s"(dst) : "memory", "m0");
            NATINF_M0_ASM_END
        }
        const bf16* base = img + hc * KT;
        int l = lane;
        asm volatile("" : "+v"(l));                                          // offsets recomputed per request (once per nine taps): no registers held
        const int prow = l >> 2, pslot = l & 3;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            if (j + 1 < j0 || j + 1 >= j1) continue;
            const int q = min(wave * PPW + j, Cfg::NPIECE - 1);           // wave-uniform piece index (the tail repeats the last piece)
            const int pp = q * 16 + prow;
            const int yy = pp / WS, xx = pp - yy * WS;
            const int y = min(max(y0 - 1 + yy, 0), RES - 1), x = min(max(xx - 1, 0), RES - 1);      // halo / pad: any readable pixel
            glds16((unsigned)((y * W + x) * g.a0_ld + ((pslot ^ ((pp >> 1) & 2)) << 3)) * 2u, base, lds_patch + buf * Cfg::PATCH_BYTES + q * 1024);
        }
    };
    auto issue_shortcut = [&](int s, int j0 = 0, int j1 = Cfg::PSW) __attribute__((always_inline)) {      // plain [256][32] tile of a1 -> patch buffer (n_half + s) & 1
        const unsigned dst = lds_patch + ((n_half + s) & 1) * Cfg::PATCH_BYTES + wave * (PSW * 1024);
        int l = lane;
        asm volatile("" : "+v"(l));                                          // recomputed per call (a few tiles per launch): no registers held
        const int prow = l >> 2, pslot = l & 3;
        const bf16* base = g.a1 + (int64_t)m0 * g.a1_ld + s * KT;
#pragma unroll
        for (int j = 0; j < PSW; ++j) {
            if (j < j0 || j >= j1) continue;
            const int pp = (wave * PSW + j) * 16 + prow;
            glds16((unsigned)(pp * g.a1_ld + ((pslot ^ ((pp >> 1) & 2)) << 3)) * 2u, base, dst + j * 1024);
        }
    };
    // ---- in-place normalisation of the patch, slice j: thread t owns the 16-byte slots t + 256 j of the REAL pixels (threads
    // ---- past the end work on a pad pixel nobody reads).  Kept as ONE block in front of the K-tile's MFMAs: spreading it one
    // ---- element per MFMA group was measured (same speed: the vector work competes with the co-resident block's MFMAs for the
    // ---- SIMD's issue slots wherever it sits) and costs 30 registers held across the tile.
    // Per slice the thread needs its slot's LDS address and whether the pixel lies inside the image: computed once per tile and
    // packed one slice per register (address in buffer 0 | inside << 31); the table row follows from the address (bit 8 of the
    // slot address is bit 2 of the patch row: both patch buffers start on multiples of 512 B).
    static_assert(Cfg::PATCH_BYTES % 512 == 0, "swizzle bit from the slot address");
    unsigned ninfo[Cfg::NROUND];
#pragma unroll
    for (int j = 0; j < Cfg::NROUND; ++j) {
        const int e = tid + THREADS * j;
        const bool live = e < Cfg::NREAL * 4;
        const int px = live ? e >> 2 : 0, s = e & 3;
        const int yy = px / WP, xx = live ? px - yy * WP : WP;              // dead threads: pad column WP of patch row 0
        const bool in = live && (unsigned)(y0 - 1 + yy) < (unsigned)RES && (unsigned)(xx - 1) < (unsigned)RES;
        ninfo[j] = (unsigned)((yy * WS + xx) * 64 + s * 16) | (in ? 0x80000000u : 0u);
    }
    // A slice = five LDS reads (norm_load, ahead of the K-tile's fragment reads), eight elements of arithmetic, one LDS write.
    // ONE wave cannot overlap its own MFMAs with anything unless the other work sits BETWEEN them: an MFMA occupies the matrix pipe
    // for 16 cycles and the issue port for 8, which leaves two vector-issue slots per MFMA (measured with one block per CU: 1,750
    // clocks per tap for 512 clocks of MFMAs when the slice ran as a block of its own; stamps in tools/conv_gn_timeline.py).  So
    // element i of the slice -- unpack, fma, exp2, add, rcp, mul: six instructions in the folded form -- is scheduled INTO MFMA
    // group i (sched_group_barrier: one MFMA, two vector instructions, four times).  Folded form (GemmArgs::gn_folded, the only one
    // the kernel implements): scale / shift carry -log2(e), so t = x*scale + shift = -log2(e) v, exp2(t) = exp(-v) and
    // t / (1 + exp2(t)) = -log2(e) silu(v); the 3x3 weights carry -ln 2.
    u32x4 nv = {0u, 0u, 0u, 0u}, ns0 = nv, ns1 = nv, nh0 = nv, nh1 = nv;
    unsigned npa = 0, npk[4] = {0u, 0u, 0u, 0u};
    float nf_even = 0.f;
    auto norm_load = [&](int j, int buf) __attribute__((always_inline)) {
        unsigned inf = ninfo[j];
        asm volatile("" : "+v"(inf));                                        // (or hipcc hoists both addresses of all six slices out of the loop, and spills them)
        npa = lds_patch + buf * Cfg::PATCH_BYTES + (inf & 0x7fffffffu);
        const unsigned ta = lds_tab + buf * Cfg::TAB_BYTES + (((inf >> 4) ^ ((inf >> 7) & 2)) & 3) * 32;
        nv = lds_read16<0>(npa);
        ns0 = lds_read16<0>(ta); ns1 = lds_read16<16>(ta); nh0 = lds_read16<128>(ta); nh1 = lds_read16<144>(ta);
        npa |= inf & 0x80000000u;                                           // bit 31: the pixel lies inside the image
    };
    // element I: after an lgkmcnt wait that covers norm_load's reads.  The first asm makes the slice's registers "new" values (to
    // hipcc an asm ds_read's result exists at once: it would hoist the arithmetic above the wait at IR level); the last one pins the
    // result here (it would otherwise sink the arithmetic into norm_store's branch).
#define NATINF_CG_NORM_PRE(I) asm volatile("" : "+v"(nv), "+v"(ns0), "+v"(ns1), "+v"(nh0), "+v"(nh1));
#define NATINF_CG_NORM_EL(I)                                                                                                \
        {                                                                                                                    \
            const unsigned w_ = nv[(I) >> 1];                                                                                \
            const float x_ = __uint_as_float(((I) & 1) ? (w_ & 0xffff0000u) : (w_ << 16));                                   \
            const float t_ = x_ * __uint_as_float((I) < 4 ? ns0[(I) & 3] : ns1[(I) & 3]) + __uint_as_float((I) < 4 ? nh0[(I) & 3] : nh1[(I) & 3]); \
            const float y_ = t_ * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t_));                                  \
            if constexpr (((I) & 1) == 0) nf_even = y_;                                                                      \
            else {                                                                                                           \
                typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));                                                 \
                const bf16x2_t pr_ = {(bf16)nf_even, (bf16)y_};                                                              \
                npk[(I) >> 1] = __builtin_bit_cast(unsigned, pr_);                                                           \
            }                                                                                                                \
        }                                                                                                                    \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                                                   \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       /* one MFMA */                                           \
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);       /* two vector instructions */                            \
        }
#define NATINF_CG_NORM_POST(I) if constexpr (((I) & 1) == 0) asm volatile("" : "+v"(nf_even)); else asm volatile("" : "+v"(npk[(I) >> 1]));
    auto norm_store = [&]() __attribute__((always_inline)) {
        u32x4 ou = {npk[0], npk[1], npk[2], npk[3]};
        if ((int)npa >= 0) ou = u32x4{0u, 0u, 0u, 0u};
        const unsigned pa = npa & 0x7fffffffu;
        asm volatile("ds_write_b128 %0, %1" :: "v"(pa), "v"(ou) : "memory");
    };
    auto norm_round = [&](int j, int buf) __attribute__((always_inline)) {   // a whole slice at once (prologue)
        norm_load(j, buf);
        wait_lgkmcnt<0>();
#define NATINF_CG_NORM_ONE(I) NATINF_CG_NORM_PRE(I) NATINF_CG_NORM_EL(I) NATINF_CG_NORM_POST(I) __builtin_amdgcn_sched_barrier(0);
        NATINF_CG_NORM_ONE(0) NATINF_CG_NORM_ONE(1) NATINF_CG_NORM_ONE(2) NATINF_CG_NORM_ONE(3)
        NATINF_CG_NORM_ONE(4) NATINF_CG_NORM_ONE(5) NATINF_CG_NORM_ONE(6) NATINF_CG_NORM_ONE(7)
#undef NATINF_CG_NORM_ONE
        norm_store();
    };
#define NATINF_CG_NO_PRE(I)
#define NATINF_CG_NO_POST(I)
#define NATINF_CG_NO_EL(I)

    // ---- fragment addresses: three per-lane bases (dx = -1, 0, +1) at dy = -1; everything else is an immediate ------------
    const int frow = lane & 15, fq = lane >> 4;
    unsigned a_dx[3];
    {
        const int ml = wm * (TM * 16) + frow;                             // first pixel row-tile of this wave
        const int pc = ((ml / W) + 1) * WS + (ml % W) + 1;                // its patch row at the centre tap
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int pp = pc - WS + d - 1;
            a_dx[d] = lds_patch + pp * 64 + ((fq ^ ((pp >> 1) & 2)) << 4);
        }
    }
    const int brow = wn * (TN * 16) + frow;
    const unsigned b_base = lds_b + brow * 64 + ((fq ^ ((brow >> 1) & 2)) << 4);
    const int arow = wm * (TM * 16) + frow;
    const unsigned a_plain = lds_patch + arow * 64 + ((fq ^ ((arow >> 1) & 2)) << 4);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // One 32-wide K-tile with a hand-counted fragment pipeline: the four resident fragments (weights; WIDE: A row-tiles) and streamed
    // fragments 0, 1 (A row-tiles; WIDE: weight col-tiles) are requested up front, streamed fragment s+2 while s is multiplied; `s_waitcnt lgkmcnt(n)` retires exactly the fragment the next
    // four MFMAs need (LDS returns in order; the five reads of a normalisation slice are older than all of them),
    // sched_barrier(0) keeps each MFMA group behind its wait (guide 5.4 rule 18).  EL(i): the vector work placed behind MFMA
    // group i.  AOFF(i): byte offset of row-tile i from `a` (compile time); BOFF: of the ring slot from `bb`.
    // Step S of eight: the streamed fragment S+2 is requested, fragment S retired, and it meets the four resident ones.
#define NATINF_CG_STEP(a, AOFF, bb, BOFF, S, EL)                                                                            \
        if constexpr ((S) + 2 < 8) {                                                                                         \
            if constexpr (WIDE) fs[((S) + 2) % 3] = lds_read16<(BOFF) + (((S) + 2) % 8) * 1024>(bb);                         \
            else fs[((S) + 2) % 3] = lds_read16<AOFF(((S) + 2) % 8)>(a);                                                     \
        }                                                                                                                    \
        wait_lgkmcnt<((S) + 2 < 8 ? 2 : 7 - (S))>();                                                                          \
        EL##_PRE(S)                                                                                                          \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {                                                                   \
            if constexpr (WIDE) acc[r_][S] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fs[(S) % 3]), __builtin_bit_cast(bf16x8, fr[r_]), acc[r_][S], 0, 0, 0); \
            else acc[S][r_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr[r_]), __builtin_bit_cast(bf16x8, fs[(S) % 3]), acc[S][r_], 0, 0, 0); \
        }                                                                                                                    \
        EL##_EL(S)                                                                                                           \
        EL##_POST(S)                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);
#define NATINF_CG_HEAD(a, AOFF, bb, BOFF)                                                                                   \
        u32x4 fr[4], fs[3];                                                                                                  \
        if constexpr (WIDE) {                                                                                                \
            fr[0] = lds_read16<AOFF(0)>(a); fr[1] = lds_read16<AOFF(1)>(a); fr[2] = lds_read16<AOFF(2)>(a); fr[3] = lds_read16<AOFF(3)>(a); \
            fs[0] = lds_read16<(BOFF)>(bb); fs[1] = lds_read16<(BOFF) + 1024>(bb);                                           \
        } else {                                                                                                             \
            fr[0] = lds_read16<(BOFF)>(bb); fr[1] = lds_read16<(BOFF) + 1024>(bb);                                           \
            fr[2] = lds_read16<(BOFF) + 2048>(bb); fr[3] = lds_read16<(BOFF) + 3072>(bb);                                    \
            fs[0] = lds_read16<AOFF(0)>(a); fs[1] = lds_read16<AOFF(1)>(a);                                                  \
        }
#define NATINF_CG_BODY(a, AOFF, bb, BOFF, EL)                                                                               \
        NATINF_CG_STEP(a, AOFF, bb, BOFF, 0, EL) NATINF_CG_STEP(a, AOFF, bb, BOFF, 1, EL) NATINF_CG_STEP(a, AOFF, bb, BOFF, 2, EL) \
        NATINF_CG_STEP(a, AOFF, bb, BOFF, 3, EL) NATINF_CG_STEP(a, AOFF, bb, BOFF, 4, EL) NATINF_CG_STEP(a, AOFF, bb, BOFF, 5, EL) \
        NATINF_CG_STEP(a, AOFF, bb, BOFF, 6, EL) NATINF_CG_STEP(a, AOFF, bb, BOFF, 7, EL)
#define NATINF_CG_TILE(a, AOFF, bb, BOFF, EL) { NATINF_CG_HEAD(a, AOFF, bb, BOFF) NATINF_CG_BODY(a, AOFF, bb, BOFF, EL) }
    static_assert(TM * TN == 32 && (WIDE ? TM : TN) == 4, "eight steps of four MFMAs");

    // head of a K-tile: wait until weight tile kt (and everything older) has landed, `allowed` younger requests stay in flight
    auto wait_tile = [&](int aux, bool next_b) __attribute__((always_inline)) {
        if (aux == 0) { if (next_b) wait_vm_lgkm_barrier<PB>(); else wait_vm_lgkm_barrier<0>(); }
        else if (aux == AUXP) { if (next_b) wait_vm_lgkm_barrier<AUXP + PB>(); else wait_vm_lgkm_barrier<AUXP>(); }
        else { if (next_b) wait_vm_lgkm_barrier<PSW + PB>(); else wait_vm_lgkm_barrier<PSW>(); }
    };

#ifdef NATINF_CG_TIMELINE
    unsigned long long dbg_wait = 0, dbg_norm = 0, dbg_mfma = 0;
    const unsigned long long dbg_t0 = cg_stamp();
#endif
    // ---- prologue: table + patch of half-chunk 0, weight tiles 0 and 1; half-chunk 0 is normalised before the loop -----------
    issue_patch(0);
    issue_b(0);
    if (NT > 1) issue_b(1);
    wait_vm_lgkm_barrier<0>();
#pragma unroll
    for (int j = 0; j < Cfg::NROUND; ++j) norm_round(j, 0);

    // ---- the nine taps of one half-chunk; BUF (its patch buffer) and the tap index are compile-time: all offsets are immediates
#define NATINF_CG_AOFF(i) (BUF * Cfg::PATCH_BYTES + ((RES == 32 ? ((i) >> 1) * WS + ((i) & 1) * 16 : (i) * WS) + (T / 3) * WS) * 64)
    auto tap = [&](auto buf_tag, auto t_tag, int hc, bool next_half, int aux0) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_tag)::value, T = decltype(t_tag)::value;
        const int kt = hc * 9 + T;
        NATINF_CG_STAMP(ts0)
        // requests younger than weight tile kt: those of taps T-2 and T-1 (aux only at tap 0) + weight tile kt+1
        wait_tile((T == 1 || T == 2) ? aux0 : 0, kt + 1 < NT);
        NATINF_CG_STAMP(ts1)
        // The requests of this step: weight tile kt+2 first, then the aux request -- the order the vmcnt counts above assume.
        // (Measured and not kept, all within noise of this form: one piece behind each MFMA group; waves 0, 1 at the head of the
        // tap and waves 2, 3 at its end -- the co-resident block already covers the issue cost of the LDS-DMA instructions.)
        if (kt + 2 < NT) issue_b(kt + 2);
        if constexpr (T == 0) {
            if (next_half) issue_patch(hc + 1);
            else if (n_sc > 0) issue_shortcut(0);
        }
        constexpr bool NORM_TAP = T >= 3 && T - 3 < Cfg::NROUND && NATINF_CG_ABL != 1;
        // the slice's five LDS reads go out FIRST (older than every fragment read: the counted waits of the steps cover them), its
        // arithmetic comes after the last MFMA group: no LDS round trip is exposed, and nothing is held across the barrier
        if constexpr (NORM_TAP) { if (next_half) norm_load(T - 3, BUF ^ 1); }
        NATINF_CG_HEAD(a_dx[T % 3], NATINF_CG_AOFF, b_base, (T % 3) * Cfg::BT_BYTES)
        NATINF_CG_STAMP(ts2)
        if constexpr (NORM_TAP) {
            NATINF_CG_BODY(a_dx[T % 3], NATINF_CG_AOFF, b_base, (T % 3) * Cfg::BT_BYTES, NATINF_CG_NORM)
            if (next_half) norm_store();
        } else {
            NATINF_CG_BODY(a_dx[T % 3], NATINF_CG_AOFF, b_base, (T % 3) * Cfg::BT_BYTES, NATINF_CG_NO)
        }
        NATINF_CG_STAMP(ts3)
        NATINF_CG_ADD(dbg_wait, ts0, ts1) NATINF_CG_ADD(dbg_norm, ts1, ts2) NATINF_CG_ADD(dbg_mfma, ts2, ts3)
    };
    auto half_chunk = [&](auto buf_tag, int hc) __attribute__((always_inline)) {
        const bool next_half = hc + 1 < n_half;
        const int aux0 = next_half ? AUXP : (n_sc > 0 ? PSW : 0);        // the aux request of tap 0
        using std::integral_constant;
        tap(buf_tag, integral_constant<int, 0>{}, hc, next_half, aux0); tap(buf_tag, integral_constant<int, 1>{}, hc, next_half, aux0);
        tap(buf_tag, integral_constant<int, 2>{}, hc, next_half, aux0); tap(buf_tag, integral_constant<int, 3>{}, hc, next_half, aux0);
        tap(buf_tag, integral_constant<int, 4>{}, hc, next_half, aux0); tap(buf_tag, integral_constant<int, 5>{}, hc, next_half, aux0);
        tap(buf_tag, integral_constant<int, 6>{}, hc, next_half, aux0); tap(buf_tag, integral_constant<int, 7>{}, hc, next_half, aux0);
        tap(buf_tag, integral_constant<int, 8>{}, hc, next_half, aux0);
    };
#ifdef NATINF_CG_TIMELINE
    const unsigned long long dbg_t1 = cg_stamp();
#endif
    for (int hc = 0; hc < n_half; hc += 2) {                              // a0_C is a multiple of 64: half-chunks come in pairs
        half_chunk(std::integral_constant<int, 0>{}, hc);
        half_chunk(std::integral_constant<int, 1>{}, hc + 1);
    }
    // ---- 1x1 shortcut segment: plain [256][32] A tiles, two-stage (tile s+1 requested at the head of tile s)
#define NATINF_CG_POFF(i) ((i) * 1024)
    for (int s = 0; s < n_sc; ++s) {
        const int kt = nk + s;
        if (s == 0) wait_tile(0, kt + 1 < NT);                             // its A tile was requested nine taps ago
        else wait_vm_lgkm_barrier<0>();                                    // A(s) was the last request of the previous step
        if (kt + 2 < NT) issue_b(kt + 2);
        if (s + 1 < n_sc) issue_shortcut(s + 1);
        const unsigned pa = a_plain + ((n_half + s) & 1) * Cfg::PATCH_BYTES, pb = b_base + (kt % NSB) * Cfg::BT_BYTES;
        NATINF_CG_TILE(pa, NATINF_CG_POFF, pb, 0, NATINF_CG_NO)
    }
#undef NATINF_CG_POFF
#undef NATINF_CG_AOFF
#undef NATINF_CG_STEP
#undef NATINF_CG_TILE
#undef NATINF_CG_HEAD
#undef NATINF_CG_BODY
#undef NATINF_CG_NO_EL
#undef NATINF_CG_NO_PRE
#undef NATINF_CG_NO_POST
#undef NATINF_CG_NORM_PRE
#undef NATINF_CG_NORM_EL
#undef NATINF_CG_NORM_POST
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave is done with the tiles before the epilogue reuses them
    // The epilogue's arguments (bias, row vector, residual, partial-sum table, ...) are fetched from the kernel-argument segment
    // HERE, through a pointer hipcc cannot see through: kept in scalar registers across the K loop they cost ~40 SGPRs, the
    // allocator spilled them into vector-register lanes, and the vector registers it took for that tipped the loop into scratch
    // spills -- whose loads and stores would corrupt the hand-counted vmcnt waits above.
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const unsigned __attribute__((address_space(4))) *kernarg_u32_t;
    kernarg_u32_t gp = (kernarg_u32_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(gp));
    GemmArgs ge;
    {
        unsigned* d = reinterpret_cast<unsigned*>(&ge);
#pragma unroll
        for (unsigned i = 0; i < sizeof(GemmArgs) / 4; ++i) d[i] = gp[i];
    }
#else
    const GemmArgs ge = g;
#endif
#ifdef NATINF_CG_TIMELINE
    const unsigned long long dbg_t2 = cg_stamp();
#endif
    tile_epilogue<Cfg::WM, Cfg::WN, TM, TN, typename Cfg::Epi, EPI>(ge, smem, acc, m0, n0, 0, tid, lane, wm, wn);
#ifdef NATINF_CG_TIMELINE
    if (ge.dbg_ts && tid == 0 && (blockIdx.x == 0 || blockIdx.x == 777)) {
        unsigned long long* o = ge.dbg_ts + (blockIdx.x ? 8 : 0);
        o[0] = dbg_t1 - dbg_t0; o[1] = dbg_wait; o[2] = dbg_norm; o[3] = dbg_mfma; o[4] = dbg_t2 - dbg_t1; o[5] = cg_stamp() - dbg_t2; o[6] = (unsigned long long)nk;
    }
#endif
}

}  // namespace ncsn

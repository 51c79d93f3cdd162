// conv_gn2.h -- k_conv_gn2: the fused GroupNorm-apply + SiLU + 3x3 convolution of conv_gn.h with the WEIGHTS STREAMED THROUGH
// REGISTERS instead of an LDS ring.
//
// What bounded k_conv_gn (tools/conv_gn_timeline.py, profiles/r02): its only per-tap stream was the 128 x 64 B weight tile, but that
// stream put an `s_waitcnt + s_barrier` in front of every 32-wide K step (nine per half-chunk) and two LDS-DMA instructions per wave
// and tap (60-185 issue cycles each) on the waves that also issue the MFMAs.  Here
//   * the weights are stored FRAGMENT-MAJOR (k_pack_frag: one contiguous 1-KiB block per (16 output channels, 32-wide K step), lane l
//     holds its MFMA operand at l * 16): a wave fetches the four fragments of its 64 output channels for tap t+1 with four coalesced
//     `global_load_dwordx4` straight into registers while it multiplies tap t (two register sets; an L2-warm load returns in 160-330
//     cycles, a tap lasts ~1,000) -- no LDS ring, no LDS fragment reads for the weights, no cross-wave dependency;
//   * a wave normalises exactly the patch pieces IT requested (piece j * 4 + wave: the LDS-DMA destination is lane-linear, so the 16
//     bytes a lane fetched are the 16 bytes it normalises in place), so the raw patch needs no hand-off between waves either;
//   * what is left is ONE barrier per half-chunk (nine taps, 288 MFMAs per wave): "the normalised patch of half-chunk h is complete
//     and everyone is done reading the buffer of h-1".
// Both tile shapes use the same wave tile (128 pixels x 64 channels: eight A row-tiles stream past four resident weight fragments):
// 256 x 128 = 2 x 2 waves, 128 x 256 (N = 256 layers at 16x16) = 1 x 4 waves, all four reading the same A fragments.
// Arithmetic (folded SiLU form, K order, swizzle, epilogues): as conv_gn.h.
// Measured on top of this form and not kept (same-box A/B, tools/bench_conv_gn.py; DESIGN.md section 4 has the numbers):
//   * the taps as ONE software pipeline (A fragments of tap t+1 requested in the last two steps of tap t, the four weight loads and the
//     request items of tap 0 spread one per MFMA group): 128x256 tile 4-5 % SLOWER, 256x128 tile equal;
//   * normalisation elements in pairs (no transcendental result consumed by the next instruction: hipcc's s_nop padding halves): equal;
//   * the 256x128 tile as 1 x 4 waves of 256 pixels x 32 channels (every weight fragment fetched by ONE wave: half the L1 traffic of the
//     weight stream, twice the A fragment reads): 0-3 % slower;
//   * 128 x 128 tiles of 2 x 2 waves with 64 x 64 wave tiles (168 registers, THREE blocks per CU = three waves per SIMD to cover each
//     other's stalls, for twice the weight bytes and 1.2x the patch per flop): 4-9 % slower;
//   * EPI 5 / 6: every thread touching two of the residual tile's 512 cache lines three taps before the end, so that the epilogue's
//     residual fetch hits L2: +-0 (the +20-30 us of a residual launch are its 134 MB of extra HBM traffic, not latency);
//   * v_mfma_f32_32x32x16_bf16 (half the MFMA instructions, 1.5x the vector-issue room per matrix-pipe cycle -- tools/probes/
//     mfma_issue_probe.hip --, swizzle by patch column, 32x32 packed epilogue): 5 % fewer shader cycles (PMC), equal wall time.
//   * (round 4) the element's chain cut into three stages ONE MFMA GROUP APART (group I issues unpack-fma-exp2 of element I + 1, add-rcp of element I, mul-pack
//     of element I - 1: nothing issued behind a group depends on a result produced behind the same group; each stage pinned with an opaque asm, or hipcc
//     sinks them back together and spills 60-200 registers; 0 spills at 255-256 registers with the pins): 32x32 K = 3,456 +2-5 %, 16x16 K = 4,608 +1.5 %,
//     everything else +-1 % -- the in-order wave's dependency chain is not what holds the loop back -- and the four extra live registers tip the
//     -DNATINF_DEV build (130 scalar spills in vector lanes) into spilling the destination of an asm load in flight: a fault.  Not kept.
//   * (round 4) THREE weight register sets for the TM = 4 instantiations (8x8 / 4x4: tap t + 2 requested while tap t is multiplied, set t % 3, counted vmcnt(TN) from
//     tap 2 on; 204 / 114 registers, 0 spills, parity green): +-1 % on every 8x8 / 4x4 shape of tools/bench_conv_gn.py -- the one-tap-ahead weight request is not what
//     a 4x4 tap's ~1,100 clocks (for 128 clocks of MFMAs) are made of.  Not kept.
// PMC picture of the 256x128 tile (tools/pmc_conv_gn.sh): matrix pipe 48-55 % busy, LDS 25 %, L1/TA ~45 %, waves 24 % in s_waitcnt and
// 38 % ready-but-not-issued: no unit is saturated; two waves per SIMD do not cover each other's dependency stalls.
#pragma once
#include "conv_gn.h"
#ifndef NATINF_CG_PK
#define NATINF_CG_PK 1            // 1 = packed-fp32 arithmetic (v_pk_fma / v_pk_add / v_pk_mul) in the PROLOGUE's normalisation rounds, 0 = scalar (A/B builds)
#endif

namespace ncsn {

// Patch geometry of k_conv_gn2: patch rows follow each other WITHOUT pad columns (row stride WS = W + 2 pixels; k_conv_gn pads the stride
// to a multiple of 8: 40 / 24).  What made the padding necessary there was the swizzle key -- bit 2 of the linear patch-row index p, which a
// dy shift (p += WS) must not change.  Keyed on bit 2 of the pixel COLUMN xx = p % WS instead, the swizzle is dy-invariant for any stride, and
// still conflict-free: a fragment read covers 16 consecutive columns of ONE patch row, rows of equal bank quarter (p & 3) are four columns
// apart, and bit 2 of xx alternates along them exactly as bit 2 of p did.  Patch rows per tile: 340 instead of 400 (32x32), 180 / 324 instead
// of 240 / 432 (16x16): 15-25 % less to fetch, to normalise and to keep in LDS.
template <int RES> struct PatchGeo2 { static constexpr int W = RES, WP = RES + 2, WS = RES + 2; };

// RES = 8 (round 3): a 128-pixel tile holds TWO whole 8x8 images.  Each image has its own patch -- (8+2) x (8+2) pixels, padded to 112 patch
// rows so that a 16-row DMA piece never straddles two images -- its own (scale | shift) table, its own row of the time-embedding vector and its
// own GroupNorm partials (tile_epilogue's NSAMP); a 16-pixel row-tile is two image rows, which the per-lane fragment bases absorb.  Swizzle key
// for 10-pixel patch rows: xx & 2 (searched over every 2 x 8 window and ds_read_b128 lane group: conflict-free; bit 2 of the column is not).
// TM_ = 4 (RES = 8 / 4): 64-pixel x 256-channel tiles, wave tile 64 x 64 -- ONE 8x8 image per tile, so that the 8x8 level launches two blocks per CU
// (512 tiles at B = 512) instead of one; a normalisation round (eight elements per lane) then spans TWO taps of four MFMA groups.
// RES = 4: FOUR whole 4x4 images per tile -- a 16-pixel row-tile is one image; per image a (4+2) x (4+2) patch padded to 48 patch rows (three DMA
// pieces), its own table, row vector and partials.  Swizzle key for 6-pixel patch rows: xx & 2 as well (checked over every tap and ds_read_b128 lane group).
// NG_ = 2 (RES = 4): TWO groups of four waves per block, group g multiplying the half-chunks (and shortcut tiles) g, g + 2, g + 4, ... of the SAME
// output tile, each with its own patch buffers, tables and weight stream; the second group's accumulators cross LDS once at the end and the first group
// runs the epilogue.  The 4x4 level launches 128 tiles at B = 512 -- one block per CU on half the chip -- and a single wave per SIMD cannot cover its
// own LDS / weight-stream latencies (tile timeline: ~930 clocks per tap for 256 clocks of MFMAs); two waves per SIMD do what the second
// co-resident block does at the other resolutions.
// TN_ = 2 (RES = 4): wave tile 64 pixels x 32 channels, block tile 64 x 128 -- TWO column tiles per row tile, so that the 4x4 level launches 256 blocks (one per
// CU) instead of 128; each block streams only its half of the weight matrix, the (small) patch is normalised by both.
template <int RES, bool WIDE_ = false, int TM_ = 8, int NG_ = 1, int TN_ = 4>
struct ConvGn2Cfg {
    using Geo = PatchGeo2<RES>;
    static constexpr bool WIDE = WIDE_;
    static constexpr int WM = WIDE ? 1 : 2, WN = WIDE ? 4 : 2, TM = TM_, TN = TN_, NW = 4, NG = NG_, THREADS = 256 * NG_, KT = 32;
    // (four groups -- 1,024 threads, four waves per SIMD -- were built too: the same 21 / 32 us as two at K = 2,304 / 4,608, and a parity failure at B = 512 that
    // was not chased; the reduction below is written for any group count)
    static_assert(NG_ == 1 || (NG_ == 2 && RES == 4), "K groups: the 4x4 level only");
    static_assert(TN_ == 4 || (TN_ == 2 && RES == 4), "32-channel wave tiles: the 4x4 level only");
    static_assert(RES == 4 ? (TM_ == 4 && WIDE_) : (TM_ == 8 || (TM_ == 4 && RES == 8 && WIDE_)), "the 4-row-tile form exists for the 8x8 and 4x4 levels only");
    static constexpr int BM_ = WM * TM * 16, BN_ = WN * TN * 16;
    static constexpr int NIMG = RES * RES >= BM_ ? 1 : BM_ / (RES * RES);   // whole images per tile (RES = 8: 2)
    static constexpr int IMGP = NIMG > 1 ? ((RES + 2) * (RES + 2) + 15) / 16 * 16 : 0;      // patch rows per image when a tile holds several
    static constexpr int PR = BM_ / Geo::W + 2;                         // image rows of a tile + the halo rows (NIMG == 1)
    static constexpr int PLAST = NIMG > 1 ? NIMG * IMGP : (PR - 1) * Geo::WS + Geo::WP;          // patch rows that are ever read
    static constexpr int NPIECE = (PLAST + 15) / 16;                    // 1-KiB DMA pieces (16 patch rows of 64 B)
    static constexpr int NROUND = (NPIECE + NW - 1) / NW;               // piece j * NW + wave belongs to wave `wave`, round j
    static constexpr int NFULL = NPIECE / NW;                           // rounds in which every wave has a piece
    static constexpr int PSW = BM_ / 16 / NW;                           // shortcut-tile pieces per wave
    static constexpr int PATCH_BYTES = (NPIECE > BM_ / 16 ? NPIECE : BM_ / 16) * 1024, TAB_IMG_BYTES = 256, TAB_BYTES = NIMG * TAB_IMG_BYTES;
    static constexpr int VOFF_BYTES = NROUND * 1024;                    // the lanes' patch-request offsets, one 32-bit word per lane, wave and round (see issue_patch)
    static constexpr int TILES_BYTES = 2 * PATCH_BYTES + 2 * TAB_BYTES + VOFF_BYTES;
    using Epi = EpiCfg<WM, WN, TM, TN, 81920>;
    static constexpr int EPI_BYTES = Epi::PACK_BYTES + (NIMG - 1) * WN * TN * 4 * 8;      // + the partial-sum rows of the further samples
    static constexpr int RED_BYTES = (NG - 1) * TM * TN * 16 * 256;   // the further groups' accumulators (fp32, one float4 per thread and MFMA tile)
    static constexpr int LDS_BYTES = (NG * TILES_BYTES > EPI_BYTES ? NG * TILES_BYTES : EPI_BYTES) > RED_BYTES ? (NG * TILES_BYTES > EPI_BYTES ? NG * TILES_BYTES : EPI_BYTES) : RED_BYTES;
    static constexpr int swz_key(int xx) { return RES <= 8 ? (xx & 2) : ((xx >> 1) & 2); }
    static_assert(RES * RES % BM_ == 0 || (BM_ % (RES * RES) == 0 && WM == 1 && (NIMG == 2 || NIMG == 4) && TM % NIMG == 0), "a tile lies inside one image, or holds two / four whole images");
    static constexpr int TAPS_PER_ROUND = 8 / TM;                       // a round = eight elements per lane, one per MFMA group
    static_assert(NROUND * TAPS_PER_ROUND <= 7, "the rounds run behind taps 2..8");
    static_assert(Epi::PACK_OK && LDS_BYTES <= (NG > 1 ? 163840 : 81920), "two blocks per CU (one with several K groups)");
};

// Weights [N][ld] bf16 in the engine's K order ((c / 64) * 9 + tap) * 64 + c % 64 (+ the c1 shortcut columns at 9 * cin) -> fragment-major
// [N / 16][NT][64 lanes][8] (N % 32 == 0): K step kt = (half-chunk hc, tap t) reads columns ((hc >> 1) * 9 + t) * 64 + (hc & 1) * 32 .. + 31, the shortcut
// steps follow; lane l of a block holds row (l & 15), columns 8 * (l >> 4) .. + 7 of the step: the A operand of v_mfma_f32_16x16x32_bf16.
__global__ __launch_bounds__(256) void k_pack_frag(const bf16* __restrict__ w, bf16* __restrict__ wf, int N, int ld, int cin, int c1)
{
    const int nk = 9 * (cin / 32), NT = nk + c1 / 32;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)(N / 16) * NT * 64) return;
    const int lane = (int)(idx & 63), kt = (int)((idx >> 6) % NT), nt = (int)((idx >> 6) / NT);
    int col;
    if (kt < nk) { const int hc = kt / 9, t = kt - 9 * hc; col = ((hc >> 1) * 9 + t) * 64 + (hc & 1) * 32; }
    else col = 9 * cin + (kt - nk) * 32;
    // rows of an n-tile pair interleaved (gemm_dma.h, PAIR): row r of tile 2 p + h is output channel 32 p + 8 (r >> 2) + 4 h + (r & 3)
    const int r = lane & 15, row = 32 * (nt >> 1) + 8 * (r >> 2) + 4 * (nt & 1) + (r & 3);
    const uint4 v = *reinterpret_cast<const uint4*>(w + (int64_t)row * ld + col + (lane >> 4) * 8);
    *reinterpret_cast<uint4*>(wf + idx * 8) = v;
}

// (functions, not asm statements inside the kernel's generic lambdas: clang rejects asm operands that name captured variables there)
__device__ __forceinline__ u32x4 gload16(unsigned voff, const void* sbase) {
    u32x4 v;
    asm volatile(NATINF_PAD_PRE "global_load_dwordx4 %0, %1, %2" NATINF_PAD_POST : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
    return v;
}
template <int OFF> __device__ __forceinline__ void lds_write16(unsigned addr, u32x4 v) {
    asm volatile(NATINF_PAD_PRE "ds_write_b128 %0, %1 offset:%2" NATINF_PAD_POST :: "v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void fresh4(u32x4& a, u32x4& b, u32x4& c, u32x4& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }

// EPI: the packed epilogues of gemm_dma.h (1 plain, 2 + GroupNorm partials, 5 + bf16 residual, 6 both).  g.b_frag = k_pack_frag's output.
template <int RES, bool WIDE, int EPI, int TMV = 8, int NGV = 1, int TNV = 4>
__global__ __launch_bounds__(256 * NGV, NGV > 1 ? 1 : 2) void k_conv_gn2(const GemmArgs g)      // (NGV = 4: 1,024 threads, <= 128 registers)
{
    using Cfg = ConvGn2Cfg<RES, WIDE, TMV, NGV, TNV>;
    using Geo = typename Cfg::Geo;
    constexpr int BM_ = Cfg::BM_, BN_ = Cfg::BN_, NW = Cfg::NW, TM = Cfg::TM, TN = Cfg::TN, KT = Cfg::KT;
    constexpr int W = Geo::W, WS = Geo::WS, HW = RES * RES;
    constexpr int PSW = Cfg::PSW, NROUND = Cfg::NROUND, NPIECE = Cfg::NPIECE, NFULL = Cfg::NFULL, NIMG = Cfg::NIMG, IMGP = Cfg::IMGP, NG = Cfg::NG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_poison();
    // LDS: [2][PATCH_BYTES] patch buffers, then [2][scale 32 | shift 32] fp32 tables (the epilogue reuses all of it as its slab)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = NG > 1 ? wave_all >> 2 : 0, wave = NG > 1 ? (wave_all & 3) : wave_all;      // K group, wave inside it
    const int tid_g = NG > 1 ? (tid & 255) : tid;
    const int wm = WIDE ? 0 : wave >> 1, wn = WIDE ? wave : wave & 1;
    const int nN = (g.N + BN_ - 1) / BN_, nM = (g.M + BM_ - 1) / BM_;      // (M is a whole number of tiles but for RES = 8 with an odd batch: the last tile holds one image)
    const int tile = xcd_remap(blockIdx.x, nM * nN);
    const int mt = tile / nN, nt = tile - mt * nN;
    const int m0 = mt * BM_, n0 = nt * BN_;
    const int b = m0 / HW, y0 = (m0 % HW) / W;                            // (first) image and first image row of this tile
    // several images per tile: how many of them exist (a batch that is no multiple of NIMG ends on a partial tile: the missing images re-read the last
    // real one, nothing of them is stored)
    const int nval = NIMG > 1 ? min(NIMG, (g.M - m0) / HW) : 1;
    const int ush = g.a0_up;                                              // 1: the source image has half the resolution (nearest up-sampling in the fetch)
    const bf16* const img = g.a0 + (int64_t)b * (HW >> (2 * ush)) * g.a0_ld;
    const float* const gsc = g.gn_scale + (int64_t)b * g.gn_ld;
    const float* const gsh = g.gn_shift + (int64_t)b * g.gn_ld;
    const int n_half = g.a0_C / KT, n_sc = g.a1 ? g.a1_C / KT : 0;      // (the loops below count a group's OWN half-chunks / shortcut tiles: k-th one = NG * k + grp)
    const int nk = 9 * n_half, NT = nk + n_sc;
    const int nh_g = n_half / NG, nsc_g = n_sc / NG;

    // Every LDS access, every LDS-DMA and every weight load of the K loop is inline asm with hand-counted waits (conv_gn.h explains
    // why: hipcc drains vmcnt in front of any LDS access it can see while an LDS-DMA is in flight).
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    auto glds16 = [](unsigned voff, const void* sbase, unsigned lds_dst) __attribute__((always_inline)) {
        NATINF_M0_ASM_BEGIN
        asm volatile(NATINF_PAD_PRE "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" NATINF_PAD_POST :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
        NATINF_M0_ASM_END
    };
    // (LDS byte addresses straight from the array: a cast of a generic pointer carries a null check against the shared aperture, which
    // hipcc has mis-selected into a vector compare on an SGPR-only operand in some variants of this kernel)
    const unsigned lds_patch = (unsigned)(uintptr_t)((lds_u8*)smem) + grp * Cfg::TILES_BYTES, lds_tab = lds_patch + 2 * Cfg::PATCH_BYTES;

    // ---- weight fragments: two register sets, set (kt & 1) holds K step kt ---------------------------------------------------
    u32x4 bw[2][TN];
    unsigned boff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) boff[j] = (unsigned)(((n0 >> 4) + wn * TN + j) * NT) * 1024u + (unsigned)lane * 16u;
    const unsigned char* const wfrag = reinterpret_cast<const unsigned char*>(g.b_frag);
    auto load_b = [&](auto set_tag, int kt) __attribute__((always_inline)) {
        constexpr int P = decltype(set_tag)::value;
        const unsigned char* base = wfrag + (int64_t)kt * 1024;
#pragma unroll
        for (int j = 0; j < TN; ++j) bw[P][j] = gload16(boff[j], base);
    };

    // ---- requests: the (scale | shift) table + this wave's patch pieces of half-chunk hc -> buffers hc & 1 ---------------------
    // The per-lane source offset of a patch request -- pixel (y, x) of patch row pp, clamped, times the row stride, plus the swizzled slot -- does
    // not depend on the half-chunk (that is the scalar base).  Recomputing it per half-chunk (a division by the patch row stride per piece) was
    // ~150 of the ~650 vector instructions a wave issues per half-chunk at 32x32 -- in a kernel whose bound is the vector issue port -- and holding
    // six offsets in registers is what the 256-register instantiations cannot afford.  So the prologue computes them once (FIRST) and parks them
    // in LDS behind the tables (one word per lane: conflict-free), and the K loop reads them back: six ds_read_b32 and one wait per nine taps.
    const unsigned lds_voff = lds_tab + 2 * Cfg::TAB_BYTES + (unsigned)tid_g * 4u;
    auto issue_patch = [&](auto first_tag, int k) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const int buf = k & 1, hc = NG * k + grp;
        int l;                                                               // the lane id, recomputed: kept alive across the K loop it is the value hipcc spills
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));                                          // addresses recomputed per request (once per nine taps): no registers held
        {
            // the table: scale by lanes 0-31, shift by lanes 32-63 -- two half-wave requests with a scalar base and a 32-bit lane offset
            // (one request with a per-lane 64-bit address kept a zero register alive across the loop, which hipcc then spilled); one table per image
            const unsigned toff = (unsigned)(l & 31) * 4u;
#pragma unroll
            for (int im = 0; im < NIMG; ++im) {
                const unsigned dst = lds_tab + buf * Cfg::TAB_BYTES + im * Cfg::TAB_IMG_BYTES;
                const int64_t io = (int64_t)min(im, nval - 1) * g.gn_ld;
                NATINF_M0_ASM_BEGIN
                if (l < 32) asm volatile(NATINF_PAD_PRE "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" NATINF_PAD_POST :: "v"(toff), "s"(gsc + io + hc * KT), "s"(dst) : "memory", "m0");
                else        asm volatile(NATINF_PAD_PRE "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" NATINF_PAD_POST :: "v"(toff), "s"(gsh + io + hc * KT), "s"(dst) : "memory", "m0");
                NATINF_M0_ASM_END
            }
        }
        const bf16* base = img + hc * KT;
        if constexpr (!FIRST) {
            unsigned vo[NROUND];
#pragma unroll
            for (int j = 0; j < NROUND; ++j) asm volatile(NATINF_PAD_PRE "ds_read_b32 %0, %1 offset:%2" NATINF_PAD_POST : "=v"(vo[j]) : "v"(lds_voff), "n"(j * 1024) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // (tap 0, behind the hand-off barrier: no other LDS read of this wave is in flight)
#pragma unroll
            for (int j = 0; j < NROUND; ++j) {
                const int q = j * NW + wave;
                if (j >= NFULL && q >= NPIECE) continue;
                asm volatile("" : "+v"(vo[j]));
                glds16(vo[j], base, lds_patch + buf * Cfg::PATCH_BYTES + q * 1024);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < NROUND; ++j) {
            const int q = j * NW + wave;                                      // wave-uniform piece index
            if (j >= NFULL && q >= NPIECE) continue;
            int lj = l;
            asm volatile("" : "+v"(lj));                                      // one piece's address arithmetic at a time: hoisted together, the pieces' temporaries spill inside the loop
            const int prow = lj >> 2, pslot = lj & 3;
            const int pp = q * 16 + prow;
            if constexpr (NIMG > 1) {
                // several whole images per tile: patch row pp = image im, pixel (yy - 1, xx - 1) of it; rows past the (RES + 2)^2 real ones are padding
                const int im = q / (IMGP / 16);                               // (a piece never straddles two images: IMGP % 16 == 0; wave-uniform)
                const int rem = pp - im * IMGP;
                const int yy = rem / WS, xx = rem - yy * WS;
                const int y = min(max(yy - 1, 0), RES - 1), x = min(max(xx - 1, 0), RES - 1);
                const unsigned vo = (unsigned)((min(im, nval - 1) * HW + y * W + x) * g.a0_ld + ((pslot ^ Cfg::swz_key(xx)) << 3)) * 2u;
                glds16(vo, base, lds_patch + buf * Cfg::PATCH_BYTES + q * 1024);
                asm volatile(NATINF_PAD_PRE "ds_write_b32 %0, %1 offset:%2" NATINF_PAD_POST :: "v"(lds_voff), "v"(vo), "n"(j * 1024) : "memory");
            } else {
                const int yy = pp / WS, xx = pp - yy * WS;
                const int y = min(max(y0 - 1 + yy, 0), RES - 1), x = min(max(xx - 1, 0), RES - 1);      // halo / pad: any readable pixel
                const unsigned vo = (unsigned)(((y >> ush) * (W >> ush) + (x >> ush)) * g.a0_ld + ((pslot ^ Cfg::swz_key(xx)) << 3)) * 2u;
                glds16(vo, base, lds_patch + buf * Cfg::PATCH_BYTES + q * 1024);
                asm volatile(NATINF_PAD_PRE "ds_write_b32 %0, %1 offset:%2" NATINF_PAD_POST :: "v"(lds_voff), "v"(vo), "n"(j * 1024) : "memory");
            }
        }
    };
    auto issue_shortcut = [&](int sk) __attribute__((always_inline)) {       // plain [BM][32] tile of a1 -> patch buffer sk & 1 (n_half is even)
        const int s = NG * sk + grp;
        const unsigned dst = lds_patch + (sk & 1) * Cfg::PATCH_BYTES + wave * (PSW * 1024);
        int l;                                                               // the lane id, recomputed: kept alive across the K loop it is the value hipcc spills
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        const int prow = l >> 2, pslot = l & 3;
        // a1_up: the shortcut operand lives at half the resolution too: tile row pp = pixel (y0 + pp / W, pp % W) <- source pixel (y >> 1, x >> 1)
        const int sup = g.a1_up;
        const bf16* base = g.a1 + (int64_t)(sup ? b * (HW >> 2) : m0) * g.a1_ld + s * KT;
#pragma unroll
        for (int j = 0; j < PSW; ++j) {
            const int pp = (wave * PSW + j) * 16 + prow;
            int src = sup ? ((y0 + pp / W) >> 1) * (W >> 1) + ((pp % W) >> 1) : pp;
            if constexpr (NIMG > 1) src = min(pp, nval * HW - 1);                  // a partial last tile: the missing images' rows re-read a real one
            glds16((unsigned)(src * g.a1_ld + ((pslot ^ ((pp >> 1) & 2)) << 3)) * 2u, base, dst + j * 1024);
        }
    };

#ifdef NATINF_CG_TIMELINE
    const unsigned long long dbg_t0 = cg_stamp();
#endif
    // Weight warm-up (GemmArgs::w_warm).  Inside a forward pass the weights of a layer are cold: every block streams the same matrix one tap
    // ahead, so the first round of blocks pays a memory round trip PER TAP (all of them missing on the same lines at the same time) -- at 4x4,
    // where one round of 128 blocks is the whole launch, that doubled the kernel's time against an L2-warm measurement.  The first <= 16 blocks of
    // an XCD (blockIdx & 7 under round-robin dispatch; a wrong guess only costs speed) touch every 128-byte line of the matrix once, in K order,
    // split over their waves: the lines are on their way into this XCD's L2 before the K loop asks for them.
    // (In FRONT of the first patch / weight requests: behind them, with their asm destination registers already "defined", the loop below raised
    // the pressure enough in the 32x32 instantiation for hipcc to spill weight registers whose loads were still in flight.)
    unsigned warm_junk = 0;                                               // ONE destination register for all of them ("+v": it stays allocated from the first request to the wait below)
    if (RES <= 8 && g.w_warm) {                                            // (32x32 / 16x16: measured +-0 -- later rounds of blocks hide the first one's misses -- and one register the 256-register instantiations do not have)
        const int jx = (int)blockIdx.x >> 3;
        const int nbx = min(((int)gridDim.x + 7 - ((int)blockIdx.x & 7)) >> 3, 16);
        if (jx < nbx) {
            const int NB16 = g.N >> 4, total = NB16 * NT;                    // 1-KiB fragment blocks; one request = 64 lines = 8 of them
            int l;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
            int i = jx * (NW * NG) + wave_all;
            for (int it = 0; it < 8 && i * 8 < total; ++it, i += nbx * (NW * NG)) {
                const int pz = min(i * 8 + (l >> 3), total - 1), kt = pz / NB16, nt = pz - kt * NB16;
                asm volatile("global_load_dword %0, %1, %2" : "+v"(warm_junk) : "v"((unsigned)((nt * NT + kt) * 1024 + (l & 7) * 128)), "s"(wfrag) : "memory");
            }
        }
    }
    // the first requests go out NOW: the index arithmetic below (masks, fragment bases, 128 accumulator registers) runs while they fly
    issue_patch(std::true_type{}, 0);
    load_b(std::integral_constant<int, 0>{}, grp * 9);                     // the group's first K step: tap 0 of half-chunk grp

    // ---- in-place normalisation, round j: the wave's piece j * NW + wave, lane l its bytes l * 16 .. + 15 = patch row q * 16 + (l >> 2),
    // ---- slot l & 3, which holds channel chunk (l & 3) ^ 2 * (bit 2 of that row's patch column: nmask bit 8 + j).
    // One register for both per-lane constants of the normalisation rounds: bits 0-15 = the lane's slot address in piece 0 of buffer 0
    // (lds_patch + wave * 1024 + lane * 16 < 64 KiB), bits 16-31 = nmask.  Held apart, the second register is the one hipcc spills in the
    // 16x16 / N = 128 instantiation -- and reloads inside the K loop behind a vmcnt(0) of its own.
    unsigned nmask = 0;                                                      // bit j: the pixel of round j lies inside the image; bit 8 + j: bit 2 of its patch column (the swizzle key)
#pragma unroll
    for (int j = 0; j < NROUND; ++j) {
        const int pp = (j * NW + wave) * 16 + (lane >> 2);
        if constexpr (NIMG > 1) {
            const int rem = pp % IMGP, yy = rem / WS, xx = rem - yy * WS;
            if ((unsigned)(yy - 1) < (unsigned)RES && (unsigned)(xx - 1) < (unsigned)RES) nmask |= 1u << j;
            nmask |= (unsigned)(Cfg::swz_key(xx) >> 1) << (8 + j);
        } else {
            const int yy = pp / WS, xx = pp - yy * WS;
            if ((unsigned)(y0 - 1 + yy) < (unsigned)RES && (unsigned)(xx - 1) < (unsigned)RES) nmask |= 1u << j;
            nmask |= (unsigned)(Cfg::swz_key(xx) >> 1) << (8 + j);
        }
    }
    const unsigned npack = (unsigned)(wave * 1024 + lane * 16) | (nmask << 16);      // (the address RELATIVE to the group's patch base: with four K groups the base passes 64 KiB)
    u32x4 nv = {0u, 0u, 0u, 0u}, ns0 = nv, ns1 = nv, nh0 = nv, nh1 = nv;
    unsigned npk[4] = {0u, 0u, 0u, 0u};
    float nf_even = 0.f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto norm_load = [&](auto j_tag, auto buf_tag) __attribute__((always_inline)) {
        constexpr int J = decltype(j_tag)::value, BUF = decltype(buf_tag)::value;
        unsigned np_ = npack;
        asm volatile("" : "+v"(np_));                                         // (opaque: or the unpacked halves are hoisted out of the loop again)
        unsigned nb = np_ & 0xffffu;
        const unsigned nm = np_ >> 16;
        nv = lds_read16<BUF * Cfg::PATCH_BYTES + J * NW * 1024>(nb + lds_patch);
        // the lane's row of the table: channel chunk (l & 3) ^ 2 * (bit 2 of the patch column).  Recomputed from the slot address (bits 4-9 = the lane) per
        // round: a register held across the K loop for it is the one hipcc spills (and its reload drains the weight stream)
        // (several images per tile: the piece's image, hence its table, is wave-uniform -- pieces do not straddle images)
        constexpr int PIECES_PER_IMG = NIMG > 1 ? IMGP / 16 : 1;                 // (one image per tile: IMGP = 0, and the arm below is dead -- no division by it)
        const unsigned timg = NIMG > 1 ? (unsigned)((J * NW + wave) / PIECES_PER_IMG) * Cfg::TAB_IMG_BYTES : 0u;
        const unsigned tbase = lds_tab + timg + ((((nb >> 4) & 3u) ^ ((nm >> (7 + J)) & 2u)) << 5);
        ns0 = lds_read16<BUF * Cfg::TAB_BYTES>(tbase); ns1 = lds_read16<BUF * Cfg::TAB_BYTES + 16>(tbase);
        nh0 = lds_read16<BUF * Cfg::TAB_BYTES + 128>(tbase); nh1 = lds_read16<BUF * Cfg::TAB_BYTES + 144>(tbase);
    };
#define NATINF_CG_NORM_PRE(I) asm volatile("" : "+v"(nv), "+v"(ns0), "+v"(ns1), "+v"(nh0), "+v"(nh1));
    // (Round 3: the in-loop elements in packed-fp32 form -- channel pairs, v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32, ten vector instructions
    // per pair instead of thirteen -- were built and measured: +-0 on every shape.  hipcc's SIPreEmitPeephole splits packed-fp32 instructions
    // that sit in the shadow of an MFMA back into scalar ones (packed fp32 cannot be co-issued behind an MFMA on gfx950, the scalar forms can),
    // and the two extra registers a pair holds across a step tipped the EPI 1 instantiation into scratch reloads inside the loop.  The packed
    // form stays where no MFMA is in flight: the prologue's norm_round below, NATINF_CG_PK.)
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
        _Pragma("unroll") for (int g_ = 0; g_ < TN; ++g_) {                                                                  \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       /* one MFMA */                                           \
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);       /* two vector instructions */                            \
        }
#define NATINF_CG_NORM_POST(I) if constexpr (((I) & 1) == 0) asm volatile("" : "+v"(nf_even)); else asm volatile("" : "+v"(npk[(I) >> 1]));
    auto norm_store = [&](auto j_tag, auto buf_tag) __attribute__((always_inline)) {
        constexpr int J = decltype(j_tag)::value, BUF = decltype(buf_tag)::value;
        u32x4 ou = {npk[0], npk[1], npk[2], npk[3]};
        unsigned np_ = npack;
        asm volatile("" : "+v"(np_));
        if (!((np_ >> (16 + J)) & 1u)) ou = u32x4{0u, 0u, 0u, 0u};
        lds_write16<BUF * Cfg::PATCH_BYTES + J * NW * 1024>((np_ & 0xffffu) + lds_patch, ou);
    };
    // a whole round at once (prologue: no MFMAs to hide behind yet, so the eight elements are left to hipcc to interleave -- one
    // dependent unpack-fma-exp2-add-rcp-mul chain after the other costs ~70 cycles per element, 4k cycles per tile)
    auto norm_round = [&](auto j_tag, auto buf_tag) __attribute__((always_inline)) {
        norm_load(j_tag, buf_tag);
        wait_lgkmcnt<0>();
        NATINF_CG_NORM_PRE(0)
#if NATINF_CG_PK
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const unsigned w_ = nv[p];
            const f32x2 x2_ = {__uint_as_float(w_ << 16), __uint_as_float(w_ & 0xffff0000u)};
            const f32x2 s2_ = {__uint_as_float(p < 2 ? ns0[2 * (p & 1)] : ns1[2 * (p & 1)]), __uint_as_float(p < 2 ? ns0[2 * (p & 1) + 1] : ns1[2 * (p & 1) + 1])};
            const f32x2 h2_ = {__uint_as_float(p < 2 ? nh0[2 * (p & 1)] : nh1[2 * (p & 1)]), __uint_as_float(p < 2 ? nh0[2 * (p & 1) + 1] : nh1[2 * (p & 1) + 1])};
            const f32x2 t2_ = __builtin_elementwise_fma(x2_, s2_, h2_);
            const f32x2 d2_ = f32x2{__builtin_amdgcn_exp2f(t2_[0]), __builtin_amdgcn_exp2f(t2_[1])} + f32x2{1.0f, 1.0f};
            const f32x2 y2_ = t2_ * f32x2{__builtin_amdgcn_rcpf(d2_[0]), __builtin_amdgcn_rcpf(d2_[1])};
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            const bf16x2_t pr_ = {(bf16)y2_[0], (bf16)y2_[1]};
            npk[p] = __builtin_bit_cast(unsigned, pr_);
        }
#else
        float y_[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned w_ = nv[i >> 1];
            const float x_ = __uint_as_float((i & 1) ? (w_ & 0xffff0000u) : (w_ << 16));
            const float t_ = x_ * __uint_as_float(i < 4 ? ns0[i & 3] : ns1[i & 3]) + __uint_as_float(i < 4 ? nh0[i & 3] : nh1[i & 3]);
            y_[i] = t_ * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t_));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            const bf16x2_t pr_ = {(bf16)y_[2 * i], (bf16)y_[2 * i + 1]};
            npk[i] = __builtin_bit_cast(unsigned, pr_);
        }
#endif
        norm_store(j_tag, buf_tag);
    };
#define NATINF_CG_NO_PRE(I)
#define NATINF_CG_NO_POST(I)
#define NATINF_CG_NO_EL(I)

    // ---- A fragment addresses: three per-lane bases (dx = -1, 0, +1) at dy = -1; everything else is an immediate -------------
    const int frow = lane & 15, fq = lane >> 4;
    unsigned a_dx[3];
    {
        const int ml = wm * (TM * 16) + frow;                             // first pixel row-tile of this wave
        // its patch row at the centre tap, and its patch column (row-tile / dy / image offsets keep the column, hence the swizzle)
        const int pc = RES <= 8 ? ((frow / W) + 1) * WS + (frow % W) + 1 : ((ml / W) + 1) * WS + (ml % W) + 1;      // (RES = 8 / 4: a 16-pixel row-tile is two / four image rows)
        const int xc = RES <= 8 ? (frow % W) : (ml % W);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int pp = pc - WS + d - 1;
            a_dx[d] = lds_patch + pp * 64 + ((fq ^ Cfg::swz_key(xc + d)) << 4);
        }
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // One 32-wide K step: A row-tiles 0, 1 are requested up front, row-tile s+2 while s is multiplied with the four resident weight
    // fragments of register set P; `s_waitcnt lgkmcnt(n)` retires exactly the fragment the next four MFMAs need (LDS returns in
    // order; the five reads of a normalisation round are older than all of them).  EL(i): the vector work placed behind MFMA group i.
    // EO: element offset of the normalisation work (TM = 4: a round spans two taps, the second one handles elements 4..7)
#define NATINF_CG_STEP(a, AOFF, P, S, EL, EO)                                                                               \
        if constexpr ((S) + 2 < TM) fs[((S) + 2) % 3] = lds_read16<AOFF((S) + 2)>(a);                                        \
        wait_lgkmcnt<((S) + 2 < TM ? 2 : TM - 1 - (S))>();                                                                    \
        EL##_PRE((EO) + (S))                                                                                                 \
        _Pragma("unroll") for (int r_ = 0; r_ < TN; ++r_)                                                                    \
            acc[S][r_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bw[P][r_]), __builtin_bit_cast(bf16x8, fs[(S) % 3]), acc[S][r_], 0, 0, 0); \
        EL##_EL((EO) + (S))                                                                                                  \
        EL##_POST((EO) + (S))                                                                                                \
        __builtin_amdgcn_sched_barrier(0);
#define NATINF_CG_HEAD(a, AOFF) u32x4 fs[3]; fs[0] = lds_read16<AOFF(0)>(a); fs[1] = lds_read16<AOFF(1)>(a);
#define NATINF_CG_BODY(a, AOFF, P, EL, EO)                                                                                  \
        NATINF_CG_STEP(a, AOFF, P, 0, EL, EO) NATINF_CG_STEP(a, AOFF, P, 1, EL, EO) NATINF_CG_STEP(a, AOFF, P, 2, EL, EO) NATINF_CG_STEP(a, AOFF, P, 3, EL, EO) \
        if constexpr (TM == 8) {                                                                                              \
        NATINF_CG_STEP(a, AOFF, P, 4, EL, EO) NATINF_CG_STEP(a, AOFF, P, 5, EL, EO) NATINF_CG_STEP(a, AOFF, P, 6, EL, EO) NATINF_CG_STEP(a, AOFF, P, 7, EL, EO) }
    // the weight fragments of set P have landed (the wait in front of this): from here on they are new values to hipcc
#define NATINF_CG_BW_READY(P) _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) asm volatile("" : "+v"(bw[P][j_]));

    // ---- prologue: this wave's table + patch pieces of half-chunk 0 and weight step 0; its pieces are normalised before the loop ----
    using std::integral_constant;
#ifdef NATINF_CG_TIMELINE
    unsigned long long dbg_wait = 0, dbg_head = 0, dbg_mfma = 0;
#endif
    NATINF_CG_STAMP(dbg_p0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" :: "v"(warm_junk));                                   // (the warm-up requests' destination register is free again from here)
    NATINF_CG_STAMP(dbg_p1)
    {
        auto b0 = integral_constant<int, 0>{};
        norm_round(integral_constant<int, 0>{}, b0);
        if constexpr (NROUND > 1) if (1 < NFULL || 1 * NW + wave < NPIECE) norm_round(integral_constant<int, 1>{}, b0);
        if constexpr (NROUND > 2) if (2 < NFULL || 2 * NW + wave < NPIECE) norm_round(integral_constant<int, 2>{}, b0);
        if constexpr (NROUND > 3) if (3 < NFULL || 3 * NW + wave < NPIECE) norm_round(integral_constant<int, 3>{}, b0);
        if constexpr (NROUND > 4) if (4 < NFULL || 4 * NW + wave < NPIECE) norm_round(integral_constant<int, 4>{}, b0);
        if constexpr (NROUND > 5) if (5 < NFULL || 5 * NW + wave < NPIECE) norm_round(integral_constant<int, 5>{}, b0);
        if constexpr (NROUND > 6) if (6 < NFULL || 6 * NW + wave < NPIECE) norm_round(integral_constant<int, 6>{}, b0);
    }

    // ---- the nine taps of one half-chunk; BUF (its patch buffer) and the tap index are compile-time: all offsets are immediates.
    // Weight set of K step kt = hc * 9 + T: (kt & 1) = (T + BUF) & 1 (hc and BUF have the same parity).
#define NATINF_CG_AOFF(i) (BUF * Cfg::PATCH_BYTES + ((RES == 32 ? ((i) >> 1) * WS + ((i) & 1) * 16 : (RES == 16 ? (i) * WS : (RES == 8 ? ((i) >> 2) * IMGP + ((i) & 3) * 2 * WS : (i) * IMGP))) + (T / 3) * WS) * 64)
    auto tap = [&](auto buf_tag, auto t_tag, int hc, bool next_half) __attribute__((always_inline)) {      // hc: the group's own count
        constexpr int BUF = decltype(buf_tag)::value, T = decltype(t_tag)::value, P = (T + BUF) & 1;
        const int kt = (NG * hc + grp) * 9 + T;
        NATINF_CG_STAMP(ts0)
        // Weight step kt has landed.  Younger requests: the aux request of tap 0 (at T = 1: it stays in flight; every wave issued at
        // least 2 + NFULL / PSW of them).  Tap 0 is the hand-off: every wave's normalised pieces of this half-chunk are written
        // (lgkmcnt) and nobody reads the other buffer any more -> the next half-chunk's raw patch may land in it.
        if constexpr (T == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if constexpr (T == 1) {
            if (next_half) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NIMG + NFULL) : "memory");
            else if (n_sc > 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PSW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        NATINF_CG_BW_READY(P)
        NATINF_CG_STAMP(ts1)
        // the group's next K step: the next tap; after tap 8 its next half-chunk, or its first shortcut tile
        // (ONE load_b call site: the same asm statements reached on two paths would make hipcc merge -- copy -- registers whose loads are in flight)
        int ktn = kt + 1;
        bool has_next = NG > 1 ? true : kt + 1 < NT;
        if constexpr (T == 8 && NG > 1) {
            if (next_half) ktn = kt + 1 + 9 * (NG - 1);
            else { ktn = nk + grp; has_next = n_sc > 0; }
        }
        if (has_next) load_b(integral_constant<int, P ^ 1>{}, ktn);
        if constexpr (T == 0) {
            if (next_half) issue_patch(std::false_type{}, hc + 1);
            else if (n_sc > 0) issue_shortcut(0);
        }
        constexpr int TPR = Cfg::TAPS_PER_ROUND;                         // taps a normalisation round spans (TM = 4: two)
        constexpr bool NORM_TAP = T >= 2 && (T - 2) / TPR < NROUND && NATINF_CG_ABL != 1;
        constexpr int J = NORM_TAP ? (T - 2) / TPR : 0, EO = NORM_TAP ? ((T - 2) % TPR) * TM : 0;
        const bool live = NORM_TAP && next_half && (J < NFULL || J * NW + wave < NPIECE);
        if constexpr (NORM_TAP && EO == 0) { if (live) norm_load(integral_constant<int, J>{}, integral_constant<int, BUF ^ 1>{}); }
        NATINF_CG_HEAD(a_dx[T % 3], NATINF_CG_AOFF)
        NATINF_CG_STAMP(ts2)
        if constexpr (NORM_TAP) {
            NATINF_CG_BODY(a_dx[T % 3], NATINF_CG_AOFF, P, NATINF_CG_NORM, EO)
            if constexpr (EO + TM == 8) { if (live) norm_store(integral_constant<int, J>{}, integral_constant<int, BUF ^ 1>{}); }
        } else {
            NATINF_CG_BODY(a_dx[T % 3], NATINF_CG_AOFF, P, NATINF_CG_NO, 0)
        }
        NATINF_CG_STAMP(ts3)
        NATINF_CG_ADD(dbg_wait, ts0, ts1) NATINF_CG_ADD(dbg_head, ts1, ts2) NATINF_CG_ADD(dbg_mfma, ts2, ts3)
    };
    auto half_chunk = [&](auto buf_tag, int hc) __attribute__((always_inline)) {
        const bool next_half = hc + 1 < nh_g;
        tap(buf_tag, integral_constant<int, 0>{}, hc, next_half); tap(buf_tag, integral_constant<int, 1>{}, hc, next_half);
        tap(buf_tag, integral_constant<int, 2>{}, hc, next_half); tap(buf_tag, integral_constant<int, 3>{}, hc, next_half);
        tap(buf_tag, integral_constant<int, 4>{}, hc, next_half); tap(buf_tag, integral_constant<int, 5>{}, hc, next_half);
        tap(buf_tag, integral_constant<int, 6>{}, hc, next_half); tap(buf_tag, integral_constant<int, 7>{}, hc, next_half);
        tap(buf_tag, integral_constant<int, 8>{}, hc, next_half);
    };
#ifdef NATINF_CG_TIMELINE
    const unsigned long long dbg_t1 = cg_stamp();
#endif
    for (int hc = 0; hc < nh_g; hc += 2) {                                // a0_C is a multiple of 64 (128 with two K groups): a group's half-chunks come in pairs
        half_chunk(integral_constant<int, 0>{}, hc);
        half_chunk(integral_constant<int, 1>{}, hc + 1);
    }
#undef NATINF_CG_AOFF
    // ---- 1x1 shortcut segment: plain [BM][32] A tiles through the two patch buffers (tile s+1 requested at the head of tile s; every
    // ---- wave fetches a quarter of a tile, so each tile is a hand-off), weights through the register sets as before; a1_C % 64 == 0
#define NATINF_CG_POFF(i) (BUF * Cfg::PATCH_BYTES + (i) * 1024)
    auto sc_tile = [&](auto buf_tag, int s) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_tag)::value, P = BUF;            // a group has run an even number of K steps: its s-th shortcut step lives in set s & 1
        const int kt = nk + NG * s + grp;
        // the lane's fragment base in a plain [BM][32] tile, recomputed from an opaque lane id per tile: held across the main loop it is one
        // register too many (the 16x16 / N = 128 instantiation then reloads a spilled value inside the loop, behind hipcc's own vmcnt(0))
        int l2;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l2));
        const int arow = wm * (TM * 16) + (l2 & 15);
        const unsigned a_plain = lds_patch + arow * 64 + (((l2 >> 4) ^ ((arow >> 1) & 2)) << 4);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        NATINF_CG_BW_READY(P)
        if (s + 1 < nsc_g) { load_b(integral_constant<int, P ^ 1>{}, kt + NG); issue_shortcut(s + 1); }
        NATINF_CG_HEAD(a_plain, NATINF_CG_POFF)
        NATINF_CG_BODY(a_plain, NATINF_CG_POFF, P, NATINF_CG_NO, 0)
    };
    for (int s = 0; s < nsc_g; s += 2) {
        sc_tile(integral_constant<int, 0>{}, s);
        sc_tile(integral_constant<int, 1>{}, s + 1);
    }
#undef NATINF_CG_POFF
#undef NATINF_CG_STEP
#undef NATINF_CG_HEAD
#undef NATINF_CG_BODY
#undef NATINF_CG_BW_READY
#undef NATINF_CG_NO_EL
#undef NATINF_CG_NO_PRE
#undef NATINF_CG_NO_POST
#undef NATINF_CG_NORM_PRE
#undef NATINF_CG_NORM_EL
#undef NATINF_CG_NORM_POST
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave is done with the tiles before the epilogue reuses them
    if constexpr (NG > 1) {
        // the second K group hands its partial sums over (fp32, [MFMA tile][thread] float4: lane-linear 16-byte accesses) and is done;
        // S_BARRIER waits on the surviving waves only, so the first group's epilogue barriers work without it
        f32x4* red = reinterpret_cast<f32x4*>(smem) + (tid & 255);
        if (grp > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) red[((grp - 1) * TM * TN + i * TN + j) * 256] = acc[i][j];
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll 1
        for (int gg = 0; gg < NG - 1; ++gg)              // (fixed order: groups 1, 2, ... -- deterministic; one group's values in flight at a time: 128 registers)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] += red[(gg * TM * TN + i * TN + j) * 256];
        __syncthreads();
    }
    // the epilogue's arguments are fetched from the kernel-argument segment HERE (conv_gn.h: kept in scalar registers across the K
    // loop they end up spilled into vector-register lanes)
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
    // (thread / lane id recomputed: the prologue's copies, kept alive across the K loops for the epilogue alone, are among the values hipcc spills in the
    // 256-register instantiations -- and a kernel with ANY scratch use starts ~2 us later in the stream than one without)
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int tid_e = wave_all * 64 + lane_e;
    tile_epilogue<Cfg::WM, Cfg::WN, TM, TN, typename Cfg::Epi, EPI, NIMG, (RES <= 8), true>(ge, smem, acc, m0, n0, 0, tid_e, lane_e, wm, wn);      // (two K groups: tid < 256 here)
#ifdef NATINF_CG_TIMELINE
    if (ge.dbg_ts && tid == 0 && (blockIdx.x == 0 || blockIdx.x == 777)) {          // development builds: tools/conv_gn_timeline.py
        // block 0: [2..4] are the epilogue's own stamps (NATINF_TS: start, slab written, copied out), [7] = the stamp in front of it
        unsigned long long* o = ge.dbg_ts + (blockIdx.x ? 16 : 8);
        o[0] = dbg_t1 - dbg_t0; o[1] = dbg_wait; o[2] = dbg_head; o[3] = dbg_mfma; o[4] = dbg_t2 - dbg_t1; o[5] = cg_stamp() - dbg_t2; o[6] = (unsigned long long)nk;
        if (blockIdx.x == 0) { ge.dbg_ts[7] = dbg_t2; ge.dbg_ts[5] = dbg_p0 - dbg_t0; ge.dbg_ts[6] = dbg_p1 - dbg_p0; }
    }
#endif
}

}  // namespace ncsn

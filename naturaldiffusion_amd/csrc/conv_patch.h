// conv_patch.h -- 3x3 implicit GEMM with an LDS-resident input patch.
//
// Measured on the two-stage / ring kernels (profiles/r01, tools/bench_gemm.py): every tile variant saturates
// at ~9-11 TB/s of aggregate L2->LDS DMA traffic, whatever its pipeline depth.  Those kernels re-fetch the A tile
// for each of the nine taps, although the nine shifted windows of a pixel tile overlap almost completely.  Here
// the block loads, once per 64-channel chunk, the PATCH = the tile's pixels plus a one-pixel halo -- a contiguous
// run of the zero-bordered [B][H+2][W+2][C] tensor -- and all nine taps read their A fragments from it with a
// per-lane row offset.  A-side DMA traffic drops 9x (to ~1.3x the tile), leaving the weight tile as the only
// per-tap stream:  bytes per K-tile  48 KB -> 21 KB (N = 128),  64 KB -> 37 KB (N = 256).
//
//   block tile BM = 256 pixels (a band of rows of one image, or whole images when H*W < 256) x BN channels
//   LDS: 2 patch buffers (<= 400 rows x 128 B) + 2 weight tiles (BN x 128 B); 8 waves
//   per chunk c:  patch(c+1) is requested piecewise during the taps of chunk c
//   per tap t:    weight tile (c, t+1) requested, MFMAs on (patch c, weights (c, t)), vmcnt(0) + barrier
//   the 1x1 shortcut segment (a1) runs after the chunks as plain 64-wide K-tiles through the same buffers.
// LDS rows are 128 B with the 16-byte chunk XOR-ed by (row>>1)&7 on the DMA source side and on the fragment read.
// Re-measured at the end of round 1 with the packed epilogue switched in (tools/scan_patch.py): 2-6 % BEHIND the tap-by-tap
// hand-pipelined kernels on every 3x3 shape (e.g. 133 vs 126 us at (131072, 256, 2304)) -- fewer DMA pieces do not shorten
// the K loop, so this stays a non-selected variant with the general epilogue.
#pragma once
#include "gemm_dma.h"

namespace ncsn {

// PMAX = patch rows the LDS buffers hold: res 32: 10*34 = 340, res 16: 18*18 = 324 (-> 344), res 8: 4*100 = 400
template <int WM, int WN, int TM, int TN, int PMAX>
struct PatchCfg {
    static constexpr int PATCH_MAX_ROWS = PMAX;
    static constexpr int NW = WM * WN, THREADS = NW * 64;
    static constexpr int BM_ = WM * TM * 16, BN_ = WN * TN * 16;
    static constexpr int PATCH_BYTES = PATCH_MAX_ROWS * 128;
    static constexpr int BT_BYTES = BN_ * 128;
    static constexpr int TILES_BYTES = 2 * PATCH_BYTES + 2 * BT_BYTES;
    static constexpr int PB = BN_ / 8 / NW;                                   // weight-tile DMA pieces per wave
    static constexpr int PP = (PATCH_MAX_ROWS / 8 + NW - 1) / NW;             // patch DMA pieces per wave (max)
    static constexpr int PS = BM_ / 8 / NW;                                   // shortcut-tile pieces per wave
    using Epi = EpiCfg<WM, WN, TM, TN, TILES_BYTES>;
    static constexpr int LDS_BYTES = TILES_BYTES;
    static_assert(BM_ == 256 && PMAX % 8 == 0, "the tile is 256 pixels");
    static_assert(PP <= 9, "at most one patch piece per tap and wave");
    static_assert(LDS_BYTES <= 163840, "LDS budget");
};

template <int WM, int WN, int TM, int TN, int PMAX>
__global__ __launch_bounds__(WM * WN * 64, 2) void k_conv_patch(const GemmArgs g)
{
    using Cfg = PatchCfg<WM, WN, TM, TN, PMAX>;
    constexpr int BM_ = Cfg::BM_, BN_ = Cfg::BN_, NW = Cfg::NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_poison();
    unsigned char* sPatch = smem;                                  // [2][PATCH_BYTES]
    unsigned char* sB = smem + 2 * Cfg::PATCH_BYTES;               // [2][BT_BYTES]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nN = (g.N + BN_ - 1) / BN_, nM = (g.M + BM_ - 1) / BM_;
    const int tile = xcd_remap(blockIdx.x, nM * nN);
    const int m0 = (tile / nN) * BM_, n0 = (tile % nN) * BN_;

    const int W = 1 << g.logW, HW = 1 << g.logHW, H = HW >> g.logW, Wp = W + 2, Hp = H + 2;
    const int n_img = (g.M + HW - 1) >> g.logHW;
    const int64_t total_pix = (int64_t)n_img * Hp * Wp;
    // patch geometry (see header): a contiguous run of padded pixels starting at `origin`
    const int b0 = m0 >> g.logHW, y0 = HW >= BM_ ? (m0 & (HW - 1)) >> g.logW : 0;
    const int P = HW >= BM_ ? (BM_ / W + 2) * Wp : (BM_ >> g.logHW) * Hp * Wp;
    const int64_t origin = (int64_t)b0 * Hp * Wp + (int64_t)y0 * Wp;
    const int n_chunk = g.a0_C / BK, n_sc = g.a1 ? g.a1_C / BK : 0;

    // ---- per-lane patch row of each of the wave's TM row-tiles (centre tap), and the plain tile row for the shortcut
    const int frow = lane & 15, fq = lane >> 4;
    int pc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = min(m0 + wm * TM * 16 + i * 16 + frow, g.M - 1);
        const int b = m >> g.logHW, p = m & (HW - 1), y = p >> g.logW, x = p & (W - 1);
        pc[i] = (b - b0) * Hp * Wp + (y - y0 + 1) * Wp + x + 1;
    }

    // ---- DMA requests.  Source addresses are recomputed per request from a few lane constants (a dozen VALU ops
    // ---- per 1-KiB piece) instead of living in registers: the accumulators need the space.
    const int prow = lane >> 3, pchunk = lane & 7;
    const int n_pp = (P + 7) / 8;                                   // patch pieces in total
    typedef __attribute__((address_space(3))) void lds_void;
    auto issue_patch_piece = [&](int j, int chunk, int buf) __attribute__((always_inline)) {
        const int q = wave + NW * j;                                // piece = patch rows 8q .. 8q+7
        if (q < n_pp) {
            const int r = 8 * q + prow;
            const int64_t pix = min(origin + r, total_pix - 1);
            const bf16* src = g.a0 + pix * g.a0_ld + chunk * BK + ((pchunk ^ ((r >> 1) & 7)) << 3);
            __builtin_amdgcn_global_load_lds(src, (lds_void*)(sPatch + buf * Cfg::PATCH_BYTES + q * 1024), 16, 0, 0);
        }
    };
    auto issue_b = [&](int kt, int buf) __attribute__((always_inline)) {        // kt = K-tile index in packed order
#pragma unroll
        for (int j = 0; j < Cfg::PB; ++j) {
            const int r = (wave * Cfg::PB + j) * 8 + prow;
            const bf16* src = g.b + (int64_t)min(n0 + r, g.N - 1) * g.b_ld + kt * BK + ((pchunk ^ ((r >> 1) & 7)) << 3);
            __builtin_amdgcn_global_load_lds(src, (lds_void*)(sB + buf * Cfg::BT_BYTES + (wave * Cfg::PB + j) * 1024), 16, 0, 0);
        }
    };
    auto issue_shortcut = [&](int sc, int buf) __attribute__((always_inline)) {  // plain [BM][64] tile into a patch buffer
#pragma unroll
        for (int j = 0; j < Cfg::PS; ++j) {
            const int r = (wave * Cfg::PS + j) * 8 + prow;
            const bf16* src = g.a1 + (int64_t)min(m0 + r, g.M - 1) * g.a1_ld + sc * BK + ((pchunk ^ ((r >> 1) & 7)) << 3);
            __builtin_amdgcn_global_load_lds(src, (lds_void*)(sPatch + buf * Cfg::PATCH_BYTES + (wave * Cfg::PS + j) * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one K-tile: A rows `arow[i]` of patch buffer `pbuf`, weights from tile buffer `bbuf`
    auto compute = [&](const int (&arow)[TM], int pbuf, int bbuf) __attribute__((always_inline)) {
        const bf16* pa = reinterpret_cast<const bf16*>(sPatch + pbuf * Cfg::PATCH_BYTES);
        const bf16* tb = reinterpret_cast<const bf16*>(sB + bbuf * Cfg::BT_BYTES) + (wn * TN * 16 + frow) * LDS_ROW;
        const int bswz = (frow >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fb[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *reinterpret_cast<const bf16x8*>(tb + j * 16 * LDS_ROW + ((((ks << 2) | fq) ^ bswz) << 3));
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = arow[i];
                const bf16x8 fa = *reinterpret_cast<const bf16x8*>(pa + r * LDS_ROW + ((((ks << 2) | fq) ^ ((r >> 1) & 7)) << 3));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa, acc[i][j], 0, 0, 0);
            }
        }
    };

    // ---- prologue: patch 0 and weight tile 0
#pragma unroll
    for (int j = 0; j < Cfg::PP; ++j) issue_patch_piece(j, 0, 0);
    issue_b(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int kt = 0;                                                     // K-tile counter in packed order
    for (int c = 0; c < n_chunk; ++c) {
        const int pbuf = c & 1;
        const bool next_is_chunk = c + 1 < n_chunk, next_is_sc = !next_is_chunk && n_sc > 0;
#pragma unroll 1
        for (int ty = 0; ty < 3; ++ty) {
#pragma unroll 1
            for (int tx = 0; tx < 3; ++tx, ++kt) {
                const int t = 3 * ty + tx;
                const bool last_tile = !next_is_chunk && !next_is_sc && t == 8;
                if (!last_tile) issue_b(kt + 1, (kt + 1) & 1);
                if (next_is_chunk) { if (t < Cfg::PP) issue_patch_piece(t, c + 1, pbuf ^ 1); }
                else if (next_is_sc && t == 0) issue_shortcut(0, pbuf ^ 1);
                const int off = (ty - 1) * Wp + (tx - 1);
                int arow[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) arow[i] = pc[i] + off;
                compute(arow, pbuf, kt & 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // LDS-DMA completions are not tracked across the back-edge
                __syncthreads();
            }
        }
    }
    for (int s = 0; s < n_sc; ++s, ++kt) {                          // 1x1 shortcut segment: plain tiles
        const int pbuf = (n_chunk + s) & 1;
        if (s + 1 < n_sc) { issue_b(kt + 1, (kt + 1) & 1); issue_shortcut(s + 1, pbuf ^ 1); }
        int arow[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) arow[i] = wm * TM * 16 + i * 16 + frow;
        compute(arow, pbuf, kt & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    dma_tile_epilogue<WM, WN, TM, TN, typename Cfg::Epi>(g, smem, acc, m0, n0, 0, tid, lane, wm, wn);
}

}  // namespace ncsn

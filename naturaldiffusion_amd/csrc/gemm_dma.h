// gemm_dma.h -- LDS-DMA implicit-GEMM kernels of the NCSN++ engine (included by ncsnpp_kernels.h users).
//
// Operand tiles go HBM/L2 -> LDS with `global_load_lds_dwordx4` (no VGPR staging, no ds_write pass -- the
// ds_write_b128 path moves only ~79 B/clk/CU and was the busiest pipe of the register-staged kernel).  One
// wave-instruction lands 64 lanes x 16 B = 1 KiB of consecutive LDS; the XOR swizzle therefore sits on the SOURCE
// address and on the fragment read, never on the LDS destination.  Loads are unconditional, which the callers
// make legal:
//   * taps == 9 operands are stored with a one-pixel zero border ([B][H+2][W+2][C], written by k_gn_apply),
//     so a shifted tap never leaves the tensor and no zero-fill is needed;
//   * rows beyond M (or N) are clamped to the last valid row -- computed, never stored;
//   * K0 and K1 are multiples of 64 (checked on the host; other shapes use k_gemm_bf16).
#pragma once
#include "ncsnpp_kernels.h"

#ifndef NATINF_SETPRIO
#define NATINF_SETPRIO 0
#endif

namespace ncsn {

// ------------------------------------------------------------------------------------------------
// k_gemm_dma<WM, WN, TM, TN>: LDS-DMA GEMM with a configurable block tile.
//
//   block tile BM x BN = (WM*TM*16) x (WN*TN*16), WM*WN waves, each wave TM x TN tiles of 16x16 (x32 in K),
//   BK = 64, two LDS stages of (BM + BN) x 128 B.
//
// Why bigger tiles: every K-tile moves (BM+BN)*128 B from L2 into LDS for 2*BM*BN*64 flops.  At 128x128 that is
// 64 flop/B, i.e. ~39 TB/s of L2->LDS traffic at the 2.5 PFLOP/s MFMA peak -- more than the ~34 TB/s the L2s
// deliver; 256x256 halves it (128 flop/B) and, with 2048 MFMA-cycles of work per K-tile per CU, the single
// prefetched tile covers the DMA latency even at one block per CU.  Same preconditions as k_gemm_bf16_dma
// (zero-bordered 3x3 operands, K multiples of 64, clamped M/N edges).
//   <2,4,8,4> 256x256, 512 threads, 128 KB LDS   (N = 256 layers with >= 1 tile per CU)
//   <4,2,4,4> 256x128, 512 threads,  96 KB LDS   (N = 128 layers)
//   <2,2,4,4> 128x128, 256 threads,  66 KB LDS   (small grids: 8x8 / 4x4 levels, attention, embeddings)
// ------------------------------------------------------------------------------------------------
// Epilogue staging geometry for a (WM x WN waves) x (TM x TN 16x16 tiles) block with AVAIL bytes of LDS.
template <int WM, int WN, int TM, int TN, int AVAIL>
struct EpiCfg {
    static constexpr int CROW = WN * TN * 16 + 4;                          // fp32 staging row stride
    static constexpr int bytes(int rb) { return WM * rb * 16 * CROW * 4; }
    // row-tiles (of 16) staged per pass: the most that fits
    static constexpr int RB = bytes(TM) <= AVAIL ? TM : (bytes(TM / 2) <= AVAIL ? TM / 2 : (bytes(TM / 4) <= AVAIL ? TM / 4 : 1));
    static constexpr int SLAB_ROWS = WM * RB * 16;
    static constexpr int SLAB_BYTES = bytes(RB);
    static_assert(TM % RB == 0 && SLAB_BYTES <= AVAIL, "epilogue staging does not fit");
    // packed (bf16) whole-tile slab of the column-terms-only epilogue: rows of BN bf16 + 16 B (a 4-bank shift per row)
    static constexpr int PROW = WN * TN * 16 * 2 + 16;
    static constexpr int PACK_SLAB = WM * TM * 16 * PROW;
    static constexpr int PACK_BYTES = PACK_SLAB + WM * WN * TN * 4 * 8;    // + the GroupNorm partials of the WM wave rows (float2 per 4-column quad), behind the slab
    static constexpr bool PACK_OK = PACK_BYTES <= AVAIL;
    static constexpr int NEED = PACK_OK && PACK_BYTES > SLAB_BYTES ? PACK_BYTES : SLAB_BYTES;
};

template <int WM, int WN, int TM, int TN>
struct DmaCfg {
    static constexpr int NW = WM * WN, THREADS = NW * 64;
    static constexpr int BM_ = WM * TM * 16, BN_ = WN * TN * 16;
    static constexpr int STAGE_BYTES = (BM_ + BN_) * BK * 2;
    static constexpr int PA = BM_ / 8 / NW, PB = BN_ / 8 / NW;            // 1-KiB DMA pieces per wave per K-tile
    using Epi = EpiCfg<WM, WN, TM, TN, 2 * STAGE_BYTES + 8192>;
    static constexpr int LDS_BYTES = Epi::NEED > 2 * STAGE_BYTES ? Epi::NEED : 2 * STAGE_BYTES;
    static_assert(BM_ % (8 * NW) == 0 && BN_ % (8 * NW) == 0, "DMA pieces must divide evenly over the waves");
};

// The main loops multiply with the operands SWAPPED (mfma(B fragment, A fragment)): a lane then holds four consecutive
// output columns of one row, and the slab below is filled with 16-byte LDS writes (a quarter of the ds_write count).
// (A register-only epilogue on this layout -- column terms hoisted, 8/16-byte stores straight from the accumulators or
// through a wave-private bf16 transpose -- was built and measured: +13 % on an isolated K = 1536 GEMM, but 3 % SLOWER end
// to end in same-box A/B runs (1,890 vs 1,950 images/s; SD3 1.57 vs 1.60), and hipcc spills the 128-register tile on
// every run-time branch that touches it.  Not kept.)
// Epilogue shared by the DMA kernels: in TM/RB passes, RB row-tiles of every wave -> LDS (fp32) -> fused adds
// (bias, per-sample row vector, residual, scale, SiLU) in fp32 -> 16-byte coalesced stores.
// Development hooks (tile timelines, epilogue ablations, K-loop ablation kernels) exist only in `make EXTRA=-DNATINF_DEV` builds:
// the shipped library carries neither the stamps nor the kernels that give wrong results by design.
#ifdef NATINF_DEV
#define NATINF_TS(i) do { if (g.dbg_ts && tid == 0 && blockIdx.x == 0) g.dbg_ts[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define NATINF_TS(i) do { } while (0)
#endif
// Column-terms-only epilogue (bias, per-sample row vector of a tile that lies inside one sample, scale, activation,
// GroupNorm partials; bf16 output): everything is applied in the accumulator registers, the tile is rounded to bf16 THERE
// and crosses LDS once as 2-byte values -- a quarter of the fp32 slab's LDS traffic, one barrier, and sweeps that are pure
// 16-byte LDS -> global copies.  (Tile timeline of the fp32-slab path at 256x256, shader clocks: slab writes 5.0k +
// sweeps 11.6k + stores 2.8k = 19.5k per tile against 3.7k per 64-wide K-tile of main loop; tools/tile_timeline.py.)
// sum over the 16 lanes of a DPP row, left in every lane of the row: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ float dpp_row_sum(float v) {
    // The opaque statements keep every stage a scalar `v_add_f32_dpp`.  Without them the SLP vectoriser pairs a (sum, sum of squares) couple into
    // `v_pk_add_f32` fed by `v_mov_b32_dpp` copies, and a DPP source then reads a packed-fp32 result two or three instructions after it issued.
    // hipcc keeps its two wait states for that (VALU write -> DPP read), but the packed add takes two passes over the wave: whenever a wave of
    // ANOTHER kernel shared the SIMD (two engines on two HIP streams), lanes 48-63 of the DPP source were still the old value -- one partial sum of
    // one tile off by 10-50 %, a whole image's GroupNorm statistics slightly off, run-to-run.  Alone on its SIMD the kernel never showed it.
    // (tools/debug_det5.py reproduces it on a single launch; tools/scan_pk_hazard.py looks for the shape in the assembly)
    // Round 4: the FIRST stage's input is whatever the caller computed last -- possibly a packed result (the gs / gq accumulation) -- so that statement
    // carries five wait states of its own: the hazard cannot occur here whatever the vectoriser does in front of the call (the scanner, run by the
    // Makefile on every build, stays as the check for the rest of the library).
    asm volatile("s_nop 4" : "+v"(v));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    asm volatile("" : "+v"(v));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    asm volatile("" : "+v"(v));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    asm volatile("" : "+v"(v));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    asm volatile("" : "+v"(v));
    return v;
}
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
// generic -> global address space: the epilogues take their pointers from a GemmArgs copy (k_conv_gn2 re-reads it from the kernel-argument segment), whose
// provenance hipcc cannot see -- it then emits flat_load / flat_store, which go through the LDS aperture check and count on lgkmcnt as well as vmcnt
template <class T> __device__ __forceinline__ __attribute__((address_space(1))) T* as_global(T* p) { return (__attribute__((address_space(1))) T*)p; }
typedef unsigned gu32x4 __attribute__((ext_vector_type(4)));       // (clang vector types: the HIP_vector_type classes cannot be read through an address-space pointer)
typedef float gf32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 gload_u4(const void* p) { const gu32x4 t = *as_global(reinterpret_cast<const gu32x4*>(p)); return make_uint4(t[0], t[1], t[2], t[3]); }
__device__ __forceinline__ float4 gload_f4(const void* p) { const gf32x4 t = *as_global(reinterpret_cast<const gf32x4*>(p)); return make_float4(t[0], t[1], t[2], t[3]); }
__device__ __forceinline__ void gstore_u4(void* p, uint4 v) { *as_global(reinterpret_cast<gu32x4*>(p)) = gu32x4{v.x, v.y, v.z, v.w}; }
// slab -> global copy of sweeps SW .. NSW-1: all LDS reads first, then the stores (template recursion instead of an array of
// kept values: hipcc left a 16-entry uint4 array in scratch memory here)
template <int SW, int NSW, int RPS, int PROW, bool EDGE>
struct SlabCopy {
    template <class T>
    static __device__ __forceinline__ void run(const unsigned char* src, T* dst, int64_t ld, int m, int M) {
        const uint4 v = *reinterpret_cast<const uint4*>(src + SW * RPS * PROW);
        SlabCopy<SW + 1, NSW, RPS, PROW, EDGE>::run(src, dst, ld, m, M);
        if (!EDGE || m + SW * RPS < M) gstore_u4(dst + (int64_t)(SW * RPS) * ld, v);
    }
};
template <int NSW, int RPS, int PROW, bool EDGE>
struct SlabCopy<NSW, NSW, RPS, PROW, EDGE> {
    template <class T> static __device__ __forceinline__ void run(const unsigned char*, T*, int64_t, int, int) {}
};
// DEQ (fp8 operands): acc * (deq_m[row] * deq_n[col]) first, and a row bias (bias_m) next to the column terms.
// OUT8: the tile leaves as e4m3 bytes with one E8M0 scale per 32 columns (OUT_FP8_MX; a block = two 16-column MFMA tiles x the
// four lanes of a row) -- the bytes cross LDS like the bf16 values do, the scale bytes are stored from the registers.
// NSAMP = 2 / 4 (k_conv_gn2 at 8x8 / 4x4: WM == 1, every TM / NSAMP row-tiles of the tile are one sample): the per-sample row vector and the GroupNorm
// partials exist once per sample; gn_part rows are then indexed by (m0 / BM) * NSAMP + sample.
// FIN (k_conv_gn2 at 8x8 / 4x4: the tile = whole samples x all BN_ = N channels): GemmArgs::fin_* -- the consumer's GroupNorm table straight from
// the tile's own partial sums, same additions in the same order as k_gn_finalize (bit-identical tables).
// PAIR (k_conv_gn2: k_pack_frag interleaves the weight rows of n-tiles 2 p, 2 p + 1 -- row r of tile 2 p + h = channel 32 p + 8 (r >> 2) + 4 h + (r & 3)): the four
// rows 4 q + e a lane holds of both tiles are EIGHT consecutive channels 32 p + 8 q .. + 7 -- 16-byte residual loads and 16-byte slab writes instead of 8-byte ones.
// SCALE_ALWAYS (k_conv_gn3: one block per CU, nothing hides the epilogue): the output scale is applied without the wave-uniform `scale != 1` branch around it -- 64 taken-or-not
// branches per tile cost more than 64 packed multiplies by 1.0 (exact: the same bytes).
template <int WM, int WN, int TM, int TN, class Cfg, int ACT, bool GN, bool RES, bool DEQ = false, bool OUT8 = false, int NSAMP = 1, bool FIN = false, bool PAIR = false, bool SCALE_ALWAYS = false>
__device__ __forceinline__ void packed_tile_epilogue(const GemmArgs& g, unsigned char* smem, f32x4 (&acc)[TM][TN],
                                                     int m0, int n0, int z, int tid, int lane, int wm, int wn)
{
    constexpr int BN_ = WN * TN * 16, BM_ = WM * TM * 16, THREADS = WM * WN * 64, PROW = OUT8 ? BN_ + 16 : Cfg::PROW;
    static_assert(!OUT8 || (!GN && !RES && TN % 2 == 0), "fp8 output: plain column terms only");
    static_assert(!PAIR || (TN % 2 == 0 && !OUT8 && !DEQ), "interleaved n-tile pairs: bf16 output");
    static_assert(NSAMP == 1 || ((NSAMP == 2 || NSAMP == 4) && WM == 1 && TM % NSAMP == 0 && !DEQ && !OUT8), "several samples per tile: one wave row, bf16 output");
    // the activation over two accumulator tiles at a time (gelu_tanh_fast8 / silu_fast8: the same operations per element, issued stage by stage)
    constexpr bool ACT8 = (ACT == ACT_GELU_TANH || ACT == ACT_SILU) && !GN && TN % 2 == 0 && NSAMP == 1;
    const int r = lane & 15, q = lane >> 4;
    // first column, inside the wave's TN * 16, of the four consecutive columns lane group q holds of accumulator tile j
    auto ncol = [&](int j) __attribute__((always_inline)) { return PAIR ? 32 * (j >> 1) + 8 * q + 4 * (j & 1) : 16 * j + 4 * q; };
    float dn[DEQ ? TN : 1][4], rsc[DEQ ? TM : 1], rbm[DEQ ? TM : 1];
    if constexpr (DEQ) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * TN * 16 + j * 16 + q * 4;
            float4 d = make_float4(1.f, 1.f, 1.f, 1.f);
            if (g.deq_n && n < g.N) d = *reinterpret_cast<const float4*>(g.deq_n + (int64_t)z * g.deq_n_bs + n);
            dn[j][0] = d.x; dn[j][1] = d.y; dn[j][2] = d.z; dn[j][3] = d.w;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = min(m0 + wm * TM * 16 + i * 16 + r, g.M - 1);
            rsc[i] = g.deq_m ? g.deq_m[(int64_t)z * g.deq_m_bs + m] : 1.f;
            rbm[i] = g.bias_m ? g.bias_m[m] : 0.f;
        }
    }
    // (two samples per tile: the row vector differs per sample.  The residual forms never carry one -- Conv_1 of a res-block has no time-embedding
    // row -- and keep ONE set of column terms: the second set is what tips their 256-register epilogue into dozens of spills)
    // (four samples per tile -- the 4x4 level: every row-tile is a sample -- : ONE set of column terms (the bias) and the sample's row vector fetched
    // per row-tile, LAZY_RV: four resident sets are 64 registers next to the 64 accumulators)
    constexpr bool LAZY_RV = NSAMP > 2 && !RES;
    constexpr int NCT = (NSAMP > 1 && !RES && !LAZY_RV) ? NSAMP : 1;
    float ct[NCT][TN][4];
#pragma unroll
    for (int sm = 0; sm < NCT; ++sm)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 16 + ncol(j);
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f), rv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < g.N) {
            if (g.bias_n) b = gload_f4(g.bias_n + n);
            // (NSAMP > 1: the further samples' rows; a partial last tile reads the last real sample's again -- those rows are not stored)
            const int srow = NSAMP > 1 ? min(m0 + sm * (BM_ / NSAMP), g.M - 1) : m0;
            if (g.rowvec && !LAZY_RV) rv = gload_f4(g.rowvec + (int64_t)((srow >> g.log_rows_per_sample) + z * g.z_samples) * g.rowvec_ld + n);
        }
        ct[sm][j][0] = b.x + rv.x; ct[sm][j][1] = b.y + rv.y; ct[sm][j][2] = b.z + rv.z; ct[sm][j][3] = b.w + rv.w;
    }
    // bf16 residual, fetched in the accumulator layout (8 bytes per lane: 4 columns of one row), all requests in flight at once
    uint2 rs[RES ? TM : 1][RES ? TN : 1];
    if constexpr (RES) {
        const bf16* rb = g.resid + (int64_t)z * g.c_bs + n0 + wn * TN * 16;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = min(m0 + wm * TM * 16 + i * 16 + r, g.M - 1);
            if constexpr (PAIR) {
#pragma unroll
                for (int p = 0; p < TN / 2; ++p) {                         // both tiles of a pair with one 16-byte load
                    const uint4 v = n0 + wn * TN * 16 + ncol(2 * p) < g.N ? gload_u4(rb + (int64_t)m * g.resid_ld + ncol(2 * p)) : make_uint4(0u, 0u, 0u, 0u);
                    rs[i][2 * p] = make_uint2(v.x, v.y); rs[i][2 * p + 1] = make_uint2(v.z, v.w);
                }
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    rs[i][j] = n0 + wn * TN * 16 + ncol(j) < g.N ? *reinterpret_cast<const uint2*>(rb + (int64_t)m * g.resid_ld + ncol(j)) : make_uint2(0u, 0u);
            }
        }
    }
    float gs[NSAMP][TN], gq[NSAMP][TN];
#pragma unroll
    for (int sm = 0; sm < NSAMP; ++sm)
#pragma unroll
        for (int j = 0; j < TN; ++j) { gs[sm][j] = 0.f; gq[sm][j] = 0.f; }
    const float scale = g.scale;
    float4 lrv[LAZY_RV ? TN : 1];                       // LAZY_RV: the row vector of the sample row-tile i belongs to
    auto fetch_rowvec = [&](int i) __attribute__((always_inline)) {
        if constexpr (LAZY_RV) {
            const int srow = min(m0 + (i / (TM / NSAMP)) * (BM_ / NSAMP), g.M - 1);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * TN * 16 + ncol(j);
                lrv[j] = (g.rowvec && n < g.N) ? *reinterpret_cast<const float4*>(g.rowvec + (int64_t)(srow >> g.log_rows_per_sample) * g.rowvec_ld + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    // the finished fp32 values of accumulator tile (i, j)
    auto value = [&](int i, int j, float (&v)[4]) __attribute__((always_inline)) {
        const int sm = NSAMP > 1 ? i / (TM / NSAMP) : 0;                  // (i is a compile-time constant at every call site)
        const int sc_ = NCT > 1 ? sm : 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // (DEQ: two vector instructions per element -- (acc * row scale) fma'd with the column scale onto the column terms -- instead of four)
            if constexpr (DEQ) v[e] = __builtin_fmaf(acc[i][j][e] * rsc[i], dn[j][e], ct[sc_][j][e]);
            else v[e] = acc[i][j][e] + ct[sc_][j][e];
        }
        if constexpr (DEQ && !OUT8) {                                     // (OUT8 -- fc1 --: no row bias, scale 1: fp8_epi() checks it)
            if (g.bias_m) {                                               // (wave-uniform: only the V^T GEMMs carry a row bias)
                asm volatile("");                                         // (a real branch: without it hipcc if-converts this into an add + four selects per value on EVERY launch)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += rbm[i];
            }
        }
        if constexpr (LAZY_RV) { v[0] += lrv[j].x; v[1] += lrv[j].y; v[2] += lrv[j].z; v[3] += lrv[j].w; }
        if constexpr (RES) {
            const bf16x4_t x = __builtin_bit_cast(bf16x4_t, rs[i][j]);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)x[e];
        }
        if constexpr (SCALE_ALWAYS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= scale;
        } else if (!OUT8 && scale != 1.0f) {                              // (wave-uniform; the transformer engines' GEMMs all have scale 1)
            asm volatile("");                                             // (a real branch, as above)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= scale;
        }
        // (tanh-GELU / SiLU: eight values -- two accumulator tiles -- at once, stage by stage, where the call sites below pair the tiles: ACT8)
        if constexpr (ACT != ACT_NONE && !ACT8) apply_act4(v, ACT);
        if constexpr (GN) {
            gs[sm][j] += (v[0] + v[1]) + (v[2] + v[3]);
            gq[sm][j] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
    };
    auto act8 = [&](float (&v0)[4], float (&v1)[4]) __attribute__((always_inline)) {
        float v8[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if constexpr (ACT == ACT_GELU_TANH) gelu_tanh_fast8(v8); else silu_fast8(v8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v0[e] = v8[e]; v1[e] = v8[4 + e]; }
    };
    constexpr int EB = OUT8 ? 1 : 2;                   // bytes per output element
    unsigned char* wbase = smem + (wm * TM * 16 + r) * PROW + (wn * TN * 16 + q * 4) * EB;
    if constexpr (!OUT8 && PAIR) {
        unsigned char* wb2 = smem + (wm * TM * 16 + r) * PROW + (wn * TN * 16) * 2;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            fetch_rowvec(i);
#pragma unroll
            for (int p = 0; p < TN / 2; ++p) {
                float v0[4], v1[4];
                value(i, 2 * p, v0);
                value(i, 2 * p + 1, v1);
                if constexpr (ACT8) act8(v0, v1);
                bf16x4_t o0, o1;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o0[e] = (bf16)v0[e]; o1[e] = (bf16)v1[e]; }
                const uint2 a = __builtin_bit_cast(uint2, o0), b = __builtin_bit_cast(uint2, o1);
                *reinterpret_cast<uint4*>(wb2 + i * 16 * PROW + ncol(2 * p) * 2) = make_uint4(a.x, a.y, b.x, b.y);
            }
        }
    } else if constexpr (!OUT8) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            fetch_rowvec(i);
            if constexpr (ACT8) {
#pragma unroll
                for (int j = 0; j < TN; j += 2) {
                    float v0[4], v1[4];
                    value(i, j, v0);
                    value(i, j + 1, v1);
                    act8(v0, v1);
                    bf16x4_t o0, o1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o0[e] = (bf16)v0[e]; o1[e] = (bf16)v1[e]; }
                    *reinterpret_cast<uint2*>(wbase + i * 16 * PROW + j * 32) = __builtin_bit_cast(uint2, o0);
                    *reinterpret_cast<uint2*>(wbase + i * 16 * PROW + (j + 1) * 32) = __builtin_bit_cast(uint2, o1);
                }
            } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float v[4];
                value(i, j, v);
                bf16x4_t o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
                *reinterpret_cast<uint2*>(wbase + i * 16 * PROW + j * 32) = __builtin_bit_cast(uint2, o);
            }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * TM * 16 + i * 16 + r;
            unsigned e8pair = 0;
#pragma unroll
            for (int b = 0; b < TN / 2; ++b) {
                float v0[4], v1[4];
                value(i, 2 * b, v0);
                value(i, 2 * b + 1, v1);
                if constexpr (ACT8) act8(v0, v1);
                float amax = fmaxf(fmaxf(fmaxf(fabsf(v0[0]), fabsf(v0[1])), fmaxf(fabsf(v0[2]), fabsf(v0[3]))),
                                   fmaxf(fmaxf(fabsf(v1[0]), fabsf(v1[1])), fmaxf(fabsf(v1[2]), fabsf(v1[3]))));
                amax = group4_max_nonneg(amax);
                float inv;
                const unsigned e8 = mx_scale_of(amax, inv);
                *reinterpret_cast<unsigned*>(wbase + i * 16 * PROW + (2 * b) * 16) = pack_fp8x4<false>(v0[0] * inv, v0[1] * inv, v0[2] * inv, v0[3] * inv);
                *reinterpret_cast<unsigned*>(wbase + i * 16 * PROW + (2 * b + 1) * 16) = pack_fp8x4<false>(v1[0] * inv, v1[1] * inv, v1[2] * inv, v1[3] * inv);
                e8pair |= e8 << (8 * b);
            }
            // the wave's 64 columns are blocks (wn & 1) * 2 + {0, 1} of their 128-column group: two adjacent scale bytes
            static_assert(TN == 4, "scale bytes are stored as one 16-bit pair per row and wave");
            const int nb = n0 + wn * 64;
            if (q == 0 && m < g.M && nb < g.N)
                *reinterpret_cast<unsigned short*>(g.c_mx + (int64_t)z * g.c_mx_bs + ((int64_t)(nb >> 7) * g.c_mx_ld + m) * 4 + ((nb >> 5) & 3)) = (unsigned short)e8pair;
        }
    }
    float2* const sred = reinterpret_cast<float2*>(smem + Cfg::PACK_SLAB);
    if constexpr (GN) {
        // fixed-order reduction: the 16 row-lanes of a lane group = one DPP row (pairs, quads, the two quads of a half row, the two half
        // rows: four v_add_f32 with a DPP source each -- the ds_bpermute butterflies this replaces cost 3-5k clocks per tile), then the
        // WM waves of a column through LDS, behind the slab: the same barrier publishes both, and nothing waits for the tile's stores
        // (the reduction used to run after the copy-out, behind two barriers -- each of which drains the outstanding global stores)
#pragma unroll
        for (int sm = 0; sm < NSAMP; ++sm)
#pragma unroll
            for (int j = 0; j < TN; ++j) { gs[sm][j] = dpp_row_sum(gs[sm][j]); gq[sm][j] = dpp_row_sum(gq[sm][j]); }
        if (r == 0) {
#pragma unroll
            for (int sm = 0; sm < NSAMP; ++sm)
#pragma unroll
                for (int j = 0; j < TN; ++j) sred[(NSAMP > 1 ? sm : wm) * (BN_ / 4) + wn * TN * 4 + (ncol(j) >> 2)] = make_float2(gs[sm][j], gq[sm][j]);
        }
    }
    __syncthreads();
    NATINF_TS(3);
    constexpr int CPR = BN_ * EB / 16, RPS = THREADS / CPR, NSW = BM_ / RPS;
    const int cchunk = tid % CPR, rsub = tid / CPR, n = n0 + cchunk * (16 / EB);
    if (n < g.N) {
        const unsigned char* src = smem + rsub * PROW + cchunk * 16;
        if constexpr (OUT8) {
            uint8_t* dst = reinterpret_cast<uint8_t*>(g.c) + (int64_t)z * g.c_bs + (int64_t)(m0 + rsub) * g.c_ld + n;
            if (m0 + BM_ <= g.M) SlabCopy<0, NSW, RPS, PROW, false>::run(src, dst, g.c_ld, m0 + rsub, g.M);
            else SlabCopy<0, NSW, RPS, PROW, true>::run(src, dst, g.c_ld, m0 + rsub, g.M);
        } else {
            bf16* dst = reinterpret_cast<bf16*>(g.c) + (int64_t)z * g.c_bs + (int64_t)(m0 + rsub) * g.c_ld + n;
            if (m0 + BM_ <= g.M) SlabCopy<0, NSW, RPS, PROW, false>::run(src, dst, g.c_ld, m0 + rsub, g.M);
            else SlabCopy<0, NSW, RPS, PROW, true>::run(src, dst, g.c_ld, m0 + rsub, g.M);
        }
    }
    NATINF_TS(4);
    if constexpr (GN) {
        if (tid < BN_ / 4 && n0 + tid * 4 < g.N) {
            if constexpr (NSAMP > 1) {
#pragma unroll
                for (int sm = 0; sm < NSAMP; ++sm)
                    if (m0 + sm * (BM_ / NSAMP) < g.M)
                        reinterpret_cast<float2*>(g.gn_part)[(int64_t)((m0 / BM_) * NSAMP + sm) * g.gn_quads + (n0 >> 2) + tid] = sred[sm * (BN_ / 4) + tid];
            } else {
                float s = 0.f, qq = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) { s += sred[w * (BN_ / 4) + tid].x; qq += sred[w * (BN_ / 4) + tid].y; }
                reinterpret_cast<float2*>(g.gn_part)[(int64_t)(m0 / BM_) * g.gn_quads + (n0 >> 2) + tid] = make_float2(s, qq);
            }
        }
        if constexpr (FIN) {
            // (round 5: also the 256 x 256 tile of k_conv_gn3 at 16x16 -- ONE sample, every channel, two wave rows: a quad's partial is the wave rows' sum in the
            // order the gn_part writer above adds them, so the table is the one k_gn_finalize computes from that single tile row)
            static_assert((WM == 1 || NSAMP == 1) && BN_ <= THREADS, "one channel per thread; several samples per tile: one wave row");
            const int n = n0 + tid;
            if (g.fin_scale && tid < BN_ && n < g.N) {
                constexpr int HWS = BM_ / NSAMP;                       // rows of a sample
                const int cg = g.fin_cg, qpg = cg >> 2, q0 = (n / cg) * qpg - (n0 >> 2);
                const float inv = 1.0f / (float)(cg * HWS), ga = g.fin_gamma[n], be = g.fin_beta[n];
#pragma unroll
                for (int sm = 0; sm < NSAMP; ++sm) {
                    if (m0 + sm * HWS >= g.M) break;
                    float s = 0.f, qq = 0.f;
                    for (int i = 0; i < qpg; ++i) {
                        if constexpr (NSAMP == 1 && WM > 1) {
                            float ps = 0.f, pq = 0.f;
#pragma unroll
                            for (int w = 0; w < WM; ++w) { ps += sred[w * (BN_ / 4) + q0 + i].x; pq += sred[w * (BN_ / 4) + q0 + i].y; }
                            s += ps; qq += pq;
                        } else { const float2 v = sred[sm * (BN_ / 4) + q0 + i]; s += v.x; qq += v.y; }
                    }
                    const float mean = s * inv;
                    float var = qq * inv - mean * mean;
                    var = var < 0.f ? 0.f : var;
                    const float sc = (1.0f / sqrtf(var + g.fin_eps)) * ga;
                    const int64_t o = (int64_t)(m0 / HWS + sm) * g.fin_ld + n;
                    g.fin_scale[o] = sc * g.fin_mul;
                    g.fin_shift[o] = (be - mean * sc) * g.fin_mul;
                }
            }
        }
    }
}

template <int WM, int WN, int TM, int TN, class Cfg>
__device__ __forceinline__ void dma_tile_epilogue(const GemmArgs& g, unsigned char* smem, f32x4 (&acc)[TM][TN],
                                                  int m0, int n0, int z, int tid, int lane, int wm, int wn)
{
    constexpr int BN_ = WN * TN * 16, BM_ = WM * TM * 16, THREADS = WM * WN * 64;
    float* sC = reinterpret_cast<float*>(smem);
    NATINF_TS(2);
#ifdef NATINF_DEV
    if (g.c_mode == 102) {                          // timing experiment: no epilogue at all (one store keeps the accumulators live)
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 123.456f) reinterpret_cast<float*>(g.c)[tid] = t;
        return;
    }
#endif
    float st[4] = {0.f, 0.f, 0.f, 0.f};            // fused GroupNorm partials of this thread's two 4-channel quads
    constexpr int CROW = Cfg::CROW, RB = Cfg::RB, CPR = BN_ / 8;            // 16-byte output chunks per row
    constexpr int ROWS_PER_SWEEP = THREADS / CPR;
    constexpr int NP = TM / RB, NSW = Cfg::SLAB_ROWS / ROWS_PER_SWEEP;
    uint4 keep[NSW][2];                            // one pass's finished outputs (bf16: [0] only), stored after its last load
    const int cchunk = tid % CPR, rsub = tid / CPR;
    const int n = n0 + cchunk * 8;
    const bool n_in = n < g.N;
    float bn[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) bn[q] = 0.f;
    if (g.bias_n && n_in && g.c_mode != OUT_F32_NCHW) {
        const float4 u = *reinterpret_cast<const float4*>(g.bias_n + n), w = *reinterpret_cast<const float4*>(g.bias_n + n + 4);
        bn[0] = u.x; bn[1] = u.y; bn[2] = u.z; bn[3] = u.w; bn[4] = w.x; bn[5] = w.y; bn[6] = w.z; bn[7] = w.w;
    }
    float dn[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) dn[q] = 1.f;
    if (g.deq_n && n_in) {
        const float* pd = g.deq_n + (int64_t)z * g.deq_n_bs + n;
        const float4 u = *reinterpret_cast<const float4*>(pd), w = *reinterpret_cast<const float4*>(pd + 4);
        dn[0] = u.x; dn[1] = u.y; dn[2] = u.z; dn[3] = u.w; dn[4] = w.x; dn[5] = w.y; dn[6] = w.z; dn[7] = w.w;
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(dn[0]), "+v"(dn[1]), "+v"(dn[2]), "+v"(dn[3]), "+v"(dn[4]), "+v"(dn[5]), "+v"(dn[6]), "+v"(dn[7]));
    // the bias is waited for here, once, and handed on as asm outputs: hipcc then attaches no pending load to bn[]
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bn[0]), "+v"(bn[1]), "+v"(bn[2]), "+v"(bn[3]), "+v"(bn[4]), "+v"(bn[5]), "+v"(bn[6]), "+v"(bn[7]));
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        if (pass) __syncthreads();                 // previous pass fully read
#pragma unroll
        for (int ii = 0; ii < RB; ++ii)
#pragma unroll
            for (int j = 0; j < TN; ++j)       // swapped-operand accumulators: a lane holds 4 consecutive columns of row (lane & 15)
                *reinterpret_cast<f32x4*>(sC + (wm * RB * 16 + ii * 16 + (lane & 15)) * CROW + wn * TN * 16 + j * 16 + (lane >> 4) * 4) = acc[pass * RB + ii][j];
        __syncthreads();
        NATINF_TS(3 + 3 * pass);
        // slab row s  <->  tile row  (s / (RB*16)) * TM*16 + pass*RB*16 + s % (RB*16)
        if (g.c_mode == OUT_F32_NCHW) {
            float* out = reinterpret_cast<float*>(g.c);
            const int nvalid = min(BN_, g.N - n0);
            for (int e = tid; e < Cfg::SLAB_ROWS * nvalid; e += THREADS) {
                const int s = e % Cfg::SLAB_ROWS, nn = e / Cfg::SLAB_ROWS;
                const int m = m0 + (s / (RB * 16)) * (TM * 16) + pass * RB * 16 + s % (RB * 16);
                if (m < g.M) {
                    float v = sC[s * CROW + nn];
                    if (g.bias_n) v += g.bias_n[n0 + nn];
                    v *= g.scale;
                    const int b = m >> g.logHW, p = m & ((1 << g.logHW) - 1);
                    out[((int64_t)b * g.N + n0 + nn) * ((int64_t)1 << g.logHW) + p] = v;
                }
            }
            continue;
        }
#pragma unroll
        for (int sw = 0; sw < NSW; ++sw) {
            const int s = sw * ROWS_PER_SWEEP + rsub;
            const int m = m0 + (s / (RB * 16)) * (TM * 16) + pass * RB * 16 + s % (RB * 16);
            if (m >= g.M || !n_in) continue;
            const float4 u = *reinterpret_cast<const float4*>(sC + s * CROW + cchunk * 8);
            const float4 w = *reinterpret_cast<const float4*>(sC + s * CROW + cchunk * 8 + 4);
            float v[8] = {u.x, u.y, u.z, u.w, w.x, w.y, w.z, w.w};
            if (g.deq_m || g.deq_n) {
                const float dm = g.deq_m ? g.deq_m[(int64_t)z * g.deq_m_bs + m] : 1.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] *= dm * dn[q];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += bn[q];
            if (g.bias_m) {
                const float bm = g.bias_m[m];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += bm;
            }
            if (g.rowvec) {
                const float* rv = g.rowvec + (int64_t)((m >> g.log_rows_per_sample) + z * g.z_samples) * g.rowvec_ld + n;
                const float4 s4 = *reinterpret_cast<const float4*>(rv), t4 = *reinterpret_cast<const float4*>(rv + 4);
                v[0] += s4.x; v[1] += s4.y; v[2] += s4.z; v[3] += s4.w; v[4] += t4.x; v[5] += t4.y; v[6] += t4.z; v[7] += t4.w;
            }
            if (g.gate) {
                const float* gv = g.gate + (int64_t)((m >> g.log_rows_per_sample) + z * g.z_samples) * g.gate_ld + n;
                const float4 s4 = *reinterpret_cast<const float4*>(gv), t4 = *reinterpret_cast<const float4*>(gv + 4);
                v[0] *= s4.x; v[1] *= s4.y; v[2] *= s4.z; v[3] *= s4.w; v[4] *= t4.x; v[5] *= t4.y; v[6] *= t4.z; v[7] *= t4.w;
            }
            if (g.resid) {
                const bf16x8 rs = *reinterpret_cast<const bf16x8*>(g.resid + (int64_t)z * g.c_bs + (int64_t)m * g.resid_ld + n);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += (float)rs[q];
            }
            if (g.resid_f32) {
                const float* rp = g.resid_f32 + (int64_t)z * g.c_bs + (int64_t)m * g.resid_f32_ld + n;
                const float4 s4 = *reinterpret_cast<const float4*>(rp), t4 = *reinterpret_cast<const float4*>(rp + 4);
                v[0] += s4.x; v[1] += s4.y; v[2] += s4.z; v[3] += s4.w; v[4] += t4.x; v[5] += t4.y; v[6] += t4.z; v[7] += t4.w;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] *= g.scale;
            apply_act8(v, g.act);
            if (g.gn_part) {
                st[0] += (v[0] + v[1]) + (v[2] + v[3]); st[1] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                st[2] += (v[4] + v[5]) + (v[6] + v[7]); st[3] += (v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]);
            }
            if (g.c_mode == OUT_BF16) {
                bf16x8 o;
#pragma unroll
                for (int q = 0; q < 8; ++q) o[q] = (bf16)v[q];
                keep[sw][0] = __builtin_bit_cast(uint4, o);
            } else if (g.c_mode == OUT_FP8_MX) {
                // a 32-column block = this thread's 8 columns and those of its three neighbours (tid ^ 1, ^ 2: same row)
                float amax = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) amax = fmaxf(amax, fabsf(v[q]));
                amax = fmaxf(amax, __shfl_xor(amax, 1));
                amax = fmaxf(amax, __shfl_xor(amax, 2));
                float inv;
                const unsigned e8 = mx_scale_of(amax, inv);
                keep[sw][0] = make_uint4(pack_fp8x4(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv),
                                         pack_fp8x4(v[4] * inv, v[5] * inv, v[6] * inv, v[7] * inv), e8, 0u);
            } else {
                keep[sw][0] = __builtin_bit_cast(uint4, make_float4(v[0], v[1], v[2], v[3]));
                keep[sw][1] = __builtin_bit_cast(uint4, make_float4(v[4], v[5], v[6], v[7]));
            }
        }
        NATINF_TS(4 + 3 * pass);
        // ---- this pass's stores, after all of its loads.  vmcnt counts stores as well as loads, and hipcc waits
        // ---- vmcnt(0) for a loaded value whenever stores are outstanding too (mixed event types retire out of order):
        // ---- with the store inside the sweep, every sweep's bias / row-vector / gate / residual use waited for the
        // ---- previous sweep's store round trip -- 11.8 us of epilogue per 256x256 tile (K-scan in profiles/r01 notes).
#pragma unroll
        for (int sw = 0; sw < NSW; ++sw) {
            const int s = sw * ROWS_PER_SWEEP + rsub;
            const int m = m0 + (s / (RB * 16)) * (TM * 16) + pass * RB * 16 + s % (RB * 16);
            if (m >= g.M || !n_in) continue;
#ifdef NATINF_DEV
            if (g.c_mode == 103) {                  // timing experiment: everything but the global stores
                if (keep[sw][0].x == 0x12345678u) *reinterpret_cast<uint4*>(g.c) = keep[sw][0];
            } else
#endif
            if (g.c_mode == OUT_BF16) {
                *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(g.c) + (int64_t)z * g.c_bs + (int64_t)m * g.c_ld + n) = keep[sw][0];
            } else if (g.c_mode == OUT_FP8_MX) {
                *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(g.c) + (int64_t)z * g.c_bs + (int64_t)m * g.c_ld + n) = make_uint2(keep[sw][0].x, keep[sw][0].y);
                if ((cchunk & 3) == 0) g.c_mx[(int64_t)z * g.c_mx_bs + ((int64_t)(n >> 7) * g.c_mx_ld + m) * 4 + ((n >> 5) & 3)] = (uint8_t)keep[sw][0].z;
            } else {
                float* o = reinterpret_cast<float*>(g.c) + (int64_t)z * g.c_bs + (int64_t)m * g.c_ld + n;
                *reinterpret_cast<uint4*>(o) = keep[sw][0];
                *reinterpret_cast<uint4*>(o + 4) = keep[sw][1];
            }
        }
        NATINF_TS(5 + 3 * pass);
    }
    if (g.gn_part) {
        // block reduction in a fixed order: [row-thread][chunk][4] through LDS, then one thread per quad
        __syncthreads();
        float* sred = reinterpret_cast<float*>(smem);
        *reinterpret_cast<float4*>(sred + (rsub * CPR + cchunk) * 4) = make_float4(st[0], st[1], st[2], st[3]);
        __syncthreads();
        if (tid < CPR * 2 && n0 + tid * 4 < g.N) {
            const int ch = tid >> 1, hf = (tid & 1) * 2;
            float s = 0.f, q = 0.f;
#pragma unroll 4
            for (int r = 0; r < ROWS_PER_SWEEP; ++r) { s += sred[(r * CPR + ch) * 4 + hf]; q += sred[(r * CPR + ch) * 4 + hf + 1]; }
            reinterpret_cast<float2*>(g.gn_part)[(int64_t)(m0 / BM_) * g.gn_quads + (n0 >> 2) + tid] = make_float2(s, q);
        }
    }
}

// Residual-stream epilogue of the transformer engines: out_f32 = (resid_f32 + gate * (acc + bias [+ row vector])) * scale with the
// gate / row vector constant over the block tile (one sample per tile), straight from the accumulator registers -- a lane holds
// four consecutive columns of a row, so the fp32 residual is read and the result written as 16-byte accesses (64 B per row and
// instruction), no LDS, no barrier.  The residual rows are fetched half a tile at a time (64 VGPRs in flight).
// HI_ (0 = by the register budget of a 256-register kernel): residual row-tiles fetched per round trip.  The w128 kernels (accumulators in AGPRs, the fragment
// registers dead by the epilogue) fetch all eight at once: one memory round trip per half tile instead of four -- with one block per CU nothing hides them.
template <int WM, int WN, int TM, int TN, bool DEQ = false, int HI_ = 0>
__device__ __forceinline__ void direct_f32_epilogue(const GemmArgs& g, f32x4 (&acc)[TM][TN], int m0, int n0, int z, int lane, int wm, int wn)
{
    const int r = lane & 15, q = lane >> 4;
    const int64_t sample = (int64_t)((m0 >> g.log_rows_per_sample) + z * g.z_samples);
    float ct[TN][4], gt[TN][4];
    float dn[DEQ ? TN : 1][4], rsc[DEQ ? TM : 1];          // fp8 operands: column / row dequantization scales
    if constexpr (DEQ) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * TN * 16 + j * 16 + q * 4;
            float4 d = make_float4(1.f, 1.f, 1.f, 1.f);
            if (g.deq_n && n < g.N) d = *reinterpret_cast<const float4*>(g.deq_n + (int64_t)z * g.deq_n_bs + n);
            dn[j][0] = d.x; dn[j][1] = d.y; dn[j][2] = d.z; dn[j][3] = d.w;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
            rsc[i] = g.deq_m ? g.deq_m[(int64_t)z * g.deq_m_bs + min(m0 + wm * TM * 16 + i * 16 + r, g.M - 1)] : 1.f;
    }
    bool n_ok[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 16 + j * 16 + q * 4;
        n_ok[j] = n < g.N;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f), rv = make_float4(0.f, 0.f, 0.f, 0.f), gv = make_float4(1.f, 1.f, 1.f, 1.f);
        if (n_ok[j]) {
            if (g.bias_n) b = gload_f4(g.bias_n + n);
            if (g.rowvec) rv = *reinterpret_cast<const float4*>(g.rowvec + sample * g.rowvec_ld + n);
            if (g.gate) gv = *reinterpret_cast<const float4*>(g.gate + sample * g.gate_ld + n);
        }
        ct[j][0] = b.x + rv.x; ct[j][1] = b.y + rv.y; ct[j][2] = b.z + rv.z; ct[j][3] = b.w + rv.w;
        gt[j][0] = gv.x; gt[j][1] = gv.y; gt[j][2] = gv.z; gt[j][3] = gv.w;
    }
    const float scale = g.scale;
    const int64_t eoff = (int64_t)z * g.c_bs + n0 + wn * TN * 16 + q * 4;                        // element offset of the lane's first column (fp32 and half streams alike)
    constexpr int HI = HI_ > 0 ? HI_ : (DEQ ? (TM > 2 ? 2 : TM) : (TM > 4 ? TM / 2 : TM));      // residual row-tiles in flight (register budget)
    typedef _Float16 f16x4_ds __attribute__((ext_vector_type(4)));
    // F16 (GemmArgs::stream_f16, kernel-uniform): the stream's rows are IEEE half -- 8 bytes per lane and access instead of 16, half the bytes of the epilogue that
    // bounds these launches (gemm_w128.h); the update itself stays fp32 with ONE rounding to half
    auto sweep = [&](auto f16_tag) __attribute__((always_inline)) {
        constexpr bool F16 = decltype(f16_tag)::value;
        const float* rb = g.resid_f32 + eoff;
        float* cb = reinterpret_cast<float*>(g.c) + eoff;
        const _Float16* rbh = reinterpret_cast<const _Float16*>(g.resid_f32) + eoff;
        _Float16* cbh = reinterpret_cast<_Float16*>(g.c) + eoff;
#pragma unroll
        for (int h = 0; h < TM / HI; ++h) {
            f32x4 rs[F16 ? 1 : HI][F16 ? 1 : TN];
            uint2 rh[F16 ? HI : 1][F16 ? TN : 1];
#pragma unroll
            for (int i = 0; i < HI; ++i) {
                const int m = min(m0 + wm * TM * 16 + (h * HI + i) * 16 + r, g.M - 1);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (F16) rh[i][j] = n_ok[j] ? *reinterpret_cast<const uint2*>(rbh + (int64_t)m * g.resid_f32_ld + j * 16) : make_uint2(0u, 0u);
                    else rs[i][j] = n_ok[j] ? *reinterpret_cast<const f32x4*>(rb + (int64_t)m * g.resid_f32_ld + j * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int i = 0; i < HI; ++i) {
                const int m = m0 + wm * TM * 16 + (h * HI + i) * 16 + r;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x4 v;
                    f16x4_ds xh;
                    if constexpr (F16) xh = __builtin_bit_cast(f16x4_ds, rh[i][j]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a0 = DEQ ? acc[h * HI + i][j][e] * (rsc[h * HI + i] * dn[j][e]) : acc[h * HI + i][j][e];
                        float x_;
                        if constexpr (F16) x_ = (float)xh[e]; else x_ = rs[i][j][e];
                        v[e] = ((a0 + ct[j][e]) * gt[j][e] + x_) * scale;
                    }
                    if (m < g.M && n_ok[j]) {
                        if constexpr (F16) {
                            const f16x4_ds o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                            *reinterpret_cast<uint2*>(cbh + (int64_t)m * g.c_ld + j * 16) = __builtin_bit_cast(uint2, o);
                        } else *reinterpret_cast<f32x4*>(cb + (int64_t)m * g.c_ld + j * 16) = v;
                    }
                }
            }
        }
    };
    // Half stream, 16-byte form (the wave's TN column tiles all inside N; TN even): the lanes q, q ^ 1 of a row -- rows of sixteen lanes that hold columns 4 q .. 4 q + 3
    // of EVERY column tile -- trade halves with v_permlane16_swap (gfx950: the odd rows of one register against the even rows of another), so that the even lane
    // owns eight consecutive columns of tile 2 p and the odd lane eight of tile 2 p + 1: one 16-byte access per lane and tile pair, 64 contiguous bytes per row and
    // instruction -- the fp32 form's access shape at half its count (the 8-byte form above moves 32-byte segments).
    auto sweep16 = [&]() __attribute__((always_inline)) {
        static_assert(TN % 2 == 0 || TN == 1, "tile pairs");
        const int jt_off = (q & 1) * 16 + (q >> 1) * 8 - q * 4;            // the lane's eight columns start at tile 2 p + (q & 1), column (q >> 1) * 8 (eoff already holds q * 4)
        const _Float16* rbh = reinterpret_cast<const _Float16*>(g.resid_f32) + eoff + jt_off;
        _Float16* cbh = reinterpret_cast<_Float16*>(g.c) + eoff + jt_off;
        constexpr int NP = TN / 2 > 0 ? TN / 2 : 1;
#pragma unroll
        for (int h = 0; h < TM / HI; ++h) {
            uint4 rq[HI][NP];
#pragma unroll
            for (int i = 0; i < HI; ++i) {
                const int m = min(m0 + wm * TM * 16 + (h * HI + i) * 16 + r, g.M - 1);
#pragma unroll
                for (int p = 0; p < NP; ++p) rq[i][p] = *reinterpret_cast<const uint4*>(rbh + (int64_t)m * g.resid_f32_ld + p * 32);
            }
#pragma unroll
            for (int i = 0; i < HI; ++i) {
                const int m = m0 + wm * TM * 16 + (h * HI + i) * 16 + r;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    // even lane: (x, y) = its tile-2p columns, (z, w) = the odd lane's; odd lane: (x, y) = the even lane's tile-2p+1 columns, (z, w) = its own.
                    // odd.x <-> even.z, odd.y <-> even.w: afterwards (x, y) = the lane's columns of tile 2 p, (z, w) = of tile 2 p + 1, on every lane
                    auto s0 = __builtin_amdgcn_permlane16_swap(rq[i][p].x, rq[i][p].z, false, false);
                    auto s1 = __builtin_amdgcn_permlane16_swap(rq[i][p].y, rq[i][p].w, false, false);
                    const f16x4_ds x0 = __builtin_bit_cast(f16x4_ds, make_uint2(s0[0], s1[0])), x1 = __builtin_bit_cast(f16x4_ds, make_uint2(s0[1], s1[1]));
                    f16x4_ds o0, o1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a0 = DEQ ? acc[h * HI + i][2 * p][e] * (rsc[h * HI + i] * dn[2 * p][e]) : acc[h * HI + i][2 * p][e];
                        const float a1 = DEQ ? acc[h * HI + i][2 * p + 1][e] * (rsc[h * HI + i] * dn[2 * p + 1][e]) : acc[h * HI + i][2 * p + 1][e];
                        o0[e] = (_Float16)(((a0 + ct[2 * p][e]) * gt[2 * p][e] + (float)x0[e]) * scale);
                        o1[e] = (_Float16)(((a1 + ct[2 * p + 1][e]) * gt[2 * p + 1][e] + (float)x1[e]) * scale);
                    }
                    const uint2 u0 = __builtin_bit_cast(uint2, o0), u1 = __builtin_bit_cast(uint2, o1);
                    auto t0 = __builtin_amdgcn_permlane16_swap(u0.x, u1.x, false, false);          // the same trade back: (x, y, z, w) = the lane's eight consecutive columns
                    auto t1 = __builtin_amdgcn_permlane16_swap(u0.y, u1.y, false, false);
                    if (m < g.M) *reinterpret_cast<uint4*>(cbh + (int64_t)m * g.c_ld + p * 32) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
                }
            }
        }
    };
    if (g.stream_f16) {
        bool wide = false;
        if constexpr (TN % 2 == 0) wide = n0 + (wn + 1) * TN * 16 <= g.N && g.resid_f32_ld % 8 == 0 && g.c_ld % 8 == 0;      // (wave-uniform)
        if (wide) { if constexpr (TN % 2 == 0) sweep16(); }
        else sweep(std::integral_constant<bool, true>{});
    } else sweep(std::integral_constant<bool, false>{});
}

// EPI (kernel template parameter, chosen on the host by packed_epi()): 0 = fp32-slab epilogue with every fused term as a run-time
// flag; 1..6 = packed epilogue: plain / + GroupNorm partials / + SiLU / + tanh-GELU / + bf16 residual / + residual and partials;
// 7 = the direct fp32 residual-stream epilogue; 8 = packed with row terms (a row bias: the V^T = W h^T + b GEMMs).  One epilogue per kernel: with both in one
// kernel behind a run-time branch hipcc spilled inside the packed register phase (measured: isolated GEMMs +15..23 %, the
// network 8 % SLOWER).
template <int WM, int WN, int TM, int TN, class Cfg, int EPI, int NSAMP = 1, bool FIN = false, bool PAIR = false>
__device__ __forceinline__ void tile_epilogue(const GemmArgs& g, unsigned char* smem, f32x4 (&acc)[TM][TN],
                                              int m0, int n0, int z, int tid, int lane, int wm, int wn)
{
    if constexpr (EPI == 0) dma_tile_epilogue<WM, WN, TM, TN, Cfg>(g, smem, acc, m0, n0, z, tid, lane, wm, wn);
    else if constexpr (EPI == 7) direct_f32_epilogue<WM, WN, TM, TN>(g, acc, m0, n0, z, lane, wm, wn);
    else {
        static_assert(Cfg::PACK_OK, "packed epilogue needs the whole bf16 tile in LDS");
        NATINF_TS(2);
        packed_tile_epilogue<WM, WN, TM, TN, Cfg, EPI == 3 ? ACT_SILU : (EPI == 4 ? ACT_GELU_TANH : ACT_NONE), EPI == 2 || EPI == 6, EPI == 5 || EPI == 6, EPI == 8, false, NSAMP, FIN && (EPI == 2 || EPI == 6), PAIR>(
            g, smem, acc, m0, n0, z, tid, lane, wm, wn);
    }
}

// ---- hand-counted LDS fragment pipeline (k_gemm_dma<..., 2>) ------------------------------------------------
// hipcc schedules "ds_read xN; s_waitcnt lgkmcnt(0); MFMA xM" batches: inside one wave LDS latency and the matrix
// pipe never overlap.  Here the fragment reads are inline-asm ds_read_b128 (invisible to hipcc's waitcnt pass) issued
// two steps ahead of their use, retired by COUNTED lgkmcnt waits, with sched_barrier(0) pinning each MFMA group
// behind its wait (guide 5.4 rule 18 / 5.7 form iii).  A step = one A row-tile x TN B tiles of one k-step.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int OFF> __device__ __forceinline__ u32x4 lds_read16(unsigned addr) {
    u32x4 v;
    asm volatile(NATINF_PAD_PRE "ds_read_b128 %0, %1 offset:%2" NATINF_PAD_POST : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int N> __device__ __forceinline__ void wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// reads still allowed in flight when step `st` starts its MFMAs.  Issue order: [B(ks0) x4, A0, A1], then per step s:
// A(s+2) if it exists, then -- during the last four steps of k-step 0 -- one B fragment of k-step 1.
constexpr int pipe_lgkm_after(int TM, int st) {
    int n = 6, pos_fa[40] = {4, 5}, pos_fb1_last = -1;
    for (int s = 0; s <= st; ++s) {
        if (s + 2 < 2 * TM) pos_fa[s + 2] = n++;
        if (s / TM == 0 && s % TM >= TM - 4) pos_fb1_last = n++;
    }
    int need = pos_fa[st];
    if (st >= TM && pos_fb1_last > need) need = pos_fb1_last;
    return n - 1 - need;
}
template <int TM, int ST, bool NOMFMA = false>       // NOMFMA: ablation (k_gemm_dma<..., 4>): the fragment reads and waits without the MFMAs
struct PipeStep {
    static __device__ __forceinline__ void run(u32x4 (&fa)[3], u32x4 (&fb)[2][4], f32x4 (&acc)[TM][4],
                                               unsigned a0, unsigned a1, unsigned b1) {
        constexpr int S = 2 * TM, ks = ST / TM, i = ST % TM;
        if constexpr (ST + 2 < S) {
            constexpr int ks2 = (ST + 2) / TM, i2 = (ST + 2) % TM;
            fa[(ST + 2) % 3] = lds_read16<i2 * 2048>(ks2 ? a1 : a0);
        }
        if constexpr (ks == 0 && i >= TM - 4) fb[1][i - (TM - 4)] = lds_read16<(i - (TM - 4)) * 2048>(b1);
        wait_lgkmcnt<pipe_lgkm_after(TM, ST)>();
        if constexpr (!NOMFMA) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[ks][j]),
                                                                    __builtin_bit_cast(bf16x8, fa[ST % 3]), acc[i][j], 0, 0, 0);
        } else {
            asm volatile("" :: "v"(fa[ST % 3]), "v"(fb[ks][0]), "v"(fb[ks][1]), "v"(fb[ks][2]), "v"(fb[ks][3]));     // keep the reads alive
        }
        if constexpr (ST + 1 < S) PipeStep<TM, ST + 1, NOMFMA>::run(fa, fb, acc, a0, a1, b1);
    }
};

// SPREAD = 1: the DMA requests of tile k+1 are not issued in one burst after the barrier (every wave of the block
// would then be issuing ~100-cycle LDS-DMA instructions at the same moment, with the matrix pipe idle) but one at
// a time between the MFMA groups of tile k.
template <int WM, int WN, int TM, int TN, int SPREAD = 0, int EPI = 0>
__global__ __launch_bounds__(WM * WN * 64, 2) void k_gemm_dma(const GemmArgs g)
{
    using Cfg = DmaCfg<WM, WN, TM, TN>;
    constexpr int BM_ = Cfg::BM_, BN_ = Cfg::BN_, THREADS = Cfg::THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_poison();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nN = (g.N + BN_ - 1) / BN_, nM = (g.M + BM_ - 1) / BM_;
    const int tile = xcd_remap(blockIdx.x, nM * nN);
    NATINF_TS(0);
    int mt_, nt_;
    tile_coords(tile, nM, nN, g.raster_g, mt_, nt_);
    const int m0 = mt_ * BM_, n0 = nt_ * BN_;
    const int z = blockIdx.z;

    const bf16* a0 = g.a0 + (int64_t)z * g.a_bs;
    const bf16* a1 = g.a1 ? g.a1 + (int64_t)z * g.a_bs : nullptr;
    const bf16* bp = g.b + (int64_t)z * g.b_bs;
    const int K0 = g.taps * g.a0_C, K1 = g.a1 ? g.a1_C : 0;
    const int nk0 = K0 / BK, nk = nk0 + K1 / BK;
    const int Wp = (1 << g.logW) + 2, Hp = (1 << (g.logHW - g.logW)) + 2;

    // SPREAD 6: only the first half of the waves (one per SIMD) issue LDS-DMA, two waves' worth each; the other wave of every
    // SIMD goes straight to its MFMAs, so the matrix pipe is not idle while the DMA burst is being issued.
    constexpr bool HALF = SPREAD == 6;
    constexpr int PAI = HALF ? 2 * Cfg::PA : Cfg::PA, PBI = HALF ? 2 * Cfg::PB : Cfg::PB;
    const bool issuer = !HALF || wave < Cfg::NW / 2;
    uint64_t a_row[PAI], a_delta[PAI], b_row[PBI];
    // the first K-tile's pieces are requested as soon as each address exists (kt = 0: tap (-1, -1) of chunk 0, or column 0)
    typedef __attribute__((address_space(3))) void lds_void;
    const int64_t ashift0 = g.taps == 9 ? (int64_t)(-Wp - 1) * g.a0_ld * 2 : 0;
#pragma unroll
    for (int j = 0; j < PAI; ++j) {
        const int r = (wave * PAI + j) * 8 + (lane >> 3);
        const int lchunk = ((lane & 7) ^ ((r >> 1) & 7)) << 3;
        const int m = min(m0 + r, g.M - 1);
        int64_t off0;
        if (g.taps == 9) {
            const int b = m >> g.logHW, p = m & ((1 << g.logHW) - 1), y = p >> g.logW, x = p & ((1 << g.logW) - 1);
            off0 = ((int64_t)(b * Hp + y + 1) * Wp + x + 1) * g.a0_ld;
        } else {
            off0 = (int64_t)m * g.a0_ld;
        }
        a_row[j] = reinterpret_cast<uint64_t>(a0 + off0 + lchunk);
        a_delta[j] = a1 ? reinterpret_cast<uint64_t>(a1 + (int64_t)m * g.a1_ld + lchunk) - a_row[j] : 0;
        if (issuer)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(a_row[j] + (uint64_t)ashift0),
                                             (lds_void*)(smem + wave * (PAI * 1024) + j * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < PBI; ++j) {
        const int r = (wave * PBI + j) * 8 + (lane >> 3);
        const int lchunk = ((lane & 7) ^ ((r >> 1) & 7)) << 3;
        b_row[j] = reinterpret_cast<uint64_t>(bp + (int64_t)min(n0 + r, g.N - 1) * g.b_ld + lchunk);
        if (issuer)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(b_row[j]), (lds_void*)(smem + BM_ * BK * 2 + wave * (PBI * 1024) + j * 1024), 16, 0, 0);
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // K-tile 0 (requested in the address loops above)
    __syncthreads();
    NATINF_TS(1);

    const int frow = lane & 15, fq = lane >> 4, fswz = (frow >> 1) & 7;
    constexpr int NP = PAI + PBI, SLOTS = 2 * TM, STEP = SLOTS / NP > 0 ? SLOTS / NP : 1;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        // tile-uniform part of the next tile's addresses (scalar registers)
        const int kn = kt + 1;
        const bool seg0n = kn < nk0;
        int64_t ashiftn; int kkn;
        if (seg0n) {
            int tap = 0, c0 = kn * BK;
            if (g.taps == 9) { const int cch = kn / 9; tap = kn - 9 * cch; c0 = cch * BK; }
            const int dy = g.taps == 9 ? tap / 3 - 1 : 0, dx = g.taps == 9 ? tap % 3 - 1 : 0;
            ashiftn = ((int64_t)(dy * Wp + dx) * g.a0_ld + c0) * 2; kkn = kn * BK;
        } else {
            ashiftn = (int64_t)(kn - nk0) * BK * 2; kkn = K0 + (kn - nk0) * BK;
        }
        unsigned char* dAn = smem + (cur ^ 1) * Cfg::STAGE_BYTES + wave * (PAI * 1024);
        unsigned char* dBn = smem + (cur ^ 1) * Cfg::STAGE_BYTES + BM_ * BK * 2 + wave * (PBI * 1024);
        auto issue_piece = [&](int p) __attribute__((always_inline)) {
            if (p < PAI) {
                const uint64_t pa = a_row[p] + (seg0n ? 0 : a_delta[p]) + (uint64_t)ashiftn;
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(pa), (lds_void*)(dAn + p * 1024), 16, 0, 0);
            } else {
                const uint64_t pb = b_row[p - PAI] + (uint64_t)kkn * 2;
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(pb), (lds_void*)(dBn + (p - PAI) * 1024), 16, 0, 0);
            }
        };
        if (SPREAD != 1 && SPREAD != 3 && more && issuer) {          // SPREAD 3 / 4: ablations of the hand-pipelined loop (no DMA after tile 0 / no MFMAs)
#pragma unroll
            for (int p = 0; p < NP; ++p) issue_piece(p);
        }
        const bf16* ta = reinterpret_cast<const bf16*>(smem + cur * Cfg::STAGE_BYTES) + (wm * TM * 16 + frow) * LDS_ROW;
        const bf16* tb = reinterpret_cast<const bf16*>(smem + cur * Cfg::STAGE_BYTES + BM_ * BK * 2) + (wn * TN * 16 + frow) * LDS_ROW;
        if constexpr (SPREAD < 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int ko = (((ks << 2) | fq) ^ fswz) << 3;
                bf16x8 fb[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(tb + j * 16 * LDS_ROW + ko);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const bf16x8 fa = *reinterpret_cast<const bf16x8*>(ta + i * 16 * LDS_ROW + ko);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa, acc[i][j], 0, 0, 0);
                    if (SPREAD == 1) {
                        const int slot = ks * TM + i;
                        if (more && slot % STEP == 0 && slot / STEP < NP) issue_piece(slot / STEP);
                    }
                }
            }
        } else {
            static_assert(SPREAD < 2 || TN == 4, "the hand-counted pipeline is written for TN = 4");
            typedef __attribute__((address_space(3))) unsigned char lds_u8;
            const unsigned a_lds = (unsigned)(uintptr_t)((lds_u8*)(unsigned char*)const_cast<bf16*>(ta));
            const unsigned b_lds = (unsigned)(uintptr_t)((lds_u8*)(unsigned char*)const_cast<bf16*>(tb));
            const unsigned ko0 = ((0 | fq) ^ fswz) << 4, ko1 = ((4 | fq) ^ fswz) << 4;      // bytes
            u32x4 fa[3], fb[2][4];
            fb[0][0] = lds_read16<0>(b_lds + ko0); fb[0][1] = lds_read16<2048>(b_lds + ko0);
            fb[0][2] = lds_read16<4096>(b_lds + ko0); fb[0][3] = lds_read16<6144>(b_lds + ko0);
            fa[0] = lds_read16<0>(a_lds + ko0); fa[1] = lds_read16<2048>(a_lds + ko0);
            PipeStep<TM, 0, SPREAD == 4>::run(fa, fb, acc, a_lds + ko0, a_lds + ko1, b_lds + ko1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile's LDS-DMA has landed (explicit: not left to the compiler's tracking)
        __syncthreads();
    }
    tile_epilogue<WM, WN, TM, TN, typename Cfg::Epi, EPI>(g, smem, acc, m0, n0, z, tid, lane, wm, wn);
}


// ------------------------------------------------------------------------------------------------
// k_gemm_ring<WM, WN, TM, TN, NS>: the same block geometry with a deeper pipeline.
//
// The 2-stage kernel keeps at most one K-tile of DMA in flight per CU, and LDS-DMA ingest is latency-bound:
// sustained bytes/s = bytes in flight / round-trip time.  Here the K loop advances in 32-wide tiles through a
// ring of NS LDS slots (64-byte rows); NS-1 tiles are requested ahead and a COUNTED `s_waitcnt vmcnt` retires
// only the oldest one, so NS-2 tiles stay in flight across every barrier (raw s_barrier, never
// __syncthreads(), whose vmcnt(0) would drain the queue).  One barrier per 32-wide tile.
//   per iteration kt:  wait(tile kt landed) -> barrier -> request tile kt+NS-1 into slot (kt-1)%NS -> MFMAs on kt
// The barrier both publishes tile kt (every wave waited for its own pieces first) and proves that every wave is
// done reading slot (kt-1)%NS (its ds_reads fed MFMAs issued before the barrier).
// LDS image: 64-byte rows, 16-byte chunk q of row r stored at chunk q ^ T[(r>>2)&3], T = {0,2,3,1}: conflict-free
// for ds_read_b128's lane groups (4 rows share a 256-byte bank line).
// ------------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN, int NS>
struct RingCfg {
    static constexpr int BKR = 32;
    static constexpr int NW = WM * WN, THREADS = NW * 64;
    static constexpr int BM_ = WM * TM * 16, BN_ = WN * TN * 16;
    static constexpr int STAGE_BYTES = (BM_ + BN_) * BKR * 2;
    static constexpr int PA = BM_ / 16 / NW, PB = BN_ / 16 / NW;          // 1-KiB pieces (16 rows x 64 B) per wave per tile
    static constexpr int DPT = PA + PB;                                    // DMA instructions per wave per tile
    using Epi = EpiCfg<WM, WN, TM, TN, NS * STAGE_BYTES + 4096>;
    static constexpr int LDS_BYTES = Epi::NEED > NS * STAGE_BYTES ? Epi::NEED : NS * STAGE_BYTES;
    static_assert(BM_ % (16 * NW) == 0 && BN_ % (16 * NW) == 0, "DMA pieces must divide evenly over the waves");
    static_assert((NS - 2) * DPT <= 63 && NS >= 3, "vmcnt immediate is 6 bits");
};

template <int N> __device__ __forceinline__ void wait_vmcnt_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(N) : "memory");
}

template <int WM, int WN, int TM, int TN, int NS, int EPI = 0>
__global__ __launch_bounds__(WM * WN * 64, 2) void k_gemm_ring(const GemmArgs g)
{
    using Cfg = RingCfg<WM, WN, TM, TN, NS>;
    constexpr int BM_ = Cfg::BM_, BN_ = Cfg::BN_, BKR = Cfg::BKR, ROW = 32;      // ROW: bf16 per LDS row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_poison();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nN = (g.N + BN_ - 1) / BN_, nM = (g.M + BM_ - 1) / BM_;
    const int tile = xcd_remap(blockIdx.x, nM * nN);
    int mt_, nt_;
    tile_coords(tile, nM, nN, g.raster_g, mt_, nt_);
    const int m0 = mt_ * BM_, n0 = nt_ * BN_;
    const int z = EPI == 9 ? 0 : blockIdx.z;

    const bf16* a0 = g.a0 + (int64_t)z * g.a_bs;
    const bf16* a1 = g.a1 ? g.a1 + (int64_t)z * g.a_bs : nullptr;
    const bf16* bp = g.b + (int64_t)z * g.b_bs;
    const int K0 = g.taps * g.a0_C, K1 = g.a1 ? g.a1_C : 0;
    const int nk0 = K0 / BKR, nk_all = nk0 + K1 / BKR;
    const int Wp = (1 << g.logW) + 2, Hp = (1 << (g.logHW - g.logW)) + 2;
    // EPI 9 = split-K: blockIdx.z is the K slice (batch must be 1), this block multiplies K-tiles [kt_lo, kt_lo + nk) and writes its
    // fp32 partial tile to g.c + slice * M * N; k_splitk_reduce sums the slices and applies the fused terms
    const int kt_lo = EPI == 9 ? (int)((int64_t)nk_all * blockIdx.z / g.splitk) : 0;
    const int nk = EPI == 9 ? (int)((int64_t)nk_all * (blockIdx.z + 1) / g.splitk) - kt_lo : nk_all;

    uint64_t a_row[Cfg::PA], a_delta[Cfg::PA], b_row[Cfg::PB];
#pragma unroll
    for (int j = 0; j < Cfg::PA; ++j) {
        const int r = (wave * Cfg::PA + j) * 16 + (lane >> 2);
        const int t4 = (r >> 2) & 3;
        const int lchunk = ((lane & 3) ^ ((0x1320 >> (t4 * 4)) & 3)) << 3;          // T = {0,2,3,1}
        const int m = min(m0 + r, g.M - 1);
        int64_t off0;
        if (g.taps == 9) {
            const int b = m >> g.logHW, p = m & ((1 << g.logHW) - 1), y = p >> g.logW, x = p & ((1 << g.logW) - 1);
            off0 = ((int64_t)(b * Hp + y + 1) * Wp + x + 1) * g.a0_ld;
        } else {
            off0 = (int64_t)m * g.a0_ld;
        }
        a_row[j] = reinterpret_cast<uint64_t>(a0 + off0 + lchunk);
        a_delta[j] = a1 ? reinterpret_cast<uint64_t>(a1 + (int64_t)m * g.a1_ld + lchunk) - a_row[j] : 0;
    }
#pragma unroll
    for (int j = 0; j < Cfg::PB; ++j) {
        const int r = (wave * Cfg::PB + j) * 16 + (lane >> 2);
        const int t4 = (r >> 2) & 3;
        const int lchunk = ((lane & 3) ^ ((0x1320 >> (t4 * 4)) & 3)) << 3;
        b_row[j] = reinterpret_cast<uint64_t>(bp + (int64_t)min(n0 + r, g.N - 1) * g.b_ld + lchunk);
    }

    typedef __attribute__((address_space(3))) void lds_void;
    auto issue_tile = [&](int kt_rel) __attribute__((always_inline)) {     // kt_rel: index inside this block's K range
        const int slot = kt_rel % NS, kt = kt_lo + kt_rel;
        unsigned char* dA = smem + slot * Cfg::STAGE_BYTES + wave * (Cfg::PA * 1024);
        unsigned char* dB = smem + slot * Cfg::STAGE_BYTES + BM_ * BKR * 2 + wave * (Cfg::PB * 1024);
        const bool seg0 = kt < nk0;
        int64_t ashift;
        if (seg0) {
            int tap = 0, c0 = kt * BKR;
            if (g.taps == 9) { const int q = kt >> 1, cch = q / 9; tap = q - 9 * cch; c0 = cch * 64 + (kt & 1) * BKR; }
            const int dy = g.taps == 9 ? tap / 3 - 1 : 0, dx = g.taps == 9 ? tap % 3 - 1 : 0;
            ashift = ((int64_t)(dy * Wp + dx) * g.a0_ld + c0) * 2;
        } else {
            ashift = (int64_t)(kt - nk0) * BKR * 2;
        }
        const uint64_t kk2 = (uint64_t)kt * BKR * 2;           // B columns are packed in K-loop order
#pragma unroll
        for (int j = 0; j < Cfg::PA; ++j) {
            const uint64_t pa = a_row[j] + (seg0 ? 0 : a_delta[j]) + (uint64_t)ashift;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(pa), (lds_void*)(dA + j * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < Cfg::PB; ++j)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(b_row[j] + kk2), (lds_void*)(dB + j * 1024), 16, 0, 0);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < nk) issue_tile(t);

    const int frow = lane & 15, fq = lane >> 4;
    const int fko = (fq ^ ((0x1320 >> (((frow >> 2) & 3) * 4)) & 3)) << 3;       // swizzled chunk of this lane's 8 k-values
    for (int kt = 0; kt < nk; ++kt) {
        // tiles requested so far: 0 .. min(nk, kt+NS-1)-1; retire tile kt, keep the younger ones in flight
        if (kt + NS - 1 <= nk) wait_vmcnt_barrier<(NS - 2) * Cfg::DPT>();
        else                   wait_vmcnt_barrier<0>();
        if (kt + NS - 1 < nk) issue_tile(kt + NS - 1);
        const int slot = kt % NS;
        const bf16* ta = reinterpret_cast<const bf16*>(smem + slot * Cfg::STAGE_BYTES) + (wm * TM * 16 + frow) * ROW + fko;
        const bf16* tb = reinterpret_cast<const bf16*>(smem + slot * Cfg::STAGE_BYTES + BM_ * BKR * 2) + (wn * TN * 16 + frow) * ROW + fko;
        bf16x8 fb[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(tb + j * 16 * ROW);
#if NATINF_SETPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const bf16x8 fa = *reinterpret_cast<const bf16x8*>(ta + i * 16 * ROW);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa, acc[i][j], 0, 0, 0);
        }
#if NATINF_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    }
    __syncthreads();               // every wave is done with the ring before the epilogue reuses it
    if constexpr (EPI == 9) {
        // split-K partial: the accumulators as they are (swapped-operand layout: a lane holds four consecutive columns of a row)
        float* part = reinterpret_cast<float*>(g.c) + (int64_t)blockIdx.z * g.M * g.N;
        const int r = lane & 15, q = lane >> 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * TM * 16 + i * 16 + r;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * TN * 16 + j * 16 + q * 4;
                if (m < g.M && n < g.N) *reinterpret_cast<f32x4*>(part + (int64_t)m * g.N + n) = acc[i][j];
            }
        }
    } else {
        tile_epilogue<WM, WN, TM, TN, typename Cfg::Epi, EPI>(g, smem, acc, m0, n0, z, tid, lane, wm, wn);
    }
}

// Second pass of a split-K GEMM: out = act((sum over slices + bias_n + rowvec[sample] + resid) * scale), bf16.  A block owns 16 rows
// (SPLITK_ROWS) x all N columns, a thread 8 columns of 16 * (N/8) / 256 rows; slices are summed in ascending order (deterministic).
// gn_part (optional): GroupNorm partial sums of the fp32 results, float2 [row tile of 16][N/4] like the GEMM epilogues write them.
constexpr int SPLITK_ROWS = 16;
__global__ __launch_bounds__(256) void k_splitk_reduce(const float* __restrict__ part, int S, int M, int N, const float* __restrict__ bias_n,
                                                       const float* __restrict__ rowvec, int rowvec_ld, int log_rows_per_sample,
                                                       const bf16* __restrict__ resid, int resid_ld, float scale, int act,
                                                       bf16* __restrict__ c, int c_ld, float2* __restrict__ gn_part, int gn_quads)
{
    __shared__ float2 red[512];                              // [row lane][quad], 256 / cpr row lanes x 2 * cpr quads = 512 entries
    lds_poison();
    const int cpr = N >> 3, tid = threadIdx.x;               // host guarantees 256 % cpr == 0
    const int cx = tid % cpr, ry = tid / cpr, rp = 256 / cpr, n = cx * 8;
    float bn[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) bn[q] = bias_n ? bias_n[n + q] : 0.f;
    float gs[2] = {0.f, 0.f}, gq[2] = {0.f, 0.f};
    for (int r = ry; r < SPLITK_ROWS; r += rp) {
        const int m = blockIdx.x * SPLITK_ROWS + r;
        if (m >= M) break;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int sl = 0; sl < S; ++sl) {
            const float* p = part + ((int64_t)sl * M + m) * N + n;
            const float4 u = *reinterpret_cast<const float4*>(p), w = *reinterpret_cast<const float4*>(p + 4);
            v[0] += u.x; v[1] += u.y; v[2] += u.z; v[3] += u.w; v[4] += w.x; v[5] += w.y; v[6] += w.z; v[7] += w.w;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += bn[q];
        if (rowvec) {
            const float* rv = rowvec + (int64_t)(m >> log_rows_per_sample) * rowvec_ld + n;
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += rv[q];
        }
        if (resid) {
            const bf16x8 rs = *reinterpret_cast<const bf16x8*>(resid + (int64_t)m * resid_ld + n);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += (float)rs[q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] *= scale;
        apply_act8(v, act);
#pragma unroll
        for (int q = 0; q < 8; ++q) { gs[q >> 2] += v[q]; gq[q >> 2] += v[q] * v[q]; }
        bf16x8 o;
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = (bf16)v[q];
        *reinterpret_cast<bf16x8*>(c + (int64_t)m * c_ld + n) = o;
    }
    if (gn_part) {
        red[ry * 2 * cpr + cx * 2] = make_float2(gs[0], gq[0]);
        red[ry * 2 * cpr + cx * 2 + 1] = make_float2(gs[1], gq[1]);
        __syncthreads();
        if (tid < 2 * cpr) {
            float a = 0.f, b = 0.f;
            for (int y = 0; y < rp; ++y) { const float2 t = red[y * 2 * cpr + tid]; a += t.x; b += t.y; }
            gn_part[(int64_t)blockIdx.x * gn_quads + tid] = make_float2(a, b);
        }
    }
}

}  // namespace ncsn

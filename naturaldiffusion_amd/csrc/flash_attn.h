// flash_attn.h -- softmax attention over long (joint) sequences, head_dim 64: the SD3 MMDiT attention.
//
// One block = 128 queries of one (sequence, head): 4 waves x 32 queries.  Keys are walked in tiles of 128; the K tile
// [128][64] and the V^T tile [64][128] of step kt+1 stream into LDS by direct global->LDS DMA while step kt is being
// multiplied (two buffers, one barrier per tile).  S and P never leave the registers (same operand trick as
// attn_fused.h): S^T = K Q^T leaves a lane with four keys of its query column per 16-key tile, and O^T = V^T P^T wants,
// per lane, eight consecutive keys of that query -- so the A-operand ROWS of S^T tile (2c+h) are taken from the K tile
// in the order  key = 32c + 8(r>>2) + 4h + (r&3),  which makes the lane's values of tiles 2c and 2c+1 exactly keys
// 32c+8q .. 32c+8q+7, the 16 bytes its V^T fragment read covers.
//   online softmax: running max m and (per-lane partial) sum l per query; exp2 with scale*log2(e) folded into one fma;
//   the O^T accumulators are rescaled by exp2((m_old-m_new)c) once per key tile.
//   LDS: 128-byte rows, 16-byte chunk index XOR-ed with a 3-bit row hash on the DMA source side and on the fragment
//   read: K uses bits (1,3,4) of the row (its rows are read in the permuted order above), V^T bits (1,2,3).
//   XCD-aware block order: the 35 query blocks of one (sequence, head) run back to back on one XCD, so its K / V^T
//   (1.1 MB at 4,429 tokens) are fetched from HBM once and re-read from that XCD's L2.
// Keys >= T_total (the padding up to a multiple of 128) are masked in the last tile; V^T must be finite there.
// Measured alternatives that did NOT pay (tools/bench_flash.py, same-box A/B; 8 x 24 heads x 4,429 tokens, 1.35-1.45 ms):
// Q pre-scaled + accumulators started at -m + lazy re-referencing (no per-element fma): -1 % in the engine, +7 % alone;
// 8-wave blocks sharing a K/V tile (half the DMA requests per wave): +-0; v_permlane16/32_swap instead of ds_bpermute for
// the 4-lane maxima: +-0; tree instead of chain reductions: slower.  Ablation of the loop: without the softmax block
// 0.90 ms, without P V 1.13 ms, S^T alone 0.47 ms (1.0 PFLOP/s), without the exp only: unchanged -- the serial
// S -> softmax -> PV order inside a wave is the cost, not any single instruction class.  Also without effect (+-2 %):
// 64 queries per wave with 64-key tiles (half the LDS fragment bytes per MFMA), 64-key tiles at 3 and 4 waves per SIMD
// (130 / 128 VGPRs).  What is common to all of them is the instruction count per tile -- ~250 VALU + 66 v_exp beside 64
// MFMAs per wave -- so the open steps are fewer VALU instructions per score (row sums through an all-ones V^T row on the
// matrix pipe, max over packed halves) and overlapping tile t's softmax with tile t+1's S^T (two S register sets).
// Built and measured after that (same-box A/B, bit-identical output): the loop software-pipelined INSIDE a wave -- S^T(kt+1)
// issued in eight groups of four MFMAs in front of the slices of softmax(kt), two S register sets, 253 VGPRs, no spills -- is
// 3-5 % SLOWER (1,228 vs 1,190 us).  The SIMD's vector issue port is the resource, not the overlap: per key tile and wave
// 64 MFMAs x 8 issue cycles + 66 v_exp x 8 + ~250 VALU x 4 = ~2.0k cycles of issue against the measured 2.6k, shared by the
// two waves of a SIMD, so moving work between the pipes buys nothing.  `-fno-slp-vectorize` (the compiler packs the row sums
// and the rescale into v_pk_add/mul_f32) is +-1 %.  What is left is fewer issue cycles per score: 32x32x16 MFMAs (half the
// MFMA issue cost).  Row sums on the matrix pipe (an all-ones A operand in front of P^T: 8 MFMAs instead of 64 adds per tile and
// lane, no final cross-lane reduction) were built too: +-1 %, not kept.  The next tile's DMA requests issued behind the S^T MFMAs
// instead of in front of them: 3 % slower.
// Round 2: that was built -- the whole kernel on v_mfma_f32_32x32x16_bf16 (S^T tiles of 32 keys x 32 queries with the K rows taken in the order
// i -> i with bits 2, 3 exchanged, so that accumulator elements 8j .. 8j+7 of a lane are the eight consecutive keys of its P^T operand for PV
// step j; one query per lane, one cross-lane step per row maximum; same LDS images and swizzles, both conflict-free for 32-row fragments;
// 168 VGPRs, bit-level agreement with this kernel to 5e-4) -- and measured in a same-box A/B at the SD3 shape: 1,169 us against 1,169 us.
// tools/probes/mfma_issue_probe.hip explains it: a 32x32x16 MFMA holds the SIMD's vector issue for ~11-12 cycles, not 8 (two waves per
// SIMD: 5 plain vector instructions per MFMA are free, the 6th is not), so the issue cost per flop only drops to 0.7x, ~280 of the ~4,100
// issue cycles a pair of waves spends per key tile (2 x (64 MFMAs x 8 + 250 VALU x 4 + 66 v_exp x 8) against 4,800 measured): inside the
// run-to-run spread.  The vector instructions per score are the cost; the kernel sits at 0.34 of the bf16 peak with an issue-bound ceiling
// of ~0.40 for this instruction mix.  Not kept.
// Reference: diffusers JointAttnProcessor2_0 as called by pipe.transformer (src/SD3NaturalInference.py:210-213).
#pragma once
#include "ncsnpp_kernels.h"

namespace ncsn {

constexpr int FA_KT = 128, FA_QB = 128, FA_STAGE = 32768, FA_LDS_BYTES = 2 * FA_STAGE;

struct FlashArgs {
    const bf16* q; const bf16* k; int ld_qk; int64_t qk_bs;        // [B][Tp][ld_qk], head h at column 64h
    const bf16* vT; int64_t vT_bs;                                 // [B][H*64][Tp]
    bf16* o; int ld_o; int64_t o_bs;                               // [B][Tp][ld_o]
    int H, Tp, Ttot; float c1;                                     // c1 = softmax scale * log2(e)
    uint8_t* o8; uint8_t* omx;                                     // optional fp8 output instead of o: e4m3 [B][Tp][64H] + E8M0 block scales, K-tile major per sequence ([H/2][Tp][4])
};

__global__ __launch_bounds__(256, 2) void k_flash_attn64(const FlashArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_poison();
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nQ = a.Tp / FA_QB, nK = a.Tp / FA_KT;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = tile / nQ, qb = tile - bh * nQ, b = bh / a.H, head = bh - b * a.H;
    const bf16* qbase = a.q + (int64_t)b * a.qk_bs + head * 64;
    const bf16* kbase = a.k + (int64_t)b * a.qk_bs + head * 64;
    const bf16* vbase = a.vT + (int64_t)b * a.vT_bs + (int64_t)head * 64 * a.Tp;

    // DMA: per wave and stage 4 K pieces + 4 V^T pieces of 1 KiB (8 rows x 128 B)
    const int prow = lane >> 3, pch = lane & 7;
    auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
        unsigned char* st = smem + buf * FA_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = wave * 4 + j, row = 8 * p + prow;
            const int f = ((row >> 1) & 1) | (((row >> 3) & 3) << 1);
            __builtin_amdgcn_global_load_lds(kbase + (int64_t)(kt * FA_KT + row) * a.ld_qk + ((pch ^ f) << 3), (lds_void*)(st + p * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = wave * 4 + j, sub = p >> 3, d = 8 * (p & 7) + prow;
            __builtin_amdgcn_global_load_lds(vbase + (int64_t)d * a.Tp + kt * FA_KT + sub * 64 + ((pch ^ ((d >> 1) & 7)) << 3),
                                             (lds_void*)(st + 16384 + p * 1024), 16, 0, 0);
        }
    };

    bf16x8 qf[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[g][ks] = *reinterpret_cast<const bf16x8*>(qbase + (int64_t)(qb * FA_QB + wave * 32 + 16 * g + r) * a.ld_qk + 32 * ks + 8 * q);

    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
    f32x4 oacc[2][4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { oacc[0][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; oacc[1][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int krow = 8 * (r >> 2) + (r & 3), swz = r >> 1;
    issue(0, 0);
    for (int kt = 0; kt < nK; ++kt) {
        // tile kt landed: the compiler does not track LDS-DMA completions across the loop back-edge, so wait explicitly;
        // the barrier then also proves every wave is done with tile kt-1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nK) issue(kt + 1, (kt + 1) & 1);
        const unsigned char* sK = smem + (kt & 1) * FA_STAGE;
        const unsigned char* sV = sK + 16384;

        f32x4 acc[2][8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { acc[0][t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int row = 32 * (t >> 1) + 4 * (t & 1) + krow;
                const bf16x8 fa = *reinterpret_cast<const bf16x8*>(sK + row * 128 + (((4 * ks + q) ^ swz) << 4));
                acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, qf[0][ks], acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, qf[1][ks], acc[1][t], 0, 0, 0);
            }
        if (kt == nK - 1 && a.Ttot < a.Tp) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = kt * FA_KT + 32 * (t >> 1) + 8 * q + 4 * (t & 1) + i < a.Ttot;
                    acc[0][t][i] = ok ? acc[0][t][i] : -INFINITY;
                    acc[1][t][i] = ok ? acc[1][t][i] : -INFINITY;
                }
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float mx = m[g];
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) mx = fmaxf(mx, acc[g][t][i]);
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float alpha = __builtin_amdgcn_exp2f((m[g] - mx) * a.c1), mc = mx * a.c1;
            m[g] = mx;
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float p = __builtin_amdgcn_exp2f(fmaf(acc[g][t][i], a.c1, -mc)); acc[g][t][i] = p; s += p; }
            l[g] = l[g] * alpha + s;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int i = 0; i < 4; ++i) oacc[g][dt][i] *= alpha;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bf16x8 pf[2];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) { pf[g][i] = (bf16)acc[g][2 * c][i]; pf[g][4 + i] = (bf16)acc[g][2 * c + 1][i]; }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 fv = *reinterpret_cast<const bf16x8*>(sV + (c >> 1) * 8192 + (16 * dt + r) * 128 + (((4 * (c & 1) + q) ^ swz) << 4));
                oacc[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, pf[0], oacc[0][dt], 0, 0, 0);
                oacc[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, pf[1], oacc[1][dt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        float s = l[g];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const float inv = 1.0f / s;
        const int64_t qrow = (int64_t)b * a.Tp + qb * FA_QB + wave * 32 + 16 * g + r;
        if (a.o8) {
            // fp8 output for the fp8 output projection: one MX block = 32 head channels of a query = two d-tiles x the four
            // lanes of the query column
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                float amax = 0.f;
#pragma unroll
                for (int dt = 2 * blk; dt < 2 * blk + 2; ++dt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) amax = fmaxf(amax, fabsf(oacc[g][dt][i] * inv));
                amax = group4_max_nonneg(amax);
                float qinv;
                const unsigned e8 = mx_scale_of(amax, qinv);
                qinv *= inv;
#pragma unroll
                for (int dt = 2 * blk; dt < 2 * blk + 2; ++dt)
                    *reinterpret_cast<unsigned*>(a.o8 + qrow * (a.H * 64) + head * 64 + 16 * dt + 4 * q) =
                        pack_fp8x4(oacc[g][dt][0] * qinv, oacc[g][dt][1] * qinv, oacc[g][dt][2] * qinv, oacc[g][dt][3] * qinv);
                if (q == 0)                           // block index kb = 2 head + blk of this row; plane of a sequence: [H/2][Tp][4]
                    a.omx[(int64_t)b * a.Tp * (a.H * 2) + ((int64_t)(head >> 1) * a.Tp + (qrow - (int64_t)b * a.Tp)) * 4 + (head & 1) * 2 + blk] = (uint8_t)e8;
            }
            continue;
        }
        bf16* orow = a.o + (int64_t)b * a.o_bs + (int64_t)(qb * FA_QB + wave * 32 + 16 * g + r) * a.ld_o + head * 64 + 4 * q;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            bf16x4 w;
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = (bf16)(oacc[g][dt][i] * inv);
            *reinterpret_cast<bf16x4*>(orow + 16 * dt) = w;
        }
    }
}

// ---- Round 4: k_flash_attn64_v2 -- fewer vector instructions per score, and no cross-lane chain on the critical path ----------------------
// Per key tile and wave the kernel above issues 64 MFMAs next to ~215 plain vector instructions and 66 v_exp_f32 (ISA census of the built loop: 64
// v_fma for scale * s - scale * m, 32 v_max3, 32 v_cvt_pk, ~41 adds for the row sums, 16 v_pk_mul for the O rescale) and, before the first exp2 of a
// tile can issue, walks max -> ds_bpermute -> max -> ds_bpermute -> max -> exp2: a serial cross-lane chain through the LDS pipe, twice per tile.
// Here:
//   * Q is multiplied by scale * log2(e) ONCE when it is loaded (one more bf16 rounding of q), and the score accumulators START at -m_ref (the MFMA's
//     C operand: a register quadruple per query group that only changes when the reference does) -- S^T leaves the matrix pipe as the exponent
//     itself: no per-score fma;
//   * the reference m_ref is the running maximum as of the last RE-REFERENCING, not of every tile: a tile re-references (cross-lane maximum, exp2,
//     rescale of O and l, subtract from the tile's scores) only when some score of the wave exceeds its reference by more than FA_THR (2^8: P stays
//     <= 256, far inside bf16 / fp32 range; softmax is invariant to the reference, so this is exact up to rounding) -- a wave-uniform branch taken in
//     the first tile and then a handful of times per query block on real score distributions; the common path has NO cross-lane operation;
//   * SUMS = 1: the row sums ride on the matrix pipe -- an all-ones A operand in front of P^T (8 more MFMAs per tile instead of ~41 vector adds;
//     the sum then uses the bf16-rounded P the numerator uses, and arrives complete in every lane: no final cross-lane reduction).
// Common path per tile and wave: 64 (+8) MFMAs, 64 v_exp_f32, 32 v_max3, 32 v_cvt_pk, a compare.  tests/test_gpu_mmdit.py forces the branch at a
// chosen tile (spiked key) and checks both kernels against an fp64 softmax.
constexpr float FA_THR = 8.0f;

template <int SUMS, int KT = FA_KT>      // KT = 64: 16-KiB stages, 136 registers -- three blocks per CU (three waves per SIMD) instead of two
__global__ __launch_bounds__(256, KT == 64 ? 3 : 2) void k_flash_attn64_v2(const FlashArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_poison();
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NT = KT / 16, NC = KT / 32, NPW = KT / 32, STAGE = 256 * KT;      // score tiles, PV steps, DMA pieces per wave and operand, bytes per stage
    const int nQ = a.Tp / FA_QB, nK = a.Tp / KT;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = tile / nQ, qb = tile - bh * nQ, b = bh / a.H, head = bh - b * a.H;
    const bf16* qbase = a.q + (int64_t)b * a.qk_bs + head * 64;
    const bf16* kbase = a.k + (int64_t)b * a.qk_bs + head * 64;
    const bf16* vbase = a.vT + (int64_t)b * a.vT_bs + (int64_t)head * 64 * a.Tp;

    // DMA requests: per wave and stage 4 K pieces + 4 V^T pieces of 1 KiB.  The per-lane part of a source address does not depend on the key tile,
    // so it is a 32-bit offset computed ONCE (eight registers) next to a scalar base that moves by a constant per tile: the builtin form with a 64-bit
    // per-lane pointer spent ~30 vector instructions per tile and wave on address arithmetic -- in a loop whose bound is the vector issue port.
    const int prow = lane >> 3, pch = lane & 7;
    unsigned koff[NPW], voff[NPW];
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        const int p = wave * NPW + j, row = 8 * p + prow;
        const int f = ((row >> 1) & 1) | (((row >> 3) & 3) << 1);
        koff[j] = (unsigned)(row * a.ld_qk + ((pch ^ f) << 3)) * 2u;
        const int sub = p >> 3, d = 8 * (p & 7) + prow;
        voff[j] = (unsigned)(d * a.Tp + sub * 64 + ((pch ^ ((d >> 1) & 7)) << 3)) * 2u;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
    auto glds = [](unsigned vo, const void* sbase, unsigned dst) __attribute__((always_inline)) {
        NATINF_M0_ASM_BEGIN
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(vo), "s"(sbase), "s"(dst) : "memory", "m0");
        NATINF_M0_ASM_END
    };
    auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
        const unsigned st = lds0 + buf * STAGE + wave * (NPW * 1024);
        const bf16* kb = kbase + (int64_t)kt * KT * a.ld_qk;
        const bf16* vb = vbase + kt * KT;
#pragma unroll
        for (int j = 0; j < NPW; ++j) glds(koff[j], kb, st + j * 1024);
#pragma unroll
        for (int j = 0; j < NPW; ++j) glds(voff[j], vb, st + KT * 128 + j * 1024);
    };
    issue(0, 0);

    bf16x8 qf[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 raw = *reinterpret_cast<const bf16x8*>(qbase + (int64_t)(qb * FA_QB + wave * 32 + 16 * g + r) * a.ld_qk + 32 * ks + 8 * q);
#pragma unroll
            for (int i = 0; i < 8; ++i) qf[g][ks][i] = (bf16)((float)raw[i] * a.c1);
        }

    float mref[2] = {0.f, 0.f}, l[2] = {0.f, 0.f};
    f32x4 cinit[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};      // -m_ref: the C operand of every score tile's first MFMA
    f32x4 lsum[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 oacc[2][4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { oacc[0][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; oacc[1][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16)1.0f;

    const int krow = 8 * (r >> 2) + (r & 3), swz = r >> 1;
    for (int kt = 0; kt < nK; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nK) issue(kt + 1, (kt + 1) & 1);
        const unsigned char* sK = smem + (kt & 1) * STAGE;
        const unsigned char* sV = sK + KT * 128;

        f32x4 acc[2][NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int row = 32 * (t >> 1) + 4 * (t & 1) + krow;
            const bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(sK + row * 128 + (((0 + q) ^ swz) << 4));
            const bf16x8 fa1 = *reinterpret_cast<const bf16x8*>(sK + row * 128 + (((4 + q) ^ swz) << 4));
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa0, qf[0][0], cinit[0], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa0, qf[1][0], cinit[1], 0, 0, 0);
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa1, qf[0][1], acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa1, qf[1][1], acc[1][t], 0, 0, 0);
        }
        if ((kt + 1) * KT > a.Ttot) {                                          // (the padding -- up to 127 keys -- can reach into the last TWO 64-key tiles)
            // (the limit is made opaque HERE: as a loop invariant hipcc evaluates the 32 comparisons in front of the loop and keeps 64 scalar registers of
            // masks alive across it -- 27 of them spilled into vector-register lanes)
            int lim = a.Ttot - kt * KT - 8 * q;
            asm volatile("" : "+v"(lim));
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = 32 * (t >> 1) + 4 * (t & 1) + i < lim;
                    acc[0][t][i] = ok ? acc[0][t][i] : -INFINITY;
                    acc[1][t][i] = ok ? acc[1][t][i] : -INFINITY;
                }
        }
        // the lane's largest exponent of each query group (relative to the group's reference)
        float lm[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float mx = acc[g][0][0];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) mx = fmaxf(mx, acc[g][t][i]);
            lm[g] = mx;
        }
        if (kt == 0 || __any(fmaxf(lm[0], lm[1]) > FA_THR)) {
            // re-reference: the FIRST tile takes the true column maximum (whatever its sign); later tiles only raise the reference
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float mx = lm[g];
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float d = kt == 0 ? mx : fmaxf(mx, 0.f);
                mref[g] += d;
                cinit[g] = f32x4{-mref[g], -mref[g], -mref[g], -mref[g]};
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[g][t][i] -= d;
                if (kt > 0) {                                                  // (first tile: O = l = 0, and exp2(-d) may overflow for a negative maximum)
                    const float alpha = __builtin_amdgcn_exp2f(-d);
                    l[g] *= alpha;
#pragma unroll
                    for (int i = 0; i < 4; ++i) lsum[g][i] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                        for (int i = 0; i < 4; ++i) oacc[g][dt][i] *= alpha;
                }
            }
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p = __builtin_amdgcn_exp2f(acc[g][t][i]);
                    acc[g][t][i] = p;
                    if constexpr (!SUMS) s += p;
                }
            if constexpr (!SUMS) l[g] += s;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            bf16x8 pf[2];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) { pf[g][i] = (bf16)acc[g][2 * c][i]; pf[g][4 + i] = (bf16)acc[g][2 * c + 1][i]; }
            if constexpr (SUMS) {
                lsum[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[0], lsum[0], 0, 0, 0);
                lsum[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[1], lsum[1], 0, 0, 0);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 fv = *reinterpret_cast<const bf16x8*>(sV + (c >> 1) * 8192 + (16 * dt + r) * 128 + (((4 * (c & 1) + q) ^ swz) << 4));
                oacc[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, pf[0], oacc[0][dt], 0, 0, 0);
                oacc[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, pf[1], oacc[1][dt], 0, 0, 0);
            }
        }
    }
    // the epilogue's arguments are fetched from the kernel-argument segment HERE: kept in scalar registers across the key loop they are what hipcc
    // spills into vector-register lanes (32 of them, read back with v_readlane inside the loop)
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const unsigned __attribute__((address_space(4))) *kernarg_u32_t;
    kernarg_u32_t gp = (kernarg_u32_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(gp));
    FlashArgs e;
    {
        unsigned* dw = reinterpret_cast<unsigned*>(&e);
#pragma unroll
        for (unsigned i = 0; i < sizeof(FlashArgs) / 4; ++i) dw[i] = gp[i];
    }
#else
    const FlashArgs e = a;
#endif
    const int tile_e = xcd_remap(blockIdx.x, gridDim.x), nQ_e = e.Tp / FA_QB;
    const int bh_e = tile_e / nQ_e, qb_e = tile_e - bh_e * nQ_e, b_e = bh_e / e.H, head_e = bh_e - b_e * e.H;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        float s;
        if constexpr (SUMS) s = lsum[g][0];                                   // every row of the all-ones product is the complete row sum
        else {
            s = l[g];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
        }
        const float inv = 1.0f / s;
        const int64_t qrow = (int64_t)b_e * e.Tp + qb_e * FA_QB + wave * 32 + 16 * g + r;
        if (e.o8) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                float amax = 0.f;
#pragma unroll
                for (int dt = 2 * blk; dt < 2 * blk + 2; ++dt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) amax = fmaxf(amax, fabsf(oacc[g][dt][i] * inv));
                amax = group4_max_nonneg(amax);
                float qinv;
                const unsigned e8 = mx_scale_of(amax, qinv);
                qinv *= inv;
#pragma unroll
                for (int dt = 2 * blk; dt < 2 * blk + 2; ++dt)
                    *reinterpret_cast<unsigned*>(e.o8 + qrow * (e.H * 64) + head_e * 64 + 16 * dt + 4 * q) =
                        pack_fp8x4(oacc[g][dt][0] * qinv, oacc[g][dt][1] * qinv, oacc[g][dt][2] * qinv, oacc[g][dt][3] * qinv);
                if (q == 0)
                    e.omx[(int64_t)b_e * e.Tp * (e.H * 2) + ((int64_t)(head_e >> 1) * e.Tp + (qrow - (int64_t)b_e * e.Tp)) * 4 + (head_e & 1) * 2 + blk] = (uint8_t)e8;
            }
            continue;
        }
        bf16* orow = e.o + (int64_t)b_e * e.o_bs + (int64_t)(qb_e * FA_QB + wave * 32 + 16 * g + r) * e.ld_o + head_e * 64 + 4 * q;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            bf16x4 w;
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = (bf16)(oacc[g][dt][i] * inv);
            *reinterpret_cast<bf16x4*>(orow + 16 * dt) = w;
        }
    }
}

// Also measured on the 64-key form and not kept (one process, three interleaved rounds, 885-891 us every one of them): `s_setprio 1` around the two MFMA clusters;
// each PV step's 32 exponentials pinned in front of its own MFMAs instead of all 64 ahead of the first step; both together.
// Built on top of that and NOT kept (round 4; correct -- it passed every parity case of tests/test_gpu_mmdit.py incl. the forced re-referencing):
// k_flash_attn64_pp, the two waves of a SIMD in OPPOSITE phases.  The counters of v2 (profiles/r04/flash_pmc_mode2.json) read as if matrix time
// (1,152 pipe cycles per wave-tile) and vector time (~800) ADD UP on a SIMD (2,160 cycles per wave-tile) instead of overlapping -- the two co-resident
// waves belong to two blocks that run the same program from the same start.  So: 8-wave blocks of 256 queries, waves w and w + 4 (one SIMD) in two
// groups one SEGMENT apart -- matrix segment = P(i-1) V(i-1) + S(i), vector segment = softmax(i) + the group's DMA requests for tile i + 2 (K by one
// group, V^T by the other; rings of three tiles, counted vmcnt(4)), one block-wide s_barrier between segments.  1,007-1,013 us against 966-974 for v2 on
// the same box (even / odd wave groups instead: 1,134 us, so the w / w + 4 pairing is the right one); SQ_WAIT_ANY 30 % of wave cycles against 18 %, matrix
// pipe 49 % busy against 53 %.  The reading that fits both: an MFMA occupies the SIMD's vector ISSUE port for 8 of its 16 cycles, so matrix and vector
// work of two waves do share one resource -- v2's waves already keep that port ~78 % busy (2 x 39 % SQ_ACTIVE_INST_ANY) -- and a barrier per segment
// only adds the imbalance between the segments.  What is left is fewer issue cycles per score (32x32x16 MFMAs: half the MFMA issue cost, at the lower
// clock that shape holds on this part), not a different arrangement of the same instructions.

}  // namespace ncsn

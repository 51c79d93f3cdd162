// attn_blk256.h -- k_attn_blk256: the WHOLE 16x16 attention block of NCSN++ / ddpm (AttnBlockpp, layerspp.py:75-91: h = GroupNorm(x); q, k, v = NIN(h);
// softmax(q k^T / sqrt(C)) v; NIN_3; (x + .) / sqrt 2) as ONE launch: k_qkv256 (attn_qkv.h) and k_attn256<true, 8> (attn256.h) in one kernel, phase after phase.
//
// Both kernels already worked on the same geometry -- one 8-wave block per sample, a wave = 32 tokens -- and between them q | k (134 MB at B = 512) and V^T (67 MB)
// went to HBM and came back: 226 MB written and 272 MB re-read per launch pair, each launch bound by memory at ~4.2 TB/s (72 + 84 us).  Here
//   * q never leaves the registers.  k_qkv256's q tiles leave a lane with EIGHT consecutive channels 32 t + 8 q' .. + 7 of token tau(g, r) = 8 (r >> 2) + 4 g + (r & 3):
//     that IS the B operand of S^T = K Q^T for K step t (lane (r, q'): eight K values of query column r) -- for the wave's queries taken in the order tau instead of
//     16 g + r.  The attention phases do not care which query a column is; the output rows (projection, residual, store) use tau too.
//   * k and V^T are still written (one block cannot keep 256 KB on chip) and re-read by the SAME block right away: the stores are complete behind `s_waitcnt vmcnt(0)`
//     (one XCD's L2; the block's own L1 holds no line of them: nothing of this launch read them before), a barrier publishes them to the other waves, and the LDS-DMA
//     of the attention phases finds them in L2 -- HBM sees the writes only.
// Arithmetic, tile loops, swizzles, projection and GroupNorm partials: the two kernels' own (see their headers).  One block per CU (256 registers), 134 KB of LDS (ABLK_LDS_BYTES: four 32-KB stages + the biases);
// the projection phase's weight tiles come three tiles ahead (four 16-KB stages: what k_qkv256's four co-resident blocks hide, one block must prefetch).
#pragma once
#include "attn_qkv.h"
#include "attn256.h"

namespace ncsn {

constexpr int ABLK_STAGES = 4, ABLK_LDS_BYTES = ABLK_STAGES * A256_STAGE + QKV_BIAS_BYTES;      // four 32-KB stages (K / V^T / W3 tiles three ahead), the projection phase's biases behind them

// Sum over the wave's 32 queries of a lane's two per-query values (a: query tau(0, r), b: tau(1, r)) IN THE ORDER k_attn256<true, 8> ADDS THEM, so that the GroupNorm
// partial sums of the block's output -- and with them every later module of the network -- are the two-launch plan's bytes (round-5 review, item 5; advisor note).
// k_attn256's lane r holds queries u = 16 g + r and adds (g = 0) + (g = 1) first, then dpp_row_sum's tree over r: u ^ 16, then u ^ 1, u ^ 2, u ^ 4, u ^ 8.  Here
// u = tau(g, r) = 8 (r >> 2) + 4 g + (r & 3): bit 4 of u is bit 3 of r, bit 2 of u is g, bit 3 of u is bit 2 of r -- so the same tree is r ^ 8 (row_ror:8) on each of the
// two values, the two quad stages on each, THEN a + b, then the half-row mirror (quads are uniform by then: r ^ 4).  fp32 addition commutes, the tree is what must match.
__device__ __forceinline__ float dpp_row_sum_tau(float a, float b) {
    asm volatile("s_nop 4" : "+v"(a), "+v"(b));            // (the inputs may be packed-fp32 results: dpp_row_sum's note in gemm_dma.h)
    a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x128, 0xF, 0xF, true));
    b += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x128, 0xF, 0xF, true));
    asm volatile("" : "+v"(a), "+v"(b));
    a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, true));
    b += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0xB1, 0xF, 0xF, true));
    asm volatile("" : "+v"(a), "+v"(b));
    a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x4E, 0xF, 0xF, true));
    b += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x4E, 0xF, 0xF, true));
    asm volatile("" : "+v"(a), "+v"(b));
    float v = a + b;
    asm volatile("s_nop 1" : "+v"(v));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    asm volatile("" : "+v"(v));
    return v;
}

// x: [B*256][x_ld] bf16 (raw block input: normalised for q | k | v, and the residual of the output); gsc / gsh: GroupNorm (scale | shift) tables [B][256] fp32;
// wf: k_pack_qkv_w's output; bqk: [512] (q then k), bv: [256]; qk: [B*256][512] scratch (only the k half, columns 256.., is written); vT: [B][256][256] scratch;
// w3f: k_pack_attn_w3's output, b3: [256]; o: [B*256][o_ld]; gn_part (may be null): float2 [B][gn_quads].  grid = B, 512 threads, ABLK_LDS_BYTES.
__global__ __launch_bounds__(512, 1) void k_attn_blk256(const bf16* __restrict__ x, int x_ld, const float* __restrict__ gsc, const float* __restrict__ gsh,
                                                       const bf16* __restrict__ wf, const float* __restrict__ bqk, const float* __restrict__ bv,
                                                       bf16* __restrict__ qk, bf16* __restrict__ vT, float scale, const bf16* __restrict__ w3f, const float* __restrict__ b3,
                                                       bf16* __restrict__ o, int o_ld, float out_scale, float2* __restrict__ gn_part, int gn_quads)
{
    constexpr int T = 256, C = 256, KT = A256_KT, NKT = T / KT;
    static_assert(4 * QKV_STAGE <= ABLK_STAGES * A256_STAGE && QKV_TILES == 24, "the projection phase's four stages inside the attention phases' LDS; its biases behind them");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_poison();
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int tok0 = wave * 32;

    // ================= phase 1: GroupNorm-apply + q | k | v projections (k_qkv256) =================
    // One block per CU here (the attention phases need 256 registers), so the weight tiles -- an L2 round trip each, 2-4k clocks against ~1k clocks of MFMAs per
    // tile -- come THREE tiles ahead through four 16-KB stages (k_qkv256 hides them behind four co-resident blocks instead).  Requests, loads and stores retire in
    // issue order: the wait for tile t leaves everything issued behind its requests in flight -- the stores of tiles t - 3 .. t - 1 (two per tile; the q tiles store
    // nothing) and the requests of tiles t + 1, t + 2 (two per wave and tile).
    constexpr int NST = 4, AHEAD = NST - 1;
    auto issue_w = [&](int i) __attribute__((always_inline)) {     // weight tile i: n-tiles 2 i, 2 i + 1, all eight K steps
        unsigned char* st = smem + (i & (NST - 1)) * QKV_STAGE;
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        const bf16* base = wf + (int64_t)i * (QKV_STAGE / 2);
#pragma unroll
        for (int j = 0; j < QKV_STAGE / 8192; ++j) {
            const int p = wave * (QKV_STAGE / 8192) + j;
            __builtin_amdgcn_global_load_lds(base + p * 512 + l * 8, (lds_void*)(st + p * 1024), 16, 0, 0);
        }
    };
    issue_w(0); issue_w(1); issue_w(2);
    float* const sBias = reinterpret_cast<float*>(smem + ABLK_STAGES * A256_STAGE);     // [512 q | k][256 v], behind the attention phases' stages
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    const unsigned lds_bias = (unsigned)(uintptr_t)((lds_u8*)smem) + ABLK_STAGES * A256_STAGE;
    if (tid < 192) reinterpret_cast<float4*>(sBias)[tid] = tid < 128 ? reinterpret_cast<const float4*>(bqk)[tid] : reinterpret_cast<const float4*>(bv)[tid - 128];

    bf16x8 qf[2][8];                                              // the wave's queries as B operands: query tau(g, r), channels 32 c + 8 q .. + 7 (filled by the q tiles)
    {
        bf16x8 hf[2][8];                                          // the wave's 32 tokens x 256 channels, normalised, in operand layout (attn_qkv.h)
        {
            const bf16* xb = x + ((int64_t)b * T + tok0 + 8 * (r >> 2) + (r & 3)) * x_ld + 8 * q;
            bf16x8 raw[2][8];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) raw[g][kc] = *reinterpret_cast<const bf16x8*>(xb + (int64_t)(4 * g) * x_ld + 32 * kc);
            const float* sc = gsc + (int64_t)b * C + 8 * q;
            const float* sh = gsh + (int64_t)b * C + 8 * q;
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                const float4 s0 = *reinterpret_cast<const float4*>(sc + 32 * kc), s1 = *reinterpret_cast<const float4*>(sc + 32 * kc + 4);
                const float4 h0 = *reinterpret_cast<const float4*>(sh + 32 * kc), h1 = *reinterpret_cast<const float4*>(sh + 32 * kc + 4);
                const float s[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, h[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int j = 0; j < 8; ++j) hf[g][kc][j] = (bf16)((float)raw[g][kc][j] * s[j] + h[j]);
            }
        }
        int le;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(le));
        const int re = le & 15, qe = le >> 4;
        // tile t has landed for every wave, and everyone is done with the stage tile t + 3 goes into (tile t - 1's)
        auto tile_begin = [&](auto t_tag) __attribute__((always_inline)) {
            constexpr int t = decltype(t_tag)::value;
            constexpr int stores = 2 * ((t - 3 >= 8) + (t - 2 >= 8) + (t - 1 >= 8)), ahead = 2 * ((t + 1 < QKV_TILES) + (t + 2 < QKV_TILES));
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(stores + ahead) : "memory");
            __builtin_amdgcn_s_barrier();
            if constexpr (t + AHEAD < QKV_TILES) issue_w(t + AHEAD);
            __builtin_amdgcn_sched_barrier(0);
        };
        // a q | k tile (two n-tiles = one 32-channel group with interleaved rows: the lane ends up with EIGHT consecutive channels of its token)
        auto qk_tile = [&](auto t_tag, bf16x8 (&w)[2]) __attribute__((always_inline)) {
            constexpr int t = decltype(t_tag)::value;
            tile_begin(t_tag);
            const unsigned char* sW = smem + (t & (NST - 1)) * QKV_STAGE;
            f32x4 a[2][2];
#pragma unroll
            for (int ntl = 0; ntl < 2; ++ntl) {
                a[ntl][0] = f32x4{0.f, 0.f, 0.f, 0.f}; a[ntl][1] = a[ntl][0];
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) {
                    const bf16x8 fa = *reinterpret_cast<const bf16x8*>(sW + (ntl * 8 + kc) * 1024 + le * 16);
                    a[ntl][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, hf[0][kc], a[ntl][0], 0, 0, 0);
                    a[ntl][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, hf[1][kc], a[ntl][1], 0, 0, 0);
                }
            }
            const int n = 32 * t + 8 * qe;
            f32x4 b0, b1;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(b0), "=&v"(b1) : "v"(lds_bias + (unsigned)n * 4u) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1) :: "memory");
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) { w[g][i] = (bf16)(a[0][g][i] + b0[i]); w[g][4 + i] = (bf16)(a[1][g][i] + b1[i]); }
        };
        using std::integral_constant;
        // q: tiles 0..7 stay in registers (K step t of the scores)
#define NATINF_BLK_Q(t) { bf16x8 w[2]; qk_tile(integral_constant<int, t>{}, w); qf[0][t] = w[0]; qf[1][t] = w[1]; }
        NATINF_BLK_Q(0) NATINF_BLK_Q(1) NATINF_BLK_Q(2) NATINF_BLK_Q(3) NATINF_BLK_Q(4) NATINF_BLK_Q(5) NATINF_BLK_Q(6) NATINF_BLK_Q(7)
#undef NATINF_BLK_Q
        // k: tiles 8..15 -> [q | k] columns 256..
#define NATINF_BLK_K(t) { bf16x8 w[2]; qk_tile(integral_constant<int, t>{}, w); const int n = 32 * t + 8 * qe;                                   \
            _Pragma("unroll") for (int g = 0; g < 2; ++g) *reinterpret_cast<bf16x8*>(qk + ((int64_t)b * T + tok0 + 8 * (re >> 2) + 4 * g + (re & 3)) * (2 * C) + n) = w[g]; }
        NATINF_BLK_K(8) NATINF_BLK_K(9) NATINF_BLK_K(10) NATINF_BLK_K(11) NATINF_BLK_K(12) NATINF_BLK_K(13) NATINF_BLK_K(14) NATINF_BLK_K(15)
#undef NATINF_BLK_K
        // v: tiles 16..23 -> V^T (the same registers as the A operand)
        auto v_tile = [&](auto t_tag) __attribute__((always_inline)) {
            constexpr int t = decltype(t_tag)::value;
            tile_begin(t_tag);
            const unsigned char* sW = smem + (t & (NST - 1)) * QKV_STAGE;
#pragma unroll
            for (int ntl = 0; ntl < QKV_NTL; ++ntl) {
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) {
                    const bf16x8 fb = *reinterpret_cast<const bf16x8*>(sW + (ntl * 8 + kc) * 1024 + le * 16);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf[0][kc], fb, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf[1][kc], fb, a1, 0, 0, 0);
                }
                const int ch = 16 * (QKV_NTL * t + ntl - 32) + re;
                float bb;
                asm volatile("ds_read_b32 %0, %1" : "=v"(bb) : "v"(lds_bias + (unsigned)(512 + ch) * 4u) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bb) :: "memory");
                bf16x8 w;
#pragma unroll
                for (int i = 0; i < 4; ++i) { w[i] = (bf16)(a0[i] + bb); w[4 + i] = (bf16)(a1[i] + bb); }
                *reinterpret_cast<bf16x8*>(vT + ((int64_t)b * C + ch) * T + tok0 + 8 * qe) = w;
            }
        };
        v_tile(integral_constant<int, 16>{}); v_tile(integral_constant<int, 17>{}); v_tile(integral_constant<int, 18>{}); v_tile(integral_constant<int, 19>{});
        v_tile(integral_constant<int, 20>{}); v_tile(integral_constant<int, 21>{}); v_tile(integral_constant<int, 22>{}); v_tile(integral_constant<int, 23>{});
    }
    // every store of k and V^T is complete (vmcnt counts a store until it is written), every wave's: the block re-reads them through L2
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // ================= phases 2-4: scores, softmax, P V, output projection (k_attn256<true, 8>) =================
    constexpr int NP = 4;                                         // 1-KiB DMA pieces per wave and 32-KB tile (8 waves)
    const bf16* kbase = qk + (int64_t)b * T * (2 * C) + C;
    const bf16* vbase = vT + (int64_t)b * A256_D * T;
    constexpr int qk_ld = 2 * C;
    auto issue = [&](int i) __attribute__((always_inline)) {
        unsigned char* st = smem + (i & (ABLK_STAGES - 1)) * A256_STAGE;
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        if (i >= 2 * NKT) {                                       // W3 tile i - 8
            const bf16* base = w3f + (int64_t)(i - 2 * NKT) * (A256_STAGE / 2);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int p = wave * NP + j;
                __builtin_amdgcn_global_load_lds(base + p * 512 + l * 8, (lds_void*)(st + p * 1024), 16, 0, 0);
            }
        } else if (i < NKT) {
            const bf16* base = kbase + (int64_t)(i * KT) * qk_ld;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int p = wave * NP + j, row = 2 * p + (l >> 5), ch = l & 31;
                const int h = (row & 3) | (((row >> 3) & 3) << 2);
                __builtin_amdgcn_global_load_lds(base + row * qk_ld + ((ch ^ h) << 3), (lds_void*)(st + p * 1024), 16, 0, 0);
            }
        } else {
            const bf16* base = vbase + (i - NKT) * KT;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int p = wave * NP + j, d = 8 * p + (l >> 3), ch = l & 7;
                __builtin_amdgcn_global_load_lds(base + d * T + ((ch ^ ((d >> 1) & 7)) << 3), (lds_void*)(st + p * 1024), 16, 0, 0);
            }
        }
    };
    // the twelve tiles (four of K, four of V^T, four of W3) come THREE ahead through four stages: an L2 round trip is 2-4k clocks, a tile 1-2k clocks of MFMAs, and one
    // block per CU has nothing else to run meanwhile.  Phases 2-3 issue no other vector-memory operation: the wait for tile i leaves the eight requests of tiles
    // i + 1, i + 2 in flight; the projection phase (residual loads, stores) waits for everything.
    issue(0); issue(1); issue(2);

    f32x4 acc[2][16];
#pragma unroll
    for (int t = 0; t < 16; ++t) { acc[0][t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int krow = 8 * (r >> 2) + (r & 3);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // (the raw barrier: __syncthreads() carries a fence that waits vmcnt(0) -- for the two tiles in flight too)
        issue(kt + 3);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sK = smem + (kt & 3) * A256_STAGE;
#pragma unroll
        for (int tl = 0; tl < 4; ++tl)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int row = 32 * (tl >> 1) + 4 * (tl & 1) + krow;
                const bf16x8 fa = *reinterpret_cast<const bf16x8*>(sK + row * 512 + (((4 * c + q) ^ r) << 4));
                acc[0][4 * kt + tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, qf[0][c], acc[0][4 * kt + tl], 0, 0, 0);
                acc[1][4 * kt + tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, qf[1][c], acc[1][4 * kt + tl], 0, 0, 0);
            }
    }
    float inv[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) mx = fmaxf(mx, acc[g][t][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float p = __expf((acc[g][t][i] - mx) * scale); acc[g][t][i] = p; sum += p; }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        inv[g] = 1.0f / sum;
        __builtin_amdgcn_sched_barrier(0);
    }
    bf16x8 pf[2][8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) { pf[g][c][i] = (bf16)(acc[g][2 * c][i] * inv[g]); pf[g][c][4 + i] = (bf16)(acc[g][2 * c + 1][i] * inv[g]); }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 oacc[2][16];
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) { oacc[0][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; oacc[1][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int vt = 0; vt < NKT; ++vt) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // (the raw barrier: __syncthreads() carries a fence that waits vmcnt(0) -- for the two tiles in flight too)
        issue(NKT + vt + 3);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sV = smem + ((NKT + vt) & 3) * A256_STAGE;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int dt = 0; dt < 16; ++dt) {
                const bf16x8 fv = *reinterpret_cast<const bf16x8*>(sV + (16 * dt + r) * 128 + (((4 * cc + q) ^ (r >> 1)) << 4));
                oacc[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, pf[0][2 * vt + cc], oacc[0][dt], 0, 0, 0);
                oacc[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, pf[1][2 * vt + cc], oacc[1][dt], 0, 0, 0);
            }
    }
    bf16x8 of[2][8];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int kc = 0; kc < 8; ++kc)
#pragma unroll
            for (int i = 0; i < 4; ++i) { of[g][kc][i] = (bf16)oacc[g][2 * kc][i]; of[g][kc][4 + i] = (bf16)oacc[g][2 * kc + 1][i]; }
    __builtin_amdgcn_sched_barrier(0);
    int le;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(le));
    const int re = le & 15, qe = le >> 4;
    float2* const sred = reinterpret_cast<float2*>(smem);
    float2 part[16];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 3 < 4) issue(2 * NKT + t + 3);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sW = smem + ((2 * NKT + t) & 3) * A256_STAGE;
        f32x4 a3[2][4];
#pragma unroll
        for (int ntl = 0; ntl < 4; ++ntl) { a3[0][ntl] = f32x4{0.f, 0.f, 0.f, 0.f}; a3[1][ntl] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ntl = 0; ntl < 4; ++ntl)
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                const bf16x8 fa = *reinterpret_cast<const bf16x8*>(sW + (ntl * 8 + kc) * 1024 + le * 16);
                a3[0][ntl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, of[0][kc], a3[0][ntl], 0, 0, 0);
                a3[1][ntl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, of[1][kc], a3[1][ntl], 0, 0, 0);
            }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int n = 64 * t + 32 * pr + 8 * qe;
            const float4 b0 = *reinterpret_cast<const float4*>(b3 + n), b1 = *reinterpret_cast<const float4*>(b3 + n + 4);
            float s0[2], ss0[2], s1[2], ss1[2];                 // per g: dpp_row_sum_tau adds them in k_attn256's order
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int64_t row = (int64_t)b * T + tok0 + 8 * (re >> 2) + 4 * g + (re & 3);      // query tau(g, re)
                const bf16x8 rx = *reinterpret_cast<const bf16x8*>(x + row * x_ld + n);
                const f32x4 lo = a3[g][2 * pr], hi = a3[g][2 * pr + 1];
                float v[8] = {lo[0] + b0.x, lo[1] + b0.y, lo[2] + b0.z, lo[3] + b0.w, hi[0] + b1.x, hi[1] + b1.y, hi[2] + b1.z, hi[3] + b1.w};
                bf16x8 w;
#pragma unroll
                for (int i = 0; i < 8; ++i) { v[i] = (v[i] + (float)rx[i]) * out_scale; w[i] = (bf16)v[i]; }
                *reinterpret_cast<bf16x8*>(o + row * o_ld + n) = w;
                s0[g] = (v[0] + v[1]) + (v[2] + v[3]);  ss0[g] = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                s1[g] = (v[4] + v[5]) + (v[6] + v[7]);  ss1[g] = (v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]);
            }
            part[4 * t + 2 * pr] = make_float2(dpp_row_sum_tau(s0[0], s0[1]), dpp_row_sum_tau(ss0[0], ss0[1]));
            part[4 * t + 2 * pr + 1] = make_float2(dpp_row_sum_tau(s1[0], s1[1]), dpp_row_sum_tau(ss1[0], ss1[1]));
        }
    }
    if (gn_part) {
        __syncthreads();
        if (re == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) sred[wave * 64 + 8 * (k >> 1) + 2 * qe + (k & 1)] = part[k];
        }
        __syncthreads();
        if (tid < 64) {
            float s = 0.f, ss = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) { s += sred[w * 64 + tid].x; ss += sred[w * 64 + tid].y; }
            gn_part[(int64_t)b * gn_quads + tid] = make_float2(s, ss);
        }
    }
}

// ============================================================================================================================================================
// k_attn_blk256_v2 (round 6): the same block with q k^T and P V taken against h ITSELF -- no k, no V^T, nothing written or re-read between the phases.
//   q k^T = (h Wq + bq)(h' Wk + bk)^T = (h Wqk + cq) h'^T + [terms that depend on the query only: softmax over the keys cancels them],  Wqk = Wq Wk^T, cq = bq Wk^T
//   (P v) W3 + b3 = P (h' Wv + bv) W3 + b3 = (P h') Wvo + bo   (the rows of P sum to one),                                            Wvo = Wv W3,  bo = bv W3 + b3
// (AttnBlockpp, layerspp.py:75-91; exact in real arithmetic: 2e-7 in fp32, and with the engine's bf16 rounding points the block's error against fp32 is the same --
// 4.30e-3 against 4.37e-3 of its output's range -- CPU study in DESIGN.md section 4 "Round 6").  Two projections instead of four (134 MFLOP per sample instead of 201),
// and the sample's normalised tokens h [256][256] bf16 = 128 KB are the ONLY operand of both attention products: they live in LDS, written once by the waves that
// hold them in registers.  HBM sees x in and the output out: 134 MB per launch at B = 512 against 495.
//   * A' = h Wqk + cq stays in registers exactly as k_attn_blk256's q does (k_pack_qkv_w's q tiles of the folded matrix: eight 16-KB tiles, three ahead through four
//     stages that sit in the image's bytes BEFORE the image is written);
//   * scores S^T = h A'^T: row reads (ds_read_b128) of the image in k_attn256's permuted key order; P V as U^T = h^T P^T: the A operand is h TRANSPOSED -- eight
//     keys of one channel per lane -- read with ds_read_b64_tr_b16 (gfx950: a 16-lane group reads a block of 4 rows x 16 columns and every lane receives one column);
//     ONE image serves both: 512-byte rows, 16-byte chunk c of row `row` at chunk c ^ key(row), key = 3 (row & 1) + 4 ((row >> 1) & 1) + 8 ((row >> 3) & 1) on the
//     chunk's low four bits -- conflict-free for the row reads' four lane groups and for the transposed reads' two halves (searched over the linear keys against
//     every tile, both read kinds and the guide's lane groups);
//   * the output projection (k_pack_attn_w3 of the folded Wvo: four 32-KB tiles): tile 0 is requested before P V into the 32 KB behind the image, tiles 1-3 into
//     the image's bytes once P V is done; residual, rescale, store and the GroupNorm partial sums as before (same order: dpp_row_sum_tau).
// Not the bytes of k_attn_blk256 (another arithmetic); the per-module taps and the isolated fp32 comparison bound it (tests/test_gpu_attn_block.py).
constexpr int ABLK2_IMG = 131072, ABLK2_LDS_BYTES = ABLK2_IMG + A256_STAGE;
static_assert(4 * QKV_STAGE <= ABLK2_IMG && ABLK2_LDS_BYTES <= 163840, "the q phase's four stages inside the image's bytes; image + one W3 tile in 160 KB");

// folded weights of one attention block: w[i] = NIN_i.W [256 in][256 out], b[i] = NIN_i.b (layers.py:546-555)
//   wqk [in][out] = sum_m w0[in][m] w1[out][m];  cq [out] = sum_m b0[m] w1[out][m];  wvo [in][out] = sum_m w2[in][m] w3[m][out];  bo [out] = sum_m b2[m] w3[m][out] + b3[out]
__global__ __launch_bounds__(256) void k_attn_fold_w(const float* __restrict__ w0, const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ w3,
                                                      const float* __restrict__ b0, const float* __restrict__ b2, const float* __restrict__ b3,
                                                      float* __restrict__ wqk, float* __restrict__ cq, float* __restrict__ wvo, float* __restrict__ bo)
{
    const int o = threadIdx.x, i = blockIdx.x;                            // grid 257: rows 0..255 of both products, block 256 = the two bias vectors
    float a = 0.f, c = 0.f;
    if (i < 256) {
        for (int m = 0; m < 256; ++m) { a += w0[i * 256 + m] * w1[o * 256 + m]; c += w2[i * 256 + m] * w3[m * 256 + o]; }
        wqk[i * 256 + o] = a; wvo[i * 256 + o] = c;
    } else {
        for (int m = 0; m < 256; ++m) { a += b0[m] * w1[o * 256 + m]; c += b2[m] * w3[m * 256 + o]; }
        cq[o] = a; bo[o] = c + b3[o];
    }
}

// x, gsc / gsh, o, gn_part: as k_attn_blk256.  wqf: k_pack_qkv_w's q tiles of the folded Wqk (the first 128 KB of its output); cq: [256]; wvof: k_pack_attn_w3 of the folded Wvo; bo: [256].
// grid = B, 512 threads, ABLK2_LDS_BYTES.
__global__ __launch_bounds__(512, 1) void k_attn_blk256_v2(const bf16* __restrict__ x, int x_ld, const float* __restrict__ gsc, const float* __restrict__ gsh,
                                                          const bf16* __restrict__ wqf, const float* __restrict__ cq, float scale, const bf16* __restrict__ wvof,
                                                          const float* __restrict__ bo, bf16* __restrict__ o, int o_ld, float out_scale, float2* __restrict__ gn_part, int gn_quads)
{
    constexpr int T = 256, C = 256, NTQ = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_poison();
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int tok0 = wave * 32;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_u8*)smem);

    // ================= phase 1: GroupNorm-apply, A' = h Wqk + cq (k_attn_blk256's q tiles) =================
    constexpr int NST = 4, AHEAD = NST - 1;
    auto issue_w = [&](int i) __attribute__((always_inline)) {     // Wqk tile i: n-tiles 2 i, 2 i + 1, all eight K steps (stages inside the image's bytes: it is written later)
        unsigned char* st = smem + (i & (NST - 1)) * QKV_STAGE;
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        const bf16* base = wqf + (int64_t)i * (QKV_STAGE / 2);
#pragma unroll
        for (int j = 0; j < QKV_STAGE / 8192; ++j) {
            const int p = wave * (QKV_STAGE / 8192) + j;
            __builtin_amdgcn_global_load_lds(base + p * 512 + l * 8, (lds_void*)(st + p * 1024), 16, 0, 0);
        }
    };
    issue_w(0); issue_w(1); issue_w(2);
    float* const sBias = reinterpret_cast<float*>(smem + ABLK2_IMG);      // cq [256], behind the image (the W3 tile that lands there later is requested after the q phase)
    const unsigned lds_bias = lds0 + ABLK2_IMG;
    if (tid < 64) reinterpret_cast<float4*>(sBias)[tid] = reinterpret_cast<const float4*>(cq)[tid];

    bf16x8 qf[2][8];                                              // A' as B operands: query tau(g, r), channels 32 c + 8 q .. + 7
    {
        bf16x8 hf[2][8];                                          // the wave's 32 tokens x 256 channels, normalised, in operand layout (attn_qkv.h)
        {
            const bf16* xb = x + ((int64_t)b * T + tok0 + 8 * (r >> 2) + (r & 3)) * x_ld + 8 * q;
            bf16x8 raw[2][8];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) raw[g][kc] = *reinterpret_cast<const bf16x8*>(xb + (int64_t)(4 * g) * x_ld + 32 * kc);
            const float* sc = gsc + (int64_t)b * C + 8 * q;
            const float* sh = gsh + (int64_t)b * C + 8 * q;
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                const float4 s0 = *reinterpret_cast<const float4*>(sc + 32 * kc), s1 = *reinterpret_cast<const float4*>(sc + 32 * kc + 4);
                const float4 h0 = *reinterpret_cast<const float4*>(sh + 32 * kc), h1 = *reinterpret_cast<const float4*>(sh + 32 * kc + 4);
                const float s[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, h[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int j = 0; j < 8; ++j) hf[g][kc][j] = (bf16)((float)raw[g][kc][j] * s[j] + h[j]);
            }
        }
        int le;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(le));
        const int qe = le >> 4;
        auto q_tile = [&](auto t_tag, bf16x8 (&w)[2]) __attribute__((always_inline)) {
            constexpr int t = decltype(t_tag)::value;
            constexpr int ahead = 2 * ((t + 1 < NTQ) + (t + 2 < NTQ));      // the requests of tiles t + 1, t + 2 stay in flight (two per wave and tile; nothing is stored in this phase)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(ahead) : "memory");
            __builtin_amdgcn_s_barrier();
            if constexpr (t + AHEAD < NTQ) issue_w(t + AHEAD);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char* sW = smem + (t & (NST - 1)) * QKV_STAGE;
            f32x4 a[2][2];
#pragma unroll
            for (int ntl = 0; ntl < 2; ++ntl) {
                a[ntl][0] = f32x4{0.f, 0.f, 0.f, 0.f}; a[ntl][1] = a[ntl][0];
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) {
                    const bf16x8 fa = *reinterpret_cast<const bf16x8*>(sW + (ntl * 8 + kc) * 1024 + le * 16);
                    a[ntl][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, hf[0][kc], a[ntl][0], 0, 0, 0);
                    a[ntl][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, hf[1][kc], a[ntl][1], 0, 0, 0);
                }
            }
            const int n = 32 * t + 8 * qe;
            f32x4 b0, b1;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(b0), "=&v"(b1) : "v"(lds_bias + (unsigned)n * 4u) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1) :: "memory");
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) { w[g][i] = (bf16)(a[0][g][i] + b0[i]); w[g][4 + i] = (bf16)(a[1][g][i] + b1[i]); }
        };
        using std::integral_constant;
#define NATINF_BLK2_Q(t) { bf16x8 w[2]; q_tile(integral_constant<int, t>{}, w); qf[0][t] = w[0]; qf[1][t] = w[1]; }
        NATINF_BLK2_Q(0) NATINF_BLK2_Q(1) NATINF_BLK2_Q(2) NATINF_BLK2_Q(3) NATINF_BLK2_Q(4) NATINF_BLK2_Q(5) NATINF_BLK2_Q(6) NATINF_BLK2_Q(7)
#undef NATINF_BLK2_Q
        // every wave is done with the weight stages: the image goes over them.  Row = token, 512 bytes; chunk (4 kc + q) of token tau(g, r) at chunk ^ key(row)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int row = tok0 + 8 * (r >> 2) + 4 * g + (r & 3);
            const int key = 3 * (row & 1) + 4 * ((row >> 1) & 1) + 8 * ((row >> 3) & 1);
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) *reinterpret_cast<bf16x8*>(smem + row * 512 + (((4 * kc + q) ^ key) << 4)) = hf[g][kc];
        }
    }
    __syncthreads();

    // ================= phase 2: scores S^T = h A'^T (the image's rows in k_attn256's permuted key order), softmax =================
    f32x4 acc[2][16];
#pragma unroll
    for (int t = 0; t < 16; ++t) { acc[0][t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    {
        const int krow = 8 * (r >> 2) + (r & 3);
        const int keyr = 3 * (r & 1) + 4 * ((r >> 1) & 1) + 8 * ((r >> 2) & 1);      // key(row) of every row this lane reads: row = 64 kt + 32 a + 4 b + krow
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int tl = 0; tl < 4; ++tl)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int row = 64 * kt + 32 * (tl >> 1) + 4 * (tl & 1) + krow;
                    const bf16x8 fa = *reinterpret_cast<const bf16x8*>(smem + row * 512 + (((4 * c + q) ^ keyr) << 4));
                    acc[0][4 * kt + tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, qf[0][c], acc[0][4 * kt + tl], 0, 0, 0);
                    acc[1][4 * kt + tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, qf[1][c], acc[1][4 * kt + tl], 0, 0, 0);
                }
    }
    float inv[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) mx = fmaxf(mx, acc[g][t][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float p = __expf((acc[g][t][i] - mx) * scale); acc[g][t][i] = p; sum += p; }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        inv[g] = 1.0f / sum;
        __builtin_amdgcn_sched_barrier(0);
    }
    bf16x8 pf[2][8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) { pf[g][c][i] = (bf16)(acc[g][2 * c][i] * inv[g]); pf[g][c][4 + i] = (bf16)(acc[g][2 * c + 1][i] * inv[g]); }
    __builtin_amdgcn_sched_barrier(0);

    // ================= phase 3: U^T = h^T P^T -- the image read TRANSPOSED; W3 tile 0 on its way meanwhile =================
    constexpr int NP = 4;                                         // 1-KiB DMA pieces per wave and 32-KB tile (8 waves)
    auto issue3 = [&](int i) __attribute__((always_inline)) {     // Wvo tile i: tile 0 behind the image, tiles 1..3 in the image's bytes (requested once P V is done)
        unsigned char* st = i == 0 ? smem + ABLK2_IMG : smem + (i - 1) * A256_STAGE;
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        const bf16* base = wvof + (int64_t)i * (A256_STAGE / 2);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int p = wave * NP + j;
            __builtin_amdgcn_global_load_lds(base + p * 512 + l * 8, (lds_void*)(st + p * 1024), 16, 0, 0);
        }
    };
    issue3(0);                                                    // (cq behind the image is dead: every wave passed the barriers behind the q phase)
    f32x4 oacc[2][16];
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) { oacc[0][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; oacc[1][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    {
        // lane 4 q' + p of its 16-lane group supplies row k0 + q' (k0 = 64 vt + 32 cc + 8 q: this lane group's eight keys), columns 16 dt + 4 p .. + 3; it receives
        // channel 16 dt + r for the four rows.  key(row) is the same for rows k0 + q' and k0 + 4 + q': 3 (q' & 1) + 4 (q' >> 1) + 8 (q & 1)
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const int qp = r >> 2, pp = r & 3;
        const int keyt = 3 * (qp & 1) + 4 * (qp >> 1) + 8 * (q & 1);
        const unsigned abase = lds0 + (unsigned)((8 * q + qp) * 512 + 8 * (pp & 1));
        const unsigned kh = (unsigned)(keyt >> 1), kl = (unsigned)(keyt & 1) ^ (unsigned)(pp >> 1);
#pragma unroll
        for (int vt = 0; vt < 4; ++vt)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int dt = 0; dt < 16; ++dt) {
                    // chunk 2 dt + (p >> 1) at (2 dt + (p >> 1)) ^ key: the pair index dt ^ (key >> 1) on its low three bits, the chunk of the pair (p >> 1) ^ (key & 1)
                    const unsigned a = abase + (unsigned)((64 * vt + 32 * cc) * 512) + ((((unsigned)dt ^ kh) << 5) | (kl << 4));
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)a);
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(a + 4 * 512));
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const bf16x8 fv = __builtin_bit_cast(bf16x8, both);
                    oacc[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, pf[0][2 * vt + cc], oacc[0][dt], 0, 0, 0);
                    oacc[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, pf[1][2 * vt + cc], oacc[1][dt], 0, 0, 0);
                }
    }
    bf16x8 of[2][8];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int kc = 0; kc < 8; ++kc)
#pragma unroll
            for (int i = 0; i < 4; ++i) { of[g][kc][i] = (bf16)oacc[g][2 * kc][i]; of[g][kc][4 + i] = (bf16)oacc[g][2 * kc + 1][i]; }
    __builtin_amdgcn_sched_barrier(0);
    // every wave is done with the image: the other three W3 tiles go over it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    issue3(1); issue3(2); issue3(3);

    // ================= phase 4: output projection, residual, rescale, GroupNorm partials (k_attn_blk256's, on the folded weights) =================
    int le;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(le));
    const int re = le & 15, qe = le >> 4;
    float2 part[16];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t == 0) {                                              // tile 0 landed during P V; this one wait covers tiles 1-3 as well (one L2 round trip per block): every wave
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // waits for ITS pieces of all four tiles, the barrier publishes them -- the later passes need neither
            __syncthreads();
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sW = t == 0 ? smem + ABLK2_IMG : smem + (t - 1) * A256_STAGE;
        f32x4 a3[2][4];
#pragma unroll
        for (int ntl = 0; ntl < 4; ++ntl) { a3[0][ntl] = f32x4{0.f, 0.f, 0.f, 0.f}; a3[1][ntl] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ntl = 0; ntl < 4; ++ntl)
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                const bf16x8 fa = *reinterpret_cast<const bf16x8*>(sW + (ntl * 8 + kc) * 1024 + le * 16);
                a3[0][ntl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, of[0][kc], a3[0][ntl], 0, 0, 0);
                a3[1][ntl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, of[1][kc], a3[1][ntl], 0, 0, 0);
            }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int n = 64 * t + 32 * pr + 8 * qe;
            const float4 b0 = *reinterpret_cast<const float4*>(bo + n), b1 = *reinterpret_cast<const float4*>(bo + n + 4);
            float s0[2], ss0[2], s1[2], ss1[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int64_t row = (int64_t)b * T + tok0 + 8 * (re >> 2) + 4 * g + (re & 3);      // query tau(g, re)
                const bf16x8 rx = *reinterpret_cast<const bf16x8*>(x + row * x_ld + n);
                const f32x4 lo = a3[g][2 * pr], hi = a3[g][2 * pr + 1];
                float v[8] = {lo[0] + b0.x, lo[1] + b0.y, lo[2] + b0.z, lo[3] + b0.w, hi[0] + b1.x, hi[1] + b1.y, hi[2] + b1.z, hi[3] + b1.w};
                bf16x8 w;
#pragma unroll
                for (int i = 0; i < 8; ++i) { v[i] = (v[i] + (float)rx[i]) * out_scale; w[i] = (bf16)v[i]; }
                *reinterpret_cast<bf16x8*>(o + row * o_ld + n) = w;
                s0[g] = (v[0] + v[1]) + (v[2] + v[3]);  ss0[g] = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                s1[g] = (v[4] + v[5]) + (v[6] + v[7]);  ss1[g] = (v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]);
            }
            part[4 * t + 2 * pr] = make_float2(dpp_row_sum_tau(s0[0], s0[1]), dpp_row_sum_tau(ss0[0], ss0[1]));
            part[4 * t + 2 * pr + 1] = make_float2(dpp_row_sum_tau(s1[0], s1[1]), dpp_row_sum_tau(ss1[0], ss1[1]));
        }
    }
    if (gn_part) {
        float2* const sred = reinterpret_cast<float2*>(smem + 3 * A256_STAGE);      // (the image's last 32 KB: no W3 tile lives there)
        __syncthreads();
        if (re == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) sred[wave * 64 + 8 * (k >> 1) + 2 * qe + (k & 1)] = part[k];
        }
        __syncthreads();
        if (tid < 64) {
            float s = 0.f, ss = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) { s += sred[w * 64 + tid].x; ss += sred[w * 64 + tid].y; }
            gn_part[(int64_t)b * gn_quads + tid] = make_float2(s, ss);
        }
    }
}

}  // namespace ncsn

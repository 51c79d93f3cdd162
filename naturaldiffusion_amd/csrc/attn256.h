// attn256.h -- k_attn256: the 16x16 self-attention of NCSN++ / ddpm (256 tokens, ONE head of 256 channels; reference AttnBlockpp,
// layerspp.py:75-91: softmax(q k^T / sqrt(C)) v) with K and V^T STREAMED through LDS.
//
// k_attn_fused<8,16,true> (attn_fused.h) kept the whole K (128 KB) and then the whole V^T of a sample in LDS: one 512-thread block per CU,
// every phase -- load K, barrier, scores, barrier, load V^T, barrier, P V -- alone on its CU with nothing to overlap it: 94 us per launch at
// B = 512 for 8 us worth of MFMAs (14 % of the matrix peak).  Here
//   * a block = 128 queries of one sample (4 waves x 32 queries; two blocks per sample), 64 KB of LDS, 256 registers: TWO blocks per CU;
//   * K and V^T arrive in tiles of 64 keys ([64][256] / [256][64] bf16 = 32 KB each) by direct global->LDS DMA into a two-stage ring,
//     tile i+1 in flight while tile i is multiplied: four K tiles (scores), then four V^T tiles (P V), one barrier per tile;
//   * all 256 scores of a query stay in registers (128 per lane), so the softmax is the exact two-pass one -- no online rescaling --
//     and runs while the first V^T tile is in flight;
//   * S^T = K Q^T with the K rows of score tile (2c + h) taken in the order key = 32c + 8(r >> 2) + 4h + (r & 3) (flash_attn.h's trick):
//     the lane's values of tiles 2c and 2c+1 are keys 32c + 8q .. 32c + 8q + 7, exactly the 16 bytes its V^T fragment read covers -- V^T
//     needs no permutation and P never leaves the registers.
//   LDS swizzles (chunk index XOR row hash, applied on the DMA source address and on the fragment read): K rows are 512 B (every row starts on
//   the same bank): hash = (row & 3) | ((row >> 3) & 3) << 2, which is the lane's row number r for the permuted rows a fragment reads -- 16
//   distinct 16-byte positions per ds_read_b128 lane group; V^T rows are 128 B: hash = (row >> 1) & 7 (flash_attn.h).
#pragma once
#include "ncsnpp_kernels.h"

namespace ncsn {

__device__ __forceinline__ int blockIdx_pair(int b, int qhalf) { return 2 * b + qhalf; }      // GroupNorm partial row of (sample, query half): 128-row tiles in token order

constexpr int A256_T = 256, A256_D = 256, A256_KT = 64, A256_STAGE = 32768, A256_LDS_BYTES = 2 * A256_STAGE;

// PROJ (round 3, late): the block's output projection in the same launch -- out = (x + O W3 + b3) * out_scale (AttnBlockpp's NIN_3 + skip, layerspp.py:88-91)
// -- so that O (67 MB at B = 512) is neither written nor read back.  out^T = W3^T O^T as a third MFMA phase: the B operand is the wave's own O, rounded
// to bf16 exactly as the separate launch stored it -- lane (query r, q) holds channels 16 dt + 4 q + i of tile dt, so K step kc takes tiles 2 kc and
// 2 kc + 1: K slot j <-> channel 32 kc + (j < 4 ? 4 q + j : 16 + 4 q + j - 4) -- and the A operand is W3 packed fragment-major IN THAT CHANNEL ORDER
// (k_pack_attn_w3: [16 n-tiles][8 K steps][64 lanes][8]), streamed through the same two LDS stages as four more 32-KB tiles (lane-linear pieces:
// no swizzle).  A lane ends up with four consecutive output channels of one query: bias, residual, scale, 8-byte store, and the GroupNorm partial
// sums of the output (one table row per block = 128 rows: a DPP row sum over the 16 queries of a lane group, then the two query groups and four waves).
// w3f: k_pack_attn_w3's output; resid: x [B*256][resid_ld]; gn_part (may be null): float2 [2 B][gn_quads].
__global__ __launch_bounds__(256) void k_pack_attn_w3(const float* __restrict__ w, bf16* __restrict__ wf)      // w: NIN weight [256 in][256 out] (layers.py:546-555)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;                  // one element: [nt][kc][lane][j]
    if (idx >= 256 * 256) return;
    const int j = idx & 7, lane = (idx >> 3) & 63, kc = (idx >> 9) & 7, nt = idx >> 12;
    const int q = lane >> 4, r = lane & 15;
    // rows: n-tiles 2 p, 2 p + 1 share the channels 32 p ..: row r of tile 2 p + h is channel 32 p + 8 (r >> 2) + 4 h + (r & 3) -- the rows 4 q + i a lane holds
    // of both tiles are eight consecutive channels
    const int c = 32 * kc + (j < 4 ? 4 * q + j : 16 + 4 * q + j - 4), n = 32 * (nt >> 1) + 8 * (r >> 2) + 4 * (nt & 1) + (r & 3);
    wf[idx] = (bf16)w[c * 256 + n];
}

// qk: [B*256][qk_ld] bf16, q at column 0, k at column k_off; vT: [B][256 channels][256 tokens]; o: [B*256][o_ld].  grid = 2 B, 256 threads.
// NWV = 8 (the PROJ form's launch): ONE sample per block, 8 waves x 32 queries, one block per CU -- every K / V^T / W3 tile crosses L2 -> LDS once per sample
// instead of once per half (393 -> 197 MB per launch at B = 512, more than the kernel's HBM traffic); NWV = 4: two blocks of 128 queries per sample.
template <bool PROJ, int NWV = 4>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 2 : 1) void k_attn256(const bf16* __restrict__ qk, int qk_ld, int k_off, const bf16* __restrict__ vT,
                                                    bf16* __restrict__ o, int o_ld, float scale, const bf16* __restrict__ w3f, const float* __restrict__ b3,
                                                    const bf16* __restrict__ resid, int resid_ld, float out_scale, float2* __restrict__ gn_part, int gn_quads)
{
    constexpr int T = A256_T, KT = A256_KT, NKT = T / KT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_poison();
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The two blocks of a sample read the same K and V^T (128 KB each): block ids i and i + 8 -- the same XCD under round-robin dispatch, launched in the
    // same wave of blocks -- take the two query halves of one sample, so the second read hits that XCD's L2 (PMC, round 3: 336 MB fetched per launch at
    // B = 512 against 201 MB of operands with the halves on neighbouring ids = different XCDs).  A tail of < 16 blocks keeps the plain order.
    const int bid = blockIdx.x, full = (int)gridDim.x & ~15;
    const int b = NWV == 8 ? bid : (bid < full ? ((bid >> 4) << 3) + (bid & 7) : bid >> 1), qhalf = NWV == 8 ? 0 : (bid < full ? (bid >> 3) & 1 : bid & 1);
    constexpr int NP = 32 / NWV;                          // 1-KiB DMA pieces per wave and 32-KB tile
    const bf16* qbase = qk + (int64_t)b * T * qk_ld;
    const bf16* kbase = qbase + k_off;
    const bf16* vbase = vT + (int64_t)b * A256_D * T;

    // tile i of the stream: i < 4: K rows [64 i, 64 i + 64) x 256 channels; i >= 4: V^T rows (channels) 0..255 x keys [64 (i - 4), + 64)
    // per wave and tile 8 DMA pieces of 1 KiB: K: 2 rows x 512 B per piece (lane l: row 2p + (l >> 5), chunk l & 31); V^T: 8 rows x 128 B
    auto issue = [&](int i) __attribute__((always_inline)) {
        unsigned char* st = smem + (i & 1) * A256_STAGE;
        int l;                                           // the lane id, recomputed per call: with every loop unrolled hipcc would otherwise keep the 16 per-lane
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));      // addresses of all eight tiles alive (22 spilled registers)
        if (PROJ && i >= 2 * NKT) {                       // W3 tile i - 8: n-tiles 4 (i - 8) .. + 3, all eight K steps: 32 fragment blocks of 1 KiB, lane-linear
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
    issue(0);

    // the wave's 32 queries as B operands: query (16 g + r), channels 32 c + 8 q .. + 7
    bf16x8 qf[2][8];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int c = 0; c < 8; ++c)
            qf[g][c] = *reinterpret_cast<const bf16x8*>(qbase + (int64_t)(qhalf * 128 + wave * 32 + 16 * g + r) * qk_ld + 32 * c + 8 * q);

    f32x4 acc[2][16];                                   // score tile t = 4 kt + tl: keys 64 kt + 32 (tl >> 1) + 8 q' + 4 (tl & 1) + i  (q' = lane >> 4)
#pragma unroll
    for (int t = 0; t < 16; ++t) { acc[0][t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int krow = 8 * (r >> 2) + (r & 3);            // (its swizzle hash is r)

#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile kt has landed (the compiler does not track LDS-DMA completions)
        __syncthreads();                                   // ... for every wave, and everyone is done with the stage tile kt + 1 goes into
        issue(kt + 1);                                     // (kt + 1 == 4: the first V^T tile)
        __builtin_amdgcn_sched_barrier(0);                 // the requests go out BEFORE this tile's MFMAs: hipcc otherwise sinks them to the end of the tile, right in front of the wait
        const unsigned char* sK = smem + (kt & 1) * A256_STAGE;
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

    // exact softmax over the 256 keys of each query: the lane's 64 values + the three other lanes of its query column
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
        __builtin_amdgcn_sched_barrier(0);               // one query group at a time: interleaved, the two exp chains and the packing overflow 256 registers
    }
    bf16x8 pf[2][8];                                    // P as the B operand of O^T = V^T P^T, 32-key chunk c: keys 32 c + 8 q .. + 7
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (PROJ || vt + 1 < NKT) issue(NKT + vt + 1);            // (PROJ: the first W3 tile follows the last V^T tile)
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sV = smem + ((NKT + vt) & 1) * A256_STAGE;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)                  // the tile's two 32-key chunks: P chunk 2 vt + cc
#pragma unroll
            for (int dt = 0; dt < 16; ++dt) {
                const bf16x8 fv = *reinterpret_cast<const bf16x8*>(sV + (16 * dt + r) * 128 + (((4 * cc + q) ^ (r >> 1)) << 4));
                oacc[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, pf[0][2 * vt + cc], oacc[0][dt], 0, 0, 0);
                oacc[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, pf[1][2 * vt + cc], oacc[1][dt], 0, 0, 0);
            }
    }
    if constexpr (PROJ) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x8 of[2][8];                                  // O as the B operand of out^T = W3^T O^T (see the head of this file for the K-slot order)
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
        float2* const sred = reinterpret_cast<float2*>(smem);           // [4 waves][64 quads], written after the last tile
        float2 part[16];                                  // this lane group's (sum, sum of squares) of quads 16 t + 8 pr + 2 qe + h, over the wave's 32 queries
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t + 1 < 4) issue(2 * NKT + t + 1);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char* sW = smem + (t & 1) * A256_STAGE;
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
            // n-tiles 2 p, 2 p + 1 are one 32-channel group with interleaved rows (k_pack_attn_w3): the lane holds EIGHT consecutive channels n .. n + 7 of its
            // query -- 16-byte residual loads and stores (64 contiguous bytes per row and instruction; the 8-byte form cost k_qkv256 10 us)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int n = 64 * t + 32 * pr + 8 * qe;
                const float4 b0 = *reinterpret_cast<const float4*>(b3 + n), b1 = *reinterpret_cast<const float4*>(b3 + n + 4);
                float s0 = 0.f, ss0 = 0.f, s1 = 0.f, ss1 = 0.f;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int64_t row = (int64_t)b * T + qhalf * 128 + wave * 32 + 16 * g + re;
                    const bf16x8 rx = *reinterpret_cast<const bf16x8*>(resid + row * resid_ld + n);
                    const f32x4 lo = a3[g][2 * pr], hi = a3[g][2 * pr + 1];
                    float v[8] = {lo[0] + b0.x, lo[1] + b0.y, lo[2] + b0.z, lo[3] + b0.w, hi[0] + b1.x, hi[1] + b1.y, hi[2] + b1.z, hi[3] + b1.w};
                    bf16x8 w;
#pragma unroll
                    for (int i = 0; i < 8; ++i) { v[i] = (v[i] + (float)rx[i]) * out_scale; w[i] = (bf16)v[i]; }
                    *reinterpret_cast<bf16x8*>(o + row * o_ld + n) = w;
                    s0 += (v[0] + v[1]) + (v[2] + v[3]);  ss0 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                    s1 += (v[4] + v[5]) + (v[6] + v[7]);  ss1 += (v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]);
                }
                part[4 * t + 2 * pr] = make_float2(dpp_row_sum(s0), dpp_row_sum(ss0));          // quads (n >> 2), (n >> 2) + 1
                part[4 * t + 2 * pr + 1] = make_float2(dpp_row_sum(s1), dpp_row_sum(ss1));
            }
        }
        if (gn_part) {
            __syncthreads();                              // every wave is done with the last W3 tile
            if (re == 0) {
#pragma unroll
                for (int k = 0; k < 16; ++k) sred[wave * 64 + 8 * (k >> 1) + 2 * qe + (k & 1)] = part[k];      // part[4 t + 2 pr + h]: quad 16 t + 8 pr + 2 qe + h
            }
            __syncthreads();
            if (tid < 64) {
                float s = 0.f, ss = 0.f;
#pragma unroll
                for (int w = 0; w < NWV; ++w) { s += sred[w * 64 + tid].x; ss += sred[w * 64 + tid].y; }
                gn_part[(int64_t)(NWV == 8 ? b : blockIdx_pair(b, qhalf)) * gn_quads + tid] = make_float2(s, ss);      // (NWV = 8: one partial row per sample)
            }
        }
        return;
    }
    // O^T's layout: a lane holds 4 consecutive channels (16 dt + 4 q + i) of query (16 g + r): 8-byte stores
    int lane_e;                                          // lane id recomputed: r / q of the prologue would otherwise be kept alive through both MFMA phases
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int re = lane_e & 15, qe = lane_e >> 4;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int dt = 0; dt < 16; ++dt) {
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            bf16x4 w;
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = (bf16)oacc[g][dt][i];
            *reinterpret_cast<bf16x4*>(o + ((int64_t)b * T + qhalf * 128 + wave * 32 + 16 * g + re) * o_ld + 16 * dt + 4 * qe) = w;
        }
}

}  // namespace ncsn

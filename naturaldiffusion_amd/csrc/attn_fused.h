// attn_fused.h -- one-launch multi-head attention for 256 tokens and head_dim <= 96 (DiT).
//
// One block = one (sample, head): its K [256][hd] and V^T [hd][256] live in LDS (95 KB), eight waves each own 32
// queries.  No S / P round trip through memory, and no LDS round trip either:
//   S^T = K Q^T        A operand = K rows from LDS, B operand = the wave's Q rows (registers, loaded once).  The MFMA
//                      result layout gives each lane, for query (lane & 15), keys 4*(lane>>4)+i of every 16-key tile.
//   softmax            per query: the lane's 64 values + xor-shuffles over lane>>4 (two steps)
//   O^T = V^T P^T      B operand = P for (query = lane & 15, 8 keys): exactly the values the lane already holds from S^T
//                      tiles 2c and 2c+1, i.e. keys {32c+4q+i} u {32c+16+4q+i}.  The contraction does not care about
//                      the key order as long as A agrees, so V^T is written to LDS with that permutation inside every
//                      32-key chunk and its fragments stay single ds_read_b128s.
//   store              O^T's layout gives a lane 4 consecutive head channels of one query: 8-byte stores.
// LDS rows are padded by 16 B so that the 16 rows of a fragment read start in 16 distinct 4-bank groups.
// Reference: timm Attention as used by deps/DiT/models.py:16,113 (softmax(q k^T * hd^-0.5) v per head).
#pragma once
#include "ncsnpp_kernels.h"

namespace ncsn {

// TWO_PHASE (head_dim 256, the NCSN++ attention): K and V^T do not fit LDS together, so V^T is loaded into K's place after
// the scores are done -- P waits in registers as packed bf16 meanwhile.
template <int NQK, int ND, bool TWO_PHASE = false>
struct AttnCfg {
    static constexpr int T = 256, THREADS = 512;
    static constexpr int KSTR = NQK * 64 + 16;            // bytes per K row (NQK*32 bf16 + pad)
    static constexpr int VSTR = T * 2 + 16;               // bytes per V^T row
    static constexpr int KB = T * KSTR, VB = ND * 16 * VSTR;
    static constexpr int LDS_BYTES = TWO_PHASE ? (KB > VB ? KB : VB) : KB + VB;
    static_assert(LDS_BYTES <= 163840, "LDS budget");
};

// qk: [B*256][qk_ld] bf16 with q at column 0 and k at column k_off (+ head*hd); vT: [B][H*hd][256]; o: [B*256][o_ld]
// VROW (round 5, the DiT engine): v comes ROW-MAJOR -- `vT` then points at the v columns of the same [B*256][qk_ld] buffer the q | k | v projection wrote as ONE
// GEMM -- and is transposed on its way into LDS (eight 2-byte LDS writes per 16-byte load, lanes along the keys: the 64 keys of a wave-instruction fill the 32
// banks twice).  What it replaces is a launch: the batched V^T = W_v h^T GEMM, 24.6 us of a 245-us DiT-XL/2 block at Validate's batch of 16, whose output columns now
// ride in the q | k GEMM's one under-filled round of tiles.  Same LDS image, same arithmetic behind it.
template <int NQK, int ND, bool TWO_PHASE = false, bool VROW = false>
__global__ __launch_bounds__(512) void k_attn_fused(const bf16* __restrict__ qk, int qk_ld, int k_off, const bf16* __restrict__ vT,
                                                    bf16* __restrict__ o, int o_ld, int H, int hd, float scale)
{
    using Cfg = AttnCfg<NQK, ND, TWO_PHASE>;
    constexpr int T = Cfg::T, KSTR = Cfg::KSTR, VSTR = Cfg::VSTR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_poison();
    unsigned char* sK = smem;
    unsigned char* sV = TWO_PHASE ? smem : smem + T * KSTR;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / H, hh = blockIdx.x - b * H;
    const bf16* qbase = qk + (int64_t)b * T * qk_ld + hh * hd;
    const bf16* kbase = qbase + k_off;
    const bf16* vbase = VROW ? vT + (int64_t)b * T * qk_ld + hh * hd : vT + ((int64_t)b * H + hh) * hd * T;
    // (zero fill by assignment, not `cond ? *p : zero4`: hipcc turns that select of two lvalues into a select of ADDRESSES and
    // parks the zero vector in scratch memory)

    for (int idx = tid; idx < T * NQK * 4; idx += 512) {                       // K rows, zero-padded to NQK*32 channels
        const int row = idx / (NQK * 4), ch = idx - row * (NQK * 4);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (ch * 8 < hd) v = *reinterpret_cast<const uint4*>(kbase + (int64_t)row * qk_ld + ch * 8);
        *reinterpret_cast<uint4*>(sK + row * KSTR + ch * 16) = v;
    }
    auto load_vT = [&]() __attribute__((always_inline)) {
        if constexpr (VROW) {
            // key k of a 32-key chunk sits at position 8 * (2 * (m & 1) + half) + 4 * (m >> 1) + i of its chunk, m = k >> 3, half = (k >> 2) & 1, i = k & 3
            // (the image the loop below builds from V^T rows); channels [hd, ND * 16) are zero rows
            for (int idx = tid; idx < T * ND * 2; idx += 512) {
                const int key = idx & (T - 1), d0 = (idx >> 8) * 8;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (d0 < hd) v = *reinterpret_cast<const uint4*>(vbase + (int64_t)key * qk_ld + d0);
                const int kk = key & 31, m = kk >> 3, pos = 8 * (2 * (m & 1) + ((kk >> 2) & 1)) + 4 * (m >> 1) + (kk & 3);
                unsigned short* dst = reinterpret_cast<unsigned short*>(sV + d0 * VSTR + ((key & ~31) + pos) * 2);
                const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) dst[e * (VSTR / 2)] = (unsigned short)(w[e >> 1] >> (16 * (e & 1)));
            }
            return;
        }
        for (int idx = tid; idx < ND * 16 * 32; idx += 512) {                      // V^T rows, keys permuted inside 32-key chunks
            const int d = idx >> 5, m8 = idx & 31, c = m8 >> 2, m = m8 & 3;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (d < hd) v = *reinterpret_cast<const uint4*>(vbase + (int64_t)d * T + m8 * 8);
            const int qa = (m & 1) * 2, jo = (m >> 1) * 4;
            unsigned char* row = sV + d * VSTR + (32 * c + jo) * 2;
            *reinterpret_cast<uint2*>(row + 8 * qa * 2) = make_uint2(v.x, v.y);
            *reinterpret_cast<uint2*>(row + 8 * (qa + 1) * 2) = make_uint2(v.z, v.w);
        }
    };
    if (!TWO_PHASE) load_vT();
    bf16x8 qf[2][NQK];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int c = 0; c < NQK; ++c) {
            const int dcol = 32 * c + 8 * q;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (dcol < hd) v = *reinterpret_cast<const uint4*>(qbase + (int64_t)(wave * 32 + 16 * g + r) * qk_ld + dcol);
            qf[g][c] = __builtin_bit_cast(bf16x8, v);
        }
    __syncthreads();

    f32x4 acc[2][16];
#pragma unroll
    for (int t = 0; t < 16; ++t) { acc[0][t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int c = 0; c < NQK; ++c) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(sK + (16 * t + r) * KSTR + (32 * c + 8 * q) * 2);
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[0][c], acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[1][c], acc[1][t], 0, 0, 0);
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
    }

    bf16x8 pf[2][8];                                 // P as the B operand of O^T = V^T P^T: the score registers are free after this
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) {      // normalised here: nothing but pf has to survive the second phase (no spills at 256 VGPRs)
                pf[g][c][i] = (bf16)(acc[g][2 * c][i] * inv[g]); pf[g][c][4 + i] = (bf16)(acc[g][2 * c + 1][i] * inv[g]);
            }
    if (TWO_PHASE) {
        __syncthreads();                             // every wave is done with K
        load_vT();
        __syncthreads();
    }
    f32x4 oacc[2][ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) { oacc[0][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; oacc[1][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(sV + (16 * dt + r) * VSTR + (32 * c + 8 * q) * 2);
            oacc[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pf[0][c], oacc[0][dt], 0, 0, 0);
            oacc[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pf[1][c], oacc[1][dt], 0, 0, 0);
        }
    }
    // lane coordinates recomputed from the hardware lane id: hipcc otherwise keeps r and q of the prologue alive through both
    // MFMA phases -- in scratch memory, at 256 VGPRs
    int lane_e;                                      // (asm: the builtin would be merged with the lane id __shfl_xor computed, and spilled)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int re = lane_e & 15, qe = lane_e >> 4;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
            const int d0 = 16 * dt + 4 * qe;
            if (d0 < hd) {
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                bf16x4 w;
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = (bf16)oacc[g][dt][i];
                *reinterpret_cast<bf16x4*>(o + ((int64_t)b * T + wave * 32 + 16 * g + re) * o_ld + hh * hd + d0) = w;
            }
        }
}

}  // namespace ncsn

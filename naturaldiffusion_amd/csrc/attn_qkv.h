// attn_qkv.h -- k_qkv256: GroupNorm-apply + the q | k | v projections of the 16x16 attention block (AttnBlockpp, layerspp.py:75-87: h = GroupNorm_0(x);
// q, k, v = NIN_0/1/2(h)) in ONE launch.
//
// Until round 3 (late) this was three launches, every one bound by HBM: k_gn_apply (x -> h: 67 + 67 MB at B = 512, 35 us), the q | k GEMM (h -> [q | k]:
// 67 + 134 MB, 66 us) and the batched V^T GEMM (h -> V^T: 67 + 67 MB, 34 us).  Here x is read once and h never exists:
//   * a block = the 256 tokens of one sample (8 waves x 32 tokens; grid B); a lane loads its wave's x fragments -- token 16 g + r, channels
//     32 kc + 8 q .. + 7 -- straight into the MFMA operand layout, normalises them in registers (x * scale + shift with the sample's table, rounded to
//     bf16 as k_gn_apply rounds h) and keeps all 256 channels of its 32 tokens as sixteen operand registers sets (64 VGPRs);
//   * the three weight matrices arrive fragment-major (k_pack_qkv_w: [48 n-tiles][8 K steps][64 lanes][8], 384 KB, L2-resident) in 24 tiles of 16 KB
//     through a two-stage LDS ring by LDS-DMA (lane-linear 1-KiB pieces: conflict-free, no swizzle), shared by the four waves;
//   * q | k tiles: out^T = W h^T (weights as the A operand) -- a lane ends up with four consecutive output channels of one token: 8-byte stores into
//     [q | k] (row-major, what k_attn256 reads); v tiles: the SAME registers as the A operand and the weights as B -- a lane ends up with four
//     consecutive TOKENS of one channel: 8-byte stores into V^T ([B][256 channels][256 tokens]).  Biases: per column for q | k, per row for V^T.
// 768 MFMAs per wave (12 us of matrix time per CU) under 67 MB in, 201 MB out: two 8-wave blocks per CU, 126 registers, 75 us per launch at B = 512 (35 + 66 + 34 us
// for the three launches it replaces), output bit-identical to theirs.  What the first forms taught (105 -> 100 -> 92 -> 82 us): 32-KB tiles at two blocks
// per CU left the next tile's DMA exposed; a per-n-tile bias LOAD inside the loop made hipcc drain the DMA and the previous stores (vmcnt(0)) at every
// n-tile; 8-byte stores (32 contiguous bytes per row and instruction) cost 10 us against 16-byte ones (64).  One sample per block (8 waves, two blocks
// per CU) instead of half a sample: the 384 KB of weights cross L2 -> LDS once per sample -- 197 MB per launch instead of 393 (more than the HBM traffic): 82 -> 75 us.
#pragma once
#include "ncsnpp_kernels.h"

namespace ncsn {

// Weight tiles of 16 KB (two n-tiles x eight K steps), two stages = 32 KB of LDS per block: FOUR blocks per CU (127 registers).  With 32-KB tiles and two
// blocks per CU the launch took 105 us: a tile is 64 MFMAs per wave (~1k clocks), its successor's DMA (an L2 round trip, ~2-4k clocks) was requested
// only one tile ahead, and two blocks per CU could not cover the difference.
// The loop carries NO ordinary global load: with an LDS-DMA in flight hipcc answers the first use of one with `s_waitcnt vmcnt(0)` -- which waits for the
// next tile's DMA and for every store of the previous tile (the first form loaded the bias per n-tile and ran at one memory round trip per n-tile:
// 100 us).  The biases sit in LDS (3 KB behind the stages), the tile wait is COUNTED -- vmcnt(4): the wave's four stores of the tile before stay in
// flight, loads / stores / LDS-DMA retire in issue order -- and the barrier is the raw s_barrier (a __syncthreads() fence drains the stores too).
constexpr int QKV_STAGE = 16384, QKV_NTL = 2, QKV_TILES = 48 / QKV_NTL, QKV_BIAS_BYTES = 768 * 4, QKV_LDS_BYTES = 2 * QKV_STAGE + QKV_BIAS_BYTES;

// w0 / w1 / w2: the NIN weights of q, k, v, each [256 in][256 out] fp32 (layers.py:546-555) -> wf[nt][kc][lane][j] = W_(nt / 16)[32 kc + 8 (lane >> 4) + j][16 (nt % 16) + (lane & 15)]
// (n_mats = 1: the q tiles only -- k_attn_blk256_v2's folded Wqk: w1 / w2 are not read)
__global__ __launch_bounds__(256) void k_pack_qkv_w(const float* __restrict__ w0, const float* __restrict__ w1, const float* __restrict__ w2, bf16* __restrict__ wf, int n_mats = 3)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_mats * 256 * 256) return;
    const int j = idx & 7, lane = (idx >> 3) & 63, kc = (idx >> 9) & 7, nt = idx >> 12;
    const float* w = nt < 16 ? w0 : (nt < 32 ? w1 : w2);
    const int r = lane & 15;
    // q / k: n-tiles 2 p, 2 p + 1 share the 32 channels 32 p ..: row r of tile 2 p + h is channel 32 p + 8 (r >> 2) + 4 h + (r & 3), so that the four rows
    // 4 q + i a lane holds of both tiles are eight consecutive channels (one 16-byte store); v: plain order
    const int c = 32 * kc + 8 * (lane >> 4) + j, n = nt < 32 ? 32 * ((nt & 15) >> 1) + 8 * (r >> 2) + 4 * (nt & 1) + (r & 3) : 16 * (nt & 15) + r;
    wf[idx] = (bf16)w[c * 256 + n];
}

// x: [B*256][x_ld] bf16 (raw block input); gsc / gsh: GroupNorm (scale | shift) tables [B][256] fp32; wf: k_pack_qkv_w's output; bqk: [512] (q then k), bv: [256];
// qk: [B*256][512]; vT: [B][256][256].  grid = B, 512 threads, QKV_LDS_BYTES.
__global__ __launch_bounds__(512, 2) void k_qkv256(const bf16* __restrict__ x, int x_ld, const float* __restrict__ gsc, const float* __restrict__ gsh,
                                                  const bf16* __restrict__ wf, const float* __restrict__ bqk, const float* __restrict__ bv,
                                                  bf16* __restrict__ qk, bf16* __restrict__ vT)
{
    constexpr int T = 256, C = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_poison();
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;                                    // a block = one sample: 8 waves x 32 tokens (the weight tiles are streamed once per sample, not per half)
    const int tok0 = wave * 32;                                  // the wave's first token inside the sample

    auto issue = [&](int i) __attribute__((always_inline)) {     // weight tile i: n-tiles 2 i, 2 i + 1, all eight K steps
        unsigned char* st = smem + (i & 1) * QKV_STAGE;
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        const bf16* base = wf + (int64_t)i * (QKV_STAGE / 2);
#pragma unroll
        for (int j = 0; j < QKV_STAGE / 8192; ++j) {
            const int p = wave * (QKV_STAGE / 8192) + j;
            __builtin_amdgcn_global_load_lds(base + p * 512 + l * 8, (lds_void*)(st + p * 1024), 16, 0, 0);
        }
    };
    issue(0);
    float* const sBias = reinterpret_cast<float*>(smem + 2 * QKV_STAGE);      // [512 q | k][256 v]
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    const unsigned lds_bias = (unsigned)(uintptr_t)((lds_u8*)smem) + 2 * QKV_STAGE;
    if (tid < 192) reinterpret_cast<float4*>(sBias)[tid] = tid < 128 ? reinterpret_cast<const float4*>(bqk)[tid] : reinterpret_cast<const float4*>(bv)[tid - 128];

    // the wave's 32 tokens x 256 channels, normalised, in operand layout: hf[g][kc] = token tau(g, r) = 8 (r >> 2) + 4 g + (r & 3), channels 32 kc + 8 q .. + 7.
    // (tau: an accumulator lane holds rows 4 q + i of a 16-row tile; with this assignment the rows of groups 0 and 1 a lane holds are EIGHT consecutive
    // tokens -- one 16-byte store per lane into V^T instead of two 8-byte ones; every 128-byte line is then written in 64-byte halves)
    bf16x8 hf[2][8];
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

    int le;                                                       // (lane id again: r / q of the prologue need not stay alive through the loop)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(le));
    const int re = le & 15, qe = le >> 4;
#pragma unroll 1
    for (int t = 0; t < QKV_TILES; ++t) {
        // tile t has landed (the compiler does not track LDS-DMA completions); younger than its requests are only the (at most) two stores of tile t - 1
        if (t == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                            // ... for every wave, and everyone is done with the stage tile t + 1 goes into
        if (t + 1 < QKV_TILES) issue(t + 1);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sW = smem + (t & 1) * QKV_STAGE;
        const bool is_v = t >= 32 / QKV_NTL;                     // (uniform: n-tiles 0-15 q, 16-31 k, 32-47 v)
        if (!is_v) {
            // the tile's two n-tiles are one 32-channel group with its rows interleaved (k_pack_qkv_w): the lane ends up with EIGHT consecutive channels
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
            const int n = 32 * t + 8 * qe;                       // lane (token tau(g, re), qe): output channels n .. n + 7
            // (bias from LDS by inline asm: a read hipcc can see, next to an LDS-DMA in flight, comes with its own vmcnt(0))
            f32x4 b0, b1;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(b0), "=&v"(b1) : "v"(lds_bias + (unsigned)n * 4u) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1) :: "memory");
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                bf16x8 w;
#pragma unroll
                for (int i = 0; i < 4; ++i) { w[i] = (bf16)(a[0][g][i] + b0[i]); w[4 + i] = (bf16)(a[1][g][i] + b1[i]); }
                *reinterpret_cast<bf16x8*>(qk + ((int64_t)b * T + tok0 + 8 * (re >> 2) + 4 * g + (re & 3)) * (2 * C) + n) = w;
            }
        } else {
#pragma unroll
            for (int ntl = 0; ntl < QKV_NTL; ++ntl) {
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) {
                    const bf16x8 fb = *reinterpret_cast<const bf16x8*>(sW + (ntl * 8 + kc) * 1024 + le * 16);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf[0][kc], fb, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf[1][kc], fb, a1, 0, 0, 0);
                }
                // lane (channel re of the tile, qe): rows 4 qe + i of group 0 are tokens 8 qe + i, of group 1 tokens 8 qe + 4 + i: eight consecutive tokens
                const int ch = 16 * (QKV_NTL * t + ntl - 32) + re;
                float bb;
                asm volatile("ds_read_b32 %0, %1" : "=v"(bb) : "v"(lds_bias + (unsigned)(512 + ch) * 4u) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bb) :: "memory");
                bf16x8 w;
#pragma unroll
                for (int i = 0; i < 4; ++i) { w[i] = (bf16)(a0[i] + bb); w[4 + i] = (bf16)(a1[i] + bb); }
                *reinterpret_cast<bf16x8*>(vT + ((int64_t)b * C + ch) * T + tok0 + 8 * qe) = w;
            }
        }
    }
}

}  // namespace ncsn

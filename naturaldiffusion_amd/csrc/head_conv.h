// head_conv.h -- k_head_conv: the output head of NCSN++ in one launch: GroupNorm-apply + SiLU + 3x3 convolution 128 -> 3 channels, fp32 NCHW
// (reference: ncsnpp.py:373-381  h = act(GroupNorm(h)); h = conv3x3(h)  at full resolution).
//
// Round 2 ran it as k_gn_apply (read the raw 32x32x128 tensor, write a normalised zero-bordered copy: ~60 us at B = 512) + a 512 x 128-tile
// implicit GEMM whose N = 3 fills 3 of 128 tile columns and leaves through the scalar NCHW branch of the fp32-slab epilogue (143 us, 25 TFLOP/s).
// The op has 3.6 GFLOP and reads 134 MB: it is bound by the normalisation arithmetic and the read, not by the matrix pipe.  Here:
//   * a block owns 8 image rows x 32 pixels of one sample; per 64-channel half it loads the (8+2) x (32+2) pixel patch straight from the RAW
//     tensor (16-byte loads), normalises it (folded form: scale / shift carry -log2 e, the weights -ln 2: t / (1 + exp2 t)), rounds to bf16 and
//     parks it in LDS (rows of 64 channels padded to 144 B: conflict-free ds_read_b128 for 16 consecutive pixels), zeros outside the image;
//   * the 3 output channels ride in a 16-wide MFMA column tile (v_mfma_f32_16x16x32_bf16, B operand = the zero-padded [16][1152] weights read
//     from L2 per K step): 36 K steps x 4 pixel tiles per wave and half -- 2 % of the kernel's time;
//   * lanes 0-2 of every 16-lane group hold 4 consecutive pixels of one output channel: 16-byte NCHW stores.
#pragma once
#include "ncsnpp_kernels.h"

namespace ncsn {

struct HeadConvCfg {
    static constexpr int RES = 32, C = 128, ROWS = 8, PW = RES + 2, PR = ROWS + 2, PSTR = 144, NPIX = PR * PW;
    static constexpr int LDS_BYTES = NPIX * PSTR + 2 * 64 * 4;
};

// x: raw bf16 [B][32][32][ld]; sc / sh: [B][128] folded GroupNorm scale / shift; w: bf16 [16][1152] (rows >= 3 zero), K order
// ((c / 64) * 9 + tap) * 64 + c % 64, times -ln 2; bias [3]; out fp32 [B][3][32][32].  grid = B * 4, 256 threads.
__global__ __launch_bounds__(256, 3) void k_head_conv(const bf16* __restrict__ x, int ld, const float* __restrict__ sc, const float* __restrict__ sh,
                                                      const bf16* __restrict__ w, const float* __restrict__ bias, float* __restrict__ out)
{
    using Cfg = HeadConvCfg;
    constexpr int RES = Cfg::RES, C = Cfg::C, ROWS = Cfg::ROWS, PW = Cfg::PW, PSTR = Cfg::PSTR, NPIX = Cfg::NPIX;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_poison();
    unsigned char* const patch = smem;
    float* const tab = reinterpret_cast<float*>(smem + NPIX * PSTR);           // [scale 64 | shift 64] of the current half
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / (RES / ROWS), y0 = (blockIdx.x % (RES / ROWS)) * ROWS;
    const int r = lane & 15, q = lane >> 4;
    const bf16* const img = x + (int64_t)b * RES * RES * ld;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < 2; ++h) {
        __syncthreads();                                                         // the previous half's fragment reads are done
        if (tid < 128) tab[tid] = (tid < 64 ? sc : sh)[(int64_t)b * C + h * 64 + (tid & 63)];
        __syncthreads();
        // all of this thread's 16-byte chunks of the half are requested BEFORE the first one is used: one HBM round trip per half instead of
        // one per chunk (the first form, a load -> normalise -> store loop, ran at 1.5 TB/s: 91 us per launch at B = 512)
        constexpr int NIT = (NPIX * 8 + 255) / 256;
        uint4 raw[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            const int p = i >> 3, c8 = i & 7;
            const int yy = p / PW, xx = p - yy * PW;
            const int y = y0 - 1 + yy, xg = xx - 1;
            raw[it] = make_uint4(0u, 0u, 0u, 0u);
            if (i < NPIX * 8 && (unsigned)y < (unsigned)RES && (unsigned)xg < (unsigned)RES)
                raw[it] = *reinterpret_cast<const uint4*>(img + (int64_t)(y * RES + xg) * ld + h * 64 + c8 * 8);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            if (i >= NPIX * 8) break;
            const int p = i >> 3, c8 = i & 7;
            const int yy = p / PW, xx = p - yy * PW;
            const int y = y0 - 1 + yy, xg = xx - 1;
            uint4 o = make_uint4(0u, 0u, 0u, 0u);
            if ((unsigned)y < (unsigned)RES && (unsigned)xg < (unsigned)RES) {
                const float4 s0 = *reinterpret_cast<const float4*>(tab + c8 * 8), s1 = *reinterpret_cast<const float4*>(tab + c8 * 8 + 4);
                const float4 h0 = *reinterpret_cast<const float4*>(tab + 64 + c8 * 8), h1 = *reinterpret_cast<const float4*>(tab + 64 + c8 * 8 + 4);
                const float ss[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, hh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                const unsigned wd[4] = {raw[it].x, raw[it].y, raw[it].z, raw[it].w};
                unsigned pk[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t0 = __uint_as_float(wd[e] << 16) * ss[2 * e] + hh[2 * e];
                    const float t1 = __uint_as_float(wd[e] & 0xffff0000u) * ss[2 * e + 1] + hh[2 * e + 1];
                    const float v0 = t0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t0));
                    const float v1 = t1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t1));
                    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                    const bf16x2_t pr = {(bf16)v0, (bf16)v1};
                    pk[e] = __builtin_bit_cast(unsigned, pr);
                }
                o = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            }
            *reinterpret_cast<uint4*>(patch + p * PSTR + c8 * 16) = o;
        }
        __syncthreads();
        const bf16* const wrow = w + (int64_t)r * (9 * C) + h * 9 * 64 + q * 8;  // lane's weight row (output channel r; rows >= 3 are zero)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 bf = *reinterpret_cast<const bf16x8*>(wrow + tap * 64 + ks * 32);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int t = wave * 4 + i;                                  // pixel tile: image row y0 + (t >> 1), pixels (t & 1) * 16 .. + 15
                    const int pp = ((t >> 1) + 1 + dy) * PW + (t & 1) * 16 + r + 1 + dx;
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(patch + pp * PSTR + ks * 64 + q * 16);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[i], 0, 0, 0);
                }
            }
        }
    }
    // D[pixel 4q + e][channel r]: lanes r < 3 store four consecutive pixels of their channel
    if (r < 3) {
        const float bb = bias[r];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = wave * 4 + i;
            float* o = out + (((int64_t)b * 3 + r) * RES + y0 + (t >> 1)) * RES + (t & 1) * 16 + q * 4;
            *reinterpret_cast<float4*>(o) = make_float4(acc[i][0] + bb, acc[i][1] + bb, acc[i][2] + bb, acc[i][3] + bb);
        }
    }
}

}  // namespace ncsn

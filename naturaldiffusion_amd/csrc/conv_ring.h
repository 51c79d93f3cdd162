// conv_ring.h -- k_conv_ring: a convolution of ANY geometry (kh x kw taps, stride, zero padding, sizes that are no powers of two) as an implicit GEMM on the block
// geometry and LDS ring of k_gemm_ring (gemm_dma.h): the Inception-V3 pool3 engine's 1x7 / 7x1 / 3x3 / 5x5 / strided convolutions (inception_engine.inc).
//
// Round 5.  Until now every such convolution was k_inc_im2col + one GEMM: the im2col matrix of a forward of 500 images is written and read once per layer --
// 13.3 ms of a 43.8 ms forward in the copy kernel alone (rocprofv3, profiles/r05/inception_before_kernel_stats.csv), and the GEMMs behind it stream an operand kh * kw
// times the size of the activation tensor from HBM.  Here the A operand is the activation tensor itself:
//   * a K-tile = one tap (ky, kx) x 32 consecutive channels = 64 bytes per output pixel: the 1-KiB LDS-DMA piece of k_gemm_ring (16 rows x 64 B) with the lane's
//     row address moved by the tap -- a block-uniform byte shift -- and the lane's pixel checked against the image: a tap that falls into the zero padding
//     fetches its 16 bytes from a page of zeros instead (every lane of a piece has its own address anyway, so a mixed piece costs nothing extra);
//   * rows are decoded once per block (two integer divisions per piece: sizes like 149, 147, 71, 35, 17 have no shift form);
//   * the tap / channel-chunk counters advance with the K-tiles (requests are issued in K order), no division in the loop;
//   * the filters' two bf16 terms (W = W_hi + W_lo, inception_engine.inc) are two passes over the same taps against columns [0, Kp) and [Kp, 2 Kp) of the
//     weight rows -- the B columns are packed in K-loop order, as for k_gemm_ring;
//   * channel counts that are no multiple of 32 (48, 80) are padded by their PRODUCER: its filter rows and bias beyond the real outputs are zero, so the pad
//     channels hold relu(0) = 0 and this kernel never looks at a channel count.
//   * F16 (the Inception engine's default since round 5, natinf_set_inception_conv(2)): activations and filters are IEEE half precision and the filters ONE term.
//     Why the two-term bf16 filters existed: a bf16-rounded filter is off by the same amount for every image, which shifts the feature means -- what a Frechet
//     distance measures (0.095 of 0.099 on 2,000 images with one bf16 term).  A half-precision filter has eleven significant bits instead of eight: the same shift
//     is 8x smaller, its square 64x -- where two bf16 terms put it -- for HALF the matrix work; half-precision activations round 8x finer than bf16 ones as well.
//     Range: every filter row is scaled by a power of two so that its largest element lies in [0.5, 1) (exact; the epilogue multiplies the column back:
//     GemmArgs::deq_n), so a row keeps 2^-14 of its maximum as NORMAL numbers whatever BatchNorm folded into it; activations behind a ReLU stayed far below 65,504 on the synthetic weights (2,000 images; parity with pytorch_fid's fp32 module is UNPINNED),
//     and the epilogue saturates at 65,504 instead of overflowing to inf should real weights ever exceed it.
// Epilogue: bias (+ the column scale) + ReLU in the accumulator registers, 16-bit values through one LDS slab, 16-byte row stores into a channel slice of the
// concat buffer (NHWC).
#pragma once
#include "gemm_dma.h"

namespace ncsn {

struct ConvGeom {
    int H, W, Ho, Wo;                 // pixels per sample: input, output
    int kh, kw, stride, ph, pw;
    int ncc;                          // 32-channel chunks per tap (padded input channels / 32)
    int passes;                       // 2: hi + lo filter terms (B has 2 * kh * kw * ncc * 32 columns), 1: one term
    const void* zeros;                // >= 16 zero bytes
};

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));

// relu(acc * deq_n + bias_n) -> 16-bit -> LDS slab [BM][BN * 2 + 16 bytes] -> 16-byte row stores.  N % 8 == 0; rows >= M and columns >= N are not stored.
template <int WM, int WN, int TM, int TN, bool F16>
__device__ __forceinline__ void conv_ring_epilogue(const GemmArgs& g, unsigned char* smem, f32x4 (&acc)[TM][TN], int m0, int n0, int tid, int lane, int wm, int wn)
{
    constexpr int BM_ = WM * TM * 16, BN_ = WN * TN * 16, THREADS = WM * WN * 64, PROW = BN_ * 2 + 16;
    const int r = lane & 15, q = lane >> 4;
    float bs[TN][4], sc[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 16 + j * 16 + q * 4;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f), d = make_float4(1.f, 1.f, 1.f, 1.f);
        if (n < g.N) {
            if (g.bias_n) b = gload_f4(g.bias_n + n);
            if (g.deq_n) d = gload_f4(g.deq_n + n);
        }
        bs[j][0] = b.x; bs[j][1] = b.y; bs[j][2] = b.z; bs[j][3] = b.w; sc[j][0] = d.x; sc[j][1] = d.y; sc[j][2] = d.z; sc[j][3] = d.w;
    }
    unsigned char* wbase = smem + (wm * TM * 16 + r) * PROW + (wn * TN * 16 + q * 4) * 2;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(__builtin_fmaf(acc[i][j][e], sc[j][e], bs[j][e]), 0.f);
            uint2 o;
            // IEEE half saturates here, it does not overflow: a value above 65,504 would become inf, then NaN features and a NaN FID without a word (round-5 advisor note).
            // With the ReLU's lower bound this is one v_med3_f32 per value; finite inputs give finite activations in every layer.
            if constexpr (F16) { f16x4_t h; for (int e = 0; e < 4; ++e) h[e] = (_Float16)fminf(v[e], 65504.f); o = __builtin_bit_cast(uint2, h); }
            else { bf16x4_t h; for (int e = 0; e < 4; ++e) h[e] = (bf16)v[e]; o = __builtin_bit_cast(uint2, h); }
            *reinterpret_cast<uint2*>(wbase + i * 16 * PROW + j * 32) = o;
        }
    __syncthreads();
    constexpr int CPR = BN_ * 2 / 16;
    if constexpr (THREADS % CPR == 0) {
        constexpr int RPS = THREADS / CPR, NSW = BM_ / RPS;
        static_assert(BM_ % RPS == 0, "whole rows per sweep");
        const int cchunk = tid % CPR, rsub = tid / CPR, n = n0 + cchunk * 8;
        if (n < g.N) {
            const unsigned char* src = smem + rsub * PROW + cchunk * 16;
            unsigned short* dst = reinterpret_cast<unsigned short*>(g.c) + (int64_t)(m0 + rsub) * g.c_ld + n;
            if (m0 + BM_ <= g.M) SlabCopy<0, NSW, RPS, PROW, false>::run(src, dst, g.c_ld, m0 + rsub, g.M);
            else SlabCopy<0, NSW, RPS, PROW, true>::run(src, dst, g.c_ld, m0 + rsub, g.M);
        }
    } else {                                                            // (192-column tiles: 24 chunks per row do not divide the block)
        static_assert((BM_ * CPR) % THREADS == 0, "whole sweeps");
#pragma unroll
        for (int sw = 0; sw < BM_ * CPR / THREADS; ++sw) {
            const int e = sw * THREADS + tid, row = e / CPR, cchunk = e - row * CPR, n = n0 + cchunk * 8;
            if (m0 + row < g.M && n < g.N)
                gstore_u4(reinterpret_cast<unsigned short*>(g.c) + (int64_t)(m0 + row) * g.c_ld + n, *reinterpret_cast<const uint4*>(smem + row * PROW + cchunk * 16));
        }
    }
}

template <int WM, int WN, int TM, int TN, int NS, bool F16 = false>
__global__ __launch_bounds__(WM * WN * 64, 2) void k_conv_ring(const GemmArgs g, const ConvGeom cv)
{
    using Cfg = RingCfg<WM, WN, TM, TN, NS>;
    constexpr int BM_ = Cfg::BM_, BN_ = Cfg::BN_, BKR = Cfg::BKR, ROW = 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_poison();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nN = (g.N + BN_ - 1) / BN_, nM = (g.M + BM_ - 1) / BM_;
    const int tile = xcd_remap(blockIdx.x, nM * nN);
    // column tiles inner: the blocks that share a row panel of the activation tensor run together
    const int mt_ = tile / nN, nt_ = tile - mt_ * nN;
    const int m0 = mt_ * BM_, n0 = nt_ * BN_;

    const int ntap = cv.kh * cv.kw;
    const int nk = cv.passes * ntap * cv.ncc;
    const uint64_t zaddr = reinterpret_cast<uint64_t>(cv.zeros);

    // per piece: the byte address of (sample, oy * stride - ph, ox * stride - pw, the lane's channel chunk) -- possibly outside the tensor, only used for
    // taps that land inside -- and that pixel's coordinates
    uint64_t a_row[Cfg::PA], b_row[Cfg::PB];
    int iy0[Cfg::PA], ix0[Cfg::PA];
    const int HoWo = cv.Ho * cv.Wo;
#pragma unroll
    for (int j = 0; j < Cfg::PA; ++j) {
        const int r = (wave * Cfg::PA + j) * 16 + (lane >> 2);
        const int t4 = (r >> 2) & 3;
        const int lchunk = ((lane & 3) ^ ((0x1320 >> (t4 * 4)) & 3)) << 3;          // T = {0,2,3,1} (k_gemm_ring's LDS image)
        const int m = min(m0 + r, g.M - 1);
        const int b = m / HoWo, p = m - b * HoWo, oy = p / cv.Wo, ox = p - oy * cv.Wo;
        iy0[j] = oy * cv.stride - cv.ph; ix0[j] = ox * cv.stride - cv.pw;
        const int64_t off = (((int64_t)b * cv.H + iy0[j]) * cv.W + ix0[j]) * g.a0_ld + lchunk;
        a_row[j] = reinterpret_cast<uint64_t>(g.a0) + (uint64_t)(off * 2);
    }
#pragma unroll
    for (int j = 0; j < Cfg::PB; ++j) {
        const int r = (wave * Cfg::PB + j) * 16 + (lane >> 2);
        const int t4 = (r >> 2) & 3;
        const int lchunk = ((lane & 3) ^ ((0x1320 >> (t4 * 4)) & 3)) << 3;
        b_row[j] = reinterpret_cast<uint64_t>(g.b + (int64_t)min(n0 + r, g.N - 1) * g.b_ld + lchunk);
    }

    typedef __attribute__((address_space(3))) void lds_void;
    // the K-tile about to be requested (requests are issued in K order: the counters advance with them)
    int q_ky = 0, q_kx = 0, q_cc = 0, q_kt = 0;
    auto issue_next = [&]() __attribute__((always_inline)) {
        const int slot = q_kt % NS;
        unsigned char* dA = smem + slot * Cfg::STAGE_BYTES + wave * (Cfg::PA * 1024);
        unsigned char* dB = smem + slot * Cfg::STAGE_BYTES + BM_ * BKR * 2 + wave * (Cfg::PB * 1024);
        const int64_t ashift = ((int64_t)(q_ky * cv.W + q_kx) * g.a0_ld + q_cc * BKR) * 2;
        const uint64_t kk2 = (uint64_t)q_kt * BKR * 2;
#pragma unroll
        for (int j = 0; j < Cfg::PA; ++j) {
            const bool in = (unsigned)(iy0[j] + q_ky) < (unsigned)cv.H && (unsigned)(ix0[j] + q_kx) < (unsigned)cv.W;
            const uint64_t pa = in ? a_row[j] + (uint64_t)ashift : zaddr;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(pa), (lds_void*)(dA + j * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < Cfg::PB; ++j)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(b_row[j] + kk2), (lds_void*)(dB + j * 1024), 16, 0, 0);
        ++q_kt;
        if (++q_cc == cv.ncc) {
            q_cc = 0;
            if (++q_kx == cv.kw) { q_kx = 0; if (++q_ky == cv.kh) q_ky = 0; }      // (behind the last tap: the second filter term starts over)
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < nk) issue_next();

    const int frow = lane & 15, fq = lane >> 4;
    const int fko = (fq ^ ((0x1320 >> (((frow >> 2) & 3) * 4)) & 3)) << 3;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + NS - 1 <= nk) wait_vmcnt_barrier<(NS - 2) * Cfg::DPT>();
        else                   wait_vmcnt_barrier<0>();
        if (kt + NS - 1 < nk) issue_next();
        const int slot = kt % NS;
        const bf16* ta = reinterpret_cast<const bf16*>(smem + slot * Cfg::STAGE_BYTES) + (wm * TM * 16 + frow) * ROW + fko;
        const bf16* tb = reinterpret_cast<const bf16*>(smem + slot * Cfg::STAGE_BYTES + BM_ * BKR * 2) + (wn * TN * 16 + frow) * ROW + fko;
        bf16x8 fb[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(tb + j * 16 * ROW);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const bf16x8 fa = *reinterpret_cast<const bf16x8*>(ta + i * 16 * ROW);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (F16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fb[j]), __builtin_bit_cast(f16x8, fa), acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa, acc[i][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();               // every wave is done with the ring before the epilogue reuses it
    static_assert(BM_ * (BN_ * 2 + 16) <= Cfg::LDS_BYTES, "the 16-bit slab fits the ring");
    conv_ring_epilogue<WM, WN, TM, TN, F16>(g, smem, acc, m0, n0, tid, lane, wm, wn);
}

}  // namespace ncsn

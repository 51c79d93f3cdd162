// gemm_fp8.h -- 256x256x128 LDS-DMA GEMM on fp8 (e4m3) operands: k_gemm_fp8.
//
// C[m][n] = deq_m[m] * deq_n[n] * sum_k A8[m][k] * B8[n][k]  (+ the usual epilogue terms), A8 / B8 row-major fp8 bytes
// with one fp32 scale per row (activations: per token, written by k_ln_modulate_fp8; weights: per output channel, written
// at load time).  A K-tile is 128 bytes per row -- the SAME LDS image, DMA pieces and swizzle as the bf16 kernel's 64-wide
// tiles -- but it feeds v_mfma_f32_16x16x128_f8f6f4 (a lane supplies 32 K bytes of its row: two 16-byte LDS
// reads), which does four times the K of the bf16 MFMA in twice its time: twice the flops per byte moved and per cycle.
// Unit block scales in the MFMA (127 = 2^0); the real scales are applied once, in the epilogue.
// Two-stage pipeline like k_gemm_dma<2,4,8,4> (one K-tile of prefetch, explicit vmcnt(0) + barrier per K-tile), swapped
// operands, slab epilogue.  Plain GEMMs only (taps = 1, one K segment); K % 128 == 0; M / N edges clamped.
#pragma once
#include "gemm_dma.h"

namespace ncsn {

typedef int i32x8 __attribute__((ext_vector_type(8)));

// EPI (chosen on the host, launch_gemm_fp8): 0 = fp32-slab epilogue; 1 = packed bf16 output (row / column scales, biases);
// 2 = packed e4m3 + E8M0 output with tanh-GELU (fc1 -> fc2's operand); 3 = direct fp32 residual-stream epilogue.
template <bool MXA, int EPI = 0>     // MXA: the A operand carries E8M0 block scales (GemmArgs::a_mx), fed to the MFMA lane by lane
__global__ __launch_bounds__(512, 2) void k_gemm_fp8(const GemmArgs g)
{
    using Cfg = DmaCfg<2, 4, 8, 4>;
    constexpr int WN = 4, TM = 8, TN = 4, BM_ = 256, BN_ = 256, BKB = 128;       // BKB: K bytes (= elements) per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_poison();
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nN = (g.N + BN_ - 1) / BN_, nM = (g.M + BM_ - 1) / BM_;
    const int tile = xcd_remap(blockIdx.x, nM * nN);
    int mt_, nt_;
    tile_coords(tile, nM, nN, g.raster_g, mt_, nt_);
    const int m0 = mt_ * BM_, n0 = nt_ * BN_;
    const int z = blockIdx.z;
    const uint8_t* a0 = reinterpret_cast<const uint8_t*>(g.a0) + (int64_t)z * g.a_bs;     // strides in bytes = elements
    const uint8_t* bp = reinterpret_cast<const uint8_t*>(g.b) + (int64_t)z * g.b_bs;
    const int nk = g.a0_C / BKB;

    // one wave per SIMD (waves 0..3) issues all LDS-DMA pieces, two waves' worth each (see k_gemm_dma SPREAD 6)
    constexpr int PAI = 2 * Cfg::PA, PBI = 2 * Cfg::PB;
    const bool issuer = wave < Cfg::NW / 2;
    uint64_t a_row[PAI], b_row[PBI];
#pragma unroll
    for (int j = 0; j < PAI; ++j) {
        const int r = (wave * PAI + j) * 8 + (lane >> 3);
        a_row[j] = reinterpret_cast<uint64_t>(a0 + (int64_t)min(m0 + r, g.M - 1) * g.a0_ld + (((lane & 7) ^ ((r >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int j = 0; j < PBI; ++j) {
        const int r = (wave * PBI + j) * 8 + (lane >> 3);
        b_row[j] = reinterpret_cast<uint64_t>(bp + (int64_t)min(n0 + r, g.N - 1) * g.b_ld + (((lane & 7) ^ ((r >> 1) & 7)) << 4));
    }
    auto issue_tile = [&](int kt, int buf) __attribute__((always_inline)) {
        if (!issuer) return;
        unsigned char* dA = smem + buf * Cfg::STAGE_BYTES + wave * (PAI * 1024);
        unsigned char* dB = smem + buf * Cfg::STAGE_BYTES + BM_ * 128 + wave * (PBI * 1024);
#pragma unroll
        for (int j = 0; j < PAI; ++j)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(a_row[j] + (uint64_t)kt * BKB), (lds_void*)(dA + j * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < PBI; ++j)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(b_row[j] + (uint64_t)kt * BKB), (lds_void*)(dB + j * 1024), 16, 0, 0);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4, fswz = (frow >> 1) & 7;
    // MX scales of the A rows: stored K-TILE MAJOR (ncsnpp_kernels.h), so the 256 rows x 4 blocks of one K-tile are 1 KiB
    // of consecutive bytes -- one more DMA piece per stage (wave 0), parked behind the operand stages.  A lane then reads
    // the dword of each of its rows from LDS and uses byte fq.  (Plain per-lane loads of the scales in the K loop cost
    // 35 % of the kernel: they share vmcnt with the DMA and sit on the critical path.)
    unsigned sc[TM];
    const uint8_t* mx_src = MXA ? g.a_mx + (int64_t)z * g.a_mx_bs + (int64_t)m0 * 4 + lane * 16 : nullptr;
    auto issue_mx = [&](int kt, int buf) __attribute__((always_inline)) {
        if (wave == 0)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(mx_src + (int64_t)kt * g.a_mx_ld * 4), (lds_void*)(smem + 131072 + buf * 1024), 16, 0, 0);
    };
    if constexpr (MXA) issue_mx(0, 0);
    issue_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // The MFMA's K order (probed, tools/fp8_probe): lane group q supplies K bytes 16q..16q+15 in its first four registers
    // and 64+16q..64+16q+15 in the other four -- two stacked 64-wide halves -- and MX block j (K bytes 32j..32j+31) takes
    // its scale from lane group j.  So a lane reads the 16-byte chunks q and 4+q of its row (not 2q, 2q+1): memory-
    // contiguous 32-blocks are then the hardware's blocks.  (Without block scales any order common to A and B would do.)
    const int c0 = (fq ^ fswz) << 4, c1 = ((4 + fq) ^ fswz) << 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue_tile(kt + 1, cur ^ 1);
        if constexpr (MXA) {
            if (kt + 1 < nk) issue_mx(kt + 1, cur ^ 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
                sc[i] = (*reinterpret_cast<const unsigned*>(smem + 131072 + cur * 1024 + (wm * TM * 16 + i * 16 + frow) * 4) >> (8 * fq)) & 0xffu;
        }
        const unsigned char* ta = smem + cur * Cfg::STAGE_BYTES + (wm * TM * 16 + frow) * 128;
        const unsigned char* tb = smem + cur * Cfg::STAGE_BYTES + BM_ * 128 + (wn * TN * 16 + frow) * 128;
        i32x8 fb[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const uint4 lo = *reinterpret_cast<const uint4*>(tb + j * 2048 + c0), hi = *reinterpret_cast<const uint4*>(tb + j * 2048 + c1);
            fb[j] = i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const uint4 lo = *reinterpret_cast<const uint4*>(ta + i * 2048 + c0), hi = *reinterpret_cast<const uint4*>(ta + i * 2048 + c1);
            const i32x8 fa = i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[j], fa, acc[i][j], 0, 0, 0, 127, 0, MXA ? (int)sc[i] : 127);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if constexpr (EPI == 1) packed_tile_epilogue<2, 4, 8, 4, typename Cfg::Epi, ACT_NONE, false, false, true, false>(g, smem, acc, m0, n0, z, tid, lane, wm, wn);
    else if constexpr (EPI == 2) packed_tile_epilogue<2, 4, 8, 4, typename Cfg::Epi, ACT_GELU_TANH, false, false, true, true>(g, smem, acc, m0, n0, z, tid, lane, wm, wn);
    else if constexpr (EPI == 3) direct_f32_epilogue<2, 4, 8, 4, true>(g, acc, m0, n0, z, lane, wm, wn);
    else dma_tile_epilogue<2, 4, 8, 4, typename Cfg::Epi>(g, smem, acc, m0, n0, z, tid, lane, wm, wn);
}

// x fp32 [rows][D] -> LayerNorm (no affine, eps 1e-6), (1 + scale) / shift modulation, then fp8 e4m3 with one scale per row:
// q = round(v / s), s = max|v| / 448 (the e4m3 maximum; values are clamped, v_cvt_pk_fp8_f32 does not saturate)
__global__ __launch_bounds__(256) void k_ln_modulate_fp8(const float* __restrict__ x, const float* __restrict__ shift,
                                                         const float* __restrict__ scale, int mod_ld, uint8_t* __restrict__ h,
                                                         float* __restrict__ row_scale, int D, int64_t rows, int rows_per_sample, int x_f16 = 0)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * D;
    const _Float16* xh = reinterpret_cast<const _Float16*>(x) + row * D;      // (x_f16: the residual stream is IEEE half, read through the same pointer)
    float v[24];                                                    // D <= 1536, D % 128 == 0: two adjacent columns per lane and step
    const int n = D >> 6;
    float s = 0.f;
    for (int i = 0; i < n; i += 2) {
        float2 u;
        if (x_f16) { u.x = (float)xh[64 * i + 2 * lane]; u.y = (float)xh[64 * i + 2 * lane + 1]; }
        else u = *reinterpret_cast<const float2*>(xr + 64 * i + 2 * lane);
        v[i] = u.x; v[i + 1] = u.y; s += u.x + u.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)D;
    float qq = 0.f;
    for (int i = 0; i < n; ++i) { const float d = v[i] - mean; qq += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) qq += __shfl_xor(qq, o);
    const float rstd = 1.0f / sqrtf(qq / (float)D + 1e-6f);
    const int64_t b = row / rows_per_sample;
    const float* sh = shift + b * mod_ld; const float* sc = scale + b * mod_ld;
    float amax = 0.f;
    for (int i = 0; i < n; i += 2) {
        const int d = 64 * i + 2 * lane;
        v[i] = (v[i] - mean) * rstd * (1.0f + sc[d]) + sh[d];
        v[i + 1] = (v[i + 1] - mean) * rstd * (1.0f + sc[d + 1]) + sh[d + 1];
        amax = fmaxf(amax, fmaxf(fabsf(v[i]), fabsf(v[i + 1])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    const float qs = amax > 0.f ? amax / 448.0f : 1.0f, inv = 1.0f / qs;
    if (lane == 0) row_scale[row] = qs;
    for (int i = 0; i < n; i += 2) {
        const float a = fminf(fmaxf(v[i] * inv, -448.f), 448.f), c = fminf(fmaxf(v[i + 1] * inv, -448.f), 448.f);
        const int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, c, 0, false);
        *reinterpret_cast<uint16_t*>(h + row * D + 64 * i + 2 * lane) = (uint16_t)(w & 0xffff);
    }
}

// The same for D = NC * 256 with 16-byte loads and 4-byte stores (k_ln_modulate_v4's access pattern, dit_engine.inc)
template <int NC, bool XH = false>                                      // XH: x is IEEE half (the MMDiT engine's 16-bit image stream): 8-byte loads of four columns
__global__ __launch_bounds__(256) void k_ln_modulate_fp8_v4(const float* __restrict__ x, const float* __restrict__ shift,
                                                            const float* __restrict__ scale, int mod_ld, uint8_t* __restrict__ h,
                                                            float* __restrict__ row_scale, int64_t rows, int rows_per_sample)
{
    constexpr int D = NC * 256;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4* xr = reinterpret_cast<const float4*>(x + row * D) + lane;
    const uint2* xh = reinterpret_cast<const uint2*>(reinterpret_cast<const _Float16*>(x) + row * D) + lane;
    float4 v[NC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        if constexpr (XH) {
            typedef _Float16 f16x4_ln8 __attribute__((ext_vector_type(4)));
            const f16x4_ln8 hv = __builtin_bit_cast(f16x4_ln8, xh[64 * i]);
            v[i] = make_float4((float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]);
        } else v[i] = xr[64 * i];
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)D;
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const float a = v[i].x - mean, b_ = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        qq += (a * a + b_ * b_) + (c * c + d * d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) qq += __shfl_xor(qq, o);
    const float rstd = 1.0f / sqrtf(qq / (float)D + 1e-6f);
    const int64_t b = row / rows_per_sample;
    const float4* sh = reinterpret_cast<const float4*>(shift + b * mod_ld) + lane;
    const float4* sc = reinterpret_cast<const float4*>(scale + b * mod_ld) + lane;
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const float4 g = sc[64 * i], t = sh[64 * i];
        v[i].x = (v[i].x - mean) * rstd * (1.0f + g.x) + t.x; v[i].y = (v[i].y - mean) * rstd * (1.0f + g.y) + t.y;
        v[i].z = (v[i].z - mean) * rstd * (1.0f + g.z) + t.z; v[i].w = (v[i].w - mean) * rstd * (1.0f + g.w) + t.w;
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    const float qs = amax > 0.f ? amax / 448.0f : 1.0f, inv = 1.0f / qs;
    if (lane == 0) row_scale[row] = qs;
    unsigned* out = reinterpret_cast<unsigned*>(h + row * D) + lane;
#pragma unroll
    for (int i = 0; i < NC; ++i) out[64 * i] = pack_fp8x4(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
}
// The half stream in 16-byte accesses (D = NC8 * 512; SD3: 1536), as k_ln_modulate_h8 (dit_engine.inc): eight consecutive columns per lane and 512-column chunk, one
// 16-byte load of x and one 8-byte store of e4m3 bytes per chunk (32 us alone; ~49 us mean inside a forward, beside the text stream's launches, like the four-column form).
template <int NC8>
__global__ __launch_bounds__(256) void k_ln_modulate_fp8_h8(const float* __restrict__ x, const float* __restrict__ shift, const float* __restrict__ scale, int mod_ld,
                                                            uint8_t* __restrict__ h, float* __restrict__ row_scale, int64_t rows, int rows_per_sample)
{
    constexpr int D = NC8 * 512;
    typedef _Float16 f16x8_ln8 __attribute__((ext_vector_type(8)));
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint4* xh = reinterpret_cast<const uint4*>(reinterpret_cast<const _Float16*>(x) + row * D) + lane;
    float v[NC8][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC8; ++i) {
        const f16x8_ln8 hv = __builtin_bit_cast(f16x8_ln8, xh[64 * i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = (float)hv[e];
        s += ((v[i][0] + v[i][1]) + (v[i][2] + v[i][3])) + ((v[i][4] + v[i][5]) + (v[i][6] + v[i][7]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)D;
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < NC8; ++i) {
        float d[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = v[i][e] - mean;
        qq += ((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) + ((d[4] * d[4] + d[5] * d[5]) + (d[6] * d[6] + d[7] * d[7]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) qq += __shfl_xor(qq, o);
    const float rstd = 1.0f / sqrtf(qq / (float)D + 1e-6f);
    const int64_t b = row / rows_per_sample;
    const float4* sh = reinterpret_cast<const float4*>(shift + b * mod_ld) + 2 * lane;
    const float4* sc = reinterpret_cast<const float4*>(scale + b * mod_ld) + 2 * lane;
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NC8; ++i) {
        const float4 g0 = sc[128 * i], g1 = sc[128 * i + 1], t0 = sh[128 * i], t1 = sh[128 * i + 1];
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, tt[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[i][e] = (v[i][e] - mean) * rstd * (1.0f + gg[e]) + tt[e]; amax = fmaxf(amax, fabsf(v[i][e])); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    const float qs = amax > 0.f ? amax / 448.0f : 1.0f, inv = 1.0f / qs;
    if (lane == 0) row_scale[row] = qs;
    uint2* out = reinterpret_cast<uint2*>(h + row * D) + lane;
#pragma unroll
    for (int i = 0; i < NC8; ++i)
        out[64 * i] = make_uint2(pack_fp8x4(v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv), pack_fp8x4(v[i][4] * inv, v[i][5] * inv, v[i][6] * inv, v[i][7] * inv));
}
inline void launch_ln_modulate_fp8(const float* x, const float* shift, const float* scale, int mod_ld, uint8_t* h, float* row_scale, int D, int64_t rows,
                                   int rows_per_sample, hipStream_t s, bool x_f16 = false)
{
    const dim3 grid((unsigned)((rows + 3) / 4));
    const bool al = mod_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(shift) | reinterpret_cast<uintptr_t>(scale)) % 16 == 0;
    if (x_f16) {
        if (D == 1536 && al) hipLaunchKernelGGL(k_ln_modulate_fp8_h8<3>, grid, dim3(256), 0, s, x, shift, scale, mod_ld, h, row_scale, rows, rows_per_sample);
        else if (D == 256 && al) hipLaunchKernelGGL((k_ln_modulate_fp8_v4<1, true>), grid, dim3(256), 0, s, x, shift, scale, mod_ld, h, row_scale, rows, rows_per_sample);
        else hipLaunchKernelGGL(k_ln_modulate_fp8, grid, dim3(256), 0, s, x, shift, scale, mod_ld, h, row_scale, D, rows, rows_per_sample, 1);
        return;
    }
    if (D == 1536 && al) hipLaunchKernelGGL(k_ln_modulate_fp8_v4<6>, grid, dim3(256), 0, s, x, shift, scale, mod_ld, h, row_scale, rows, rows_per_sample);
    else if (D == 256 && al) hipLaunchKernelGGL(k_ln_modulate_fp8_v4<1>, grid, dim3(256), 0, s, x, shift, scale, mod_ld, h, row_scale, rows, rows_per_sample);
    else hipLaunchKernelGGL(k_ln_modulate_fp8, grid, dim3(256), 0, s, x, shift, scale, mod_ld, h, row_scale, D, rows, rows_per_sample, 0);
}

// weights: W fp32 [N][K] -> fp8 bytes [N][K] + one scale per output channel
__global__ __launch_bounds__(256) void k_pack_fp8_rows(const float* __restrict__ W, uint8_t* __restrict__ q, float* __restrict__ row_scale, int N, int K)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float* w = W + (int64_t)row * K;
    float amax = 0.f;
    for (int k = lane; k < K; k += 64) amax = fmaxf(amax, fabsf(w[k]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    const float qs = amax > 0.f ? amax / 448.0f : 1.0f, inv = 1.0f / qs;
    if (lane == 0) row_scale[row] = qs;
    for (int k = 2 * lane; k < K; k += 128) {
        const float a = fminf(fmaxf(w[k] * inv, -448.f), 448.f), c = fminf(fmaxf(w[k + 1] * inv, -448.f), 448.f);
        *reinterpret_cast<uint16_t*>(q + (int64_t)row * K + k) = (uint16_t)(__builtin_amdgcn_cvt_pk_fp8_f32(a, c, 0, false) & 0xffff);
    }
}

}  // namespace ncsn

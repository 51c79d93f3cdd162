// ni_step.hip -- the Natural Inference recurrence as fused gfx950 streaming kernels.
//
// One launch per sampling step does everything the reference does between two denoiser
// calls: model output -> x0_hat, append to the history slab, coefficient-row weighted sum
// over the history (+ noise mixing), cast.  The kernels are HBM-bound (0.1-0.25 flop/B), so
// the design rules are the streaming ones: 16-byte vector accesses, one coalesced stream per
// history row ([slot][E] slab), coefficients through the scalar cache (wave-uniform s_load, no
// LDS needed), unrolled term loop so several row loads are in flight, <= 2048 resident blocks
// with a grid-stride loop.
//
// Arithmetic contract (include/natinf.h): operand types and operation ORDER of the reference,
// one IEEE rounding per reference operation.  The file is compiled -ffp-contract=off and the
// pragma below repeats it, so no mul+add pair is ever fused.
//
// Reference lines replaced: src/CIFAR10NaturalInference.py:219-238,299-304;
// src/ValidateNaturalInference.py:193,198-204,355,362-366; src/SD3NaturalInference.py:61-69,
// 117-129,157-168,209,215-219; deps/score_sde_pytorch/models/utils.py:157.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "natinf.h"

#pragma clang fp contract(off)

namespace {

constexpr int kBlock = 256;
constexpr int kMaxGrid = 2048;            // 256 CUs x 8 blocks: enough to fill the chip, grid-stride the rest

inline int grid_for(int64_t nvec) {
    int64_t g = (nvec + kBlock - 1) / kBlock;
    return (int)(g < 1 ? 1 : (g > kMaxGrid ? kMaxGrid : g));
}

struct alignas(16) d2 { double x, y; };
typedef _Float16 h16;
struct alignas(16) h8 { h16 v[8]; };

// fp32 product of an fp32 scalar and an fp16 value, rounded to fp32 and THEN to fp16 (what eager
// PyTorch does).  The empty asm pins the fp32 product in a VGPR so the backend cannot select
// v_fma_mixlo_f16 (x*y + (+0)), whose +0 addend would turn a -0 product into +0.
__device__ __forceinline__ h16 hmulf(float s, h16 a) {
    float p = s * (float)a;
    asm("" : "+v"(p));
    return (h16)p;
}
__device__ __forceinline__ h16 hadd(h16 a, h16 b) { return (h16)((float)a + (float)b); }
__device__ __forceinline__ h16 hsub(h16 a, h16 b) { return (h16)((float)a - (float)b); }

// acc <- fp16 chain over the sparse row; hand-unrolled by 4 so four row loads are in flight
// (the asm pin in hmulf keeps the compiler from unrolling the loop itself).
__device__ __forceinline__ void chain_terms(h8& acc, const h16* __restrict__ hist, const int32_t* __restrict__ idx,
                                            const float* __restrict__ val, int n_terms, int64_t v, int64_t E)
{
    int t = 0;
    for (; t + 4 <= n_terms; t += 4) {
        const float c0 = val[t], c1 = val[t + 1], c2 = val[t + 2], c3 = val[t + 3];
        const h8 h0 = reinterpret_cast<const h8*>(hist + (int64_t)idx[t] * E)[v];
        const h8 h1 = reinterpret_cast<const h8*>(hist + (int64_t)idx[t + 1] * E)[v];
        const h8 h2 = reinterpret_cast<const h8*>(hist + (int64_t)idx[t + 2] * E)[v];
        const h8 h3 = reinterpret_cast<const h8*>(hist + (int64_t)idx[t + 3] * E)[v];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc.v[i] = hadd(acc.v[i], hmulf(c0, h0.v[i]));
            acc.v[i] = hadd(acc.v[i], hmulf(c1, h1.v[i]));
            acc.v[i] = hadd(acc.v[i], hmulf(c2, h2.v[i]));
            acc.v[i] = hadd(acc.v[i], hmulf(c3, h3.v[i]));
        }
    }
    for (; t < n_terms; ++t) {
        const float c = val[t];
        const h8 h = reinterpret_cast<const h8*>(hist + (int64_t)idx[t] * E)[v];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc.v[i] = hadd(acc.v[i], hmulf(c, h.v[i]));
    }
}

// ------------------------------------------------------------------------------------------
// CIFAR10 form, fp64 history
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_step_f64hist(
    const float4* __restrict__ x_k, const float4* __restrict__ mout, const float4* __restrict__ noise,
    double* __restrict__ hist, float4* __restrict__ x_next,
    const int32_t* __restrict__ idx, const double* __restrict__ val, int n_terms, double c_diag,
    int k, double alpha, double sigma2, float stdv, float b0, int64_t nvec, int64_t E)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        const float4 xv = x_k[v], ov = mout[v], nv = noise[v];
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        const float os[4] = {ov.x, ov.y, ov.z, ov.w};
        const float ns[4] = {nv.x, nv.y, nv.z, nv.w};
        double x0[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s = (-os[i]) / stdv;                          // score, fp32
            x0[i] = ((double)s * sigma2 + (double)xs[i]) / alpha;     // fp64, three roundings
        }
        d2* hk = reinterpret_cast<d2*>(hist + (int64_t)k * E) + 2 * v;
        hk[0] = d2{x0[0], x0[1]};
        hk[1] = d2{x0[2], x0[3]};

        double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (int t = 0; t < n_terms; ++t) {
            const double c = val[t];
            const d2* hj = reinterpret_cast<const d2*>(hist + (int64_t)idx[t] * E) + 2 * v;
            const d2 a = hj[0], b = hj[1];
            acc[0] = acc[0] + a.x * c; acc[1] = acc[1] + a.y * c;
            acc[2] = acc[2] + b.x * c; acc[3] = acc[3] + b.y * c;
        }
        float r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = acc[i] + x0[i] * c_diag;
            r[i] = (float)acc[i] + b0 * ns[i];
        }
        x_next[v] = make_float4(r[0], r[1], r[2], r[3]);
    }
}

__global__ __launch_bounds__(kBlock) void k_wsum_f64(
    const double* __restrict__ hist, float4* __restrict__ out,
    const int32_t* __restrict__ idx, const double* __restrict__ val, int n_terms, int64_t nvec, int64_t E)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (int t = 0; t < n_terms; ++t) {
            const double c = val[t];
            const d2* hj = reinterpret_cast<const d2*>(hist + (int64_t)idx[t] * E) + 2 * v;
            const d2 a = hj[0], b = hj[1];
            acc[0] = acc[0] + a.x * c; acc[1] = acc[1] + a.y * c;
            acc[2] = acc[2] + b.x * c; acc[3] = acc[3] + b.y * c;
        }
        out[v] = make_float4((float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]);
    }
}

// fast mode: fp32 history, fp32 FMA accumulate (contraction re-enabled locally)
__global__ __launch_bounds__(kBlock) void k_step_f32hist(
    const float4* __restrict__ x_k, const float4* __restrict__ mout, const float4* __restrict__ noise,
    float* __restrict__ hist, float4* __restrict__ x_next,
    const int32_t* __restrict__ idx, const float* __restrict__ val, int n_terms, float c_diag,
    int k, float inv_alpha, float sigma2_over_std, float b0, int64_t nvec, int64_t E)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        const float4 xv = x_k[v], ov = mout[v], nv = noise[v];
        float4 x0;
        x0.x = __builtin_fmaf(-ov.x, sigma2_over_std, xv.x) * inv_alpha;
        x0.y = __builtin_fmaf(-ov.y, sigma2_over_std, xv.y) * inv_alpha;
        x0.z = __builtin_fmaf(-ov.z, sigma2_over_std, xv.z) * inv_alpha;
        x0.w = __builtin_fmaf(-ov.w, sigma2_over_std, xv.w) * inv_alpha;
        reinterpret_cast<float4*>(hist + (int64_t)k * E)[v] = x0;
        float4 acc = make_float4(b0 * nv.x, b0 * nv.y, b0 * nv.z, b0 * nv.w);
#pragma unroll 4
        for (int t = 0; t < n_terms; ++t) {
            const float c = val[t];
            const float4 h = reinterpret_cast<const float4*>(hist + (int64_t)idx[t] * E)[v];
            acc.x = __builtin_fmaf(h.x, c, acc.x); acc.y = __builtin_fmaf(h.y, c, acc.y);
            acc.z = __builtin_fmaf(h.z, c, acc.z); acc.w = __builtin_fmaf(h.w, c, acc.w);
        }
        acc.x = __builtin_fmaf(x0.x, c_diag, acc.x); acc.y = __builtin_fmaf(x0.y, c_diag, acc.y);
        acc.z = __builtin_fmaf(x0.z, c_diag, acc.z); acc.w = __builtin_fmaf(x0.w, c_diag, acc.w);
        x_next[v] = acc;
    }
}

__global__ __launch_bounds__(kBlock) void k_to_pixel(
    const float* __restrict__ x, uint8_t* __restrict__ out, int C, int HW, int centered)
{
    // one sample per blockIdx.y; out[(p*C + c)] <- in[c*HW + p]: the C reads of a pixel are HW floats apart
    // (same few cache lines across neighbouring lanes), the byte writes are fully coalesced.
    const float* xs = x + (int64_t)blockIdx.y * C * HW;
    uint8_t* os = out + (int64_t)blockIdx.y * C * HW;
    const int n = C * HW;
    for (int o = blockIdx.x * kBlock + threadIdx.x; o < n; o += gridDim.x * kBlock) {
        const int p = o / C, c = o - p * C;
        float v = xs[c * HW + p];
        if (centered) v = (v + 1.0f) / 2.0f;                 // inverse scaler, datasets.py:32-38
        v = v * 255.0f;
        v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
        os[o] = (uint8_t)(int)v;
    }
}

// ------------------------------------------------------------------------------------------
// Validate form: fp32 products, fp64 accumulate
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ld_eps(const float* p, int64_t v, int64_t svec, int64_t sstride) {
    // vector v of the logical [B][sample_elems] tensor inside a [B][sstride] buffer
    const int64_t n = v / svec, r = v - n * svec;
    return *reinterpret_cast<const float4*>(p + n * sstride + 4 * r);
}

__global__ __launch_bounds__(kBlock) void k_step_f32prod(
    const float4* __restrict__ z, const float* __restrict__ cond, const float* __restrict__ uncond, float cfg,
    int64_t svec, int64_t sstride,
    float* __restrict__ hist_x0, const float* __restrict__ hist_eps, float4* __restrict__ z_next,
    const int32_t* __restrict__ idx_c, const float* __restrict__ val_c, int n_c, float c_diag,
    const int32_t* __restrict__ idx_b, const float* __restrict__ val_b, int n_b,
    int k, float c1, float c2, int64_t nvec, int64_t E)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        const float4 zv = z[v];
        float4 ev = ld_eps(cond, v, svec, sstride);
        if (uncond) {
            const float4 uv = ld_eps(uncond, v, svec, sstride);
            float d, m;
            d = ev.x - uv.x; m = cfg * d; ev.x = uv.x + m;
            d = ev.y - uv.y; m = cfg * d; ev.y = uv.y + m;
            d = ev.z - uv.z; m = cfg * d; ev.z = uv.z + m;
            d = ev.w - uv.w; m = cfg * d; ev.w = uv.w + m;
        }
        float4 x0;
        { const float p = c1 * zv.x, q = c2 * ev.x; x0.x = p - q; }
        { const float p = c1 * zv.y, q = c2 * ev.y; x0.y = p - q; }
        { const float p = c1 * zv.z, q = c2 * ev.z; x0.z = p - q; }
        { const float p = c1 * zv.w, q = c2 * ev.w; x0.w = p - q; }
        reinterpret_cast<float4*>(hist_x0 + (int64_t)k * E)[v] = x0;

        double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (int t = 0; t < n_c; ++t) {
            const float c = val_c[t];
            const float4 h = reinterpret_cast<const float4*>(hist_x0 + (int64_t)idx_c[t] * E)[v];
            const float p0 = h.x * c, p1 = h.y * c, p2 = h.z * c, p3 = h.w * c;
            a[0] = a[0] + (double)p0; a[1] = a[1] + (double)p1; a[2] = a[2] + (double)p2; a[3] = a[3] + (double)p3;
        }
        {
            const float p0 = x0.x * c_diag, p1 = x0.y * c_diag, p2 = x0.z * c_diag, p3 = x0.w * c_diag;
            a[0] = a[0] + (double)p0; a[1] = a[1] + (double)p1; a[2] = a[2] + (double)p2; a[3] = a[3] + (double)p3;
        }
#pragma unroll 4
        for (int t = 0; t < n_b; ++t) {
            const float c = val_b[t];
            const float4 h = reinterpret_cast<const float4*>(hist_eps + (int64_t)idx_b[t] * E)[v];
            const float p0 = h.x * c, p1 = h.y * c, p2 = h.z * c, p3 = h.w * c;
            b[0] = b[0] + (double)p0; b[1] = b[1] + (double)p1; b[2] = b[2] + (double)p2; b[3] = b[3] + (double)p3;
        }
        z_next[v] = make_float4((float)a[0] + (float)b[0], (float)a[1] + (float)b[1],
                                (float)a[2] + (float)b[2], (float)a[3] + (float)b[3]);
    }
}

__global__ __launch_bounds__(kBlock) void k_wsum_f32prod(
    const float* __restrict__ hist, float4* __restrict__ out,
    const int32_t* __restrict__ idx, const float* __restrict__ val, int n_terms, int64_t nvec, int64_t E)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (int t = 0; t < n_terms; ++t) {
            const float c = val[t];
            const float4 h = reinterpret_cast<const float4*>(hist + (int64_t)idx[t] * E)[v];
            const float p0 = h.x * c, p1 = h.y * c, p2 = h.z * c, p3 = h.w * c;
            a[0] = a[0] + (double)p0; a[1] = a[1] + (double)p1; a[2] = a[2] + (double)p2; a[3] = a[3] + (double)p3;
        }
        out[v] = make_float4((float)a[0], (float)a[1], (float)a[2], (float)a[3]);
    }
}

// ------------------------------------------------------------------------------------------
// SD3 form: all-fp16 chain (every op = fp32 math on fp16 operands, rounded to fp16)
// ------------------------------------------------------------------------------------------
template <bool kVelocityCfg>
__global__ __launch_bounds__(kBlock) void k_step_f16chain(
    const h8* __restrict__ x, const h8* __restrict__ v_text, const h8* __restrict__ v_null,
    const h8* __restrict__ noise, h16* __restrict__ hist, h8* __restrict__ mean_out, h8* __restrict__ x_next,
    const int32_t* __restrict__ idx, const float* __restrict__ val, int n_terms, float c_diag, float w_total,
    int k, float sig, float sig_next, float oms_next, float cfg, int64_t nvec, int64_t E)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        const h8 xv = x[v], tv = v_text[v], uv = v_null[v];
        h8 f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (kVelocityCfg) {
                const h16 d = hsub(tv.v[i], uv.v[i]);
                const h16 vv = hadd(uv.v[i], hmulf(cfg, d));
                f.v[i] = hsub(xv.v[i], hmulf(sig, vv));
            } else {
                const h16 x0n = hsub(xv.v[i], hmulf(sig, uv.v[i]));
                const h16 x0t = hsub(xv.v[i], hmulf(sig, tv.v[i]));
                const h16 d = hsub(x0t, x0n);
                f.v[i] = hadd(x0n, hmulf(cfg, d));
            }
        }
        reinterpret_cast<h8*>(hist + (int64_t)k * E)[v] = f;

        h8 acc;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc.v[i] = (h16)0.0f;
        chain_terms(acc, hist, idx, val, n_terms, v, E);
        h8 mean;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc.v[i] = hadd(acc.v[i], hmulf(c_diag, f.v[i]));
            mean.v[i] = (h16)((float)acc.v[i] / w_total);
        }
        if (mean_out) mean_out[v] = mean;
        if (x_next) {
            const h8 nz = noise[v];
            h8 r;
#pragma unroll
            for (int i = 0; i < 8; ++i) r.v[i] = hadd(hmulf(sig_next, nz.v[i]), hmulf(oms_next, mean.v[i]));
            x_next[v] = r;
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_flow_input_f16(
    const h8* __restrict__ noise, const h8* __restrict__ mean, h8* __restrict__ out, float sig, float oms, int64_t nvec)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        const h8 nz = noise[v];
        h8 m, r;
        if (mean) m = mean[v];
        else {
#pragma unroll
            for (int i = 0; i < 8; ++i) m.v[i] = (h16)0.0f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r.v[i] = hadd(hmulf(sig, nz.v[i]), hmulf(oms, m.v[i]));
        out[v] = r;
    }
}

__global__ __launch_bounds__(kBlock) void k_wmean_f16(
    const h16* __restrict__ hist, h8* __restrict__ out,
    const int32_t* __restrict__ idx, const float* __restrict__ val, int n_terms, float w_total,
    int64_t nvec, int64_t E)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        h8 acc;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc.v[i] = (h16)0.0f;
        chain_terms(acc, hist, idx, val, n_terms, v, E);
        h8 mean;
#pragma unroll
        for (int i = 0; i < 8; ++i) mean.v[i] = (h16)((float)acc.v[i] / w_total);
        out[v] = mean;
    }
}

// ------------------------------------------------------------------------------------------
// Counter-based initial noise: Philox4x32-10 keyed by (seed, global image index).  The reference draws one
// sequential torch.randn stream (src/CIFAR10NaturalInference.py:285-290), which cannot be sharded; keying
// the generator by the image's GLOBAL index makes image i identical for any GPU count / batch split.
// counter = (index lo, index hi, element quad, 0), key = (seed lo, seed hi); 4 x uint32 -> 4 normals (Box-Muller).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&o)[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__global__ __launch_bounds__(kBlock) void k_randn_philox(
    float4* __restrict__ out, const int64_t* __restrict__ index, int64_t first_index, int64_t index_stride,
    int64_t quads_per_image, int64_t total_quads, uint32_t k0, uint32_t k1)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < total_quads; v += stride) {
        const int64_t img = v / quads_per_image, q = v - img * quads_per_image;
        const uint64_t gi = (uint64_t)(index ? index[img] : first_index + img * index_stride);
        uint32_t r[4];
        philox4x32_10((uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)q, (uint32_t)(q >> 32), k0, k1, r);
        float z[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = ((float)(r[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);       // (0, 1), 24 bits
            const float u2 = ((float)(r[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float rad = sqrtf(-2.0f * logf(u1));
            float sn, cs;
            sincosf(6.28318530717958647692f * u2, &sn, &cs);
            z[2 * h] = rad * cs; z[2 * h + 1] = rad * sn;
        }
        out[v] = make_float4(z[0], z[1], z[2], z[3]);
    }
}

inline int launched() { return hipGetLastError() == hipSuccess ? NATINF_OK : NATINF_ELAUNCH; }
inline bool terms_ok(const void* idx, const void* val, int n) { return n >= 0 && (n == 0 || (idx && val)); }

}  // namespace

extern "C" {

int natinf_abi_version(void) { return NATINF_ABI_VERSION; }

const char* natinf_strerror(int code) {
    switch (code) {
        case NATINF_OK: return "ok";
        case NATINF_EINVAL: return "invalid argument";
        case NATINF_ELAUNCH: return "HIP launch failed";
        case NATINF_ENODEV: return "no gfx950 device or code object";
        case NATINF_ESTATE: return "handle in wrong state";
        default: return "unknown natinf error";
    }
}

int natinf_probe(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return NATINF_ENODEV; }
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_wsum_f64)) != hipSuccess) {
        (void)hipGetLastError();
        return NATINF_ENODEV;
    }
    return NATINF_OK;
}

int natinf_step_f64hist(const float* x_k, const float* model_out, const float* noise,
                        double* hist, float* x_next,
                        const int32_t* idx, const double* val, int n_terms, double c_diag,
                        int k, double alpha, double sigma, float std_f32, float b0_f32,
                        int64_t E, natinf_stream_t stream)
{
    if (!x_k || !model_out || !noise || !hist || !x_next || !terms_ok(idx, val, n_terms) || k < 0 || E <= 0 || (E & 3))
        return NATINF_EINVAL;
    const int64_t nvec = E / 4;
    hipLaunchKernelGGL(k_step_f64hist, dim3(grid_for(nvec)), dim3(kBlock), 0, (hipStream_t)stream,
                       (const float4*)x_k, (const float4*)model_out, (const float4*)noise, hist, (float4*)x_next,
                       idx, val, n_terms, c_diag, k, alpha, sigma * sigma, std_f32, b0_f32, nvec, E);
    return launched();
}

int natinf_step_f32hist(const float* x_k, const float* model_out, const float* noise,
                        float* hist, float* x_next,
                        const int32_t* idx, const float* val, int n_terms, float c_diag,
                        int k, float alpha, float sigma, float std_f32, float b0_f32,
                        int64_t E, natinf_stream_t stream)
{
    if (!x_k || !model_out || !noise || !hist || !x_next || !terms_ok(idx, val, n_terms) || k < 0 || E <= 0 || (E & 3))
        return NATINF_EINVAL;
    const int64_t nvec = E / 4;
    hipLaunchKernelGGL(k_step_f32hist, dim3(grid_for(nvec)), dim3(kBlock), 0, (hipStream_t)stream,
                       (const float4*)x_k, (const float4*)model_out, (const float4*)noise, hist, (float4*)x_next,
                       idx, val, n_terms, c_diag, k, 1.0f / alpha, sigma * sigma / std_f32, b0_f32, nvec, E);
    return launched();
}

int natinf_randn_philox_f32(float* out, int64_t n_images, int64_t elems_per_image, const int64_t* image_index,
                            int64_t first_index, int64_t index_stride, uint64_t seed, natinf_stream_t stream)
{
    if (!out || n_images <= 0 || elems_per_image <= 0 || (elems_per_image & 3)) return NATINF_EINVAL;
    const int64_t qpi = elems_per_image / 4, total = qpi * n_images;
    hipLaunchKernelGGL(k_randn_philox, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, (float4*)out,
                       image_index, first_index, index_stride, qpi, total, (uint32_t)seed, (uint32_t)(seed >> 32));
    return launched();
}

int natinf_weighted_sum_f64(const double* hist, float* out, const int32_t* idx, const double* val, int n_terms,
                            int64_t E, natinf_stream_t stream)
{
    if (!hist || !out || !terms_ok(idx, val, n_terms) || E <= 0 || (E & 3)) return NATINF_EINVAL;
    const int64_t nvec = E / 4;
    hipLaunchKernelGGL(k_wsum_f64, dim3(grid_for(nvec)), dim3(kBlock), 0, (hipStream_t)stream,
                       hist, (float4*)out, idx, val, n_terms, nvec, E);
    return launched();
}

int natinf_to_pixel_u8(const float* x, uint8_t* out, int B, int C, int H, int W, int centered, natinf_stream_t stream)
{
    if (!x || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0) return NATINF_EINVAL;
    if ((int64_t)C * H * W > (1 << 30) || B > 65535) return NATINF_EINVAL;
    const int per = C * H * W;
    int gx = (per + kBlock - 1) / kBlock;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_to_pixel, dim3(gx, B), dim3(kBlock), 0, (hipStream_t)stream, x, out, C, H * W, centered);
    return launched();
}

int natinf_step_f32prod(const float* z, const float* cond, const float* uncond, float cfg,
                        int64_t sample_elems, int64_t eps_sample_stride,
                        float* hist_x0, const float* hist_eps, float* z_next,
                        const int32_t* idx_c, const float* val_c, int n_c, float c_diag,
                        const int32_t* idx_b, const float* val_b, int n_b,
                        int k, float c1_f32, float c2_f32, int64_t E, natinf_stream_t stream)
{
    if (!z || !cond || !hist_x0 || !hist_eps || !z_next || !terms_ok(idx_c, val_c, n_c) || !terms_ok(idx_b, val_b, n_b) ||
        k < 0 || E <= 0 || (E & 3) || sample_elems <= 0 || (sample_elems & 3) || (E % sample_elems) ||
        eps_sample_stride < sample_elems || (eps_sample_stride & 3))
        return NATINF_EINVAL;
    const int64_t nvec = E / 4;
    hipLaunchKernelGGL(k_step_f32prod, dim3(grid_for(nvec)), dim3(kBlock), 0, (hipStream_t)stream,
                       (const float4*)z, cond, uncond, cfg, sample_elems / 4, eps_sample_stride,
                       hist_x0, hist_eps, (float4*)z_next, idx_c, val_c, n_c, c_diag, idx_b, val_b, n_b,
                       k, c1_f32, c2_f32, nvec, E);
    return launched();
}

int natinf_weighted_sum_f32prod(const float* hist, float* out, const int32_t* idx, const float* val, int n_terms,
                                int64_t E, natinf_stream_t stream)
{
    if (!hist || !out || !terms_ok(idx, val, n_terms) || E <= 0 || (E & 3)) return NATINF_EINVAL;
    const int64_t nvec = E / 4;
    hipLaunchKernelGGL(k_wsum_f32prod, dim3(grid_for(nvec)), dim3(kBlock), 0, (hipStream_t)stream,
                       hist, (float4*)out, idx, val, n_terms, nvec, E);
    return launched();
}

int natinf_step_f16chain(const void* x, const void* v_text, const void* v_null, const void* noise,
                         void* hist, void* mean_out, void* x_next,
                         const int32_t* idx, const float* val, int n_terms, float c_diag, float w_total,
                         int k, float sig, float sig_next, float one_minus_sig_next, float cfg,
                         int flags, int64_t E, natinf_stream_t stream)
{
    if (!x || !v_text || !v_null || !hist || !terms_ok(idx, val, n_terms) || k < 0 || E <= 0 || (E & 7) ||
        (x_next && !noise) || (flags & ~NATINF_SD3_CFG_ON_VELOCITY))
        return NATINF_EINVAL;
    const int64_t nvec = E / 8;
    if (flags & NATINF_SD3_CFG_ON_VELOCITY)
        hipLaunchKernelGGL(k_step_f16chain<true>, dim3(grid_for(nvec)), dim3(kBlock), 0, (hipStream_t)stream,
                           (const h8*)x, (const h8*)v_text, (const h8*)v_null, (const h8*)noise, (h16*)hist,
                           (h8*)mean_out, (h8*)x_next, idx, val, n_terms, c_diag, w_total, k, sig, sig_next,
                           one_minus_sig_next, cfg, nvec, E);
    else
        hipLaunchKernelGGL(k_step_f16chain<false>, dim3(grid_for(nvec)), dim3(kBlock), 0, (hipStream_t)stream,
                           (const h8*)x, (const h8*)v_text, (const h8*)v_null, (const h8*)noise, (h16*)hist,
                           (h8*)mean_out, (h8*)x_next, idx, val, n_terms, c_diag, w_total, k, sig, sig_next,
                           one_minus_sig_next, cfg, nvec, E);
    return launched();
}

int natinf_flow_input_f16(const void* noise, const void* mean, void* out, float sig, float one_minus_sig,
                          int64_t E, natinf_stream_t stream)
{
    if (!noise || !out || E <= 0 || (E & 7)) return NATINF_EINVAL;
    const int64_t nvec = E / 8;
    hipLaunchKernelGGL(k_flow_input_f16, dim3(grid_for(nvec)), dim3(kBlock), 0, (hipStream_t)stream,
                       (const h8*)noise, (const h8*)mean, (h8*)out, sig, one_minus_sig, nvec);
    return launched();
}

int natinf_weighted_mean_f16(const void* hist, void* out, const int32_t* idx, const float* val, int n_terms,
                             float w_total, int64_t E, natinf_stream_t stream)
{
    if (!hist || !out || !terms_ok(idx, val, n_terms) || E <= 0 || (E & 7)) return NATINF_EINVAL;
    const int64_t nvec = E / 8;
    hipLaunchKernelGGL(k_wmean_f16, dim3(grid_for(nvec)), dim3(kBlock), 0, (hipStream_t)stream,
                       (const h16*)hist, (h8*)out, idx, val, n_terms, w_total, nvec, E);
    return launched();
}

}  // extern "C"

// ncsnpp_kernels.h -- gfx950 kernels of the NCSN++ denoiser engine.
//
// Layout: every activation is NHWC bf16 ("pixel rows" of C channels, pixel stride ld >= C so a
// tensor can live inside a wider concat buffer).  With channels innermost, a 3x3 convolution is an
// implicit GEMM  C[m, n] = sum_k A[m, k] * Wp[n, k]  with m = (b, y, x), k = (tap, c): each K-tile of
// 64 channels of one tap is a contiguous 128-byte run per pixel row.  All matmul-shaped work of the
// network (3x3 / 1x1 convolutions, NIN, linear layers, attention products) goes through ONE
// MFMA kernel, `k_gemm_bf16`, with fused epilogues.
//
//   tile 128(M) x 128(N) x 64(K), 256 threads = 4 waves in 2x2, each wave 64x64 via 4x4
//   v_mfma_f32_16x16x32_bf16; LDS double buffer, 128-byte rows with the 16-byte chunk index XOR-ed by
//   (row>>1)&7 (conflict-free for ds_read_b128's 16-lane groups: worked out in DESIGN.md);
//   global->register prefetch of tile k+1 while tile k is multiplied; one barrier per K-tile;
//   epilogue staged through LDS as fp32 so bias / time-embedding / residual adds happen in fp32 and
//   stores are 16-byte coalesced; XCD-aware tile order (neighbouring N-tiles of one M-tile share an L2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace ncsn {

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Debug build (-DNATINF_ASM_PAD): wait states in front of and behind every hand-written memory instruction.  Padding cannot change a
// result unless a hazard the hardware does not interlock (and hipcc cannot see inside inline asm) is being hit.
#ifdef NATINF_ASM_PAD
#define NATINF_PAD_PRE "s_nop 7\n\ts_nop 7\n\t"
#define NATINF_PAD_POST "\n\ts_nop 7\n\ts_nop 7"
#else
#define NATINF_PAD_PRE ""
#define NATINF_PAD_POST ""
#endif

// An LDS-DMA statement that writes M0 names it as clobbered: hipcc keeps values of its own there (spill code), and a -DNATINF_DEV build faulted until the clobber
// was declared.  clang answers every such statement, in every instantiation, with "inline asm clobber list contains reserved registers: m0"
// -- 2,260 lines per build that buried anything new (round-5 review, item 8).  The clobber is deliberate; the warning is silenced around exactly these statements.
#define NATINF_M0_ASM_BEGIN _Pragma("clang diagnostic push") _Pragma("clang diagnostic ignored \"-Winline-asm\"")
#define NATINF_M0_ASM_END _Pragma("clang diagnostic pop")

// Debug build (-DNATINF_LDS_POISON): every kernel first fills the whole 160-KiB LDS address range of its workgroup with NaN
// patterns (writes past the allocation are dropped by the hardware), so that a read of LDS the block has not written itself
// -- whatever the previous workgroup on that compute unit left there -- shows up as NaN in the parity tests.
__device__ __forceinline__ void lds_poison()
{
#ifdef NATINF_LDS_POISON
    typedef uint32_t poison_u4 __attribute__((ext_vector_type(4)));
    const poison_u4 v = {0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u};
    for (uint32_t a = threadIdx.x * 16; a < 163840u; a += blockDim.x * 16)
        asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(v) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
#endif
}

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDS_ROW = BK;                           // 64 bf16 = 128 B rows, chunk-swizzled
constexpr int TILE_ELEMS = BM * LDS_ROW;              // per operand per buffer
constexpr int C_ROW = BN + 4;                         // fp32 epilogue staging row stride
constexpr int GEMM_LDS_BYTES = BM * C_ROW * 4;        // 67,584 B: max(tiles 65,536 B, epilogue staging)
static_assert(4 * TILE_ELEMS * 2 <= GEMM_LDS_BYTES, "tile buffers must fit");

enum { OUT_BF16 = 0, OUT_F32 = 1, OUT_F32_NCHW = 2, OUT_FP8_MX = 3 };      // OUT_FP8_MX: e4m3 bytes + a block scale per 32 columns
enum { ACT_NONE = 0, ACT_SILU = 1, ACT_GELU_TANH = 2, ACT_RELU = 3 };      // ACT_RELU: general (fp32-slab) epilogue only -- the Inception engine

struct GemmArgs {
    // A operand: segment 0 = `taps` (1 or 9) shifted views of a0, segment 1 = a1 (1x1), concatenated along K
    const bf16* a0; int a0_ld; int a0_C;
    const bf16* a1; int a1_ld; int a1_C;
    int taps; int logW; int logHW;                    // spatial decode of m for taps == 9 (power-of-two H, W)
    int a0_padded;                                    // taps == 9: a0 is [B][H+2][W+2][C] with a zero border (k_gemm_bf16_dma only)
    int M, N;
    const bf16* b; int b_ld;                          // [N][K0+K1], K contiguous
    int64_t a_bs, b_bs, c_bs; int batch;              // per-batch element strides (blockIdx.z)
    const float* bias_n; const float* bias_m;
    const float* rowvec; int rowvec_ld; int log_rows_per_sample;   // + rowvec[((m >> log) + z*z_samples)*ld + n]
    int z_samples;                                    // samples per batch index z (0: rowvec / gate ignore z)
    const float* gate; int gate_ld;                   // * gate[(m >> log)*ld + n]   (adaLN-Zero gates; applied before the residual)
    const bf16* resid; int resid_ld;                  // + resid[m*ld + n]
    const float* resid_f32; int resid_f32_ld;         // + resid_f32[z*c_bs + m*ld + n]  (fp32 residual stream, batch stride of c)
    float scale; int act;
    // fp8 operands (k_gemm_fp8): the accumulator is multiplied by deq_m[z*deq_m_bs + m] * deq_n[z*deq_n_bs + n] first
    // (per-row scale of the A operand x per-row scale of the B operand); nullptr = 1
    const float* deq_m; const float* deq_n; int64_t deq_m_bs, deq_n_bs;
    // MX block scales (one E8M0 byte per row and 32 elements, value 2^(e-127)), stored K-TILE MAJOR: the byte of
    // (row r, block kb) is at [(kb >> 2) * R * 4 + r * 4 + (kb & 3)], R = rows per batch plane (*_mx_ld), so that the scales
    // one 128-wide K-tile needs for 256 rows are 1 KiB of consecutive bytes (one DMA piece).  a_mx: of the A operand
    // (k_gemm_fp8<true>; the plane must be readable up to row m0 + 255); c_mx: of the OUTPUT when c_mode == OUT_FP8_MX.
    const uint8_t* a_mx; int a_mx_ld; int64_t a_mx_bs;
    uint8_t* c_mx; int c_mx_ld; int64_t c_mx_bs;
    int raster_g;                       // > 1: tiles walk groups of raster_g row-tiles, columns outer inside a group (wide-N GEMMs; tile_coords)
    int epi_fp32_slab;                  // A/B switch: 1 = always the fp32-slab epilogue (g_epi_fp32_slab)
    unsigned long long* dbg_ts;         // timing experiments: s_memtime stamps of block 0 / thread 0 (null in production)
    void* c; int c_ld; int c_mode;
    // fused GroupNorm statistics of the OUTPUT (DMA kernels, block tile inside one sample): per block tile and per
    // 4-channel quad, (sum, sum of squares) of the fp32 results -> gn_part[(m0/BM)*gn_quads + n/4] (float2)
    float* gn_part; int gn_quads;
    // ... and, where a block tile holds WHOLE samples and every channel of the tensor (k_conv_gn2 at 8x8 / 4x4), the finished GroupNorm table of the
    // tensor's (single) consumer, written by the same epilogue instead of a k_gn_finalize launch: fin_scale / fin_shift [sample][fin_ld] get
    // rstd * gamma * fin_mul and (beta - mean * rstd * gamma) * fin_mul with the consumer's gamma / beta; fin_cg = channels per group
    float* fin_scale; float* fin_shift; const float* fin_gamma; const float* fin_beta; int fin_ld; int fin_cg; float fin_mul; float fin_eps;
    // fused GroupNorm-apply + SiLU of the INPUT (k_conv_gn, conv_gn.h): a0 is the RAW, unpadded [B][H][W][a0_ld] tensor and every
    // element is read as silu(a0 * gn_scale[b*gn_ld + c] + gn_shift[b*gn_ld + c]); a1 (1x1 shortcut segment) stays raw
    // gn_folded: scale / shift arrive multiplied by -log2(e) and the 3x3 weights by -ln 2 (the kernel then computes t = x*scale + shift,
    // t / (1 + exp2(t)) = -log2(e) * silu(v): two vector instructions per element fewer); 0: plain scale / shift / weights
    const float* gn_scale; const float* gn_shift; int gn_ld; int gn_folded;
    // k_conv_gn2: the weights of a gn_scale launch once more, fragment-major (k_pack_frag); NULL -> k_conv_gn (LDS weight ring)
    const bf16* b_frag;
    // k_conv_gn2: 1 = the first blocks of every XCD request the whole fragment-major weight matrix once at kernel start (one 4-byte load per 128-byte
    // line, results discarded), in K order: the K loop's one-tap-ahead weight stream then hits L2 instead of paying a memory round trip per tap
    int w_warm;
    // 2x nearest up-sampling folded into the operand fetch of a gn_scale launch (k_conv_gn2 only): a0 is the tensor at HALF the resolution,
    // patch pixel (y, x) <- (y >> 1, x >> 1); a1_up: the same for the rows of the 1x1 shortcut operand a1
    int a0_up; int a1_up;
    // split-K (launch_gemm decides; small-M, long-K launches): `splitk_ws` = fp32 workspace for splitk_max * M * N partial sums
    float* splitk_ws; int splitk_max; int splitk;
    // direct residual-stream epilogue (EPI 7 / fp8 EPI 3) on a 16-bit stream: resid_f32 and c point at IEEE-half rows (same element strides); the arithmetic stays
    // fp32, one rounding to half per update (the MMDiT engine's image stream, natinf_set_mmdit_stream16: the reference's SD3 pipeline is fp16)
    int stream_f16;
};

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }
// v_rcp_f32 instead of the IEEE division sequence (1 ulp; the result is rounded to bf16 right after)
__device__ __forceinline__ float silu_fast(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ float gelu_tanh_f(float v) {             // nn.GELU(approximate="tanh")
    const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
    const float e = __expf(2.0f * u);                               // tanh(u) = 1 - 2/(e^{2u}+1)
    return 0.5f * v * (2.0f - 2.0f / (e + 1.0f));
}
// Epilogue form of the same function: 0.5 v (1 + tanh u) = v / (1 + e^(-2u)), u = sqrt(2/pi) (v + 0.044715 v^3), with the constants folded
// into the exponent's polynomial -- v * rcp(1 + exp2(v * (A + B v^2))), A = -2 sqrt(2/pi) log2(e), B = 0.044715 A: five plain vector instructions
// and two transcendentals per element (the tanh form above: eight and two).  An fc1 tile's epilogue is 128 elements per lane: its vector work was as
// long as the K = 1,536 loop it follows.  exp2 -> inf gives rcp -> 0 (v -> -0 for large negative v), exp2 -> 0 gives v: no NaN for finite v.
__device__ __forceinline__ float gelu_tanh_fast(float v) {
    const float t = v * __builtin_fmaf(v * v, -0.10294324f, -2.3022082f);
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}
// Eight values at once, STAGE BY STAGE (round 5): the same operations per element as gelu_tanh_fast -- the same bytes -- but every stage is issued for all eight values
// before the next one starts (the one-statement pins order the stages; hipcc otherwise emits each element as one serial chain exp2 -> s_nop -> add -> rcp -> s_nop -> mul
// through a single register, which a lone wave per SIMD pays in full: nothing else fills the transcendental unit's latency there).
__device__ __forceinline__ void pin8(float (&x)[8]) {
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
}
__device__ __forceinline__ void gelu_tanh_fast8(float (&v)[8]) {
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = v[k] * __builtin_fmaf(v[k] * v[k], -0.10294324f, -2.3022082f);
    pin8(t);
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = __builtin_amdgcn_exp2f(t[k]);
    pin8(t);
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = 1.0f + t[k];
    pin8(t);
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = __builtin_amdgcn_rcpf(t[k]);
    pin8(t);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] *= t[k];
}
// (SiLU the same way: v * rcp(1 + __expf(-v)), the operations of apply_act4 / apply_act8)
__device__ __forceinline__ void silu_fast8(float (&v)[8]) {
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = __expf(-v[k]);
    pin8(t);
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = 1.0f + t[k];
    pin8(t);
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = __builtin_amdgcn_rcpf(t[k]);
    pin8(t);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] *= t[k];
}
__device__ __forceinline__ float apply_act(float v, int act) {
    return act == ACT_SILU ? silu_f(v) : (act == ACT_GELU_TANH ? gelu_tanh_f(v) : (act == ACT_RELU ? fmaxf(v, 0.f) : v));
}
// E8M0 scale of a 32-value block with magnitude `amax`: the smallest power of two 2^e with amax * 2^-e <= 448 (e4m3 max).
// Returns the biased byte (e + 127, clamped) and writes 2^-e.  amax = m * 2^x, m in [1,2): 2^(x-8) maps it into [256,512);
// one more halving when m > 1.75.  An all-zero block gets scale 1.
__device__ __forceinline__ unsigned mx_scale_of(float amax, float& inv) {
    const unsigned bits = __float_as_uint(amax);
    int e = (int)((bits >> 23) & 0xff) - 127 - 8 + ((bits & 0x7fffff) > 0x600000 ? 1 : 0);
    e = amax > 0.f ? (e < -127 ? -127 : (e > 127 ? 127 : e)) : 0;
    inv = __uint_as_float((unsigned)(127 - e) << 23);
    return (unsigned)(e + 127);
}
// max over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (the four lane groups that hold one accumulator row) of a NON-NEGATIVE value, on the vector pipe:
// v_permlane16_swap (odd rows of one copy <-> even rows of the other) and v_permlane32_swap (the wave's halves), unsigned integer maxima (the bit pattern of a
// non-negative float orders like the float: no canonicalising v_max in front).  The ds_bpermute form of __shfl_xor costs an LDS round trip per stage behind an
// s_waitcnt lgkmcnt(0) that also waits for every LDS store in flight -- in an epilogue that is writing its slab.
__device__ __forceinline__ float group4_max_nonneg(float a) {
    unsigned u = __float_as_uint(a);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    u = r[0] > r[1] ? r[0] : r[1];
    const auto t = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    u = t[0] > t[1] ? t[0] : t[1];
    return __uint_as_float(u);
}
// CLAMP = false: the caller guarantees |value| <= 448 (block-scaled values: mx_scale_of's 2^-e maps the block maximum into (224, 448], exactly -- a power of two)
template <bool CLAMP = true>
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {      // values already scaled; clamp: the cvt does not saturate
    if constexpr (CLAMP) {
        a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);      // (one v_med3_f32 per value)
        c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
    }
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}

// Epilogue form: ONE uniform branch per 8 values (the per-value ternary compiles to a scalar branch chain per element)
// and v_rcp_f32 instead of the 15-instruction IEEE division -- 1 ulp, far inside the bf16 output rounding.
__device__ __forceinline__ void apply_act8(float (&v)[8], int act) {
    if (act == ACT_SILU) {
        silu_fast8(v);                                   // (stage by stage: the same operations per element)
    } else if (act == ACT_GELU_TANH) {
        gelu_tanh_fast8(v);
    } else if (act == ACT_RELU) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
    }
}

__device__ __forceinline__ void apply_act4(float (&v)[4], int act) {
    if (act == ACT_SILU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = v[q] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[q]));
    } else if (act == ACT_GELU_TANH) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = gelu_tanh_fast(v[q]);
    } else if (act == ACT_RELU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
    }
}

// Linear tile index -> (row-tile, column-tile).  Row-major by default; with g (= GemmArgs::raster_g) > 1 the tiles of a group of
// g row-tiles are visited column by column, so that the ~32 tiles an XCD works on at one time form a g x (32/g) patch: they
// share g + 32/g operand panels instead of 1 + 32 when the matrix is wide (nN >> 32/g) -- L2 hits for the LDS fill.
__device__ __forceinline__ void tile_coords(int tile, int nM, int nN, int g, int& mt, int& nt) {
    if (g > 1) {
        const int per = g * nN, grp = tile / per, r = tile - grp * per, rows = min(g, nM - grp * g);
        nt = r / rows; mt = grp * g + (r - nt * rows);
    } else { mt = tile / nN; nt = tile - mt * nN; }
}

// XCD-aware bijective remap of a linear block id: consecutive ids land on different XCDs (round-robin
// dispatch), so give every XCD a contiguous run of tiles.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, l = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + l;
}

// Shared epilogue: accumulators -> LDS (fp32) -> fused adds -> coalesced stores.  All waves must have finished
// reading the operand tiles (the caller's last barrier) before this overwrites the same LDS.
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, unsigned char* smem, f32x4 (&acc)[4][4],
                                              int m0, int n0, int z, int tid, int lane, int wm, int wn)
{
    float* sC = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                sC[(wm * 64 + i * 16 + (lane >> 4) * 4 + r) * C_ROW + wn * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
    __syncthreads();

    if (g.c_mode == OUT_F32_NCHW) {
        // out[b][n][p] fp32, N small (final 3-channel conv): consecutive threads -> consecutive pixels
        float* out = reinterpret_cast<float*>(g.c);
        const int nvalid = min(BN, g.N - n0);
        for (int e = tid; e < BM * nvalid; e += 256) {
            const int r = e & (BM - 1), n = e >> 7;
            const int m = m0 + r;
            if (m < g.M) {
                float v = sC[r * C_ROW + n];
                if (g.bias_n) v += g.bias_n[n0 + n];
                v *= g.scale;
                const int b = m >> g.logHW, p = m & ((1 << g.logHW) - 1);
                out[((int64_t)b * g.N + n0 + n) * ((int64_t)1 << g.logHW) + p] = v;
            }
        }
        return;
    }

    const int cchunk = tid & 15;                   // 8 consecutive columns
    const int n = n0 + cchunk * 8;
    const bool n_in = n < g.N;                     // N is a multiple of 8 whenever this path is used
    float bn[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) bn[q] = 0.f;
    if (g.bias_n && n_in) {
        const float4 u = *reinterpret_cast<const float4*>(g.bias_n + n), w = *reinterpret_cast<const float4*>(g.bias_n + n + 4);
        bn[0] = u.x; bn[1] = u.y; bn[2] = u.z; bn[3] = u.w; bn[4] = w.x; bn[5] = w.y; bn[6] = w.z; bn[7] = w.w;
    }
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const int r = pass * 16 + (tid >> 4);
        const int m = m0 + r;
        if (m >= g.M || !n_in) continue;
        const float4 u = *reinterpret_cast<const float4*>(sC + r * C_ROW + cchunk * 8);
        const float4 w = *reinterpret_cast<const float4*>(sC + r * C_ROW + cchunk * 8 + 4);
        float v[8] = {u.x, u.y, u.z, u.w, w.x, w.y, w.z, w.w};
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += bn[q];
        if (g.bias_m) {
            const float bm = g.bias_m[m];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += bm;
        }
        if (g.rowvec) {
            const float* rv = g.rowvec + (int64_t)((m >> g.log_rows_per_sample) + z * g.z_samples) * g.rowvec_ld + n;
            const float4 s = *reinterpret_cast<const float4*>(rv), t = *reinterpret_cast<const float4*>(rv + 4);
            v[0] += s.x; v[1] += s.y; v[2] += s.z; v[3] += s.w; v[4] += t.x; v[5] += t.y; v[6] += t.z; v[7] += t.w;
        }
        if (g.gate) {
            const float* gv = g.gate + (int64_t)((m >> g.log_rows_per_sample) + z * g.z_samples) * g.gate_ld + n;
            const float4 s = *reinterpret_cast<const float4*>(gv), t = *reinterpret_cast<const float4*>(gv + 4);
            v[0] *= s.x; v[1] *= s.y; v[2] *= s.z; v[3] *= s.w; v[4] *= t.x; v[5] *= t.y; v[6] *= t.z; v[7] *= t.w;
        }
        if (g.resid) {
            const bf16x8 rs = *reinterpret_cast<const bf16x8*>(g.resid + (int64_t)z * g.c_bs + (int64_t)m * g.resid_ld + n);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += (float)rs[q];
        }
        if (g.resid_f32) {
            const float* rp = g.resid_f32 + (int64_t)z * g.c_bs + (int64_t)m * g.resid_f32_ld + n;
            const float4 s = *reinterpret_cast<const float4*>(rp), t = *reinterpret_cast<const float4*>(rp + 4);
            v[0] += s.x; v[1] += s.y; v[2] += s.z; v[3] += s.w; v[4] += t.x; v[5] += t.y; v[6] += t.z; v[7] += t.w;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] *= g.scale;
        apply_act8(v, g.act);
        if (g.c_mode == OUT_BF16) {
            bf16x8 o;
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] = (bf16)v[q];
            *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(g.c) + (int64_t)z * g.c_bs + (int64_t)m * g.c_ld + n) = o;
        } else {
            float* o = reinterpret_cast<float*>(g.c) + (int64_t)z * g.c_bs + (int64_t)m * g.c_ld + n;
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

// Branch-free masked 16-byte load: the address is always a readable one (the operand's base when
// masked), the select zeroes the value.  Keeps the staging registers out of scratch.
__device__ __forceinline__ uint4 ld_or_zero(const bf16* p, bool ok) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    return ok ? v : make_uint4(0u, 0u, 0u, 0u);
}

__global__ __launch_bounds__(256, 2) void k_gemm_bf16(const GemmArgs g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_poison();
    bf16* sA = reinterpret_cast<bf16*>(smem);
    bf16* sB = sA + 2 * TILE_ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nN = (g.N + BN - 1) / BN, nM = (g.M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, nM * nN);
    const int m0 = (tile / nN) * BM, n0 = (tile % nN) * BN;
    const int z = blockIdx.z;

    const bf16* a0 = g.a0 + (int64_t)z * g.a_bs;
    const bf16* a1 = g.a1 ? g.a1 + (int64_t)z * g.a_bs : nullptr;
    const bf16* bp = g.b + (int64_t)z * g.b_bs;

    const int K0 = g.taps * g.a0_C, K1 = g.a1 ? g.a1_C : 0;
    const int nk0 = (K0 + BK - 1) / BK, nk1 = (K1 + BK - 1) / BK, nk = nk0 + nk1;

    // ---- per-thread load slots: 4 rows x one 16-byte column chunk, for A and for B
    const int colc = tid & 7;                      // chunk column (8 bf16)
    const int row0 = tid >> 3;                     // rows row0 + 32*i
    int a_y[4], a_x[4];
    int64_t a_off0[4], a_off1[4], b_off[4];
    bool a_ok[4], b_ok[4];
    const int Wd = 1 << g.logW, Hd = 1 << (g.logHW - g.logW);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + row0 + 32 * i;
        a_ok[i] = m < g.M;
        const int p = m & ((1 << g.logHW) - 1);
        a_y[i] = p >> g.logW; a_x[i] = p & (Wd - 1);
        a_off0[i] = (int64_t)m * g.a0_ld;
        a_off1[i] = (int64_t)m * g.a1_ld;
        const int n = n0 + row0 + 32 * i;
        b_ok[i] = n < g.N;
        b_off[i] = (int64_t)n * g.b_ld;
    }

    uint4 ra[4], rb[4];
    auto load_tile = [&](int kt) __attribute__((always_inline)) {
        if (kt < nk0) {
            const int kbase = kt * BK;
            int tap = 0, c0 = kbase;
            if (g.taps == 9) { const int cch = kt / 9; tap = kt - 9 * cch; c0 = cch * BK; }   // chunk outer, tap inner
            const int dy = g.taps == 9 ? tap / 3 - 1 : 0, dx = g.taps == 9 ? tap % 3 - 1 : 0;
            const int cc = c0 + colc * 8;
            const bool kin = cc < (g.taps == 9 ? g.a0_C : K0);
            const int64_t shift = (int64_t)(dy * Wd + dx) * g.a0_ld + cc;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = a_ok[i] && kin && (unsigned)(a_y[i] + dy) < (unsigned)Hd && (unsigned)(a_x[i] + dx) < (unsigned)Wd;
                ra[i] = ld_or_zero(ok ? a0 + a_off0[i] + shift : a0, ok);
            }
        } else {
            const int cc = (kt - nk0) * BK + colc * 8;
            const bool kin = cc < K1;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                ra[i] = ld_or_zero((a_ok[i] && kin) ? a1 + a_off1[i] + cc : a1, a_ok[i] && kin);
        }
        {
            // B columns follow the same K order: segment 0 at [0, K0), segment 1 at [K0, K0+K1)
            const int kk = (kt < nk0 ? kt * BK : K0 + (kt - nk0) * BK) + colc * 8;
            const bool kin = kt < nk0 ? (kk < K0) : (kk < K0 + K1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rb[i] = ld_or_zero((b_ok[i] && kin) ? bp + b_off[i] + kk : bp, b_ok[i] && kin);
        }
    };
    auto store_tile = [&](int buf) __attribute__((always_inline)) {
        bf16* da = sA + buf * TILE_ELEMS;
        bf16* db = sB + buf * TILE_ELEMS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = row0 + 32 * i;
            const int o = r * LDS_ROW + ((colc ^ ((r >> 1) & 7)) << 3);
            *reinterpret_cast<uint4*>(da + o) = ra[i];
            *reinterpret_cast<uint4*>(db + o) = rb[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int frow = lane & 15, fq = lane >> 4, fswz = (frow >> 1) & 7;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const bf16* ta = sA + cur * TILE_ELEMS + (wm * 64 + frow) * LDS_ROW;
        const bf16* tb = sB + cur * TILE_ELEMS + (wn * 64 + frow) * LDS_ROW;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
            const int ko = (((ks << 2) | fq) ^ fswz) << 3;       // swizzled chunk of this lane's 8 k-values
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = *reinterpret_cast<const bf16x8*>(ta + i * 16 * LDS_ROW + ko);
                fb[i] = *reinterpret_cast<const bf16x8*>(tb + i * 16 * LDS_ROW + ko);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    gemm_epilogue(g, smem, acc, m0, n0, z, tid, lane, wm, wn);
}

// ------------------------------------------------------------------------------------------------
// GroupNorm (32 groups, eps 1e-6): statistics pass -> per-(sample, channel) scale / shift
// ------------------------------------------------------------------------------------------------
// One 256-thread block per sample.  scale[b][c] = rstd*gamma[c]*out_mul, shift[b][c] = (beta[c] - mean*rstd*gamma[c])*out_mul.
// Deterministic: per-thread partial sums are parked in LDS and reduced in a fixed order (no atomics), so a
// sample's statistics do not depend on timing, batch size or batch neighbours.
__global__ __launch_bounds__(256) void k_gn_stats(const bf16* __restrict__ x, int ld, int C, int HW,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  float* __restrict__ scale, float* __restrict__ shift, float eps, float out_mul)
{
    __shared__ float s_part[2][16 * 128 + 64];     // [sum|sq][lane * C + c]; lanes*C <= 2048 for every C in use
    __shared__ float s_sum[512], s_sq[512], s_mean[32], s_rstd[32];
    lds_poison();
    const int tid = threadIdx.x, b = blockIdx.x;
    const int cpp = C >> 3;                        // 16-byte chunks per pixel
    const int lanes = 256 / cpp;                   // pixel lanes (16 / 8 / 5 / 4 for C = 128 / 256 / 384 / 512)
    if (tid < cpp * lanes) {
        const int chunk = tid % cpp, pl = tid / cpp;
        float s[8], q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { s[i] = 0.f; q[i] = 0.f; }
        const bf16* base = x + (int64_t)b * HW * ld + chunk * 8;
        for (int p = pl; p < HW; p += lanes) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(base + (int64_t)p * ld);
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float f = (float)v[i]; s[i] += f; q[i] += f * f; }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { s_part[0][pl * C + chunk * 8 + i] = s[i]; s_part[1][pl * C + chunk * 8 + i] = q[i]; }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float s = 0.f, q = 0.f;
        for (int l = 0; l < lanes; ++l) { s += s_part[0][l * C + c]; q += s_part[1][l * C + c]; }
        s_sum[c] = s; s_sq[c] = q;
    }
    __syncthreads();
    const int cg = C >> 5;
    if (tid < 32) {
        float s = 0.f, q = 0.f;
        for (int i = 0; i < cg; ++i) { s += s_sum[tid * cg + i]; q += s_sq[tid * cg + i]; }
        const float inv = 1.0f / (float)(cg * HW);
        const float mean = s * inv;
        float var = q * inv - mean * mean;
        var = var < 0.f ? 0.f : var;
        s_mean[tid] = mean;
        s_rstd[tid] = 1.0f / sqrtf(var + eps);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const int gi = c / cg;
        const float sc = s_rstd[gi] * gamma[c];
        // out_mul: 1, or -log2(e) when the consumer is k_conv_gn in folded form (it then gets exp(-v) = exp2(x*scale + shift) at once)
        scale[(int64_t)b * C + c] = sc * out_mul;
        shift[(int64_t)b * C + c] = (beta[c] - s_mean[gi] * sc) * out_mul;
    }
}

// GroupNorm statistics from the per-tile quad partials the producing GEMM(s) wrote (up to two channel-wise
// concatenated sources, e.g. [h, skip] of an up-path block): fixed summation order -> deterministic.
// One block per sample.  P*: float2 [tiles][quads*]; sample b owns tiles b*tps* .. (b+1)*tps*-1.
__global__ __launch_bounds__(256) void k_gn_finalize(const float2* __restrict__ P0, int tps0, int quads0,
                                                     const float2* __restrict__ P1, int tps1, int quads1, int C, int HW,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ scale, float* __restrict__ shift, float eps, float out_mul)
{
    // A launch of this kernel is pure latency (a few KB per sample): ONE barrier, and every global read requested before anything waits --
    // the thread's gamma / beta (C <= 512: two channels per thread) first, the partial rows of its quad four at a time.  Every channel thread
    // then sums its group's quads itself (<= 4 LDS reads) instead of waiting for a 32-thread middle phase behind a second barrier.  Same
    // additions in the same order as the three-phase form it replaces: bit-identical tables (5.2 -> ~3.5 us per launch, 68 launches per forward).
    __shared__ float2 s_p[128];
    lds_poison();
    const int tid = threadIdx.x, b = blockIdx.x, nq = quads0 + quads1;
    float ga[2] = {0.f, 0.f}, be[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = tid + 256 * k;
        if (c < C) { ga[k] = gamma[c]; be[k] = beta[c]; }
    }
    for (int q = tid; q < nq; q += 256) {
        const bool first = q < quads0;
        const float2* P = first ? P0 + (int64_t)b * tps0 * quads0 + q : P1 + (int64_t)b * tps1 * quads1 + (q - quads0);
        const int tps = first ? tps0 : tps1, st = first ? quads0 : quads1;
        float s = 0.f, qq = 0.f;
        int t = 0;
        for (; t + 4 <= tps; t += 4) {
            const float2 v0 = P[(int64_t)t * st], v1 = P[(int64_t)(t + 1) * st], v2 = P[(int64_t)(t + 2) * st], v3 = P[(int64_t)(t + 3) * st];
            s += v0.x; qq += v0.y; s += v1.x; qq += v1.y; s += v2.x; qq += v2.y; s += v3.x; qq += v3.y;
        }
        for (; t < tps; ++t) { const float2 v = P[(int64_t)t * st]; s += v.x; qq += v.y; }
        s_p[q] = make_float2(s, qq);
    }
    __syncthreads();
    const int cg = C >> 5, qpg = cg >> 2;
    const float inv = 1.0f / (float)(cg * HW);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = tid + 256 * k;
        if (c >= C) break;
        const int gi = c / cg;
        float s = 0.f, q = 0.f;
        for (int i = 0; i < qpg; ++i) { const float2 v = s_p[gi * qpg + i]; s += v.x; q += v.y; }
        const float mean = s * inv;
        float var = q * inv - mean * mean;
        var = var < 0.f ? 0.f : var;
        const float sc = (1.0f / sqrtf(var + eps)) * ga[k];
        // out_mul: 1, or -log2(e) when the consumer is k_conv_gn in folded form (it then gets exp(-v) = exp2(x*scale + shift) at once)
        scale[(int64_t)b * C + c] = sc * out_mul;
        shift[(int64_t)b * C + c] = (be[k] - mean * sc) * out_mul;
    }
}

// y = act(x*scale + shift) with optional 2x nearest up-sampling / 2x2 mean down-sampling of BOTH the
// activated tensor (-> y) and the raw input (-> xr), as ResnetBlockBigGANpp does (layerspp.py:245-257).
// Destination-centric: a block owns GN_ROWS consecutive destination pixel rows of one sample (gridDim.y = B); a thread
// owns ONE 16-byte channel chunk for the whole block -- its 8 scale / 8 shift values live in registers -- and walks
// the pixels of those rows.  With pad = 1 the destination carries a one-pixel zero border ([B][Hd+2][Wd+2][C]) which
// this kernel writes too, so the 3x3 implicit GEMM that consumes y can fetch every tap unconditionally.
// xr (raw input at the output resolution, feeds the 1x1 shortcut) is never padded.
enum { RS_NONE = 0, RS_UP = 1, RS_DOWN = 2 };
constexpr int GN_ROWS = 4;
#ifndef GN_UNROLL
#define GN_UNROLL 4
#endif
__global__ __launch_bounds__(256) void k_gn_apply(const bf16* __restrict__ x, int ld, int C, int logW, int logHW,
                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                  bf16* __restrict__ y, bf16* __restrict__ xr, int act, int mode, int pad)
{
    const int cpp = C >> 3, lanes = 256 / cpp;
    const int chunk = threadIdx.x % cpp, pl = threadIdx.x / cpp;
    if (pl >= lanes) return;                          // C = 384: 5 pixel lanes x 48 chunks = 240 active threads
    const int Ws = 1 << logW, Hs = 1 << (logHW - logW);
    const int Wd = mode == RS_UP ? 2 * Ws : (mode == RS_DOWN ? Ws >> 1 : Ws);
    const int Hd = mode == RS_UP ? 2 * Hs : (mode == RS_DOWN ? Hs >> 1 : Hs);
    const int Wp = Wd + 2 * pad, Hp = Hd + 2 * pad;
    const int b = blockIdx.y, row0 = blockIdx.x * GN_ROWS;
    const int nrows = min(GN_ROWS, Hp - row0);
    float sc[8], sh[8];
    {
        const float* ps = scale + (int64_t)b * C + chunk * 8;
        const float* ph = shift + (int64_t)b * C + chunk * 8;
        const float4 s0 = *reinterpret_cast<const float4*>(ps), s1 = *reinterpret_cast<const float4*>(ps + 4);
        const float4 h0 = *reinterpret_cast<const float4*>(ph), h1 = *reinterpret_cast<const float4*>(ph + 4);
        sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
        sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
    }
    const bf16* xb = x + (int64_t)b * Hs * Ws * ld + chunk * 8;
    bf16* yb = y + ((int64_t)b * Hp + row0) * Wp * C + chunk * 8;
    bf16* xrb = xr ? xr + (int64_t)b * Hd * Wd * C + chunk * 8 : nullptr;
    // pixel walk, two pixels per iteration so that two independent 16-byte loads are in flight per thread
    int rr = 0, xx = pl;
    while (xx >= Wp) { xx -= Wp; ++rr; }
    auto advance = [&](int& r, int& c) __attribute__((always_inline)) { c += lanes; while (c >= Wp) { c -= Wp; ++r; } };
    auto interior = [&](int r, int c) __attribute__((always_inline)) {
        const int Y = row0 + r - pad, X = c - pad;
        return r < nrows && Y >= 0 && Y < Hd && X >= 0 && X < Wd;
    };
    auto finish = [&](int r, int c, bool in, const bf16x8& v) __attribute__((always_inline)) {
        if (r >= nrows) return;
        bf16x8 o;
        if (!in) {
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] = (bf16)0.0f;
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float f = (float)v[q] * sc[q] + sh[q];
                if (act == ACT_SILU) f = silu_fast(f);
                o[q] = (bf16)f;
            }
            if (xrb) *reinterpret_cast<bf16x8*>(xrb + ((int64_t)(row0 + r - pad) * Wd + (c - pad)) * C) = v;
        }
        *reinterpret_cast<bf16x8*>(yb + ((int64_t)r * Wp + c) * C) = o;
    };
    if (mode != RS_DOWN) {
        const int sh_ = mode == RS_UP ? 1 : 0;
        constexpr int U = GN_UNROLL;                   // independent 16-byte loads in flight per thread
        while (rr < nrows) {
            int r[U], c[U];
            bool in[U];
            bf16x8 v[U];
            r[0] = rr; c[0] = xx;
#pragma unroll
            for (int u = 1; u < U; ++u) { r[u] = r[u - 1]; c[u] = c[u - 1]; advance(r[u], c[u]); }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                in[u] = interior(r[u], c[u]);
                if (in[u]) v[u] = *reinterpret_cast<const bf16x8*>(xb + ((int64_t)((row0 + r[u] - pad) >> sh_) * Ws + ((c[u] - pad) >> sh_)) * ld);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) finish(r[u], c[u], in[u], v[u]);
            rr = r[U - 1]; xx = c[U - 1];
            advance(rr, xx);
        }
        return;
    }
    for (; rr < nrows; ) {
        const int Y = row0 + rr - pad, X = xx - pad;
        bf16x8 o;
        if (Y < 0 || Y >= Hd || X < 0 || X >= Wd) {
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] = (bf16)0.0f;
        } else {
            float ay[8], ax[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { ay[q] = 0.f; ax[q] = 0.f; }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(xb + ((int64_t)(2 * Y + (d >> 1)) * Ws + 2 * X + (d & 1)) * ld);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float raw = (float)v[q];
                    float f = raw * sc[q] + sh[q];
                    if (act == ACT_SILU) f = silu_fast(f);
                    ay[q] += f; ax[q] += raw;
                }
            }
            bf16x8 ox;
#pragma unroll
            for (int q = 0; q < 8; ++q) { o[q] = (bf16)(ay[q] * 0.25f); ox[q] = (bf16)(ax[q] * 0.25f); }
            if (xrb) *reinterpret_cast<bf16x8*>(xrb + ((int64_t)Y * Wd + X) * C) = ox;
        }
        *reinterpret_cast<bf16x8*>(yb + ((int64_t)rr * Wp + xx) * C) = o;
        xx += lanes;
        while (xx >= Wp) { xx -= Wp; ++rr; }
    }
}

// row softmax: S fp32 [rows][T] -> P bf16 [rows][T]; one wave per row, T <= 256, T % 4 == 0 or T == 16
__global__ __launch_bounds__(256) void k_softmax_rows(const float* __restrict__ S, bf16* __restrict__ P, int T, int64_t rows)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* s = S + row * T;
    float v[4];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < T ? s[c] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = (lane + 64 * i < T) ? __expf(v[i] - mx) : 0.f; sum += v[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < T) P[row * T + c] = (bf16)(v[i] * inv);
    }
}

// sinusoidal embedding (layers.py:515-530): labels [B] -> emb bf16 [B][128]
__global__ void k_time_embed(const float* __restrict__ labels, bf16* __restrict__ emb, int B)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 128) return;
    const int b = i >> 7, j = i & 127, h = j & 63;
    const float freq = expf((float)h * -(9.210340371976184f / 63.0f));        // ln(10000)/(half-1)
    const float a = labels[b] * freq;
    emb[i] = (bf16)(j < 64 ? sinf(a) : cosf(a));
}

// stem im2col: x fp32 NCHW [B][3][32][32] -> A bf16 [B*1024][64], k = tap*3 + c (27 used, rest zero)
__global__ __launch_bounds__(256) void k_stem_im2col(const float* __restrict__ x, bf16* __restrict__ A, int64_t rows)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one thread per (row, 8-wide chunk)
    if (idx >= rows * 8) return;
    const int64_t m = idx >> 3; const int ch = (int)(idx & 7);
    const int b = (int)(m >> 10), p = (int)(m & 1023), yy = p >> 5, xx = p & 31;
    bf16x8 o;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = ch * 8 + q;
        float v = 0.f;
        if (k < 27) {
            const int tap = k / 3, c = k - tap * 3, sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
            if ((unsigned)sy < 32u && (unsigned)sx < 32u) v = x[(((int64_t)b * 3 + c) << 10) + sy * 32 + sx];
        }
        o[q] = (bf16)v;
    }
    *reinterpret_cast<bf16x8*>(A + m * 64 + ch * 8) = o;
}

// ------------------------------------------------------------------------------------------------
// weight packing (fp32 reference layouts -> bf16 GEMM layouts), run once per load
// ------------------------------------------------------------------------------------------------
// src [N][Cin][taps] (OIHW flattened) -> packed K order.
//   chunked = 0: k = tap*tap_stride_c + c                       (stem: 27 taps*channels padded to 64; 1x1; linear)
//   chunked = 1: k = ((c/64)*taps + tap)*64 + c%64              (3x3 convs: 64-channel chunk OUTER, tap INNER, so the
//                nine shifted reads of one chunk are back to back in the K loop and hit L1/L2 instead of thrashing it)
__global__ void k_pack_conv(const float* __restrict__ src, bf16* __restrict__ dst, int N, int Cin, int taps,
                            int dst_ld, int koff, int tap_stride_c, int chunked, float wmul)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * Cin * taps) return;
    const int tap = (int)(i % taps); const int64_t r = i / taps; const int c = (int)(r % Cin); const int n = (int)(r / Cin);
    const int k = chunked ? ((c >> 6) * taps + tap) * 64 + (c & 63) : tap * tap_stride_c + c;
    dst[(int64_t)n * dst_ld + koff + k] = (bf16)(src[i] * wmul);      // wmul: 1, or -ln 2 for the 3x3 weights of a folded k_conv_gn launch
}
// src [K][N] (NIN.W) -> dst[n*dst_ld + k]
__global__ void k_pack_transpose(const float* __restrict__ src, bf16* __restrict__ dst, int K, int N, int dst_ld)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)K * N) return;
    const int n = (int)(i % N), k = (int)(i / N);
    dst[(int64_t)n * dst_ld + k] = (bf16)src[i];
}
__global__ void k_fill_bf16_zero(bf16* __restrict__ dst, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (bf16)0.0f;
}
__global__ void k_fill_f32(float* __restrict__ p, float v, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// generic im2col: x [B][H][W][ld] (channels coff .. coff+C-1, C % 8 == 0) -> out [B*Ho*Wo][Kp], k = (ky*kw + kx)*C + c, zero for padding and k >= K.
// One thread per 8 consecutive k (16 bytes).
__global__ __launch_bounds__(256) void k_inc_im2col(const bf16* __restrict__ x, int H, int W, int ld, int C, int kh, int kw, int stride, int ph, int pw,
                                                    int Ho, int Wo, int Kp, bf16* __restrict__ out, int64_t total)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int kc = Kp >> 3;
    const int k = (int)(i % kc) * 8; const int64_t row = i / kc;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (k < kh * kw * C) {
        const int tap = k / C, c = k - tap * C, ky = tap / kw, kx = tap - ky * kw;
        const int ox = (int)(row % Wo), oy = (int)((row / Wo) % Ho); const int64_t b = row / ((int64_t)Wo * Ho);
        const int y = oy * stride - ph + ky, xx = ox * stride - pw + kx;
        if ((unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W) v = *reinterpret_cast<const uint4*>(x + (((int64_t)b * H + y) * W + xx) * ld + c);
    }
    *reinterpret_cast<uint4*>(out + row * Kp + k) = v;
}

// dst = a (+ b)
__global__ void k_copy_add_f32(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ dst, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = a[i] + (b ? b[i] : 0.f);
}
// tap: NHWC bf16 (ld) -> NCHW fp32
__global__ void k_nhwc_to_nchw_f32(const bf16* __restrict__ x, int ld, int C, int HW, float* __restrict__ out, int64_t total)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // over out index (b, c, p)
    if (i >= total) return;
    const int p = (int)(i % HW); const int64_t r = i / HW; const int c = (int)(r % C); const int64_t b = r / C;
    out[i] = (float)x[(b * HW + p) * ld + c];
}

}  // namespace ncsn

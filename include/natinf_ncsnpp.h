/*
 * natinf_ncsnpp.h -- C ABI of the NCSN++ / DDPM++ denoiser engine inside libnatinf.so.
 *
 * Replaces the `model(x, labels)` call made by the reference's score function
 * (deps/score_sde_pytorch/models/utils.py:144-160 -> models/utils.py:118-123), i.e.
 * NCSNpp.forward (deps/score_sde_pytorch/models/ncsnpp.py:232-381) under the configuration the
 * CIFAR10 script imports (configs/vp/cifar10_ddpmpp_continuous.py:41-64: nf 128, ch_mult (1,2,2,2),
 * 4 BigGAN res-blocks per level, attention at 16 px, positional embedding, fir=False,
 * skip_rescale=True, progressive none, centered data, scale_by_sigma=False, eval mode).
 *
 * Arithmetic: bf16 operands on the gfx950 matrix cores (v_mfma_f32_16x16x32_bf16), fp32
 * accumulation, fp32 GroupNorm statistics / softmax, bf16 activations between layers.  The parity
 * bar against the reference's fp32 forward is a floating-point tolerance (stated in the tests),
 * not bit equality.
 *
 * Conventions are those of natinf.h: device pointers owned by the caller, explicit stream, int
 * return codes, no device allocation (the caller supplies the packed-weight buffer and the
 * activation workspace; sizes are queried below).
 */
#ifndef NATINF_NCSNPP_H
#define NATINF_NCSNPP_H

#include <stdint.h>
#include "natinf.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct natinf_ncsnpp* natinf_ncsnpp_t;

/* keep every module's output alive in the workspace (bump allocation, no reuse) so that
 * natinf_ncsnpp_debug_tap can read any of them after a forward; tests only. */
#define NATINF_NCSNPP_KEEP_ACTIVATIONS 1
/* the `ddpm` score network (deps/score_sde_pytorch/models/ddpm.py:39-181 under configs/vp/ddpm/cifar10_continuous.py: two ResnetBlockDDPM per
 * level, AttnBlock, Downsample / Upsample convolutions; 35,218,947 parameters) instead of NCSN++ / DDPM++ -- the network of the checkpoint the
 * reference's docstring names (src/CIFAR10NaturalInference.py:416).  Sizes of such a handle: natinf_ncsnpp_handle_param_count / _packed_bytes. */
#define NATINF_NCSNPP_DDPM 2

/* Number of fp32 parameters (61,804,419) of the NCSN++ plan, in the flat order documented at natinf_ncsnpp_load. */
int64_t natinf_ncsnpp_param_count(void);
/* Bytes of device memory for the packed (bf16, GEMM-ready) weights + fp32 biases / affine terms: the LARGEST NCSN++ plan (every
 * plan-build-time fusion on), whatever the natinf_set_* switches are when this is first asked -- a buffer of this size fits every
 * NCSN++ plan.  Both handle-less queries answer for NCSN++ only; prefer the natinf_ncsnpp_handle_* forms (exact for THIS handle,
 * and the only ones that know the `ddpm` network). */
int64_t natinf_ncsnpp_packed_bytes(void);
int64_t natinf_ncsnpp_handle_param_count(natinf_ncsnpp_t h);      /* of THIS handle's network (NCSN++ or `ddpm`) */
int64_t natinf_ncsnpp_handle_packed_bytes(natinf_ncsnpp_t h);
/* Bytes of device workspace a forward at batch <= max_batch needs (depends on the handle's flags). */
int64_t natinf_ncsnpp_workspace_bytes(natinf_ncsnpp_t h, int max_batch);

/* Host-side object: the static execution plan (op list, arena offsets, weight offsets).  No GPU needed. */
int natinf_ncsnpp_create(natinf_ncsnpp_t* out, int flags);
int natinf_ncsnpp_destroy(natinf_ncsnpp_t h);

/* One line per all_modules entry: "idx kind cin cout up down res param_offset"; returns the number of
 * bytes written (excluding the NUL) or NATINF_EINVAL if `cap` is too small.  Host only. */
int natinf_ncsnpp_describe(natinf_ncsnpp_t h, char* buf, int cap);

/* Pack the weights.  `params_f32` = every parameter of NCSNpp, fp32, concatenated in
 * `all_modules` order with the leaves of each module in registration order -- exactly the order of
 * `model.parameters()` / of the EMA `shadow_params` list the reference restores
 * (deps/score_sde_pytorch/models/ema.py:53-64); n_params must equal natinf_ncsnpp_param_count().
 * `packed` must stay alive and untouched for the life of the handle's forwards. */
int natinf_ncsnpp_load(natinf_ncsnpp_t h, const float* params_f32, int64_t n_params,
                       void* packed, int64_t packed_bytes, natinf_stream_t stream);

/* A second handle of the same network (same flags, built under the same natinf_set_* switches) adopts the packed weights `loaded` was
 * loaded with instead of packing its own copy: the packed buffer is read-only during forwards, so one copy serves any number of handles
 * (one handle per HIP stream: a handle's launch plan writes one workspace at a time).  The buffer must outlive both handles' forwards.
 * NATINF_ESTATE if `loaded` has no weights yet, NATINF_EINVAL if the two plans do not address the buffer identically. */
int natinf_ncsnpp_share(natinf_ncsnpp_t h, natinf_ncsnpp_t loaded);

/* out = model(x, labels): x, out [B,3,32,32] fp32 NCHW; labels [B] fp32 (= t*999). */
int natinf_ncsnpp_forward(natinf_ncsnpp_t h, const float* x, const float* labels, float* out, int B,
                          void* workspace, int64_t workspace_bytes, natinf_stream_t stream);

/* One line per GEMM launch of a forward at batch B, in launch order: "M N K0 K1 taps batch kernel".
 * Host only (nothing is launched); used to attribute rocprofv3 kernel-trace rows to layer shapes. */
int natinf_ncsnpp_describe_gemms(natinf_ncsnpp_t h, int B, char* buf, int cap);

/* Tuning hooks.  natinf_debug_gemm runs `iters` launches of one GEMM-kernel variant on caller-supplied operands
 * (A [batch][M][K0/taps channels] bf16 -- zero-bordered [B][H+2][W+2][C] when taps == 9, W = 1 << logW --,
 * optional 1x1 segment a1 [M][K1], B [N][K0+K1] bf16 in the engine's K order, optional fp32 bias, C [M][N] bf16 or
 * fp32).  Variants: 0 auto, 1 generic, 2/3/4 two-stage DMA 256x256 / 256x128 / 128x128, 5/6/7/8 ring
 * 256x256 / 256x128 / 128x128 / 64x128.  natinf_set_gemm_variant forces a variant for every DMA-eligible launch of
 * subsequent forwards (0 = automatic). */
int natinf_debug_gemm(int variant, int M, int N, int K0, int K1, int taps, int logW, int batch,
                      const void* a0, const void* a1, const void* b, const float* bias_n, void* c, int c_f32, float scale,
                      int iters, natinf_stream_t stream);
/* fp8 (e4m3) GEMM of the transformer engines, on its own (tests / micro-benchmarks): c = (a_scale[m] * b_scale[n]) * a8 b8^T
 * (+ bias_n); a8 [M][K], b8 [N][K] fp8 bytes with one fp32 scale per row, K % 128 == 0, N % 8 == 0.
 * natinf_debug_quant_fp8_rows produces such a pair from fp32 rows: scale = max|row| / 448, q = e4m3(row / scale). */
int natinf_debug_quant_fp8_rows(const float* w, void* q, float* row_scale, int rows, int cols, natinf_stream_t stream);
/* a_mx (optional): E8M0 block scales of a8 (then a_scale is normally NULL), stored K-tile major: the byte of (row r, 32-block
 * kb) at [(kb / 4) * M * 4 + r * 4 + kb % 4], readable up to 256-row granularity (allocate ceil(M / 256) * 256 rows per
 * plane).  c_mode 0 bf16, 1 fp32, 3 = fp8 e4m3 bytes [M][N] + block scales c_mx in the same layout (N % 128 == 0);
 * c_mode | (2 << 8) applies the tanh-GELU first (3 | 2 << 8 = the fc1 epilogue of the MMDiT engine's fp8 path). */
/* One plain GEMM C[M][N] = A[M][K] B[N][K]^T (bf16 operands) with the fused epilogue terms, any of which may be NULL: column bias,
 * row bias, per-sample row vector and gate ([samples][N], sample = row >> log_rows_per_sample), bf16 / fp32 residual [M][N], scale,
 * activation (0 none, 1 SiLU, 2 tanh-GELU); output bf16 or fp32 [M][N]; gn_part (optional) receives (sum, sum of squares) per
 * block tile of *bm_out rows and 4-column quad.  fp32_slab = 1 forces the general fp32-slab epilogue.  N % 8 == 0. */
int natinf_debug_gemm_fused(int variant, int M, int N, int K, const void* a, const void* b, const float* bias_n, const float* bias_m,
                            const float* rowvec, const float* gate, int log_rows_per_sample, const void* resid_bf16, const float* resid_f32,
                            float scale, int act, void* c, int c_f32, float* gn_part, int* bm_out, int fp32_slab, natinf_stream_t stream);
int natinf_debug_gemm_fp8(int M, int N, int K, const void* a8, const float* a_scale, const void* a_mx, const void* b8, const float* b_scale,
                          const float* bias_n, void* c, void* c_mx, int c_mode, int iters, natinf_stream_t stream);
/* Test hook for the up-sampling fetch paths of k_conv_gn2 (the up-sampling res-blocks, natinf_set_fuse_up): bit 0 = the following
 * natinf_debug_conv_gn calls take x at HALF the resolution ([B][res/2][res/2][cin]) and up-sample it (nearest, 2x) inside the patch fetch,
 * bit 1 = the same for the 1x1 shortcut operand a1.  0 restores the plain fetch. */
int natinf_debug_conv_gn_up(int flags);
/* The fused GroupNorm-apply + SiLU + 3x3 convolution kernel (csrc/conv_gn.h) on caller-supplied operands:
 *   out[m, n] = (sum_{tap, c} silu(x[pixel(m) + tap, c] * scale[b, c] + shift[b, c]) * w[n, c, tap] + sum_c a1[m, c] * w1[n, c]
 *                + bias_n[n] + resid[m, n]) * out_scale,  zero padding applied after the activation (layerspp.py:242-274).
 * x: bf16 [B][res][res][cin] (res 32 or 16, cin % 64 == 0); w_packed: bf16 [N][9*cin + c1], K order ((c/64)*9 + tap)*64 + c%64
 * followed by the c1 shortcut columns; a1: bf16 [B*res*res][c1] or NULL; resid: bf16 [M][N] or NULL; out: bf16 [M][N];
 * gn_part: NULL or [M/rows][N/4] (sum, sum of squares) of the fp32 outputs per pixel tile and 4-channel quad; rows = 128 for
 * 16x16 images with N % 256 == 0 (unless natinf_set_conv_gn_wide(0)), else 256.
 * Operands are taken in the kernel's FOLDED form: `scale` and `shift` must be the GroupNorm scale / shift multiplied by -log2(e)
 * and the 3x3 columns of w_packed multiplied by -ln 2 (the shortcut columns are plain); the kernel evaluates t / (1 + exp2(t)),
 * t = x*scale + shift = -log2(e) v, i.e. -log2(e) silu(v) -- the same function with two vector instructions fewer per element.
 * w_frag: NULL -> k_conv_gn (weights through an LDS ring); else a buffer of N * (9*cin + c1) bf16 that receives the fragment-major
 * copy of w_packed (k_pack_frag) and k_conv_gn2 runs (weights streamed through registers, csrc/conv_gn2.h) when N is a multiple of
 * its column tile (128; 256 for the 128-row tile) and natinf_set_conv_gn_regw is 1 (default). */
int natinf_debug_conv_gn(int res, int B, int N, int cin, int c1, const void* x, const float* scale, const float* shift, const void* w_packed, void* w_frag,
                         const void* a1, const float* bias_n, const void* resid, float out_scale, void* out, float* gn_part, int iters,
                         natinf_stream_t stream);
int natinf_set_gemm_variant(int variant);
/* A/B switch for tuning: 1 = every k_gemm_* launch takes the fp32-slab epilogue, 0 (default) = the packed bf16 epilogue where it applies.
   The fused GroupNorm + 3x3 convolution kernels (k_conv_gn2) have packed epilogues only and ignore the switch. */
int natinf_set_gemm_epilogue(int fp32_slab);
/* A/B switch for tuning: 0 = N <= 128 layers on the 4-wave 256x128 ring tile, 1 (default) = on the 512x128 hand-pipelined tile. */
int natinf_set_gemm_pref512(int on);
/* A/B switch for tuning: 1 (default) = in the 256x256 / 512x128 kernels one wave per SIMD issues all LDS-DMA pieces, 0 = every wave its own. */
int natinf_set_gemm_half_issue(int on);
/* A/B switch for tuning: 1 (default) = small-M plain GEMMs (fewer than two rounds of 256 x 256 tiles) choose between 256 x 256 and 128 x 128 tiles by the number of
 * ROUNDS of blocks each needs (128 x 128: two blocks per CU; a 256 x 256 round costs 1.5 of a 128 x 128 one, 1.3 on the four-wave tile of csrc/gemm_w128.h),
 * 0 = by the pre-round-4 rules; a value >= 10 sets the four-wave tile's ratio to value / 10 (tuning runs). */
int natinf_set_gemm_round_model(int on);
/* A/B switch for tuning: 1 (default) = plain GEMMs that took the 256 x 256 tile of eight waves (two per SIMD, 128 x 64 wave tiles) take the 256 x 256 x 64 tile of FOUR
 * waves (one per SIMD, 128 x 128 wave tiles, accumulators in AGPRs: csrc/gemm_w128.h) -- bf16 and e4m3 operands alike, the e4m3 GEMMs that write e4m3 + E8M0 behind a
 * tanh-GELU (fc1 of the MMDiT) included since round 5; 2 = all but those (the round-4 rule); 0 = the eight-wave tiles as before round 4. */
int natinf_set_gemm_w128(int on);
/* Tuning: row-tiles per raster group of launches with >= 8 column tiles (default 8; 0 = plain row-major tile order). */
int natinf_set_gemm_raster(int rows);
/* 1 (default): plans built from now on run GroupNorm-apply + SiLU inside the consuming 3x3 convolution where a fused kernel
 * exists (32x32 and 16x16 levels); 0: the separate normalisation pass everywhere (A/B runs, tests).  Read by natinf_ncsnpp_create. */
int natinf_set_fuse_gn(int on);
/* Bit mask (default 3; NATINF_EINVAL outside 0..3): fused-convolution launches with N % 256 == 0 use the 128-pixel x 256-channel tile (the patch is
 * normalised once for all 256 output channels) on 16x16 images (bit 0) and on 32x32 images (bit 1: the 16 -> 32 up-sampling block); 0: 256 x 128 tiles. */
int natinf_set_conv_gn_wide(int mask);
/* Bit mask (NATINF_EINVAL outside 0..7): fused-convolution launches on k_conv_gn3 (csrc/conv_gn3.h: four waves per block, one per SIMD, 128-pixel x
 * 128-channel wave tiles, AGPR accumulators, a K loop written slot by slot) -- bit 0: 32x32 images, N % 256 != 0, 512-pixel x 128-channel tiles; bit 1:
 * 32x32 images, N % 256 == 0, 256 x 256 tiles; bit 2: 16x16 images, N % 256 == 0, 256 x 256 tiles (one image per tile).  The convolution sums are the
 * bytes k_conv_gn2 gives (same K order, same normalisation arithmetic); GroupNorm partial rows then cover 512 / 256 pixels (natinf_debug_conv_gn: rows). */
int natinf_set_conv_gn_w128(int mask);
/* Smallest K (9 * cin + shortcut channels) at which a launch of shape 0 (32x32, 512 x 128 tiles), 1 (32x32, 256 x 256) or 2 (16x16, 256 x 256) takes k_conv_gn3
 * (defaults 2304 / 0 / 2304: with one block per CU a tile's prologue and epilogue are exposed, so short-K launches stay on k_conv_gn2); 0 = every K (tests, A/B runs). */
int natinf_set_conv_gn_w128_min_k(int shape, int k);
int natinf_set_conv_gn_regw(int on);
/* 1 (default; read when a plan is built): the up-sampling blocks at 16x16 / 32x32 read their half-resolution input inside the fused
 * convolution (nearest up-sampling in the patch fetch and in the residual fetch); 0: through the separate GroupNorm-apply + up-sample pass. */
int natinf_set_fuse_up(int on);
/* 1 (default; read when a plan is built): the output head -- GroupNorm + SiLU + the 128 -> 3 convolution, fp32 NCHW -- is ONE launch
 * (k_head_conv); 0: GroupNorm-apply pass + implicit GEMM on an N = 3 column tile. */
int natinf_set_fuse_head(int on);
/* 1 (default; read when a plan is built): the 8x8 level's 3x3 convolutions run on the fused GroupNorm + SiLU + convolution kernel as well (two whole
 * images per 128-pixel x 256-channel tile, GroupNorm partials per sample); 0: GroupNorm-apply / statistics passes + implicit GEMM. */
int natinf_set_fuse_gn8(int on);
/* 1 (default; read when a plan is built): the 4x4 level's 3x3 convolutions run on the fused kernel too (four whole images per 64-pixel x
 * 256-channel tile; per-sample tables, row vectors and GroupNorm partials); 0: GroupNorm-apply pass + split-K implicit GEMM + reduce pass +
 * statistics pass. */
int natinf_set_fuse_gn4(int on);
/* 1 (default; read when a plan is built): where a tile of the fused convolution holds whole samples and every output channel, its epilogue writes the GroupNorm
 * (scale | shift) table of the tensor's single consumer itself instead of a k_gn_finalize launch behind it -- at 8x8 / 4x4 (k_conv_gn2's 64-pixel x 256-channel
 * tiles) and, since round 5, at 16x16 (k_conv_gn3's 256 x 256 tile IS one sample; when a tuning knob routes the launch to another kernel the consumer's op launches
 * k_gn_finalize after all).  The tables are the same bytes either way.  0: a k_gn_finalize launch per table, as at 32x32; 2: 8x8 / 4x4 only (the round-4 plan);
 * 3: 16x16 only. */
int natinf_set_fuse_fin(int on);
/* Bit mask by resolution (1: 4x4, 2: 8x8; bits 4 / 8 -- 16x16 / 32x32 -- are accepted and ignored: measured +-0 there, compiled out; read at launch) of the fused-convolution launches whose first blocks request the
 * whole weight matrix once at kernel start, so that the K loop's one-tap-ahead weight stream hits L2 inside a forward pass (where every layer's
 * weights are cold).  NATINF_EINVAL outside 0..15. */
int natinf_set_conv_gn_warm(int mask);
/* 1 (default): the 16x16 attention (256 tokens, one head of 256 channels) runs as k_attn256 -- K and V^T streamed through a two-stage LDS ring by
 * LDS-DMA, two blocks per CU; 0: k_attn_fused<8,16,true> (whole K, then whole V^T, resident in LDS; one block per CU) -- a -DNATINF_DEV kernel:
 * NATINF_ESTATE in the shipped library. */
int natinf_set_attn256(int on);
/* 1 (default; read when a plan is built): k_attn256 also applies the attention block's output projection, skip connection and rescale and writes the
 * GroupNorm partials of the block's output (the O tensor is never stored); 0: a separate GEMM launch for the projection. */
int natinf_set_attn_proj(int on);
/* 1 (default; read at launch): the projection-fused attention kernel runs one 8-wave block per sample (each K / V^T / W3 tile crosses L2 -> LDS once per
 * sample); 0: two 4-wave blocks of 128 queries per sample, two blocks per CU. */
int natinf_set_attn_waves8(int on);
/* 1 (default; read when a plan is built): GroupNorm-apply and the q | k | v projections of the 16x16 attention block run as ONE launch (k_qkv256:
 * the block input is read once, the normalised tensor is never stored); 0: k_gn_apply + the q | k GEMM + the batched V^T GEMM. */
/* (read when a plan is built; a negative value: the library's default = 2 since round 6)  1: the whole 16x16 attention block -- GroupNorm-apply + q | k | v projections, scores, softmax, P V, output projection, skip
 * and the GroupNorm partials of its output -- is ONE launch (csrc/attn_blk256.h: q stays in registers; k and V^T are written and re-read through L2 by the same block);
 * 0: k_qkv256 + k_attn256 (A/B runs); 2 (round 6): k_attn_blk256_v2 -- q k^T and P V taken against the normalised tokens h themselves (folded matrices Wq Wk^T and Wv W3,
 * exact in real arithmetic): two projections instead of four, h resident in LDS (read row-wise and, for P V, transposed: ds_read_b64_tr_b16), nothing written between the
 * phases.  Needs natinf_set_attn_qkv, natinf_set_attn_proj, natinf_set_attn_waves8 and natinf_set_attn256 at 1. */
int natinf_set_attn_block(int on);
int natinf_set_attn_qkv(int on);
/* Tile of the fused kernel on the 8x8 level: 1 (default) = 64 pixels x 256 channels (one image per tile, wave tile 64 x 64, two blocks per CU at
 * B = 512), 0 = 128 x 256 (two images per tile, one block per CU; 0.6 % slower per forward: a -DNATINF_DEV kernel -- NATINF_ESTATE in the shipped
 * library). */
int natinf_set_conv_gn8_tile(int one_image);
/* 1 (default): small-M, long-K launches (the 8x8 and 4x4 levels) run as 128 x 128 tiles x 2..4 K slices + a reduce pass; 0: never. */
int natinf_set_gemm_splitk(int on);
/* natinf_debug_gemm / natinf_debug_gemm_fused with variant 0 may split K when given a workspace of max_slices * M * N floats
 * (NULL = off, the default); the engines carry their own workspace. */
int natinf_debug_set_splitk_workspace(float* ws, int max_slices);
/* Timing experiments (tools/tile_timeline.py): device buffer of 16 uint64 shader-clock stamps that block 0 / thread 0 of
 * every natinf_debug_gemm launch writes (kernel start, first tile landed, main loop done, per epilogue pass: slab written,
 * sweeps done, stores issued).  NULL switches it off. */
int natinf_debug_timestamps(void* dev_buf16);

/* Measurement hooks (bench.py): while enabled, every launch group of a forward is bracketed by a HIP event
 * pair on `stream`.  natinf_ncsnpp_profile_read waits for the recorded events and returns, per class
 * (0 = the implicit-GEMM kernels k_gemm_*: unfused convolutions / NIN / linear / attention products;
 * 1 = everything else: GroupNorm statistics + apply, softmax, embedding, stem im2col;
 * 2 = k_conv_gn2 at 32x32 / 16x16: the 3x3 convolutions with GroupNorm-apply + SiLU fused into their operand path;
 * 3 = its 8x8 instantiation, two images per tile), the summed device time in milliseconds and the number of launches since the previous read. */
int natinf_ncsnpp_profile(natinf_ncsnpp_t h, int enable);
int natinf_ncsnpp_profile_read(natinf_ncsnpp_t h, double* ms_by_class /*[4]*/, int64_t* launches_by_class /*[4]*/);

/* Per-shape timing of the matmul-shaped launches WHERE THEY RUN (every engine of the library: NCSN++, DiT, MMDiT, VAE, Inception): while enabled, each
 * launch -- bf16 or fp8, with the epilogue the engine gives it, on the stream it runs on -- is bracketed by a HIP event pair.  natinf_gemm_profile_read waits
 * for the recorded events and writes one line per distinct launch description since the previous read, in first-seen order:
 *   "M N K K_shortcut taps batch kernel/eEPILOGUE launches total_ms\n"   (the first seven fields are natinf_ncsnpp_describe_gemms' line)
 * and returns the number of bytes written (negative: error; NATINF_EINVAL when `cap` is too small -- the records are consumed either way).
 * Process-global switch, one host thread at a time (as natinf_attention_profile).  bench.py's `sd3*.gemm` objects are these numbers. */
int natinf_gemm_profile(int enable);
int natinf_gemm_profile_read(char* buf, int cap);

/* After a forward on a KEEP_ACTIVATIONS handle: copy the output of all_modules[module_idx]
 * (module_idx >= 2) as fp32 NCHW into `out` (capacity in elements).  Same B / workspace as the forward. */
int natinf_ncsnpp_debug_tap(natinf_ncsnpp_t h, int module_idx, float* out, int64_t capacity_elems,
                            natinf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NATINF_NCSNPP_H */

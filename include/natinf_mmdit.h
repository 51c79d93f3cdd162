/*
 * natinf_mmdit.h -- C ABI of the SD3 MMDiT denoiser engine inside libnatinf.so.
 *
 * Replaces `pipe.transformer(hidden_states, timestep, encoder_hidden_states, pooled_projections)` at
 * src/SD3NaturalInference.py:111-114,210-213.  The reference takes that module from the un-vendored, un-pinned
 * `diffusers` (requirements.txt:13: SD3Transformer2DModel of stable-diffusion-3-medium); its arithmetic is restated in
 * oracle/mmdit_oracle.py from the published architecture (PARITY UNPINNED -- see that file's header) and this engine is
 * tested against that restatement.  Joint (image + text) transformer blocks with adaLN-Zero modulation, head_dim 64,
 * patch 2, latent channels `in_ch`; layers / heads / text dims / token counts are create-time parameters
 * (SD3-medium at 1024x1024: 24 layers, 24 heads, joint_dim 4096, pooled_dim 2048, in_ch 16, grid 64, 333 text tokens).
 *
 * Arithmetic: bf16 operands on the matrix cores, fp32 accumulation, fp32 residual streams, LayerNorm / softmax /
 * modulation in fp32.  Conventions of natinf.h (device pointers, explicit stream, int return codes, caller-owned
 * packed-weight buffer and workspace).
 */
#ifndef NATINF_MMDIT_H
#define NATINF_MMDIT_H

#include <stdint.h>
#include "natinf.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct natinf_mmdit* natinf_mmdit_t;

/* flags.  NATINF_MMDIT_FP8 (BASELINE config 5): every image-stream projection -- q|k, v, attention output (both streams),
 * fc1, fc2 -- runs on fp8 e4m3 operands with fp32 accumulation.  Weights: one scale per output channel, at load time.
 * LayerNorm-modulate outputs: one scale per token, by the kernel that produces them.  Attention and GELU outputs: MX block
 * scales (a power of two per row and 32 channels) written by their producers' epilogues and applied by the matrix
 * instruction itself.  Attention (Q K^T, P V) and the text-stream projections other than the attention output stay bf16.
 * Needs an even head count (hidden % 128 == 0). */
#define NATINF_MMDIT_FP8 1

/* grid = image tokens per side (latent side / 2), grid*grid % 8 == 0; ctx_tokens = text tokens per sequence;
 * hidden = 64*heads <= 1536; joint_dim % 8 == 0, pooled_dim % 8 == 0; in_ch even, <= 16. */
int natinf_mmdit_create(natinf_mmdit_t* out, int layers, int heads, int joint_dim, int pooled_dim, int in_ch, int grid,
                        int ctx_tokens, int flags);
int natinf_mmdit_destroy(natinf_mmdit_t h);
int64_t natinf_mmdit_param_count(natinf_mmdit_t h);
int64_t natinf_mmdit_packed_bytes(natinf_mmdit_t h);
int64_t natinf_mmdit_workspace_bytes(natinf_mmdit_t h, int max_batch);

/* params_f32: fp32, concatenated in this order (diffusers state-dict names; D = 64*heads):
 *   pos_embed.pos_embed CROPPED to the grid ([grid*grid][D], the centre window of the checkpoint's table),
 *   pos_embed.proj.{weight,bias}, time_text_embed.timestep_embedder.linear_1.{weight,bias}, .linear_2.{weight,bias},
 *   time_text_embed.text_embedder.linear_1.{weight,bias}, .linear_2.{weight,bias}, context_embedder.{weight,bias},
 *   per block i: norm1.linear.{w,b}, norm1_context.linear.{w,b}, attn.to_q, to_k, to_v, add_k_proj, add_v_proj, add_q_proj,
 *                to_out.0 (each {w,b}), [attn.to_add_out.{w,b}], ff.net.0.proj.{w,b}, ff.net.2.{w,b},
 *                [ff_context.net.0.proj.{w,b}, ff_context.net.2.{w,b}]      ([..]: absent in the last block),
 *   norm_out.linear.{weight,bias}, proj_out.{weight,bias}. */
int natinf_mmdit_load(natinf_mmdit_t h, const float* params_f32, int64_t n_params, void* packed, int64_t packed_bytes,
                      natinf_stream_t stream);

/* out = transformer(latents, timestep, text, pooled): latents / out [B, in_ch, 2*grid, 2*grid] fp32 NCHW,
 * timestep [B] fp32, text [B, ctx_tokens, joint_dim] fp32, pooled [B, pooled_dim] fp32. */
int natinf_mmdit_forward(natinf_mmdit_t h, const float* latents, const float* timestep, const float* text, const float* pooled,
                         float* out, int B, void* workspace, int64_t workspace_bytes, natinf_stream_t stream);

/* The attention kernel on its own: o = softmax(q k^T * scale) v per (sequence, head), head_dim 64, bf16.
 * q, k: [B][Tp][ld_qk] (head h at columns 64h..64h+63; per-sequence stride qk_bs elements); vT: [B][64*H][Tp] (V
 * TRANSPOSED: keys contiguous; must be finite at keys >= T); o: [B][Tp][ld_o].  Tp % 128 == 0, Tp - 128 < T <= Tp (the padding
 * must fit inside the last 128-key tile, the only one the kernel masks; anything else is NATINF_EINVAL); keys >= T are ignored,
 * query rows >= T produce unspecified values. */
int natinf_attention_hd64_bf16(const void* q, const void* k, int ld_qk, int64_t qk_bs, const void* vT, void* o, int ld_o,
                               int64_t o_bs, int B, int H, int Tp, int T, float scale, natinf_stream_t stream);

/* 1 (read when an engine is CREATED): the image tokens' residual stream x [Tx][D] is kept in IEEE half instead of fp32 -- the reference's SD3 pipeline is
 * fp16 end to end (src/SD3NaturalInference.py:175-176) --: every update x += gate * (W h + b) is computed in fp32 from the half row and rounded to half once
 * (2^-11 per update); the attention output projection / fc2 epilogues and the LayerNorm-modulate passes move half the bytes.  The text stream (ctx_tokens rows)
 * stays fp32.  0: both streams fp32; a negative value: the library's default (1 since round 6).  Workspace bytes per sequence differ: query natinf_mmdit_workspace_bytes after creating the engine. */
int natinf_set_mmdit_stream16(int on);
/* 1 (default): the text stream's fc1 (bias + tanh-GELU: no per-sequence term, operands contiguous over the sequences) runs as ONE GEMM over B * ctx_tokens rows; 0: batched
 * per sequence like the text stream's other GEMMs (A/B runs).  Read at every forward. */
int natinf_set_mmdit_text_flat(int on);

/* Measurement hook: while enabled, every k_flash_attn64 launch of this process (engine forwards and natinf_attention_hd64_bf16
 * alike) is bracketed by a HIP event pair on its own stream.  natinf_attention_profile_read synchronises on them and returns the
 * summed duration (ms) and the launch count since the last read.  Not thread-safe; for bench.py / tools. */
int natinf_attention_profile(int enable);
/* Which head_dim-64 attention kernel subsequent launches use (engine forwards and natinf_attention_hd64_bf16 alike): 0 = k_flash_attn64 (online
 * softmax, the running maximum updated every key tile; rounds 1-3), 3 (default) = k_flash_attn64_v2<1, 64>: q pre-multiplied by scale * log2 e,
 * scores leave the matrix pipe as exponents relative to a reference that is only moved when a score exceeds it by more than 2^8 (no per-score fma, no
 * cross-lane step on the common path), row sums on the matrix pipe, 64-key tiles (three to four waves per SIMD).  1 / 2 = intermediate forms of it
 * (vector-pipe row sums / 128-key tiles): -DNATINF_DEV builds only, NATINF_ESTATE elsewhere.  Same function up to rounding; NATINF_EINVAL outside 0..3. */
int natinf_set_flash_mode(int mode);
/* 1 (default): a forward puts the text stream's launches (M = ctx_tokens rows per sequence: small GEMMs that fill a fraction of the chip) on a HIP stream of the
 * engine's own, forked from and joined to the caller's stream by events -- in front of the joint attention of every block and behind it, and at both ends of the
 * forward, so that on return everything the forward enqueued is ordered on the caller's stream as before.  0: every launch on the caller's stream, one after the
 * other.  Same launches, same arguments: the outputs are the same bytes. */
int natinf_set_mmdit_text_stream(int on);
int natinf_attention_profile_read(double* ms_total, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* NATINF_MMDIT_H */

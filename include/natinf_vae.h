/*
 * natinf_vae.h -- C ABI of the AutoencoderKL decoder engine inside libnatinf.so.
 *
 * Replaces `vae.decode(latents)` at src/ValidateNaturalInference.py:231-236,298-303,366-371 (the step right after the
 * sampling loop).  The reference takes the module from the un-vendored, un-pinned `diffusers`
 * (AutoencoderKL of stabilityai/sd-vae-ft-ema); its arithmetic is restated in oracle/vae_oracle.py from the published
 * architecture (PARITY UNPINNED -- see that file's header) and this engine is tested against that restatement.
 * Decoder: conv_in, mid block (ResnetBlock, single-head attention over the pixels, ResnetBlock), four up blocks of three
 * ResnetBlocks (512, 512, 256, 128 channels; nearest 2x + conv after the first three), GroupNorm + SiLU + conv_out.
 * Latent resolution r is a power of two, 8 <= r <= 64 (the mid-block attention materialises r^2 x r^2 scores per image); the
 * 128x128 latents of SD3 at 1024x1024 need a flash attention at head_dim 512 and are not supported yet.
 *
 * Arithmetic: bf16 operands on the matrix cores, fp32 accumulation, GroupNorm statistics / softmax in fp32.
 */
#ifndef NATINF_VAE_H
#define NATINF_VAE_H

#include <stdint.h>
#include "natinf.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct natinf_vae* natinf_vae_t;

int natinf_vae_create(natinf_vae_t* out, int latent_ch, int latent_res);     /* latent_ch 1..64, latent_res 8, 16, 32, 64 or 128 */
int natinf_vae_destroy(natinf_vae_t h);
int64_t natinf_vae_param_count(natinf_vae_t h);
int64_t natinf_vae_packed_bytes(natinf_vae_t h);
int64_t natinf_vae_workspace_bytes(natinf_vae_t h, int max_batch);

/* params_f32: fp32, `post_quant_conv.{weight [C][C], bias}` of the AutoencoderKL (identity / zero if the model has none),
 * then the decoder's parameters in state-dict order of diffusers' `Decoder`:
 *   conv_in.{weight,bias}; mid_block.resnets.0.{norm1.{w,b}, conv1.{w,b}, norm2.{w,b}, conv2.{w,b}};
 *   mid_block.attentions.0.{group_norm.{w,b}, to_q.{w,b}, to_k.{w,b}, to_v.{w,b}, to_out.0.{w,b}}; mid_block.resnets.1.{..};
 *   up_blocks.i.resnets.j.{norm1, conv1, norm2, conv2, [conv_shortcut]} (i = 0..3, j = 0..2), up_blocks.i.upsamplers.0.conv (i < 3);
 *   conv_norm_out.{w,b}; conv_out.{w,b}. */
int natinf_vae_load(natinf_vae_t h, const float* params_f32, int64_t n_params, void* packed, int64_t packed_bytes, natinf_stream_t stream);

/* images = decoder(post_quant_conv(latents)) = AutoencoderKL.decode: latents [B, latent_ch, r, r] fp32 NCHW (already divided
 * by the scaling factor), images [B, 3, 8r, 8r] fp32 NCHW. */
int natinf_vae_decode(natinf_vae_t h, const float* latents, float* images, int B, void* workspace, int64_t workspace_bytes,
                      natinf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NATINF_VAE_H */

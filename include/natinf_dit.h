/*
 * natinf_dit.h -- C ABI of the DiT denoiser engine inside libnatinf.so.
 *
 * Replaces `model.forward(zt, timesteps, classlabels)` of src/ValidateNaturalInference.py:190-191, i.e.
 * DiT.forward (deps/DiT/models.py:237-253) with its blocks (:105-146), embedders (:27-99), fixed sin-cos position
 * embedding (:279-326) and unpatchify (:222-235); `timm`'s PatchEmbed / Attention / Mlp (models.py:16) follow their
 * published definitions.  Input size 32, patch 2, 4 input channels, 1000 classes (+1 null), learn_sigma (8 output
 * channels); depth / hidden size / head count are create-time parameters (DiT-XL/2 = 28 / 1152 / 16).
 *
 * Arithmetic: bf16 operands on the matrix cores, fp32 accumulation, fp32 residual stream, LayerNorm / softmax /
 * modulation in fp32.  Same conventions as natinf.h (device pointers, explicit stream, int return codes, caller-owned
 * packed-weight buffer and workspace).
 */
#ifndef NATINF_DIT_H
#define NATINF_DIT_H

#include <stdint.h>
#include "natinf.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct natinf_dit* natinf_dit_t;

/* Attention runs as one fused launch per block when head_dim <= 96; this flag selects the per-head GEMM / softmax /
 * GEMM path instead (the one used for larger heads), for testing one against the other. */
#define NATINF_DIT_UNFUSED_ATTENTION 1

/* hidden % 64 == 0, hidden <= 1536, hidden % heads == 0, (hidden / heads) % 8 == 0 */
int natinf_dit_create(natinf_dit_t* out, int depth, int hidden, int heads, int flags);
/* 1 (read when an engine is CREATED): the residual stream x [256 tokens][hidden] is kept in IEEE half instead of fp32; every update
 * x += gate * (W h + b) is computed in fp32 from the half row and rounded to half once (the MMDiT engine's natinf_set_mmdit_stream16,
 * include/natinf_mmdit.h).  0: fp32; a negative value: the library's default (1 since round 6). */
int natinf_set_dit_stream16(int on);
int natinf_dit_destroy(natinf_dit_t h);
int64_t natinf_dit_param_count(natinf_dit_t h);           /* incl. the frozen pos_embed */
int64_t natinf_dit_packed_bytes(natinf_dit_t h);
int64_t natinf_dit_workspace_bytes(natinf_dit_t h, int max_batch);

/* params_f32: all parameters, fp32, concatenated in this order (reference state-dict names):
 *   pos_embed, x_embedder.proj.{weight,bias}, t_embedder.mlp.0.{weight,bias}, t_embedder.mlp.2.{weight,bias},
 *   y_embedder.embedding_table.weight,
 *   blocks.<i>.{attn.qkv.weight, attn.qkv.bias, attn.proj.weight, attn.proj.bias, mlp.fc1.weight, mlp.fc1.bias,
 *               mlp.fc2.weight, mlp.fc2.bias, adaLN_modulation.1.weight, adaLN_modulation.1.bias}  for i = 0..depth-1,
 *   final_layer.linear.{weight,bias}, final_layer.adaLN_modulation.1.{weight,bias}. */
int natinf_dit_load(natinf_dit_t h, const float* params_f32, int64_t n_params, void* packed, int64_t packed_bytes,
                    natinf_stream_t stream);

/* out = model.forward(z, t, y): z [B,4,32,32] fp32 NCHW, t [B] fp32 timesteps, y [B] int32 class labels (1000 = null),
 * out [B,8,32,32] fp32 NCHW. */
int natinf_dit_forward(natinf_dit_t h, const float* z, const float* t, const int32_t* y, float* out, int B,
                       void* workspace, int64_t workspace_bytes, natinf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NATINF_DIT_H */

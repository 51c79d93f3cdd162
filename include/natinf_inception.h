/*
 * natinf_inception.h -- C ABI of the FID Inception-V3 pool3 engine inside libnatinf.so.
 *
 * Replaces `InceptionV3([BLOCK_INDEX_BY_DIM[2048]])(batch)[0]` in the reference's FID epilogue -- get_activation / calc_fid,
 * src/CIFAR10NaturalInference.py:44-86: uint8 HWC images / 255 -> NCHW -> pool3 features [n, 2048] -> (mean, covariance) -> Frechet
 * distance (naturaldiffusion_amd/fid_stats.py).  The reference takes the module from un-vendored, un-pinned `pytorch_fid`
 * (a subclass of torchvision's Inception3 with three FID patches); its arithmetic is restated in oracle/inception_oracle.py from the
 * published architecture (PARITY UNPINNED -- see that file's header) and this engine is tested against that restatement.
 * Network: bilinear resize to 299x299 (align_corners = False), 2x - 1, Conv2d_1a .. Conv2d_4a with two 3x3 / stride-2 max pools,
 * Mixed_5b-5d (FIDInceptionA), Mixed_6a, Mixed_6b-6e (FIDInceptionC), Mixed_7a, Mixed_7b (FIDInceptionE_1), Mixed_7c (FIDInceptionE_2),
 * global average pool.  BatchNorm (eval, eps 1e-3) is folded into the filters when the weights are packed.
 *
 * Arithmetic (default plan; natinf_set_inception_conv below): IEEE-half operands on the matrix cores -- activations, and filters as one half-precision term per
 * power-of-two-scaled row --, fp32 accumulation, column scale + bias + ReLU in fp32, activations stored as half; features fp32.  Plans 0 / 1: bf16 operands, filters as
 * two bf16 terms, activations stored as bf16.
 */
#ifndef NATINF_INCEPTION_H
#define NATINF_INCEPTION_H

#include <stdint.h>
#include "natinf.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct natinf_inception* natinf_inception_t;

enum { NATINF_INCEPTION_U8_HWC = 0,     /* images: uint8 [B][H][W][3], the tensor the reference's samplers produce (to_pixel) */
       NATINF_INCEPTION_F32_CHW = 1 };  /* images: fp32 [B][3][H][W] in [0, 1], what the reference hands to the module */

int natinf_inception_create(natinf_inception_t* out, int in_h, int in_w);    /* input image size (32 x 32 for CIFAR10) */
int natinf_inception_destroy(natinf_inception_t h);
/* A/B switch, read by natinf_inception_create: 2 (default) = every convolution is one implicit-GEMM launch (csrc/conv_ring.h: the activation tensor is the A
 * operand, a K-tile = one tap x 32 channels) on HALF-PRECISION activations and one half-precision filter term (rows scaled by a power of two); 1 = the same launches
 * on bf16 activations with the filters as two bf16 terms; 0 = the round-3 / 4 plan (bf16, two terms, an im2col pass + a GEMM per convolution).  The parameter order
 * and the features' meaning are the same; packed_bytes / workspace_bytes differ (ask the handle). */
int natinf_set_inception_conv(int mode);
/* The engine's convolution kernel on caller-supplied operands (tests): out[b, oy, ox, n] = relu(col_scale[n] * sum_{ky, kx, c} x[b, oy*stride - ph + ky, ox*stride - pw + kx, c]
 * * w[n, (ky*kw + kx)*cin_p + c] + bias[n]), zero padding.  x: 16-bit [B][H][W][x_ld] (bf16, or IEEE half when f16 = 1), cin_p % 32 == 0 channels read per pixel;
 * w_packed: 16-bit [cout][passes * kh*kw*cin_p] (passes = 2: the second block of columns is a second filter term over the same taps); bias / col_scale: fp32 [cout] or NULL;
 * zeros: >= 16 zero bytes on the device (what padding taps fetch); out: 16-bit [B*Ho*Wo][out_ld], cout % 8 == 0.  tile: 0 = 256 x 64, 1 = 256 x 128, 2 = 128 x 128,
 * 3 = 128 x 192 (f16 only). */
int natinf_debug_conv_ring(int tile, int f16, int B, int H, int W, int cin_p, int x_ld, int cout, int kh, int kw, int stride, int ph, int pw, int passes,
                           const void* x, const void* w_packed, const float* bias, const float* col_scale, const void* zeros, void* out, int out_ld,
                           natinf_stream_t stream);
int64_t natinf_inception_param_count(natinf_inception_t h);
int64_t natinf_inception_packed_bytes(natinf_inception_t h);
int64_t natinf_inception_workspace_bytes(natinf_inception_t h, int max_batch);

/* params_f32: for every BasicConv2d on the pool3 path, in torchvision registration order (Conv2d_1a_3x3, Conv2d_2a_3x3, Conv2d_2b_3x3,
 * Conv2d_3b_1x1, Conv2d_4a_3x3, Mixed_5b.{branch1x1, branch5x5_1, branch5x5_2, branch3x3dbl_1..3, branch_pool}, Mixed_5c, Mixed_5d,
 * Mixed_6a.{branch3x3, branch3x3dbl_1..3}, Mixed_6b..6e.{branch1x1, branch7x7_1..3, branch7x7dbl_1..5, branch_pool},
 * Mixed_7a.{branch3x3_1, branch3x3_2, branch7x7x3_1..4}, Mixed_7b / 7c.{branch1x1, branch3x3_1, branch3x3_2a, branch3x3_2b,
 * branch3x3dbl_1, branch3x3dbl_2, branch3x3dbl_3a, branch3x3dbl_3b, branch_pool}):
 *   conv.weight [out][in][kh][kw], bn.weight, bn.bias, bn.running_mean, bn.running_var  (AuxLogits and fc are not on the path). */
int natinf_inception_load(natinf_inception_t h, const float* params_f32, int64_t n_params, void* packed, int64_t packed_bytes,
                          natinf_stream_t stream);

/* features[b][0..2047] = pool3 activations of image b (fp32). */
int natinf_inception_forward(natinf_inception_t h, const void* images, int input_kind, float* features, int B, void* workspace,
                             int64_t workspace_bytes, natinf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NATINF_INCEPTION_H */

/*
 * natinf.h -- C ABI of libnatinf.so, the MI355X (gfx950) Natural Inference engine.
 *
 * The reference (blairstar/NaturalDiffusion) has no FFI of its own: its hot path is
 * eager PyTorch inside three scripts.  Each entry point below replaces the Python
 * lines cited next to it; INTEGRATION.md shows the ctypes stub a maintainer of the
 * reference would add to route those lines here.
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer owned by the caller (PyTorch allocator);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL =
 *     the null stream); nothing here allocates, frees, synchronises or throws;
 *   - return value: NATINF_OK (0) or a negative NATINF_E* code; natinf_strerror() names it;
 *   - "E" is the element count of one state tensor (B*C*H*W), must be a multiple of the
 *     vector width stated per call; tensors are contiguous unless a stride is given;
 *   - coefficient rows are passed as *sparse rows*: n_terms (index, value) pairs in
 *     ASCENDING index order, the order in which the reference's Python loops accumulate.
 *     Dropping a zero coefficient is bit-identical to keeping it for finite data
 *     (x*0 = +-0, acc + +-0 = acc); pass the zeros too ("dense rows") to reproduce the
 *     reference for non-finite data as well.  The arrays live in device memory.
 *   - all arithmetic is performed in the operand types and in the ORDER of the reference,
 *     one IEEE rounding per reference operation (the library is built -ffp-contract=off):
 *     results are bit-identical to the reference's CPU PyTorch path.
 */
#ifndef NATINF_H
#define NATINF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NATINF_ABI_VERSION 1

#define NATINF_OK        0
#define NATINF_EINVAL   (-1)   /* bad argument (null pointer, E not a multiple of the vector width, ...) */
#define NATINF_ELAUNCH  (-2)   /* the HIP runtime rejected the launch */
#define NATINF_ENODEV   (-3)   /* no gfx950 device / code object for this device */
#define NATINF_ESTATE   (-4)   /* handle in the wrong state */

typedef void* natinf_stream_t;              /* hipStream_t */

int         natinf_abi_version(void);
const char* natinf_strerror(int code);
/* 0 if a gfx950 device is current and the code object loads; NATINF_ENODEV otherwise. */
int         natinf_probe(void);

/* ------------------------------------------------------------------------------------------
 * CIFAR10 form: fp32 model I/O, fp64 x0 history, fp64 accumulate.
 * Replaces, for one step k of one batch,
 *   src/CIFAR10NaturalInference.py:219-230 (data_fn: score -> x0_hat in fp64),
 *   deps/score_sde_pytorch/models/utils.py:157 (score = -out/std, fp32),
 *   src/CIFAR10NaturalInference.py:233-238 (weighted_sum) and :299-304 (append, B[k,0]*noise, add).
 *
 *   s      = (-model_out) / std_f32                       (fp32)
 *   x0_k   = ((double)s * (sigma*sigma) + (double)x_k) / alpha        (fp64; hist[k] <- x0_k)
 *   acc    = sum over terms t (ascending idx[t] < k) of hist[idx[t]] * val[t], then + x0_k * c_diag
 *   x_next = (float)acc + b0_f32 * noise                  (fp32)
 *
 * hist: [n_slots][E] fp64, slot k is written, slots idx[t] are read.  E % 4 == 0.
 * c_diag = C[k][k] (the coefficient of the x0 computed in this very call; pass 0.0 for none).
 * ------------------------------------------------------------------------------------------ */
int natinf_step_f64hist(const float* x_k, const float* model_out, const float* noise,
                        double* hist, float* x_next,
                        const int32_t* idx, const double* val, int n_terms, double c_diag,
                        int k, double alpha, double sigma, float std_f32, float b0_f32,
                        int64_t E, natinf_stream_t stream);

/* The same step with an fp32 history and fp32 FMA accumulation ("fast mode": half the
 * history bytes; NOT bit-identical to the reference, error <= a few fp32 ulp per term). */
int natinf_step_f32hist(const float* x_k, const float* model_out, const float* noise,
                        float* hist, float* x_next,
                        const int32_t* idx, const float* val, int n_terms, float c_diag,
                        int k, float alpha, float sigma, float std_f32, float b0_f32,
                        int64_t E, natinf_stream_t stream);

/* src/CIFAR10NaturalInference.py:233-238 on its own: out = (float) sum_t hist[idx[t]]*val[t]. */
int natinf_weighted_sum_f64(const double* hist, float* out,
                            const int32_t* idx, const double* val, int n_terms,
                            int64_t E, natinf_stream_t stream);

/* Initial noise for batch-sharded generation (replaces the sequential torch.manual_seed(888) + torch.randn of
 * src/CIFAR10NaturalInference.py:285-290, which cannot be split over GPUs): out[i][e], i < n_images, e <
 * elems_per_image, ~ N(0,1) from Philox4x32-10 with key = seed and counter = (global image index, e/4), Box-Muller.
 * The global index of row i is image_index[i] (device array) or, when image_index is NULL, first_index +
 * i*index_stride.  The same (seed, global index) gives the same image for any GPU count or batch split.
 * elems_per_image % 4 == 0. */
int natinf_randn_philox_f32(float* out, int64_t n_images, int64_t elems_per_image, const int64_t* image_index,
                            int64_t first_index, int64_t index_stride, uint64_t seed, natinf_stream_t stream);

/* src/CIFAR10NaturalInference.py:212-216 (to_pixel): x [B,C,H,W] fp32 -> uint8 [B,H,W,C] =
 * trunc(clip(x*255, 0, 255)); with centered != 0 the inverse scaler of datasets.py:32-38,
 * x <- (x+1)/2, is applied first (the reference calls the two back to back, :308-309). */
int natinf_to_pixel_u8(const float* x, uint8_t* out, int B, int C, int H, int W, int centered,
                       natinf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Validate (DiT, eps-prediction) form: fp32 tensors, fp32 products accumulated in fp64.
 * Replaces src/ValidateNaturalInference.py:193 (CFG fuse), :355 (pred_x0), :198-204 x2
 * (weighted_sum over x0 and over eps histories) and :362-366.
 *
 *   eps    = uncond ? uncond + cfg*(cond - uncond) : cond                 (fp32, three roundings)
 *   x0_k   = c1_f32*z - c2_f32*eps                                        (fp32; hist_x0[k] <- x0_k)
 *   a      = sum_t (double)(hist_x0[idx_c[t]] * val_c[t])  [+ (double)(x0_k*c_diag)]   (fp64 adds)
 *   b      = sum_t (double)(hist_eps[idx_b[t]] * val_b[t])                              (fp64 adds)
 *   z_next = (float)a + (float)b
 *
 * cond/uncond may be strided views of a [B, 2*C, H, W] DiT output (learn_sigma): element e of
 * sample n sits at n*eps_sample_stride + (e % sample_elems).  hist_eps [n_slots+1][E] already
 * holds every noise the row refers to (slot 0 = initial noise, slot j = the draw after step j-1).
 * E % 4 == 0, sample_elems % 4 == 0.
 * ------------------------------------------------------------------------------------------ */
int natinf_step_f32prod(const float* z, const float* cond, const float* uncond, float cfg,
                        int64_t sample_elems, int64_t eps_sample_stride,
                        float* hist_x0, const float* hist_eps, float* z_next,
                        const int32_t* idx_c, const float* val_c, int n_c, float c_diag,
                        const int32_t* idx_b, const float* val_b, int n_b,
                        int k, float c1_f32, float c2_f32,
                        int64_t E, natinf_stream_t stream);

/* src/ValidateNaturalInference.py:198-204 on its own. */
int natinf_weighted_sum_f32prod(const float* hist, float* out,
                                const int32_t* idx, const float* val, int n_terms,
                                int64_t E, natinf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * SD3 (flow, all-fp16 chain) form.  Replaces src/SD3NaturalInference.py:157-168 (row-normalised
 * weighted mean), :209 (next model input), :215-219 (x0 from velocity, CFG fuse, append); with
 * NATINF_SD3_CFG_ON_VELOCITY it is the Euler twin, :61-69 and :117-129.
 *
 *   default:  x0n = x - sig*v_null ; x0t = x - sig*v_text ; f = x0n + cfg*(x0t - x0n)
 *   velocity: v = v_null + cfg*(v_text - v_null) ; f = x - sig*v
 *   hist[k] <- f
 *   acc    = fp16 chain over terms t (ascending idx[t] < k) of hist[idx[t]]*val[t], then f*c_diag
 *   mean   = acc / w_total                                   (-> mean_out, may be NULL)
 *   x_next = sig_next*noise + one_minus_sig_next*mean        (-> x_next, may be NULL on the last step)
 *
 * Every product / sum / quotient is formed in fp32 from fp16 operands and rounded to fp16, as
 * eager PyTorch does; scalars are fp32 (`sig`, `sig_next`, `one_minus_sig_next` must already hold
 * the fp16-rounded value of the 0-d tensor the reference multiplies with).  E % 8 == 0.
 * ------------------------------------------------------------------------------------------ */
#define NATINF_SD3_CFG_ON_VELOCITY 1
int natinf_step_f16chain(const void* x, const void* v_text, const void* v_null, const void* noise,
                         void* hist, void* mean_out, void* x_next,
                         const int32_t* idx, const float* val, int n_terms, float c_diag, float w_total,
                         int k, float sig, float sig_next, float one_minus_sig_next, float cfg,
                         int flags, int64_t E, natinf_stream_t stream);

/* src/SD3NaturalInference.py:209 on its own: out = sig*noise + one_minus_sig*mean (fp16 ops; mean may be
 * NULL = zeros, the k = 0 case of :207). */
int natinf_flow_input_f16(const void* noise, const void* mean, void* out, float sig, float one_minus_sig,
                          int64_t E, natinf_stream_t stream);

/* src/SD3NaturalInference.py:157-168 on its own (fp16 in, fp16 out). */
int natinf_weighted_mean_f16(const void* hist, void* out,
                             const int32_t* idx, const float* val, int n_terms, float w_total,
                             int64_t E, natinf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * NCSN++ / DDPM++ denoiser (CIFAR10 VP continuous) -- replaces the `model(x, labels)` call of
 * deps/score_sde_pytorch/models/utils.py:144-160, i.e. NCSNpp.forward
 * (deps/score_sde_pytorch/models/ncsnpp.py:232-381) under
 * configs/vp/cifar10_ddpmpp_continuous.py:41-64.  Declared in natinf_ncsnpp.h.
 * ------------------------------------------------------------------------------------------ */

#ifdef __cplusplus
}
#endif
#endif /* NATINF_H */

/*
 * ia_hip.h -- C ABI of libia_hip.so, the MI355X (gfx950) backend for the InvertAvatar
 * generator forward pass.
 *
 * This is the drop-in boundary.  Each entry point replaces one
 * pybind11 plugin function or one PyTorch-level stage of the reference and is what a
 * reference-side binding (ctypes) would call.  Conventions:
 *
 *   - plain pointers + explicit sizes/strides, no torch types;
 *   - every pointer is a DEVICE pointer unless its name starts with `h_`;
 *   - the CALLER allocates all outputs and scratch; the library keeps no pointer past
 *     the call and owns no device memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - return value: 0 on success, negative ia_status on failure; the message of the last
 *     failure on the calling thread is available from ia_last_error();
 *   - nothing throws across the ABI; the library is re-entrant: it keeps no mutable state that a result depends on.  The one
 *     mutable device word per kernel module is the fp16 range watch (a sticky diagnostic flag, OR-ed by the kernels that write the
 *     split format when a value leaves the fp16 range, read and cleared by ia_split_saturation_poll -- an errno, not an input).
 *
 * Element types are named by ia_dtype.  "f16" is IEEE binary16 (the reference's c10::Half).
 */
#ifndef IA_HIP_H_
#define IA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IA_HIP_ABI_VERSION 7      /* 7 (r06, additive): ia_tokens_split / _t, ia_layernorm_split, ia_dwconv3x3_tokens_split, ia_im2col_split, ia_linear_sx / _splitk / _splitk_plan, ia_matmul_sx, ia_softmax_split, ia_attention_sx / _supported; 6 (r06): ia_render_rays (+ rgb_split, rgb_split_styles, rgb_split_planes), + ia_render_rays_box, ia_ray_limits_box / _parts; 5 (r05; ia_conv2d_mfma_sx_rgb narrowed to n <= 3 fused ToRGB channels, otherwise additive): ia_conv2d_down_sx / _plan, ia_conv3x3_s2_tiny, ia_bn_train_split, ia_convgru_gates_split / _update_split, ia_dwconv3x3_tokens, ia_se_gate_split, ia_upsample_bilinear_add; 4 (r04): + ia_upconv2d_rows_sx / _plan, ia_mouth_edge_blur, ia_split_saturation_poll, ia_conv2d_sx_supported; - ia_conv2d_small; 3 (r03, additive): ia_torgb, ia_upconv2d_fir_sx; 2 (r03): ia_render_rays (+ u_importance), ia_act_split (+ shift), ia_conv2d_mfma_sx (+ prelu_alpha), ia_uv_rasterize (+ binarize_mask) */

typedef enum ia_status {
    IA_OK = 0,
    IA_ERR_INVALID_ARG = -1,   /* the reference's TORCH_CHECK failures            */
    IA_ERR_UNSUPPORTED = -2,   /* valid per the reference API, no kernel here yet  */
    IA_ERR_LAUNCH = -3,        /* hipGetLastError() after the launch               */
    IA_ERR_NO_DEVICE = -4
} ia_status;

typedef enum ia_dtype { IA_F32 = 0, IA_F16 = 1, IA_F64 = 2 } ia_dtype;

/* Activation ids are the reference's `cuda_idx` (torch_utils/ops/bias_act.py:23-33). */
typedef enum ia_act {
    IA_ACT_LINEAR = 1, IA_ACT_RELU = 2, IA_ACT_LRELU = 3, IA_ACT_TANH = 4, IA_ACT_SIGMOID = 5,
    IA_ACT_ELU = 6, IA_ACT_SELU = 7, IA_ACT_SOFTPLUS = 8, IA_ACT_SWISH = 9
} ia_act;

/* ABI version of the loaded library (== IA_HIP_ABI_VERSION it was built with). */
int ia_version(void);

/* Copies the calling thread's last error message (NUL-terminated, truncated to n) and
 * returns its full length.  Replaces the C++ exception text of TORCH_CHECK
 * (torch_utils/ops/bias_act.cpp:39-55). */
size_t ia_last_error(char* h_buf, size_t n);

/* Number of HIP devices visible, or a negative ia_status. */
int ia_device_count(void);

/*
 * Fused bias + activation + gain + clamp.
 * Replaces bias_act_plugin.bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp)
 * (torch_utils/ops/bias_act.cpp:36-93; kernel bias_act.cu:27-151).
 *   x, y       : `numel` elements of `dtype`, any dense layout (indexing is by memory offset)
 *   b          : `size_b` elements of `dtype` or NULL; element i uses b[(i / step_b) % size_b],
 *                step_b = x.stride(dim) exactly as bias_act.cpp:77
 *   xref/yref/dy : NULL for grad == 0; same layout as x otherwise (bias_act.py:181,200)
 *   grad       : 0 forward, 1 first-order, 2 second-order gradient kernel
 *   clamp < 0  : disabled
 */
int ia_bias_act(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y,
                int dtype, int64_t numel, int size_b, int64_t step_b,
                int grad, int act, float alpha, float gain, float clamp, void* stream);

/*
 * Up-sample (zero insert) -> pad/crop -> 2-D FIR -> down-sample, per channel.
 * Replaces upfirdn2d_plugin.upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1,
 * flip, gain) (torch_utils/ops/upfirdn2d.cpp:20-102; kernels upfirdn2d.cu:33-204).
 *   x          : [n, c, in_h, in_w] of `dtype`, element strides x_stride[4] (N,C,H,W order)
 *   f          : [f_h, f_w] float32, element strides f_stride[2] (H,W order)
 *   y          : [n, c, out_h, out_w] of `dtype`, element strides y_stride[4]; the caller sizes
 *                it with out = (in*up + pad0 + pad1 - f + down) / down (upfirdn2d.cpp:39-40)
 *   flip != 0  : correlate with f as given; flip == 0 : true convolution (f flipped)
 */
int ia_upfirdn2d(const void* x, const float* f, void* y, int dtype,
                 int n, int c, int in_h, int in_w, const int64_t* h_x_stride,
                 int f_h, int f_w, const int64_t* h_f_stride,
                 int out_h, int out_w, const int64_t* h_y_stride,
                 int upx, int upy, int downx, int downy, int padx0, int pady0,
                 int flip, float gain, void* stream);

/*
 * ia_upfirdn2d with the tail of a StyleGAN2 SynthesisLayer fused in:
 *     y = clamp(act(FIR(x) + noise * noise_strength + bias[c]) * act_gain)
 * Replaces the upfirdn2d call of conv2d_resample (torch_utils/ops/conv2d_resample.py:128) plus the noise add
 * and bias_act of SynthesisLayer.forward (training/networks_stylegan2.py:318-329) for up-sampling layers.
 *   x  : [n, c, in_h, in_w] contiguous NCHW of `dtype` (f32/f16);  f : [f_h, f_w] contiguous float32
 *   noise : [out_h*out_w] float32 or NULL;  noise_strength : device scalar float32 or NULL (= 1)
 *   bias  : [c] of `dtype` or NULL;  act : IA_ACT_LINEAR | IA_ACT_LRELU;  clamp < 0 disables
 *   up in {1, 2}, 4x4 filter; other shapes return IA_ERR_UNSUPPORTED (use ia_upfirdn2d + ia_bias_act).
 */
int ia_upfirdn2d_bias_act(const void* x, const float* f, const float* noise, const float* noise_strength,
                          const void* bias, void* y, int dtype, int n, int c, int in_h, int in_w,
                          int f_h, int f_w, int out_h, int out_w, int up, int padx0, int pady0,
                          int flip, float fir_gain, int act, float alpha, float act_gain, float clamp, void* stream);

/*
 * Dense fp32 convolution of one StyleGAN2 layer on MFMA (v_mfma_f32_32x32x2_f32), with the modulation,
 * demodulation and the layer tail fused:
 *     y[b,o] = clamp(act(demod[b,o] * conv(x[b] * styles[b,:], w)[o] + noise * noise_strength + bias[o]) * gain) + residual
 * Replaces modulated_conv2d (training/networks_stylegan2.py:34-91) -> conv2d_resample
 * (torch_utils/ops/conv2d_resample.py:114-136) -> cuDNN conv2d / conv_transpose2d
 * (torch_utils/ops/conv2d_gradfix.py:37-45), and for the stride-1 form the bias_act after it (:327-329).
 *   x        : [B, I, H, W] float32 contiguous
 *   wk       : weights repacked as [ksize*ksize][I][O] float32 (tap-major, O contiguous) from the reference's
 *              [O, I, kh, kw]; tap = ky*ksize + kx, no spatial flip
 *   styles   : [B, I] or NULL;  demod : [B, O] or NULL (see ia_modconv_demod)
 *   transposed = 0 : stride 1, zero padding ksize/2 (correlation, as F.conv2d);  y : [B, O, H, W]
 *   transposed = 1 : 3x3, stride 2, no padding = F.conv_transpose2d(x, w^T, stride=2);  y : [B, O, 2H+1, 2W+1];
 *                    only `demod` is applied (noise/bias/residual must be NULL, act linear) -- the FIR and the
 *                    tail follow in ia_upfirdn2d_bias_act
 *   ksplit   : number of stream-K workers per batch element (from ia_conv2d_plan; 0 when the plan has none).  Tiles that
 *              fill whole rounds of the machine run one per workgroup; the (tile, K-chunk) units of the remaining tiles are
 *              cut into `ksplit` equal ranges, and tiles shared between workers are summed in worker order by a
 *              deterministic fix-up pass
 *   scratch  : caller-owned accumulator slabs for that pass, `scratch_bytes` >= the planned size (NULL/0 when the plan needs
 *              none).  Launches that may run concurrently must not share a scratch buffer.
 */
int ia_conv2d_mfma(const float* x, const float* wk, const float* styles, const float* demod,
                   const float* noise, const float* noise_strength, const float* bias, const float* residual,
                   float* y, float* scratch, size_t scratch_bytes,
                   int B, int I, int O, int H, int W, int ksize, int transposed,
                   int act, float alpha, float gain, float clamp, int ksplit, void* stream);

/*
 * ia_conv2d_mfma with fp16 operands on v_mfma_f32_32x32x16_f16 and fp32 accumulation: the arithmetic of the reference's
 * fp16 blocks (modulated_conv2d with x.dtype == float16, training/networks_stylegan2.py:34-91; the SR head with
 * sr_num_fp16_res > 0, training_avatar_texture/superresolution.py:209-216).  The style-scaled input and the weights are
 * rounded to fp16, everything else (demodulation, noise, bias, activation, clamp, storage) stays fp32.
 *   wk_h : weights as fp16, packed [ksize*ksize][I/8][O][8] (channel octets innermost), from the reference's [O, I, kh, kw]
 * Same arguments, plan and scratch as ia_conv2d_mfma.  Covers 3x3 layers that run on the two-stage tiles (stride-1 layers
 * with O >= 128 and >= 32^2 outputs; every stride-2 transposed layer) with I % 8 == 0 and O % 4 == 0; other shapes return
 * IA_ERR_INVALID_ARG and the caller uses ia_conv2d_mfma.
 */
int ia_conv2d_mfma_h(const float* x, const void* wk_h, const float* styles, const float* demod,
                     const float* noise, const float* noise_strength, const float* bias, const float* residual,
                     float* y, float* scratch, size_t scratch_bytes,
                     int B, int I, int O, int H, int W, int ksize, int transposed,
                     int act, float alpha, float gain, float clamp, int ksplit, void* stream);

/*
 * ia_conv2d_mfma with fp32-equivalent products formed from fp16 pairs on v_mfma_f32_32x32x16_f16 (fp32 accumulation):
 * every operand v is split as hi = fp16(v), lo = fp16(v - hi) -- 22 mantissa bits together -- and a*b is taken as
 * a_hi*b_hi + a_hi*b_lo + a_lo*b_hi.  The dropped a_lo*b_lo is <= 2^-22 |a*b|, the size of fp32's own rounding, so the
 * result is an fp32 convolution (measured against an fp64 convolution it is as close as ia_conv2d_mfma) at 3/16 of the fp32
 * MFMA's cycles.  The MFMA flushes fp16 denormals, so every factor is kept a normal number: the weights are scaled by
 * 2^wk_exp at pack time (wk_exp chosen per layer so that max|w| * 2^wk_exp <= 32768), the low parts of the activations by
 * 2^11, and the high part of a weight is multiplied by 2^-11 (exact) where it meets an activation's low part -- all three
 * products then carry the factor 2^wk_exp and go to ONE fp32 accumulator, which is multiplied by 2^-wk_exp at the end.
 * Range: x * style saturates at +-65504 (StyleGAN2 activations are O(1)-O(100); the reference clamps its fp16 blocks at
 * 256); weights below 2^(-3-wk_exp) lose the a_hi*b_lo correction, activations below 2^-14 = 6.1e-5 are carried by their
 * (scaled) low part alone, i.e. with 11 mantissa bits.
 *   wk_split : fp16 [2 (hi, lo)][ksize*ksize][I/8][O][8] of w * 2^wk_exp
 *   wk_exp   : the power of two above
 * Other arguments, plan, scratch and shape coverage as ia_conv2d_mfma_h.
 */
int ia_conv2d_mfma_s(const float* x, const void* wk_split, int wk_exp, const float* styles, const float* demod,
                     const float* noise, const float* noise_strength, const float* bias, const float* residual,
                     float* y, float* scratch, size_t scratch_bytes,
                     int B, int I, int O, int H, int W, int ksize, int transposed,
                     int act, float alpha, float gain, float clamp, int ksplit, void* stream);

/* Host-only planner for ia_conv2d_mfma (form 0), ia_conv2d_mfma_h (form 1) and ia_conv2d_mfma_s (form 2): stream-K worker count
 * (0: none) and the scratch bytes the fix-up pass needs.  The tile family, and with it the plan, depends on the form. */
int ia_conv2d_plan(int B, int I, int O, int H, int W, int ksize, int transposed, int form, int* h_ksplit, size_t* h_scratch_bytes);

/*
 * Demodulation coefficients demod[b,o] = rsqrt(sum_i styles[b,i]^2 * wsq[o,i] + 1e-8) with
 * wsq[o,i] = sum_{ky,kx} w[o,i,ky,kx]^2 (training/networks_stylegan2.py:60-64).
 */
int ia_modconv_demod(const float* styles, const float* wsq, float* demod, int B, int I, int O, void* stream);

#define IA_RENDER_WHITE_BACK 1
#define IA_RENDER_RGB_CHANNEL_MAJOR 2
#define IA_RENDER_DIST_PER_FRAME 4
#define IA_RENDER_FLIP_Z 8            /* ia_render_rays_box only */

/*
 * The fused importance renderer: one launch replaces ImportanceRenderer_bsMotion.forward(evaluation=True)
 * (training_avatar_texture/volumetric_rendering/renderer.py:309-351) together with sample_from_planes (:85-97),
 * OSGDecoder.forward (training_avatar_texture/triplane_v20.py:426-438) and MipRayMarcher2.run_forward
 * (volumetric_rendering/ray_marcher.py:25-57).
 *   planes_cl : [B, 3, plane_h, plane_w, 32] float32 -- the tri-planes in CHANNELS-LAST order (the reference's
 *               [B, 3, 32, H, W] permuted so that a texel's 32 channels are one 128-byte line)
 *   rays_o/d  : [B, R, 3];  jitter : [B, R, 48] in [0,1), the stratified-sampling noise of renderer.py:406
 *   u_importance : NULL = evaluation mode, the importance pass inverts the CDF on the grid linspace(0, 1, 48) (renderer.py:450);
 *               else [B, R, 48] uniform draws in [0,1) (the torch.rand of renderer.py:453), SORTED ascending per ray (the inverse
 *               CDF is monotone, so the fine depths then come out sorted and the merged order equals torch.sort of :361)
 *   dist      : device scalar = mean |ray origin| over the WHOLE batch (renderer.py:311); ray_start/end are
 *               derived from it in-kernel exactly as :313 does in Python doubles
 *   w0,b0,w1,b1 : OSGDecoder parameters net.0.weight [64,32], net.0.bias [64], net.2.weight [33,64], net.2.bias [33]
 *               (un-scaled; lr_multiplier = decoder_lr_mul is applied as FullyConnectedLayer does)
 *   n_coarse / n_importance : must both be 48 (train_avatar_texture.py:341-342); others -> IA_ERR_UNSUPPORTED
 *   flags : bit 0 (IA_RENDER_WHITE_BACK) = rendering_kwargs['white_back']; bit 1 (IA_RENDER_RGB_CHANNEL_MAJOR) = store rgb as
 *          [B, 32, R] -- the feature image [B, 32, nrr, nrr] the super-resolution head reads (triplane_v20.py:313) -- instead of
 *          the renderer's [B, R, 32], which saves the permute + copy between the renderer and the head; bit 2
 *          (IA_RENDER_DIST_PER_FRAME) = `dist` holds B values, frame b uses dist[b]: a batch of frames that the script renders one
 *          call each (eval_seq.py:206-212, where `dist` is every frame's own |ray origin|) keeps those results when batched
 *   rgb  : [B, R, 32] (or [B, 32, R], see flags) composited features scaled to (-1, 1);  depth : [B, R] clamped to the
 *          batch-global sample range (ray_marcher.py:50), with IA_RENDER_DIST_PER_FRAME to every frame's own sample range (what the
 *          one-call-per-frame script computes);  wsum : [B, R] sum of compositing weights
 *   minmax_scratch : 2 * ia_render_rays_grid(B, R) floats of caller scratch; with IA_RENDER_DIST_PER_FRAME 8 * B times that
 *          (one range per wave and frame)
 *   rgb_split : NULL, or a second copy of rgb in the operand format of the convolution that consumes the feature image (the SR head's
 *          first layer, superresolution.py:287 -> block0.conv0): fp16 [B, rgb_split_planes, 4, R, 8] (ia_act_split's format: 2 = hi / lo
 *          planes, 1 = one rounded plane) of rgb[b, c, r] * rgb_split_styles[b, c] ([B, 32] or NULL = unscaled) -- bit for bit what
 *          ia_act_split makes of the channel-major image, without the launch between the renderer and the head (r06)
 *   dbg_* : optional stage outputs for parity tests (NULL in production): fine depths [B,R,48], searchsorted
 *          indices [B,R,48] (int32), merge order [B,R,96] (int32, < 48 = coarse sample, >= 48 = fine sample),
 *          coarse weights [B,R,47], coarse densities [B,R,48]
 */
int ia_render_rays(const float* planes_cl, const float* rays_o, const float* rays_d, const float* jitter,
                   const float* u_importance, const float* dist, const float* w0, const float* b0, const float* w1, const float* b1,
                   float lr_multiplier, float box_warp, int flags,
                   int B, int R, int plane_h, int plane_w, int n_coarse, int n_importance,
                   float* rgb, float* depth, float* wsum, float* minmax_scratch,
                   float* dbg_z_fine, int* dbg_inds, int* dbg_order, float* dbg_w_coarse, float* dbg_sigma_coarse,
                   void* rgb_split, const float* rgb_split_styles, int rgb_split_planes, void* stream);

/* Number of persistent workgroups ia_render_rays launches for (B, R); sizes minmax_scratch. Host-only. */
int ia_render_rays_grid(int B, int R);

/*
 * The same fused launch for the EG3D renderer class, ImportanceRenderer.forward
 * (training_avatar_texture/volumetric_rendering/renderer.py:129-201; run_model :195-202, sample_stratified :224-247,
 * sample_importance / sample_pdf :249-293).  Differences from ia_render_rays:
 *   ray_limits : [B, R, 2] per-ray (near, far) -- rendering_options['ray_start'] == ['ray_end'] == 'auto', the output of
 *               ia_ray_limits_box -- sampled with math_utils.linspace (math_utils.py:101-118) and per-ray noise scale (:236-238);
 *               NULL = the fixed range [ray_start, ray_end] (the options' python doubles) through torch.linspace (:240-242)
 *   u_importance : required -- this class always draws its importance samples (sample_pdf det=False, :280); sorted per ray
 *   flags : IA_RENDER_WHITE_BACK, IA_RENDER_RGB_CHANNEL_MAJOR as above; IA_RENDER_FLIP_Z = the class's flip_z (the z coordinate of
 *          every sample is negated before the plane lookup, :196-197); IA_RENDER_DIST_PER_FRAME is rejected
 * Everything else (decoder, ray marcher, batch-global depth clamp, scratch, debug outputs) as ia_render_rays.
 */
int ia_render_rays_box(const float* planes_cl, const float* rays_o, const float* rays_d, const float* jitter,
                       const float* u_importance, const float* ray_limits, double ray_start, double ray_end,
                       const float* w0, const float* b0, const float* w1, const float* b1,
                       float lr_multiplier, float box_warp, int flags,
                       int B, int R, int plane_h, int plane_w, int n_coarse, int n_importance,
                       float* rgb, float* depth, float* wsum, float* minmax_scratch,
                       float* dbg_z_fine, int* dbg_inds, int* dbg_order, float* dbg_w_coarse, float* dbg_sigma_coarse,
                       void* stream);

/*
 * math_utils.get_ray_limits_box (volumetric_rendering/math_utils.py:46-98): slab test of n_rays rays (rays_o / rays_d [n_rays, 3])
 * against the axis-aligned cube of side box_side_length centred at the origin -> ray_limits [n_rays, 2] = (t_near, t_far),
 * (-1, -2) for a ray that misses.  repair_misses != 0 additionally applies ImportanceRenderer.forward's repair (renderer.py:133-136)
 * on the device, without the reference's `.item()` round trip: if any ray hits, every missing ray gets (min, max) of the hit rays'
 * NEAR limits.  part_scratch: 2 * ia_ray_limits_box_parts(n_rays) floats of caller scratch.
 */
int ia_ray_limits_box(const float* rays_o, const float* rays_d, double box_side_length, int n_rays, int repair_misses,
                      float* ray_limits, float* part_scratch, void* stream);
int ia_ray_limits_box_parts(int n_rays);      /* host-only */

/*
 * Stage entry for parity tests: smoothed inverse-CDF importance resampling (renderer.py:410-469, det=True) and the
 * stable merge order (unify_samples :372-382) from GIVEN coarse depths [nrays,48] and coarse weights [nrays,47].
 * Runs the same device code as ia_render_rays.  z_fine [nrays,48], inds [nrays,48] int32, order [nrays,96] int32.
 */
int ia_importance_stage(const float* z_coarse, const float* w_coarse, float* z_fine, int* inds, int* order,
                        int nrays, void* stream);

/*
 * Mouth-hole mask of the rasterised face alpha, entirely on the device.
 * Replaces the cv2.floodFill round trip of fill_mouth(images, blur_mouth_edge=False)
 * (training_avatar_texture/volumetric_rendering/renderer.py:716-741): v = alpha*255; pixels with
 * v(0,0) <= v <= v(0,0)+254 that are 4-connected to pixel (0,0) are "outside";  mouth = outside ? 0 : (255 - v)/255.
 *   alpha, mouth : [B, H, W] float32 contiguous.  H*(W+4) bytes must fit in LDS (<= 150 KiB; 256x256 is 65 KiB).
 */
int ia_fill_mouth(const float* alpha, float* mouth, int B, int H, int W, void* stream);

/*
 * The soft mouth mask of fill_mouth(images, blur_mouth_edge=True) -- the function's default -- from ia_fill_mouth's result
 * (renderer.py:732-736): copyImg = cv2.blur(cv2.erode(filled, ones(3,3), iterations=3), (5,5));  out = (255 - copyImg) / 255,
 * where filled = 255 on reached pixels (mouth == 0) and alpha*255 elsewhere.  cv2 semantics restated (OpenCV 4.6 is not in
 * this image: parity unpinned by reference tests): erosion border = +inf, box filter on CV_32F sums in double, multiplies by
 * the double 1/25, rounds to float, border BORDER_REFLECT_101.
 *   alpha, mouth, out : [B, H, W] float32 contiguous, H, W >= 3.
 */
int ia_mouth_edge_blur(const float* alpha, const float* mouth, float* out, int B, int H, int W, void* stream);

/*
 * One pyramid level of TriPlaneGenerator.rasterize (training_avatar_texture/triplane_v20.py:328-337) in one pass:
 *     out[:, :C] = AA(grid_sample(texture, uv)) * AA(alpha) + AA(static[:, :, bbox]) * (1 - AA(alpha));  out[:, C] = AA(upper_alpha)
 * where AA = F.interpolate(bilinear, antialias=True) to res x res and grid_sample = bilinear / zeros / align_corners False.
 *   tex_cl      : [B, tex_res, tex_res, C] float32 -- the texture level in CHANNELS-LAST order
 *   uv          : [B, 256, 256, 3] float32 = mesh_condition['uvcoords_image'] (u, v, mask)
 *   upper_alpha : [B, 256, 256] float32 = clamp(mask + upper-mouth mask, 0, 1) (triplane_v20.py:324-326)
 *   sta         : static feature level, NCHW with batch stride `sta_batch_stride` elements (lets plane 0 of a 96-channel
 *                 level be passed without a copy); rows by0:by1, columns bx0:bx1 are resized to res x res
 *   out         : [B, C+1, res, res] float32;   res in {32, 64, 128}
 */
int ia_rasterize_level(const float* tex_cl, const float* uv, const float* upper_alpha, const float* sta, int64_t sta_batch_stride,
                       float* out, int B, int C, int tex_res, int sta_res, int res, int by0, int by1, int bx0, int bx1, void* stream);

/*
 * Tri-plane blend of triplane_v20.py:119-128, written straight in the renderer's layout: the 256^2 face stitch and the
 * mouth-filled alpha are AA-resized to (y1-y0)^2 = 128^2, pasted at rows y0:y1 / columns x0:x1 of plane 0 and blended over
 * the static planes; planes 1-2 are the static planes.
 *   stitch [B,32,256,256], full_alpha [B,256,256], static_planes [B,3*32,256,256] (batch stride in elements given),
 *   planes_cl [B,3,256,256,32] float32.
 */
int ia_blend_planes(const float* stitch, const float* full_alpha, const float* static_planes, int64_t sta_batch_stride,
                    float* planes_cl, int B, int y0, int y1, int x0, int x1, void* stream);

/*
 * Channels-last copy of a feature map, [B,C,H,W] -> [B,H,W,C] (float32, contiguous): the layout ia_rasterize_level gathers its
 * texture level from (`tex_cl`).  Replaces the permute + contiguous copy the gather formulation of TriPlaneGenerator.rasterize
 * (training_avatar_texture/triplane_v20.py:328-337: F.grid_sample over an NCHW texture) needs on this backend; 64 x 64 tiles through
 * LDS, reads and writes in 256-byte runs.
 */
int ia_channels_last(const float* x, float* y, int B, int C, int H, int W, void* stream);

/*
 * Paste of a rasterised condition over features or the skip image inside the face backbone
 * (training_avatar_texture/networks_stylegan2_new.py:537-540):  y = cond[:, :C] * a + x * (1 - a),  a = cond[:, C:C+1].
 *   cond : [B, C+1, H, W] float32 (last channel = alpha);  x, y : [B, C, H, W] float32;  H*W % 4 == 0.
 */
int ia_cond_blend(const float* cond, const float* x, float* y, int B, int C, int H, int W, void* stream);

/*
 * All affine style vectors and demodulation coefficients of one synthesis network in two launches.
 * Replaces, per layer, FullyConnectedLayer.forward of `.affine` (training/networks_stylegan2.py:114-127, called at :318
 * and :353) and the demodulation reduction of modulated_conv2d (:60-64):
 *     styles[b, soff_l + i] = dot(ws[b, widx_l, :], A_l[i, :]) * wgain_l + bias_l[i] * bgain_l
 *     demod[b, doff_l + o]  = rsqrt(sum_i styles[b, soff_l + i]^2 * wsq_l[o, i] + 1e-8)
 *   ws          : [B, num_ws, w_dim] float32
 *   layer_table : device int64 [L][8] = {A ptr ([I,w_dim] f32), bias ptr ([I] f32), wsq ptr ([O,I] f32 or 0), I, O, widx, soff, doff}
 *   layer_gains : device float [L][2] = {weight gain, bias gain}
 *   style_row_layer [style_rows] / demod_row_layer [demod_rows] : device int32 maps from output row to layer
 *   styles : style_rows * B floats, LAYER-major: layer l occupies [B*soff_l, B*(soff_l+I_l)) as a contiguous [B, I_l] matrix;
 *   demod  : demod_rows * B floats, same scheme with doff_l / O_l.  All tables are caller-owned and only read during the call.
 */
int ia_styles_demod(const float* ws, int B, int num_ws, int w_dim, const int64_t* layer_table, const float* layer_gains,
                    const int* style_row_layer, int style_rows, const int* demod_row_layer, int demod_rows,
                    float* styles, float* demod, void* stream);

/*
 * Camera labels -> ray bundles.  Replaces RaySampler_zxc.forward
 * (training_avatar_texture/volumetric_rendering/ray_sampler.py:70-107).
 *   cam : [B, cam_stride] float32, each row = 16 floats cam2world (row-major 4x4) followed by 9 floats intrinsics (row-major 3x3,
 *         normalised by image size) -- i.e. the last 25 entries of the reference's conditioning label `c`
 *   rays_o, rays_d : [B, resolution^2, 3] float32, pixel order row-major (y, x), integer pixel centres (no +0.5)
 */
int ia_ray_sampler(const float* cam, int cam_stride, float* rays_o, float* rays_d, int B, int resolution, int normalize, void* stream);

/*
 * Activations as fp16 hi/lo pairs ("split format"), the input of ia_conv2d_mfma_sx:
 *     xs[b][plane][c/8][y][x][c%8]   fp16, plane 0 = hi = fp16(v), plane 1 = lo = fp16((v - hi) * 2^11),  v = x[b,c,y,x] * styles[b,c]
 * (v saturates at +-65504; |v| < 2^-14 rides entirely in the low plane: the MFMA flushes fp16 denormals).  4 bytes per element,
 * the size of the fp32 tensor; hi + lo * 2^-11 carries 22 mantissa bits.  C % 8 == 0.  styles [B,C] or NULL (= 1).
 * ia_act_split is the stand-alone producer; ia_fir_tail_split and ia_conv2d_mfma_sx emit the format from their epilogues.
 * planes = 2 is the format above; planes = 1 keeps plane 0 only, rounded once (hi = fp16(v), saturating): the operand format of the
 * fp16 blocks (ia_conv2d_mfma_h arithmetic) and the fp16-STORAGE form of their activations, xs[b][c/8][y][x][c%8], 2 bytes / element.
 * shift [B,C] or NULL: v = x * styles + shift -- an eval-mode BatchNorm folded into the staging of the convolution that follows it
 * (inversion encoders, encoder_inversion/models/helpers.py:102-124: BatchNorm2d -> Conv2d 3x3; the zero padding then applies to
 * the normalised tensor, as in the reference).
 */
int ia_act_split(const float* x, const float* styles, const float* shift, void* xs, int planes, int B, int C, int H, int W, void* stream);

/*
 * torch.nn.BatchNorm2d in TRAIN mode followed by ia_act_split, in two launches: xs = split((x - mean_c) * weight_c / sqrt(var_c + eps) +
 * bias_c) with mean / biased variance over (B, H, W) per channel, and the running-statistics update of a train-mode call
 * (running = running + momentum * (batch - running), unbiased variance; num_batches_tracked += 1).  The inversion encoders run the
 * BatchNorms of the e4e trunk and of the decoders' DoubleConv on batch statistics at evaluation time (the reference's eval_seq.py:91-97
 * leaves those modules in train()): encoder_inversion/models/helpers.py:102-124 (BatchNorm2d -> Conv2d 3x3), unet_encoders.py:52-66.
 *   weight, bias [C] or NULL (1 / 0); running_mean, running_var [C] or both NULL; num_batches_tracked: device int64 or NULL;
 *   partials: caller-owned scratch of C * chunks * 2 doubles; chunks = partial sums per channel (1 .. 1024: enough workgroups for
 *   pass 1, e.g. ceil(B*H*W / 8192)); xs, planes as ia_act_split.  C % 8 == 0, B*H*W > 1.
 */
int ia_bn_train_split(const float* x, const float* weight, const float* bias, float* running_mean, float* running_var,
                      long long* num_batches_tracked, double* partials, int chunks, void* xs, int planes, int B, int C, int H, int W,
                      float eps, float momentum, void* stream);

/* 1 for the layer shapes ia_conv2d_mfma_sx covers (3x3, I % 8 == 0, O % 8 == 0, from 8^2; stride-1 layers need O >= 128, W <= 512). */
int ia_conv2d_sx_supported(int I, int O, int H, int W, int ksize, int transposed);

/*
 * ia_conv2d_mfma_s on split-format activations: same layer semantics, tiles, ksplit / scratch (ia_conv2d_plan with form = 3) and
 * results of the same arithmetic (the same products; since r03 the stride-1 kernels pair the odd tap of a chunk with the next chunk's
 * instead of an all-zero tap, so the fp32 sums are added in another order: equal to summation-order level, 2e-6 of a layer's magnitude), but the operand split was done by the producer and both operands reach LDS by DMA
 * (buffer_load ... lds) instead of through registers.  Replaces the same reference chain as ia_conv2d_mfma
 * (training/networks_stylegan2.py:34-91, conv2d_resample.py:114-136) for the 3x3 layers of >= 32^2 (I % 8 == 0, O % 8 == 0).
 *   xs, planes  : input in split format, already multiplied by THIS layer's styles (there is no `styles` argument); planes = 2 with
 *                 wk_split from pack_conv_weight_split (three fp16 products per fp32 product: ia_conv2d_mfma_s arithmetic), planes = 1
 *                 with the one-plane fp16 weights [tap][I/8][O][8] and wk_exp = 0 (fp16 operands: ia_conv2d_mfma_h arithmetic)
 *   y           : [B,O,OH,OW] fp32 or NULL (stride-1 form only: a layer whose result is consumed only in split format)
 *   ys, ys_planes : the result in split format (ys_planes planes), multiplied by styles_next [B,O] (the styles of the consuming layer; NULL = 1), or
 *                 NULL; stride-1 form only (the transposed form's (2H+1)^2 image goes through ia_fir_tail_split)
  * prelu_alpha : NULL, or [O] per-output-channel negative slopes used by IA_ACT_LRELU in place of `alpha` (torch.nn.PReLU of the
 *               inversion encoders' residual units, helpers.py:105).  With demod = BatchNorm scale and bias = BatchNorm shift (per output
 *               channel) the epilogue is conv -> BatchNorm(eval) -> PReLU of those networks; styles / noise stay NULL.
 */
int ia_conv2d_mfma_sx(const void* xs, int planes, const void* wk_split, int wk_exp, const float* demod, const float* noise,
                      const float* noise_strength, const float* bias, const float* residual, float* y, void* ys, int ys_planes,
                      const float* styles_next, float* scratch, size_t scratch_bytes, int B, int I, int O, int H, int W,
                      int transposed, int act, float alpha, const float* prelu_alpha, float gain, float clamp, int ksplit, void* stream);

/*
 * y = bilinear resize of x [planes, H, W] to [planes, OH, OW] with align_corners = True (torch.nn.functional.interpolate's arithmetic)
 * + addend [planes, OH, OW] (NULL: the resize alone).  Replaces `_upsample_add` of the e4e feature pyramid
 * (encoder_inversion/models/e4e.py:48-65): F.interpolate(x, size=y.shape, mode='bilinear', align_corners=True) + y.
 */
int ia_upsample_bilinear_add(const float* x, const float* addend, float* y, int planes, int H, int W, int OH, int OW, void* stream);

/*
 * Depth-wise 3x3 convolution (stride 1, padding 1) of a Mix-FFN on the token grid, channels-last: y[b, n, c] = bias[c] + sum_k
 * w9c[k][c] * x[b, n + offset_k, c] over the H x W grid of tokens n, optionally followed by GELU (erf form).  Replaces DWConv.forward of
 * the transformer-refined decoders of the one-shot encoders (encoder_inversion/models/mmseg/mix_transformer.py:49-58: transpose -> view
 * -> nn.Conv2d(groups = C) -> flatten -> transpose) and the activation after it (:70-77).
 *   x, y [B, H*W, C] float32; w9c [9][C]: the module's weight [C, 1, 3, 3] transposed; bias [C] or NULL; act 0 = none, 1 = GELU.  C % 4 == 0.
 */
int ia_dwconv3x3_tokens(const float* x, const float* w9c, const float* bias, float* y, int B, int H, int W, int C, int act, void* stream);
/* The same convolution (+ GELU) with its result written as the operand of the linear layer behind it (Mix-FFN: dwconv -> GELU -> fc2):
 * xs fp16 [2][C/8][B*H*W][8], ia_tokens_split's format (one launch for both).  C % 16 == 0. */
int ia_dwconv3x3_tokens_split(const float* x, const float* w9c, const float* bias, void* xs, int B, int H, int W, int C, int act, void* stream);

/*
 * 3x3 convolution with stride 2 and padding 1 on an image of 2^2, 4^2 or 8^2 pixels (outputs 1^2, 2^2, 4^2): the last layers of a
 * GradualStyleBlock of the e4e encoder (encoder_inversion/models/e4e.py:22-45: Conv2d(512, 512, 3, 2, 1) + LeakyReLU down to 1x1), cuDNN
 * through torch.nn.Conv2d in the reference.  A matrix-vector product that streams the weight once; fp32 FMAs, deterministic.
 *   x [B,I,H,W], w [O,I,3,3] (the module's own layout), bias [O] or NULL, y [B,O,H/2,W/2]; act IA_ACT_LINEAR or IA_ACT_LRELU (alpha).
 * H == W in {2, 4, 8} and I * (H*W + 1) * 4 bytes <= 160 KB (ia_conv3x3_s2_tiny_supported = 1), else IA_ERR_UNSUPPORTED.
 */
int ia_conv3x3_s2_tiny_supported(int I, int O, int H, int W);
int ia_conv3x3_s2_tiny(const float* x, const float* w, const float* bias, float* y, int B, int I, int O, int H, int W, int act, float alpha,
                       void* stream);

/*
 * 3x3 convolution with STRIDE 2 and padding 1 on split-format activations: y[b,o,r,c] = sum w[o,i,ky,kx] * x[b,i,2r+ky-1,2c+kx-1], on
 * the stride-1 tile families of ia_conv2d_mfma_sx with the point grid laid over every second pixel of the input window (same
 * products as torch.nn.functional.conv2d(stride 2, padding 1): a quarter of the stride-1 layer's).  Replaces the stride-2 layers of
 * the inversion encoders (encoder_inversion/models/helpers.py:102-124, second convolution of the first residual unit of a stage;
 * e4e.py:22-45, GradualStyleBlock), which the reference hands to cuDNN through torch.nn.Conv2d.
 *   xs [B][planes][I/8][H][W][8], wk_split, wk_exp, demod, bias, residual, y, ys, ys_planes, styles_next, act, alpha, prelu_alpha, gain,
 *   clamp: as ia_conv2d_mfma_sx; y / ys / residual have the output size ((H-1)/2+1) x ((W-1)/2+1).
 *   ksplit, scratch: from ia_conv2d_down_plan (same stream-K slabs as ia_conv2d_plan).
 * I % 8 == 0, O % 8 == 0, O >= 64, outputs from 8^2 up, input windows that fit the LDS stages (W <= ~1000); else IA_ERR_UNSUPPORTED
 * from ia_conv2d_down_plan (callers evaluate the layer at stride 1 and sub-sample, or use the library).
 */
int ia_conv2d_down_plan(int B, int I, int O, int H, int W, int* h_ksplit, size_t* h_scratch_bytes);
int ia_conv2d_down_sx(const void* xs, int planes, const void* wk_split, int wk_exp, const float* demod, const float* bias, const float* residual,
                      float* y, void* ys, int ys_planes, const float* styles_next, float* scratch, size_t scratch_bytes, int B, int I, int O,
                      int H, int W, int act, float alpha, const float* prelu_alpha, float gain, float clamp, int ksplit, void* stream);

/*
 * ia_upfirdn2d_bias_act for the 4x4 filter at up = 1 (the FIR + noise + bias + activation tail of an up-sampling SynthesisLayer,
 * training/networks_stylegan2.py:318-329 after conv2d_resample.py:114-131) with the result stored in SPLIT format for the next
 * convolution: ys = split(tail(fir(x)) * styles_next[b, c]) (see ia_act_split; same FIR sums in the same order as
 * ia_upfirdn2d_bias_act).  y (fp32 [n,c,out_h,out_w]) may be NULL when only the split result is consumed.  c % 8 == 0.
 */
int ia_fir_tail_split(const float* x, const float* f, const float* noise, const float* noise_strength, const float* bias,
                      const float* styles_next, float* y, void* ys, int ys_planes, int n, int c, int in_h, int in_w, int out_h, int out_w,
                      int padx0, int pady0, int flip, float fir_gain, int act, float alpha, float act_gain, float clamp, void* stream);

/*
 * ia_conv2d_mfma_sx (stride 1) with the ToRGB layer that consumes its result evaluated in the epilogue: the last block of the SR head
 * (SynthesisBlock.forward, training/networks_stylegan2.py:448-457: conv1, then torgb(x) added to the up-sampled image) reads its
 * 128 x 512^2 activation only through ToRGB, so neither that tensor nor its re-read has to exist.
 *   rgb_out[b,c] = clamp( sum_o (rgb_wk[o,c] * rgb_styles[b,o]) * v[b,o] + rgb_bias[c], rgb_clamp ) + rgb_residual[b,c]
 * with v the layer's own result (after noise / bias / activation / clamp).  rgb_wk [O, rgb_channels] is ia_conv2d_mfma's ksize-1
 * packing of the ToRGB weight (weight_gain folded in), rgb_channels <= 3.  y and ys may both be NULL.  The layer must run in whole
 * rounds of tiles that hold every output channel (O <= 128 at the sizes the planner gives the 128 x 256 tile), else
 * IA_ERR_UNSUPPORTED (callers use ia_conv2d_mfma_sx + ia_conv1x1).  Other arguments as ia_conv2d_mfma_sx.
 */
int ia_conv2d_mfma_sx_rgb(const void* xs, int planes, const void* wk_split, int wk_exp, const float* demod, const float* noise,
                          const float* noise_strength, const float* bias, float* y, void* ys, int ys_planes, const float* styles_next,
                          const float* rgb_wk, const float* rgb_styles, const float* rgb_bias, const float* rgb_residual, float* rgb_out,
                          int rgb_channels, float rgb_clamp, int B, int I, int O, int H, int W, int act, float alpha, float gain, float clamp,
                          void* stream);

/*
 * The stride-2 transposed 3x3 convolution of an up-sampling SynthesisLayer (torch_utils/ops/conv2d_resample.py:114-131 ->
 * conv_transpose2d; training/networks_stylegan2.py:296-305) on split-format activations, evaluated per output ROW phase on the
 * stride-1 tile (128 channels x 256 / 128 points, two accumulator sets = the two column phases): the same products as
 * ia_conv2d_mfma_sx(transposed = 1) -- 9 per input pixel and channel pair, the minimum -- and the same result image
 *   y [B, O, 2H+1, 2W+1] float32 = demod[b][o] * conv_transpose2d(x * styles, w, stride 2)   (summation order differs),
 * which ia_fir_tail_split / ia_upfirdn2d_bias_act turn into the layer's output.  (csrc/conv_up.hip)
 *   xs       : activations as ia_act_split(planes = 2) stores them, already multiplied by this layer's styles
 *   wk_split : the weights of pack_conv_weight_split ([2][9][I/8][O][8] fp16 of w * 2^wk_exp), the ones ia_conv2d_mfma_sx takes
 *   scratch  : ia_upconv2d_rows_plan's byte count (partial sums of the thin edge grids: bottom row / last column of the image)
 * Needs I % 16 == 0, O % 128 == 0, 16 <= H, W <= 1024: ia_upconv2d_rows_plan returns IA_ERR_UNSUPPORTED otherwise.
 */
int ia_upconv2d_rows_plan(int B, int I, int O, int H, int W, size_t* h_scratch_bytes);
int ia_upconv2d_rows_sx(const void* xs, const void* wk_split, int wk_exp, const float* demod, float* y, float* scratch,
                        size_t scratch_bytes, int B, int I, int O, int H, int W, void* stream);

/*
 * An up-sampling SynthesisLayer with FEW input channels as ONE stride-1 launch: the transposed convolution (stride 2) and the 4x4
 * resample FIR that follows it are linear, so their composition is, per output phase (py, px), a 3x3 convolution of the INPUT image
 * with the weights  W'[py,px][dy,dx] = gain * sum_{i,ky: py+i-1-ky = 2dy} sum_{j,kx: px+j-1-kx = 2dx} F[i,j] * w[ky,kx]  (F the
 * flipped filter).  The layer then is a stride-1 convolution with 4 x O output channels -- row ((py * O/32 + o/32) * 2 + px) * 32 +
 * o % 32, so that a wave holds both horizontal phases of its channels -- whose epilogue stores point (r, c) of phase (py, px) at
 * pixel (2r + py, 2c + px): four times the products, but no (2H+1) x (2W+1) fp32 image, no FIR launch
 * and no stream-K tail -- a win when the K loop is short (the 32 -> 256 @128^2 layer of the SR head: 2.4 GFLOP that moved 200 MB).
 * Replaces modulated_conv2d(up=2) -> conv2d_resample (conv_transpose2d, stride 2) -> upfirdn2d(pad [1,1,1,1], gain 4) -> noise ->
 * bias_act of SynthesisLayer.forward (training/networks_stylegan2.py:296-305, torch_utils/ops/conv2d_resample.py:114-131) for such
 * layers; results agree with the two-launch route to fp32 rounding (the same sums in another association).
 *   xs        : split input [B, planes, I/8, H, W, 8], already multiplied by the layer's styles (ia_act_split)
 *   wk_split  : pack_conv_weight_split / _h layout of the COMPOSED weight [4*O, I, 3, 3] (row order above), wk_exp its exponent
 *   demod [B,O], noise [2H*2W], noise_strength, bias [O], styles_next [B,O]: per REAL channel / OUTPUT pixel, any may be NULL
 *   y [B,O,2H,2W] fp32 and / or ys [B, ys_planes, O/8, 2H, 2W, 8] split (at least one)
 * O % 32 == 0 and the 4*O-channel layer must run in whole 128-channel tiles (ia_conv2d_plan(B, I, 4*O, H, W, 3, 0, 3) gives 0
 * workers, at least one such tile per CU); else IA_ERR_UNSUPPORTED (callers use ia_conv2d_mfma_sx(transposed) + ia_fir_tail_split).
 */
int ia_upconv2d_fir_sx(const void* xs, int planes, const void* wk_split, int wk_exp, const float* demod, const float* noise,
                       const float* noise_strength, const float* bias, float* y, void* ys, int ys_planes, const float* styles_next,
                       int B, int I, int O, int H, int W, int act, float alpha, float gain, float clamp, void* stream);

/*
 * ToRGB layer as a streaming kernel: y = clamp((w * styles) (*) x + bias) + residual for a 1x1 modulated convolution without
 * demodulation.  Replaces ToRGBLayer.forward (training/networks_stylegan2.py:353-362: modulated_conv2d(demodulate=False) :34-91 +
 * bias_act(clamp) ) and the skip-image add of SynthesisBlock.forward (:457).  x [B,I,H,W], wk [I,O] (ia_conv2d_mfma's packing
 * for ksize 1, weight_gain folded in), styles [B,I] or NULL, bias [O] or NULL, residual [B,O,H,W] or NULL (added after the
 * clamp), y [B,O,H,W]; clamp < 0 = none.  O <= 96, I % 32 == 0, H*W % 4 == 0, else IA_ERR_UNSUPPORTED (callers use
 * ia_conv2d_mfma).  One launch, no scratch.
 */
int ia_conv1x1(const float* x, const float* wk, const float* styles, const float* bias, const float* residual, float* y,
               int B, int I, int O, int H, int W, float clamp, void* stream);

/*
 * ToRGB layer AND the skip-image branch of its block in one launch:
 *   y = clamp((w * styles) (*) x + bias) + (residual | upsample2d(skip))
 * Replaces ToRGBLayer.forward (training/networks_stylegan2.py:353-362) + `img = upfirdn2d.upsample2d(img, resample_filter)` +
 * `img = img.add_(y)` of SynthesisBlock.forward (:454-457; upsample2d = torch_utils/ops/upfirdn2d.py:315-350: zero insertion x2,
 * padding [2,1,2,1], the 4x4 filter, gain 4).  Same arguments as ia_conv1x1, plus skip [B,O,H/2,W/2] (or NULL) with its resample
 * filter skip_filter [4,4] (upfirdn2d.setup_filter([1,3,3,1])); residual and skip are exclusive.  Every load of a wave (128 input
 * channels x 32 pixels, its weights and styles) is in flight before the first MFMA, wider layers split their channels over the waves
 * of a workgroup; the up-sampling evaluates the 2 x 2 non-zero taps per output pixel in ia_upfirdn2d's order (bit-identical image).
 * C_in in {128, 256, 512, 1024}, C_out <= 96, H and W even when skip is given; else IA_ERR_UNSUPPORTED (callers use ia_conv1x1 +
 * ia_upfirdn2d).  ia_torgb_supported answers that question without a launch (1 = covered).  One launch, no scratch.
 */
int ia_torgb_supported(int I, int O, int H, int W, int with_skip);
int ia_torgb(const float* x, const float* wk, const float* styles, const float* bias, const float* residual, const float* skip,
             const float* skip_filter, float* y, int B, int I, int O, int H, int W, float clamp, void* stream);

/*
 * ia_cond_blend with the result in SPLIT format (ia_act_split) for the one layer that consumes it, multiplied by that layer's
 * styles [B,C] (NULL = 1): ys = split((cond[:, :C] * a + x * (1 - a)) * styles_next), a = cond[:, C]
 * (training_avatar_texture/networks_stylegan2_new.py:539-540 feeding the next block's conv0).  C % 8 == 0.
 */
int ia_cond_blend_split(const float* cond, const float* x, const float* styles_next, void* ys, int B, int C, int H, int W, void* stream);

/*
 * Fused filtered leaky ReLU: bias -> up-sample (zero insert, pad/crop, up-FIR x up^2) -> lrelu(slope) x gain -> clamp -> down-FIR ->
 * decimate, the up-sampled intermediate kept in LDS.
 * Replaces filtered_lrelu_plugin.filtered_lrelu(x, fu, fd, b, si, up, down, px0, px1, py0, py1, sx, sy, gain, slope, clamp,
 * flip_filter, writeSigns) -> (y, so, rc) (torch_utils/ops/filtered_lrelu.cpp:20-22,217; kernels filtered_lrelu.cu:144-1103) for
 * the forward pass without sign tensors (si = None, writeSigns = false: what inference calls); semantics = the reference's own
 * definition of the op, _filtered_lrelu_ref (filtered_lrelu.py:123-155).
 *   x, y   : [n, c, in_h, in_w] / [n, c, out_h, out_w] contiguous NCHW of `dtype` (IA_F32 or IA_F16; f64 -> IA_ERR_UNSUPPORTED,
 *            the caller composes bias_act + upfirdn2d, as the reference does when its plugin returns rc = -1, :225-231)
 *   fu, fd : 2-D float32 filters [fu_h, fu_w] / [fd_h, fd_w], contiguous (separable filters are passed as their outer product),
 *            or NULL with size 1x1 for identity
 *   b      : [c] of `dtype` or NULL
 *   out size must equal (in*up + pad0 + pad1 - (fu - 1) - (fd - 1) + (down - 1)) / down per axis (filtered_lrelu.py:142-143)
 *   clamp < 0 : disabled
 */
int ia_filtered_lrelu(const void* x, const float* fu, const float* fd, const void* b, void* y, int dtype,
                      int n, int c, int in_h, int in_w, int out_h, int out_w, int fu_h, int fu_w, int fd_h, int fd_w,
                      int up, int down, int px0, int px1, int py0, int py1, float gain, float slope, float clamp,
                      int flip_filter, void* stream);

/*
 * ConvGRU cell of the inversion encoder's recurrent up-path (encoder_inversion/models/unet_encoders.py:8-49): the element-wise
 * work around its two convolutions, which stay library convolutions on the caller's side.
 *   ia_convgru_gates : xrh = cat[x, sigmoid(gates_pre[:, :C]) * h]                  ([B,2C,H,W]; the input of conv_hh, :26-27)
 *   ia_convgru_update: z = sigmoid(gates_pre[:, C:]); c = tanh(cand_pre) (PReLU with prelu_weight[C] when non-NULL, :19-20);
 *                      h_out = (1 - z) * h + z * c (:28); with x_next / xh_next also xh_next = cat[x_next, h_out], the input of
 *                      conv_ih of the next time step (:25)
 * gates_pre = conv_ih(cat[x, h]) + bias [B,2C,H,W]; cand_pre = conv_hh(xrh) + bias [B,C,H,W]; all fp32 contiguous, H*W % 4 == 0.
 */
int ia_convgru_gates(const float* gates_pre, const float* x, const float* h, float* xrh, int B, int C, int H, int W, void* stream);
int ia_convgru_update(const float* gates_pre, const float* cand_pre, const float* h, const float* prelu_weight, float* h_out,
                      const float* x_next, float* xh_next, int B, int C, int H, int W, void* stream);

/*
 * ia_convgru_gates / ia_convgru_update with the convolution inputs they produce written in SPLIT format (ia_act_split, two planes):
 * xrh_split = split(cat[x, sigmoid(r_pre) * h]) and xh_next_split = split(cat[x_next, h']) are read by the cell's two
 * ia_conv2d_mfma_sx launches only, so neither their fp32 copies nor the two ia_act_split launches of a step exist.  h_out stays fp32
 * (the cell's output and the next step's state).  Same arithmetic as the fp32 forms.  C % 8 == 0, H * W % 4 == 0.
 */
int ia_convgru_gates_split(const float* gates_pre, const float* x, const float* h, void* xrh_split, int B, int C, int H, int W, void* stream);
int ia_convgru_update_split(const float* gates_pre, const float* cand_pre, const float* h, const float* prelu_weight, float* h_out,
                            const float* x_next, void* xh_next_split, int B, int C, int H, int W, void* stream);

/*
 * Squeeze-and-excitation gate + residual add of an IR-SE50 unit:  out = v * sigmoid(W2 relu(W1 mean_hw(v))) + shortcut.
 * Replaces SEModule.forward and the add of bottleneck_IR_SE.forward (encoder_inversion/models/helpers.py:84-100, :121-124).
 *   v, shortcut : float32 [B, C, H, W] VIEWS given by 4 strides each (floats; batch, channel, row, column) -- the stride-2 output of
 *                 the unit's second convolution and MaxPool2d(1, s) shortcuts are strided views of larger tensors
 *   w1 [R, C], w2 [C, R] : the two bias-free 1x1 convolutions (R = C / reduction <= 64);  pooled_scratch : B * C floats
 *   out : [B, C, H, W] contiguous
 */
int ia_se_gate(const float* v, const int64_t* v_strides, const float* shortcut, const int64_t* shortcut_strides, const float* w1,
               const float* w2, float* pooled_scratch, float* out, int B, int C, int R, int H, int W, void* stream);

/*
 * ia_se_gate with the result ALSO in split format for the convolution that consumes it: ys = split(out * next_scale[b][c] +
 * next_shift[b][c]) (ia_act_split's two-plane format; the eval-mode BatchNorm in front of the next residual unit's first convolution,
 * helpers.py:102-124), or ys = next_scale = next_shift = NULL for the fp32 result alone.  C % 8 == 0.  Same arithmetic as ia_se_gate.
 */
int ia_se_gate_split(const float* v, const int64_t* v_strides, const float* shortcut, const int64_t* shortcut_strides, const float* w1,
                     const float* w2, float* pooled_scratch, float* out, const float* next_scale, const float* next_shift, void* ys,
                     int B, int C, int R, int H, int W, void* stream);

/*
 * Multi-head self-attention in one launch: out = softmax(Q K^T * scale) V per (batch, head), without the [N, M] score matrix.
 * Replaces Attention.forward of the transformer-refined UNet decoders between the projections
 * (encoder_inversion/models/mmseg/mix_transformer.py:83-116 with sr_ratio 1: two batched matmuls, the softmax over [N, M] and
 * the head permutes).  fp32 operands and accumulation (v_mfma_f32_32x32x2_f32), online softmax, fixed summation order.
 *   q   : [B, N, heads * head_dim] float32 (the output of the q projection; head h = columns [h * head_dim, (h + 1) * head_dim))
 *   k, v: [B, M, heads * head_dim] views with their own batch / row strides (the reference's fused kv projection is [B, M, 2 C]:
 *         k = columns [0, C), v = columns [C, 2 C), row stride 2 C)
 *   out : [B, N, heads * head_dim] float32 = (attn @ v).transpose(1, 2).reshape(B, N, C) of the reference
 *   strides in floats, multiples of 4; head_dim must be 256 (the 1024-dim / 4-head blocks) and N * M <= 131072 (the 8^2 / 16^2 token
 *   grids of the first two decoder stages, where the ATen sequence is launch-bound: 21 vs 66 us, 65 vs 139 us; beyond that the
 *   library GEMMs are faster and the caller keeps them): others -> IA_ERR_UNSUPPORTED (ia_attention_supported tells)
 */
int ia_attention_supported(int head_dim, int N, int M);
int ia_attention(const float* q, const float* k, const float* v, float* out, int B, int heads, int N, int M, int head_dim,
                 int64_t q_batch_stride, int64_t q_row_stride, int64_t k_batch_stride, int64_t k_row_stride,
                 int64_t v_batch_stride, int64_t v_row_stride, int64_t out_batch_stride, int64_t out_row_stride,
                 float scale, void* stream);

/*
 * Token-major linear layers of the transformer blocks (encoder_inversion/models/mmseg/mix_transformer.py:18-116: Mlp.fc1 / fc2,
 * Attention.q / kv / proj -- nn.Linear on [B, tokens, C]; :158-190 the blocks that chain them) as fp32-equivalent GEMMs on the fp16
 * pipe: fp32 products from fp16 hi / lo pairs like the 3x3 convolutions (three v_mfma_f32_32x32x16_f16 per k-step, fp32
 * accumulation, lo x lo ~ 2^-22 dropped).  Replaces F.linear (rocBLAS fp32 GEMM) + bias [+ GELU] [+ the residual add of Block.forward].
 *   ia_tokens_split: x [M][K] float32 (M = B * tokens; rows `ld` floats apart, ld = K for a dense matrix) -> xs fp16 [2][K/8][M][8]: hi = fp16(v),
 *       lo = fp16((v - hi) * 2^11) -- the split format of the convolutions with the token in the pixel's place.  One call serves
 *       every linear layer that reads x (q and kv; fc1).  |v| >= 65504 saturates and raises the range-watch word
 *       (ia_split_saturation_poll).  K % 16 == 0, ld % 4 == 0, x 16-byte aligned, else IA_ERR_UNSUPPORTED.
 *   ia_linear_sx: y [M][N] float32 = act(xs . w^T * 2^-wk_exp + bias) + residual
 *       w_split : fp16 [2][1][K/8][N][8], the nn.Linear weight [N][K] as a 1x1 kernel in the convolution weight format
 *                 (hi = fp16(w * 2^wk_exp), lo = fp16(w * 2^wk_exp - hi); the host-side packing of ia_conv2d_mfma_s)
 *       bias    : [N] or NULL;  residual: [M][N] or NULL, added after the activation (x + drop_path(f(x)) of Block.forward)
 *       act     : 0 none, 1 GELU (erf form, nn.GELU default)
 *       deterministic (fixed summation order for a given shape); no workspace.
 */
int ia_tokens_split(const float* x, int64_t ld, void* xs, int M, int K, void* stream);
/*
 * nn.LayerNorm over the last dimension + ia_tokens_split of its result in one launch (Block.forward, mix_transformer.py:140-142: norm1
 * feeds only q / kv, norm2 only fc1): xs = split((x - mean) * rsqrt(var + eps) * gamma + beta), biased variance, two passes in fp32.
 * x [M][K] float32 dense; K 512, 1024 or 2048; gamma, beta [K].
 */
int ia_layernorm_split(const float* x, const float* gamma, const float* beta, float eps, void* xs, int M, int K, void* stream);
/*
 * The overlapping patch embeddings of the same encoders (mix_transformer.py:155-190 OverlapPatchEmbed: a 7x7 stride-2 / stride-4
 * convolution whose output is flattened to tokens) as im2col-free GEMMs: ia_im2col_split writes the patches of an NCHW float32 image
 * straight into the split format of a token matrix -- row m = (b, oy, ox), column k = (c, ky, kx) in the order of
 * conv.weight.reshape(N, C * ksize * ksize), zero columns up to Kp = K rounded up to 16 -- and ia_linear_sx (M = B * OH * OW, K = Kp,
 * weight rows zero-padded the same way) produces the tokens [B, OH * OW, N] + bias directly: no NCHW result, no flatten / transpose.
 *   xs: fp16 [2][Kp/8][M][8];  ksize 7 or 3, zero padding;  OH = (H + 2 pad - ksize) / stride + 1.
 */
int ia_im2col_split(const float* x, void* xs, int B, int C, int H, int W, int ksize, int stride, int pad, void* stream);
/*
 * The same product with K cut into `ksplit` slices over the launch (few rows, very long K: the deepest patch embedding is 64 tokens x
 * 50 176 x 1 024): slice products into scratch [ksplit][M][N], then one pass sums them in slice order and adds the bias.
 * ia_linear_splitk_plan gives the split the library would choose (1 = none) and the scratch it needs; K % (16 * ksplit) == 0.
 */
int ia_linear_splitk_plan(int M, int K, int N, int* ksplit, size_t* scratch_bytes);
int ia_linear_sx_splitk(const void* xs, const void* w_split, int wk_exp, const float* bias, float* y, int M, int K, int N, int ksplit,
                        float* scratch, size_t scratch_bytes, void* stream);
int ia_linear_sx(const void* xs, const void* w_split, int wk_exp, const float* bias, const float* residual, float* y, int M, int K, int N,
                 int act, void* stream);

/*
 * Attention of the same blocks on token grids too large for ia_attention (32^2 / 64^2 tokens: Attention.forward,
 * mix_transformer.py:83-116 -- q @ k^T * scale, softmax, attn @ v as two batched rocBLAS GEMMs, an elementwise scale, a softmax over
 * [heads, N, M] and the head permutes) as three launches on the fp16-pair GEMM, the score matrix in HBM once:
 *   S[z] = scale * Q[z] K[z]^T   (ia_matmul_sx on ia_tokens_split(q), ia_tokens_split(k): head z = columns [z * hd, (z + 1) * hd))
 *   P = softmax(S) written as the next product's operand  (ia_softmax_split)
 *   O[:, z * hd : (z + 1) * hd] = P[z] V[z]   (ia_matmul_sx on P and ia_tokens_split_t(v)), straight into the [N, C] token layout.
 * ia_tokens_split_t: v [M keys][ld] float32, C columns -> vt fp16 [2][M/8][C][8], octets ALONG THE KEYS (M % 16 == 0); perm = 1: the keys of
 *   a 16-key step in the row order of the 32 x 32 MFMA accumulator (octet 2t + h = keys 16t + {0..3, 8..11} + 4h), the order ia_attention_sx reads.
 * ia_softmax_split: s [Z][N][M] float32 -> ps fp16 [2][Z][M/8][N][8]; M % 16 == 0, M <= 4096.
 * ia_matmul_sx: y[z][m][n] = scale * sum_k a[z][m][k] b[z][n][k] for z < batch, both operands token-side splits (low parts at 2^11):
 *   a_rows / b_rows      : rows of the split tensors the operands live in (their octet stride is rows * 16 bytes)
 *   a_plane / b_plane    : bytes from the hi plane to the lo plane;  a_batch / b_batch: bytes per z inside a plane
 *                          (head z of [2][C/8][rows][8]: z * (hd / 8) * rows * 16;  P: (M / 8) * N * 16;  vt: z * hd * 16)
 *   y_batch_stride / y_row_stride in floats.  K % 16 == 0.  Deterministic; no workspace.
 */
int ia_tokens_split_t(const float* v, int64_t ld, void* vt, int M, int C, int perm, void* stream);
int ia_softmax_split(const float* s, void* ps, int Z, int N, int M, void* stream);
/*
 * The same attention in ONE launch, the score matrix never in memory (head_dim 256): a first pass over the keys finds every query's maximum,
 * a second one forms exp(s - max), its sum and the output over 32-key blocks; both products on fp16 pairs.  A wave computes S^T = K Q^T so that a query's keys sit in the registers of two lanes (row maximum / sum are register
 * reductions + one exchange) and its probabilities are, as they stand, a B fragment of O^T = V^T P^T -- no transposition anywhere.
 *   q_split = ia_tokens_split(q [N][C]), k_split = ia_tokens_split(k [M][C]), v_split_t = ia_tokens_split_t(v [M][C], perm = 1);
 *   out [N][C] float32 = (attn @ v).transpose(1, 2).reshape(N, C) of the reference; M % 16 == 0; deterministic.
 */
int ia_attention_sx_supported(int head_dim, int N, int M);
int ia_attention_sx(const void* q_split, const void* k_split, const void* v_split_t, float* out, int heads, int N, int M, int head_dim,
                    float scale, void* stream);
int ia_matmul_sx(const void* a_split, const void* b_split, float* y, int batch, int M, int N, int K, int a_rows, int64_t a_plane_bytes,
                 int64_t a_batch_bytes, int b_rows, int64_t b_plane_bytes, int64_t b_batch_bytes, int64_t y_batch_stride, int64_t y_row_stride,
                 float scale, void* stream);

/*
 * Driver-side UV rasteriser: projected FaceVerse mesh -> uvcoords_image, the mesh condition of TriPlaneGenerator.synthesis.
 * Replaces Faceverse_manager.make_driven_rendering from the rasteriser call on (data_preprocess/FaceVerse/renderer.py:66-82:
 * pytorch3d MeshRasterizer, ortho camera K = [-1,-1,0,0], T = [0,0,10], faces_per_pixel 1 -> render_after_rasterize
 * (volumetric_rendering/renderer.py:556-571) -> x (vis * mask) -> crop -> HWC (u, v, mask >= 0.5)).
 *   verts          : [B, V, 3] float32, the vertices handed to Meshes() (after batch_orth_proj and the z flip, :62-64)
 *   tris           : [F, 3] int32
 *   face_attrs     : [F, 3, 3] float32 = face_vertices(cat[uv * 2 - 1, mask], tris): (u, v, mask) of every corner (:33-34)
 *   zbuf_scratch   : B * crop_w * crop_h * 8 bytes (caller-owned)
 *   uvcoords_image : [B, crop_h, crop_w, 3] float32
 *   binarize_mask  : 1 = channel 2 is (mask >= 0.5) as the script returns it at the native size (:82); 0 = the continuous
 *                    mask * vis * mask product, for a caller that interpolates to another `res` first and thresholds afterwards (:78-82)
 *   raster_size 512, crop (128, 114, 256, 256), blur_radius 1e-6 in the reference (:13-14, :43)
 */
int ia_uv_rasterize(const float* verts, const int* tris, const float* face_attrs, void* zbuf_scratch, float* uvcoords_image,
                    int B, int V, int F, int raster_size, int crop_left, int crop_top, int crop_w, int crop_h, float blur_radius,
                    int binarize_mask, void* stream);

/*
 * Output side: float image batch -> uint8 picture grid, one pass.
 * Replaces layout_grid(img, grid_w, grid_h, float_to_uint8=True, chw_to_hwc) of the reference's scripts
 * (reenact_avatar_next3d.py:117-131): (img * 127.5 + 128).clamp(0, 255).to(uint8), frames tiled row-major into a
 * grid_h x grid_w mosaic, channels moved last when chw_to_hwc != 0.
 *   img : [B, C, H, W] float32 contiguous, B == grid_w * grid_h, C in {1, 3, 4}, W % 4 == 0
 *   out : uint8, [grid_h*H, grid_w*W, C] (chw_to_hwc) or [C, grid_h*H, grid_w*W]; grid_w = 1 gives the batch of
 *         HWC frames [B, H, W, C] that is all-gathered between GPUs / handed to the encoder
 * Bit-exact with the reference's torch expression (one multiply, one add, clamp, truncation).
 */
int ia_layout_grid_u8(const float* img, uint8_t* out, int B, int C, int H, int W, int grid_w, int grid_h, int chw_to_hwc, void* stream);

/*
 * Range check of the split format (debug / test aid).  The hi plane saturates at +-65504 (ia_act_split and every epilogue that writes
 * the format clamp x * style there: ia_common.h split_f16); a value that hit the clamp is no longer the fp32 number the consumer
 * should multiply.  Counts the elements of plane 0 whose magnitude is the fp16 maximum.
 *   xs : split tensor [B][planes][C/8][H*W][8] fp16;  count : device uint32 (overwritten)
 * The generator's activations stay far below the bound (|x * style| < 10^3 on the synthetic and on trained checkpoints); the tests
 * assert a zero count over a full-width frame, and hipops.CHECK_SPLIT_RANGE makes every producer check itself.
 */
int ia_split_saturation_count(const void* xs, int planes, int B, int C, int H, int W, unsigned int* count, void* stream);

/*
 * Always-on range watch of the hi / lo split.  Every producer of split-format activations (ia_act_split, the FIR tails, the
 * convolution epilogues, ia_cond_blend_split) sets a device flag when a value it splits lies outside +-65504 or is not finite
 * (it is clamped, as before).  This call reads the flags of the current device into *h_flagged (0: every split since the last
 * reset was in range), optionally clearing them; it synchronises `stream` (a host read: call it once per frame / clip, outside
 * captured graphs).  h_flagged == NULL with reset != 0 clears the flags in stream order WITHOUT a host synchronisation (the top of
 * a clip).  Real checkpoints cannot clamp silently.
 */
int ia_split_saturation_poll(unsigned int* h_flagged, int reset, void* stream);

/*
 * Input side of a captured frame: n <= 8 device-to-device segment copies in ONE launch (src[k] -> dst[k], nbytes[k] bytes).
 * The frame loop of the reference hands synthesis() fresh tensors every frame (reenact_avatar_next3d.py:196-214: ws, camera,
 * uvcoords_image); a replayed hipGraph reads fixed buffers, so the per-frame inputs are copied into them first -- with this entry
 * point as one kernel instead of one copy launch per tensor (4 x ~12 us on the frame's critical path).
 *   src, dst, nbytes : HOST arrays of n device pointers / byte counts (any alignment; 16-byte aligned segments move as uint4)
 */
int ia_stage_inputs(const void* const* src, void* const* dst, const int64_t* nbytes, int n, void* stream);

#ifdef __cplusplus
}
#endif

#endif /* IA_HIP_H_ */

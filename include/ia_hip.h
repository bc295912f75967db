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
 *   - nothing throws across the ABI; the library is re-entrant (no mutable globals).
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

#define IA_HIP_ABI_VERSION 1

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

#ifdef __cplusplus
}
#endif

#endif /* IA_HIP_H_ */

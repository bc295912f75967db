// ia_convgru_gates / ia_convgru_update: the element-wise halves of a ConvGRU cell, fused around its two convolutions.
//
// Reference cell (encoder_inversion/models/unet_encoders.py:8-49, used by recurrent_Up :85-98 in both UNets of the inversion
// encoder):      r, z = sigmoid(conv_ih(cat[x, h])).split(C)
//                c    = tanh(conv_hh(cat[x, r * h]))            (PReLU instead of tanh when out_act_prelu)
//                h'   = (1 - z) * h + z * c
// In the reference every line is a separate ATen kernel (cat, sigmoid, split, mul, cat, tanh, rsub, mul, mul, add: ten passes
// over [B, C..2C, H, W] tensors per time step).  Here the two dense convolutions stay library GEMM-convolutions and everything
// between them is two launches:
//   gates : conv_ih output -> cat[x, sigmoid(r_pre) * h]                       (the input of conv_hh, written once)
//   update: conv_hh output -> h' ; optionally also cat[x_next, h'] = the input of conv_ih of the NEXT time step
// HBM-bound: gates reads 3C and writes 2C floats per pixel, update reads 3C (+C) and writes C (+2C).
#include "ia_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// one thread = 4 consecutive pixels of one (b, c)
__global__ __launch_bounds__(256) void convgru_gates_kernel(const float4* __restrict__ gates_pre, const float4* __restrict__ x, const float4* __restrict__ h,
                                                           float4* __restrict__ xrh, int C, int64_t hw4, int64_t total4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int64_t p = i % hw4, bc = i / hw4, b = bc / C, c = bc - b * C;
    const float4 g = gates_pre[(b * 2 * C + c) * hw4 + p], xv = x[i], hv = h[i];
    float4 o;
    o.x = sigmoidf_(g.x) * hv.x; o.y = sigmoidf_(g.y) * hv.y; o.z = sigmoidf_(g.z) * hv.z; o.w = sigmoidf_(g.w) * hv.w;
    xrh[(b * 2 * C + c) * hw4 + p] = xv;
    xrh[(b * 2 * C + C + c) * hw4 + p] = o;
}

__global__ __launch_bounds__(256) void convgru_update_kernel(const float4* __restrict__ gates_pre, const float4* __restrict__ cand_pre,
                                                            const float4* __restrict__ h, const float* __restrict__ prelu_w, float4* __restrict__ h_out,
                                                            const float4* __restrict__ x_next, float4* __restrict__ xh_next, int C, int64_t hw4,
                                                            int64_t total4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int64_t p = i % hw4, bc = i / hw4, b = bc / C, c = bc - b * C;
    const float4 g = gates_pre[(b * 2 * C + C + c) * hw4 + p], cp = cand_pre[i], hv = h[i];
    const float zz[4] = {sigmoidf_(g.x), sigmoidf_(g.y), sigmoidf_(g.z), sigmoidf_(g.w)};
    const float cc[4] = {cp.x, cp.y, cp.z, cp.w}, hh[4] = {hv.x, hv.y, hv.z, hv.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float cand = prelu_w ? (cc[k] >= 0.f ? cc[k] : cc[k] * prelu_w[c]) : tanhf(cc[k]);
        const float keep = (1.f - zz[k]) * hh[k], take = zz[k] * cand;        // the reference's order of operations (:28)
        o[k] = keep + take;
    }
    const float4 ov = make_float4(o[0], o[1], o[2], o[3]);
    h_out[i] = ov;
    if (xh_next) {
        xh_next[(b * 2 * C + c) * hw4 + p] = x_next[i];
        xh_next[(b * 2 * C + C + c) * hw4 + p] = ov;
    }
}

}  // namespace

extern "C" int ia_convgru_gates(const float* gates_pre, const float* x, const float* h, float* xrh, int B, int C, int H, int W, void* stream) {
    IA_REQUIRE(gates_pre && x && h && xrh, "null pointer argument");
    IA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "empty tensor");
    IA_REQUIRE(((int64_t)H * W) % 4 == 0, "H * W must be a multiple of 4");
    IA_REQUIRE((int64_t)B * 2 * C * H * W <= INT32_MAX, "tensor is too large");
    const int64_t hw4 = (int64_t)H * W / 4, total4 = (int64_t)B * C * hw4;
    hipLaunchKernelGGL(convgru_gates_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)gates_pre, (const float4*)x, (const float4*)h, (float4*)xrh, C, hw4, total4);
    return ia::check_launch("ia_convgru_gates");
}

extern "C" int ia_convgru_update(const float* gates_pre, const float* cand_pre, const float* h, const float* prelu_weight, float* h_out,
                                 const float* x_next, float* xh_next, int B, int C, int H, int W, void* stream) {
    IA_REQUIRE(gates_pre && cand_pre && h && h_out, "null pointer argument");
    IA_REQUIRE((x_next == nullptr) == (xh_next == nullptr), "x_next and xh_next come together");
    IA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "empty tensor");
    IA_REQUIRE(((int64_t)H * W) % 4 == 0, "H * W must be a multiple of 4");
    IA_REQUIRE((int64_t)B * 2 * C * H * W <= INT32_MAX, "tensor is too large");
    const int64_t hw4 = (int64_t)H * W / 4, total4 = (int64_t)B * C * hw4;
    hipLaunchKernelGGL(convgru_update_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)gates_pre, (const float4*)cand_pre, (const float4*)h, prelu_weight, (float4*)h_out, (const float4*)x_next,
                       (float4*)xh_next, C, hw4, total4);
    return ia::check_launch("ia_convgru_update");
}

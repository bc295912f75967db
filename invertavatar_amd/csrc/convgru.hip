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

// The same two kernels with their convolution inputs written in SPLIT format (ia_act_split's: [b][plane][c/8][pixel][c%8] fp16 hi / lo),
// the operand format of ia_conv2d_mfma_sx: cat[x, r * h] and cat[x_next, h'] are read by those convolutions only, so the fp32 copies and
// the two ia_act_split launches of a step go away (64 of the ~430 launches of a group's decoder chains).  One thread = 4 consecutive
// pixels of one 8-channel group of the 2C-channel result.
typedef _Float16 h16x8_g __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void load8x4(const float* base, int64_t hw, int64_t p4, float (&v)[8][4]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float4 t = reinterpret_cast<const float4*>(base + k * hw)[p4];
        v[k][0] = t.x; v[k][1] = t.y; v[k][2] = t.z; v[k][3] = t.w;
    }
}

__device__ __forceinline__ void store_split8x4(h16x8_g* __restrict__ out, int b, int C8out, int c8, int64_t hw, int64_t p4, const float (&v)[8][4],
                                               ia::SatWatch& watch) {
    h16x8_g* dh = out + ((int64_t)(b * 2) * C8out + c8) * hw + 4 * p4;
    h16x8_g* dl = out + ((int64_t)(b * 2 + 1) * C8out + c8) * hw + 4 * p4;
#pragma unroll
    for (int px = 0; px < 4; ++px) {
        h16x8_g hi, lo;
#pragma unroll
        for (int k = 0; k < 8; ++k) { _Float16 a, l; ia::split_f16(v[k][px], a, l, watch); hi[k] = a; lo[k] = l; }
        dh[px] = hi;
        dl[px] = lo;
    }
}

// xrh_s = split(cat[x, sigmoid(r_pre) * h]); grid over (b, 2C/8 groups, pixel quads): groups [0, C/8) copy x, groups [C/8, 2C/8) gate h
__global__ __launch_bounds__(256) void convgru_gates_split_kernel(const float* __restrict__ gates_pre, const float* __restrict__ x, const float* __restrict__ h,
                                                                 h16x8_g* __restrict__ xrh_s, int C, int64_t hw, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t hw4 = hw / 4, p4 = i % hw4;
    const int C8 = C / 8, g8 = (int)((i / hw4) % (2 * C8)), b = (int)(i / (hw4 * 2 * C8));
    ia::SatWatch watch;
    float v[8][4];
    if (g8 < C8) {
        load8x4(x + ((int64_t)b * C + g8 * 8) * hw, hw, p4, v);
    } else {
        const int c0 = (g8 - C8) * 8;
        float g[8][4];
        load8x4(gates_pre + ((int64_t)b * 2 * C + c0) * hw, hw, p4, g);
        load8x4(h + ((int64_t)b * C + c0) * hw, hw, p4, v);
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int px = 0; px < 4; ++px) v[k][px] = sigmoidf_(g[k][px]) * v[k][px];
    }
    store_split8x4(xrh_s, b, 2 * C8, g8, hw, p4, v, watch);
    watch.report();
}

// h' (fp32) and, with x_next, xh_s = split(cat[x_next, h']); grid over (b, C/8 groups, pixel quads)
__global__ __launch_bounds__(256) void convgru_update_split_kernel(const float* __restrict__ gates_pre, const float* __restrict__ cand_pre,
                                                                  const float* __restrict__ h, const float* __restrict__ prelu_w, float* __restrict__ h_out,
                                                                  const float* __restrict__ x_next, h16x8_g* __restrict__ xh_s, int C, int64_t hw,
                                                                  int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t hw4 = hw / 4, p4 = i % hw4;
    const int C8 = C / 8, c8 = (int)((i / hw4) % C8), b = (int)(i / (hw4 * C8));
    const int c0 = c8 * 8;
    ia::SatWatch watch;
    float z[8][4], cp[8][4], hv[8][4], o[8][4];
    load8x4(gates_pre + ((int64_t)b * 2 * C + C + c0) * hw, hw, p4, z);
    load8x4(cand_pre + ((int64_t)b * C + c0) * hw, hw, p4, cp);
    load8x4(h + ((int64_t)b * C + c0) * hw, hw, p4, hv);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float slope = prelu_w ? prelu_w[c0 + k] : 0.f;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const float zz = sigmoidf_(z[k][px]), cc = cp[k][px];
            const float cand = prelu_w ? (cc >= 0.f ? cc : cc * slope) : tanhf(cc);
            const float keep = (1.f - zz) * hv[k][px], take = zz * cand;          // the reference's order of operations (:28)
            o[k][px] = keep + take;
        }
        reinterpret_cast<float4*>(h_out + ((int64_t)b * C + c0 + k) * hw)[p4] = make_float4(o[k][0], o[k][1], o[k][2], o[k][3]);
    }
    if (xh_s) {
        float xv[8][4];
        load8x4(x_next + ((int64_t)b * C + c0) * hw, hw, p4, xv);
        store_split8x4(xh_s, b, 2 * C8, c8, hw, p4, xv, watch);
        store_split8x4(xh_s, b, 2 * C8, C8 + c8, hw, p4, o, watch);
    }
    watch.report();
}

}  // namespace

extern "C" int ia_convgru_gates_split(const float* gates_pre, const float* x, const float* h, void* xrh_split, int B, int C, int H, int W, void* stream) {
    IA_REQUIRE(gates_pre && x && h && xrh_split, "null pointer argument");
    IA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "empty tensor");
    IA_REQUIRE(C % 8 == 0 && ((int64_t)H * W) % 4 == 0, "C must be a multiple of 8 and H * W a multiple of 4");
    IA_REQUIRE((int64_t)B * 2 * C * H * W <= INT32_MAX, "tensor is too large");
    const int64_t hw = (int64_t)H * W, total = (int64_t)B * (2 * C / 8) * (hw / 4);
    hipLaunchKernelGGL(convgru_gates_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gates_pre, x, h,
                       static_cast<h16x8_g*>(xrh_split), C, hw, total);
    return ia::check_launch("ia_convgru_gates_split");
}

extern "C" int ia_convgru_update_split(const float* gates_pre, const float* cand_pre, const float* h, const float* prelu_weight, float* h_out,
                                       const float* x_next, void* xh_next_split, int B, int C, int H, int W, void* stream) {
    IA_REQUIRE(gates_pre && cand_pre && h && h_out, "null pointer argument");
    IA_REQUIRE((x_next == nullptr) == (xh_next_split == nullptr), "x_next and xh_next_split come together");
    IA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "empty tensor");
    IA_REQUIRE(C % 8 == 0 && ((int64_t)H * W) % 4 == 0, "C must be a multiple of 8 and H * W a multiple of 4");
    IA_REQUIRE((int64_t)B * 2 * C * H * W <= INT32_MAX, "tensor is too large");
    const int64_t hw = (int64_t)H * W, total = (int64_t)B * (C / 8) * (hw / 4);
    hipLaunchKernelGGL(convgru_update_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gates_pre, cand_pre, h,
                       prelu_weight, h_out, x_next, static_cast<h16x8_g*>(xh_next_split), C, hw, total);
    return ia::check_launch("ia_convgru_update_split");
}

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

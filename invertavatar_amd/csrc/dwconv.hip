// ia_dwconv3x3_tokens: the depth-wise 3x3 convolution of a Mix-FFN on the token grid.
//
// Replaces DWConv.forward of the transformer-refined decoders (encoder_inversion/models/mmseg/mix_transformer.py:49-58 in the reference:
// tokens [B, N, C] -> transpose -> view [B, C, H, W] -> nn.Conv2d(C, C, 3, 1, 1, groups=C) -> flatten -> transpose), optionally with the
// GELU that follows it in Mlp.forward (:70-77).  The tokens ARE the image in channels-last order, so the convolution reads and writes
// them in place: one thread = four channels of one token, nine float4 neighbour loads (zero padding at the border of the H x W grid),
// weights pre-arranged [9][C].  HBM / latency bound; the library runs its naive grouped kernel on the transposed view (60 - 130 us per
// call for 0.5 - 33 MB of tokens, 27 calls per one-shot inversion).
#include "ia_common.h"

namespace {

__global__ __launch_bounds__(256) void dwconv_tokens_kernel(const float4* __restrict__ x, const float4* __restrict__ w9c, const float4* __restrict__ bias,
                                                           float4* __restrict__ y, int H, int W, int C4, int64_t total, int act) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    const int64_t tok = i / C4;
    const int n = (int)(tok % ((int64_t)H * W));
    const int64_t b = tok / ((int64_t)H * W);
    const int py = n / W, px = n - py * W;
    float4 acc = bias ? bias[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* xb = x + b * (int64_t)H * W * C4;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = py + ky - 1;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = px + kx - 1;
            if (xx < 0 || xx >= W) continue;
            const float4 v = xb[((int64_t)yy * W + xx) * C4 + c4], wv = w9c[(ky * 3 + kx) * C4 + c4];
            acc.x = fmaf(v.x, wv.x, acc.x); acc.y = fmaf(v.y, wv.y, acc.y); acc.z = fmaf(v.z, wv.z, acc.z); acc.w = fmaf(v.w, wv.w, acc.w);
        }
    }
    if (act == 1) {          // GELU (erf form: torch.nn.GELU(approximate='none'))
        acc.x = 0.5f * acc.x * (1.f + erff(acc.x * 0.70710678118654752f)); acc.y = 0.5f * acc.y * (1.f + erff(acc.y * 0.70710678118654752f));
        acc.z = 0.5f * acc.z * (1.f + erff(acc.z * 0.70710678118654752f)); acc.w = 0.5f * acc.w * (1.f + erff(acc.w * 0.70710678118654752f));
    }
    y[i] = acc;
}

}  // namespace

extern "C" int ia_dwconv3x3_tokens(const float* x, const float* w9c, const float* bias, float* y, int B, int H, int W, int C, int act, void* stream) {
    IA_REQUIRE(x && w9c && y, "x, w9c and y must be device pointers");
    IA_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, "empty tensor");
    IA_REQUIRE(C % 4 == 0, "C must be a multiple of 4 (C = %d)", C);
    IA_REQUIRE(act == 0 || act == 1, "act: 0 = none, 1 = GELU (erf form)");
    IA_REQUIRE((int64_t)B * H * W * C <= INT32_MAX, "tensor is too large");
    const int64_t total = (int64_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(dwconv_tokens_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (const float4*)w9c,
                       (const float4*)bias, (float4*)y, H, W, C / 4, total, act);
    return ia::check_launch("ia_dwconv3x3_tokens");
}

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

typedef _Float16 dw_h16x8 __attribute__((ext_vector_type(8)));

// The same convolution (+ GELU) writing its result as the operand of the linear layer behind it (Mix-FFN: dwconv -> GELU -> fc2): fp16
// hi / lo pairs [2][C/8][B*N][8] (ia_tokens_split's format).  A workgroup takes 32 tokens x 8 octets: 256-byte runs of a token row in,
// 512-byte runs of an octet plane out, through LDS.
__global__ __launch_bounds__(256) void dwconv_tokens_split_kernel(const float4* __restrict__ x, const float4* __restrict__ w9c, const float4* __restrict__ bias,
                                                                 dw_h16x8* __restrict__ xs, int B, int H, int W, int C8, int act) {
    __shared__ dw_h16x8 s_hi[8][33], s_lo[8][33];
    const int tid = threadIdx.x;
    const int64_t M = (int64_t)B * H * W;
    const int64_t m0 = (int64_t)blockIdx.x * 32;
    const int o0 = blockIdx.y * 8;
    {
        const int ml = tid >> 3, ol = tid & 7, o = o0 + ol;
        const int64_t m = m0 + ml;
        ia::SatWatch watch;
        dw_h16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) { hi[j] = (_Float16)0.f; lo[j] = (_Float16)0.f; }
        if (m < M && o < C8) {
            const int n = (int)(m % ((int64_t)H * W));
            const int64_t b = m / ((int64_t)H * W);
            const int py = n / W, px = n - py * W, C4 = C8 * 2;
            float4 a0 = bias ? bias[2 * o] : make_float4(0.f, 0.f, 0.f, 0.f), a1 = bias ? bias[2 * o + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4* xb = x + b * (int64_t)H * W * C4;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = py + ky - 1;
                if (yy < 0 || yy >= H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int xx = px + kx - 1;
                    if (xx < 0 || xx >= W) continue;
                    const float4* p = xb + ((int64_t)yy * W + xx) * C4 + 2 * o;
                    const float4* wp = w9c + (ky * 3 + kx) * C4 + 2 * o;
                    const float4 v0 = p[0], v1 = p[1], w0 = wp[0], w1 = wp[1];
                    a0.x = fmaf(v0.x, w0.x, a0.x); a0.y = fmaf(v0.y, w0.y, a0.y); a0.z = fmaf(v0.z, w0.z, a0.z); a0.w = fmaf(v0.w, w0.w, a0.w);
                    a1.x = fmaf(v1.x, w1.x, a1.x); a1.y = fmaf(v1.y, w1.y, a1.y); a1.z = fmaf(v1.z, w1.z, a1.z); a1.w = fmaf(v1.w, w1.w, a1.w);
                }
            }
            float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (act == 1) v[j] = 0.5f * v[j] * (1.f + erff(v[j] * 0.70710678118654752f));
                _Float16 h, l;
                ia::split_f16(v[j], h, l, watch);
                hi[j] = h;
                lo[j] = l;
            }
        }
        watch.report();
        s_hi[ol][ml] = hi;
        s_lo[ol][ml] = lo;
    }
    __syncthreads();
    const int ol = tid >> 5, ml = tid & 31, o = o0 + ol;
    const int64_t m = m0 + ml;
    if (m < M && o < C8) {
        xs[(int64_t)o * M + m] = s_hi[ol][ml];
        xs[((int64_t)C8 + o) * M + m] = s_lo[ol][ml];
    }
}

}  // namespace

extern "C" int ia_dwconv3x3_tokens_split(const float* x, const float* w9c, const float* bias, void* xs, int B, int H, int W, int C, int act, void* stream) {
    IA_REQUIRE(x && w9c && xs, "x, w9c and xs must be device pointers");
    IA_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, "empty tensor");
    IA_REQUIRE(act == 0 || act == 1, "act: 0 = none, 1 = GELU (erf form)");
    if (C % 16 != 0) return ia::fail(IA_ERR_UNSUPPORTED, "ia_dwconv3x3_tokens_split needs C %% 16 == 0 (got %d)", C);
    IA_REQUIRE((int64_t)B * H * W * C <= (int64_t)1 << 30, "tensor is too large");
    const int64_t M = (int64_t)B * H * W;
    hipLaunchKernelGGL(dwconv_tokens_split_kernel, dim3((unsigned)((M + 31) / 32), (unsigned)((C / 8 + 7) / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)x, (const float4*)w9c, (const float4*)bias, static_cast<dw_h16x8*>(xs), B, H, W, C / 8, act);
    return ia::check_launch("ia_dwconv3x3_tokens_split");
}

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

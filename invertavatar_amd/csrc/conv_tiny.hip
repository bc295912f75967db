// ia_conv3x3_s2_tiny: 3x3 convolution with stride 2 and padding 1 on images of 2^2, 4^2 or 8^2 pixels (outputs 1^2, 2^2, 4^2).
//
// Replaces the last layers of a GradualStyleBlock of the e4e encoder (encoder_inversion/models/e4e.py:22-45: Conv2d(512, 512, 3, stride 2,
// padding 1) + LeakyReLU down to 1x1, batch 1), which the reference hands to cuDNN through torch.nn.Conv2d.  With 1 .. 16 output pixels
// such a layer is a matrix-vector product that streams its 9.4 MB weight once: no tile of the MFMA kernels applies (the split-DMA form
// starts at 8^2 outputs), and the library's kernels for it measured 60 - 90 us per launch (profiles/r05_library_convs.txt).
//
// One wave per output channel.  The image of the batch element is staged in LDS, channel-major with a stride of H*W + 1 floats (lanes
// hold different input channels: conflict-free); a lane loads the pixels of its channel and the nine weights w[o][i][:] (36 contiguous
// bytes per lane, 2304 contiguous bytes per wave) and accumulates every output pixel with compile-time tap / pixel indices; the wave
// reduces the partial sums by lane shuffles in a fixed order (deterministic).  fp32 FMAs throughout.
#include "ia_common.h"

namespace {

constexpr size_t kLdsBytesTiny = 160 * 1024;

template <int ON>      // output side; input side 2 * ON
__global__ __launch_bounds__(256) void conv_tiny_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ y, int I, int O, int act, float alpha) {
    constexpr int N = 2 * ON, HW = N * N, PITCH = HW + 1;
    extern __shared__ float xs[];                                   // [I][PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, o = blockIdx.x * 4 + wave;
    const float* xb = x + (int64_t)b * I * HW;
    for (int e = tid; e < I * HW; e += 256) xs[(e / HW) * PITCH + (e % HW)] = xb[e];
    __syncthreads();
    if (o >= O) return;
    float acc[ON * ON];
#pragma unroll
    for (int p = 0; p < ON * ON; ++p) acc[p] = 0.f;
    const float* wo = w + (int64_t)o * I * 9;
    for (int i = lane; i < I; i += 64) {
        float wv[9], px[HW];
#pragma unroll
        for (int k = 0; k < 9; ++k) wv[k] = wo[i * 9 + k];
#pragma unroll
        for (int q = 0; q < HW; ++q) px[q] = xs[i * PITCH + q];
#pragma unroll
        for (int r = 0; r < ON; ++r)
#pragma unroll
            for (int c = 0; c < ON; ++c)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int iy = 2 * r + ky - 1, ix = 2 * c + kx - 1;      // (compile-time: the padding taps vanish)
                        if (iy >= 0 && iy < N && ix >= 0 && ix < N) acc[r * ON + c] = fmaf(wv[ky * 3 + kx], px[iy * N + ix], acc[r * ON + c]);
                    }
    }
#pragma unroll
    for (int p = 0; p < ON * ON; ++p) {
        float a = acc[p];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
        acc[p] = a;
    }
    if (lane == 0) {
        const float bo = bias ? bias[o] : 0.f;
#pragma unroll
        for (int p = 0; p < ON * ON; ++p) {
            float v = acc[p] + bo;
            if (act == IA_ACT_LRELU) v = v > 0.f ? v : v * alpha;
            y[((int64_t)b * O + o) * (ON * ON) + p] = v;
        }
    }
}

template <int ON>
int launch_tiny(const float* x, const float* w, const float* bias, float* y, int B, int I, int O, int act, float alpha, hipStream_t s) {
    const size_t lds = (size_t)I * (4 * ON * ON + 1) * sizeof(float);
    auto k = conv_tiny_kernel<ON>;
    if (const int rs = ia::reserve_lds((const void*)k, lds, "ia_conv3x3_s2_tiny")) return rs;
    hipLaunchKernelGGL(k, dim3((unsigned)((O + 3) / 4), (unsigned)B), dim3(256), lds, s, x, w, bias, y, I, O, act, alpha);
    return ia::check_launch("ia_conv3x3_s2_tiny");
}

}  // namespace

extern "C" int ia_conv3x3_s2_tiny_supported(int I, int O, int H, int W) {
    if (I < 1 || O < 1 || H != W || (H != 2 && H != 4 && H != 8)) return 0;
    return (size_t)I * (H * W + 1) * sizeof(float) <= kLdsBytesTiny ? 1 : 0;
}

extern "C" int ia_conv3x3_s2_tiny(const float* x, const float* w, const float* bias, float* y, int B, int I, int O, int H, int W, int act,
                                  float alpha, void* stream) {
    IA_REQUIRE(x && w && y, "x, w and y must be device pointers");
    IA_REQUIRE(B > 0 && I > 0 && O > 0, "empty tensor");
    IA_REQUIRE(act == IA_ACT_LINEAR || act == IA_ACT_LRELU, "the epilogue supports linear and lrelu");
    if (!ia_conv3x3_s2_tiny_supported(I, O, H, W))
        return ia::fail(IA_ERR_UNSUPPORTED, "ia_conv3x3_s2_tiny takes square images of 2, 4 or 8 pixels a side whose channels fit the LDS (I %d, %dx%d)", I, H, W);
    IA_REQUIRE((int64_t)O * I * 9 <= INT32_MAX && B <= 65535, "tensor is too large");
    hipStream_t s = (hipStream_t)stream;
    if (H == 2) return launch_tiny<1>(x, w, bias, y, B, I, O, act, alpha, s);
    if (H == 4) return launch_tiny<2>(x, w, bias, y, B, I, O, act, alpha, s);
    return launch_tiny<4>(x, w, bias, y, B, I, O, act, alpha, s);
}

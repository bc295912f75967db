// ia_torgb: ToRGBLayer + the skip-image up-sampling and add of a 'skip' SynthesisBlock in ONE launch.
//
// Replaces (training/networks_stylegan2.py:340-357 and :452-458 of SynthesisBlock.forward)
//     img = upsample2d(img_prev, resample_filter)                       (upfirdn2d, up 2, pad [2,1,2,1], gain 4)
//     y   = bias_act(modulated_conv2d(x, w, styles, demodulate=False), b, clamp)      (1x1 convolution, styles * 1/sqrt(I))
//     img = img + y
// which this backend ran as three to four launches (stream-K 1x1 MFMA convolution + fix-up + the FIR).  A 1x1 convolution to
// 3 / 32 / 96 channels is a streaming read of x: out[o, p] = sum_i w[o, i] * (s[i] * x[i, p]).  Here a wave owns 64 consecutive
// pixels; x is read straight from global memory, 256 bytes per wave and channel, coalesced; the weights of a channel are
// wave-uniform and come through the scalar cache as SGPR operands of the FMAs: no LDS, no MFMA, no staging in the channel loop.
//   large images : the four waves of a workgroup split the OUTPUT channels (x is re-read by the sister waves from L1/L2);
//   images <= 32^2: the four waves split the INPUT channels (the loop is load-latency bound there) and reduce through LDS.
// The skip image is up-sampled on the fly in the epilogue (the 2 x 2 non-zero polyphase taps of the 4 x 4 filter, summed in the
// order of upfirdn2d_tiled<.., 2, 4>).  HBM-bound: reads x once (+ the quarter-size skip image), writes the image.
#include "ia_common.h"

namespace {

struct RgbGeo {
    int B, I, O, H, W;
    int has_prev;
    float clamp;
};

// up-sampled skip image at (oy, ox): upfirdn2d(up 2, pad0 2, 4x4 filter f, gain 4, true convolution)
__device__ __forceinline__ float upsample_tap_sum(const float* __restrict__ prev, int ph, int pw, int oy, int ox, const float* __restrict__ f) {
    const int by = oy - 2, bx = ox - 2;
    const int ky0 = ((-by) % 2 + 2) % 2, kx0 = ((-bx) % 2 + 2) % 2;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int ky = ky0 + 2 * a, iy = (by + ky) >> 1;          // by + ky is even
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            const int kx = kx0 + 2 * bb, ix = (bx + kx) >> 1;
            const float v = (iy >= 0 && iy < ph && ix >= 0 && ix < pw) ? prev[iy * pw + ix] : 0.f;
            acc = fmaf(v, f[15 - (ky * 4 + kx)] * 4.f, acc);          // flipped tap (true convolution) times the up-sampling gain
        }
    }
    return acc;
}

// OPW = output channels per wave (O-split) or all output channels (K-split: KSPLIT = true).
// Weights and styles of a block of UN channels are fetched with VECTOR loads (lane l holds weight (channel l / OPW, output l % OPW)
// of the block: consecutive lanes = consecutive outputs of a weight row) and handed to the FMAs with v_readlane: every load of a
// block -- UN pixels rows, the weights, the styles -- is independent and in flight together, one wait per block.  (A first version
// read the weights through the scalar cache: hipcc turned the per-output guards into branches around single s_load_dword's with a
// wait each -- 200 us for a 4x4 image.)
template <int OPW, bool KSPLIT>
__global__ __launch_bounds__(256) void torgb_kernel(const float* __restrict__ x, const float* __restrict__ wk, const float* __restrict__ styles,
                                                   const float* __restrict__ bias, const float* __restrict__ prev, const float* __restrict__ f,
                                                   float* __restrict__ y, RgbGeo g) {
    extern __shared__ float red[];                         // K-split only: [4][OPW][64]
    constexpr int UN = 8, NV = UN * OPW, NL = (NV + 63) / 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int64_t hw = (int64_t)g.H * g.W;
    const int64_t p = (int64_t)blockIdx.x * 64 + lane, pc = p < hw ? p : hw - 1;
    const int o0 = KSPLIT ? 0 : wave * OPW;
    const int i_lo = KSPLIT ? wave * (g.I / 4) : 0, i_hi = KSPLIT ? (wave == 3 ? g.I : (wave + 1) * (g.I / 4)) : g.I;
    const float* xb = x + (int64_t)b * g.I * hw + pc;
    const float* sb = styles + (int64_t)b * g.I;
    float acc[OPW];
#pragma unroll
    for (int j = 0; j < OPW; ++j) acc[j] = 0.f;
    // per-lane weight slots of a block: value v = lane + 64 k  ->  (channel u = v / OPW, output j = v % OPW)
    int w_u[NL], w_off[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int v = lane + 64 * k, u = v / OPW, j = v - u * OPW;
        w_u[k] = (v < NV && o0 + j < g.O) ? u : -1;
        w_off[k] = u * g.O + o0 + j;
    }
    for (int i = i_lo; i < i_hi; i += UN) {
        float xv[UN], wv[NL];
#pragma unroll
        for (int u = 0; u < UN; ++u) xv[u] = xb[(int64_t)min(i + u, i_hi - 1) * hw];
#pragma unroll
        for (int k = 0; k < NL; ++k) wv[k] = (w_u[k] >= 0 && i + w_u[k] < i_hi) ? wk[(int64_t)i * g.O + w_off[k]] : 0.f;     // channels past the range weigh 0
        const float sv = sb[min(i + (lane & (UN - 1)), g.I - 1)];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const float xs = xv[u] * __builtin_amdgcn_readlane(sv, u);
#pragma unroll
            for (int j = 0; j < OPW; ++j) {
                const int v = u * OPW + j;
                acc[j] = fmaf(xs, __builtin_amdgcn_readlane(wv[v / 64], v % 64), acc[j]);
            }
        }
    }
    const int oy = (int)(pc / g.W), ox = (int)(pc - (int64_t)oy * g.W);
    auto finish = [&](int o, float v) {
        v += bias[o];
        if (g.clamp >= 0.f) v = fminf(fmaxf(v, -g.clamp), g.clamp);
        if (g.has_prev) v += upsample_tap_sum(prev + ((int64_t)b * g.O + o) * (g.H / 2) * (g.W / 2), g.H / 2, g.W / 2, oy, ox, f);
        y[((int64_t)b * g.O + o) * hw + p] = v;
    };
    if constexpr (KSPLIT) {
#pragma unroll
        for (int j = 0; j < OPW; ++j) red[(wave * OPW + j) * 64 + lane] = acc[j];
        __syncthreads();
        if (p >= hw) return;
        for (int o = wave; o < g.O; o += 4) {                              // this wave finishes outputs o = wave, wave + 4, ...
            const float v = ((red[(0 * OPW + o) * 64 + lane] + red[(1 * OPW + o) * 64 + lane]) + red[(2 * OPW + o) * 64 + lane]) + red[(3 * OPW + o) * 64 + lane];
            finish(o, v);
        }
    } else {
        if (p >= hw) return;
#pragma unroll
        for (int j = 0; j < OPW; ++j)
            if (o0 + j < g.O) finish(o0 + j, acc[j]);
    }
}

template <int OPW, bool KSPLIT>
int launch_rgb(const float* x, const float* wk, const float* styles, const float* bias, const float* prev, const float* f, float* y,
               const RgbGeo& g, hipStream_t s) {
    const size_t lds = KSPLIT ? (size_t)4 * OPW * 64 * sizeof(float) : 0;
    auto k = torgb_kernel<OPW, KSPLIT>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const dim3 grid((unsigned)(((int64_t)g.H * g.W + 63) / 64), g.B);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, x, wk, styles, bias, prev, f, y, g);
    return ia::check_launch("ia_torgb");
}

}  // namespace

extern "C" int ia_torgb(const float* x, const float* wk, const float* styles, const float* bias, const float* prev_img, const float* f,
                        float* y, int B, int I, int O, int H, int W, float clamp, void* stream) {
    IA_REQUIRE(x && wk && styles && bias && y, "null pointer argument");
    IA_REQUIRE(B > 0 && I > 0 && O > 0 && H > 0 && W > 0, "empty tensor");
    IA_REQUIRE(prev_img == nullptr || (f != nullptr && H % 2 == 0 && W % 2 == 0), "the skip image is [B,O,H/2,W/2] and needs the 4x4 resample filter");
    IA_REQUIRE((int64_t)B * I * H * W <= INT32_MAX && (int64_t)B * O * H * W <= INT32_MAX, "tensor is too large");
    if ((int64_t)H * W < 64) return ia::fail(IA_ERR_UNSUPPORTED, "ia_torgb: images below 8x8 run on ia_conv2d_mfma (ksize 1)");
    RgbGeo g{B, I, O, H, W, prev_img ? 1 : 0, clamp};
    hipStream_t s = (hipStream_t)stream;
    const bool small = (int64_t)H * W <= 1024 && I % 4 == 0;
    if (small) {
        if (O <= 4) return launch_rgb<4, true>(x, wk, styles, bias, prev_img, f, y, g, s);
        if (O <= 32) return launch_rgb<32, true>(x, wk, styles, bias, prev_img, f, y, g, s);
        if (O <= 96) return launch_rgb<96, true>(x, wk, styles, bias, prev_img, f, y, g, s);
    } else {
        if (O <= 4) return launch_rgb<1, false>(x, wk, styles, bias, prev_img, f, y, g, s);
        if (O <= 32) return launch_rgb<8, false>(x, wk, styles, bias, prev_img, f, y, g, s);
        if (O <= 96) return launch_rgb<24, false>(x, wk, styles, bias, prev_img, f, y, g, s);
    }
    return ia::fail(IA_ERR_UNSUPPORTED, "ia_torgb: up to 96 output channels (got %d); use ia_conv2d_mfma with ksize 1", O);
}

// ia_filtered_lrelu: bias -> zero-insert up-sampling -> pad/crop -> up-FIR (x up^2) -> leaky ReLU x gain -> clamp -> down-FIR ->
// decimation, per channel, in ONE kernel with the up-sampled intermediate held in LDS.
//
// Replaces filtered_lrelu_plugin.filtered_lrelu (torch_utils/ops/filtered_lrelu.cpp:20-22, kernels filtered_lrelu.cu:144-1103);
// semantics are those of the reference's own definition of the op, _filtered_lrelu_ref (filtered_lrelu.py:123-155):
//     t = upfirdn2d(x + b, fu, up = up, padding = [px0, px1, py0, py1], gain = up^2)
//     t = clamp(lrelu(t, slope) * gain, +-clamp)
//     y = upfirdn2d(t, fd, down = down)
// with the filters applied as true convolutions unless flip_filter (upfirdn2d semantics).
//
// A workgroup produces a TOY x TOX tile of one (n, c) plane in three LDS-resident steps: (1) the input window (+ bias), (2) the
// window of the up-sampled, filtered and activated intermediate that the tile's down-filter footprint covers -- only the
// non-zero polyphase taps of the up-filter are visited --, (3) the down-filtered, decimated outputs.  The intermediate
// (up^2 x the input, the tensor the unfused composition writes to and re-reads from HBM twice) never leaves the CU.
// HBM traffic = input + output; arithmetic = (fu taps / up^2) MACs per intermediate sample + fd taps per output.
#include "ia_common.h"

namespace {

struct FlrGeo {
    int n, c, ih, iw, oh, ow;
    int up, down, px0, py0;
    int fuh, fuw, fdh, fdw;
    int mh, mw;                  // size of the full intermediate image: ih*up + py0 + py1 - (fuh - 1)
    int tih, tiw, tmh, tmw;      // LDS window sizes: input, intermediate
    float gain_up, gain, slope, clamp;
    int flip;
};

constexpr int TOX = 32, TOY = 8;

__device__ __forceinline__ int fdiv(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

template <class T>
__global__ __launch_bounds__(256) void filtered_lrelu_kernel(const T* __restrict__ x, const float* __restrict__ fu, const float* __restrict__ fd,
                                                            const T* __restrict__ b, T* __restrict__ y, FlrGeo g) {
    extern __shared__ float lds[];
    float* ku = lds;                               // [fuh][fuw] flipped (true convolution) unless g.flip, times up^2
    float* kd = ku + g.fuh * g.fuw;                // [fdh][fdw]
    float* s_in = kd + g.fdh * g.fdw;              // [tih][tiw]
    float* s_mid = s_in + g.tih * g.tiw;           // [tmh][tmw]
    const int tid = threadIdx.x;
    const int tiles_x = (g.ow + TOX - 1) / TOX;
    const int ox0 = (blockIdx.x % tiles_x) * TOX, oy0 = (blockIdx.x / tiles_x) * TOY;
    const int64_t plane = blockIdx.y;
    const int ch = (int)(plane % g.c);
    for (int i = tid; i < g.fuh * g.fuw; i += 256) {
        const int ky = i / g.fuw, kx = i - ky * g.fuw;
        ku[i] = (fu ? fu[(g.flip ? ky : g.fuh - 1 - ky) * g.fuw + (g.flip ? kx : g.fuw - 1 - kx)] : 1.f) * g.gain_up;
    }
    for (int i = tid; i < g.fdh * g.fdw; i += 256) {
        const int ky = i / g.fdw, kx = i - ky * g.fdw;
        kd[i] = fd ? fd[(g.flip ? ky : g.fdh - 1 - ky) * g.fdw + (g.flip ? kx : g.fdw - 1 - kx)] : 1.f;
    }
    // (1) input window: intermediate rows [my0, my0 + tmh) read up-sampled rows my0 .. my0 + tmh - 1 + fuh - 1, i.e. input rows
    // floor((my0 - py0) / up) ..
    const int my0 = oy0 * g.down, mx0 = ox0 * g.down;
    const int iy0 = fdiv(my0 - g.py0, g.up), ix0 = fdiv(mx0 - g.px0, g.up);
    const float bias = b ? (float)ia::Num<T>::load(b + ch) : 0.f;
    const T* xp = x + plane * (int64_t)g.ih * g.iw;
    for (int i = tid; i < g.tih * g.tiw; i += 256) {
        const int r = i / g.tiw, cidx = i - r * g.tiw;
        const int iy = iy0 + r, ix = ix0 + cidx;
        float v = 0.f;     // outside the image the UP-SAMPLED signal is zero padding: no bias there
        if (iy >= 0 && iy < g.ih && ix >= 0 && ix < g.iw) v = (float)ia::Num<T>::load(xp + (int64_t)iy * g.iw + ix) + bias;
        s_in[i] = v;
    }
    __syncthreads();
    // (2) intermediate window: up-FIR over the non-zero polyphase taps, then gain, leaky ReLU, clamp
    for (int i = tid; i < g.tmh * g.tmw; i += 256) {
        const int r = i / g.tmw, cidx = i - r * g.tmw;
        const int gy = my0 + r, gx = mx0 + cidx;          // position in the full intermediate image
        float v = 0.f;
        if (gy < g.mh && gx < g.mw) {
            const int by = gy - g.py0, bx = gx - g.px0;   // up-sampled coordinate of tap (0, 0)
            const int ky0 = ((-by) % g.up + g.up) % g.up, kx0 = ((-bx) % g.up + g.up) % g.up;
            float acc = 0.f;
            for (int ky = ky0; ky < g.fuh; ky += g.up) {
                const int ly = fdiv(by + ky, g.up) - iy0;
                for (int kx = kx0; kx < g.fuw; kx += g.up) {
                    const int lx = fdiv(bx + kx, g.up) - ix0;
                    acc = fmaf(s_in[ly * g.tiw + lx], ku[ky * g.fuw + kx], acc);
                }
            }
            v = acc > 0.f ? acc : acc * g.slope;
            v *= g.gain;
            if (g.clamp >= 0.f) v = fminf(fmaxf(v, -g.clamp), g.clamp);
        }
        s_mid[i] = v;
    }
    __syncthreads();
    // (3) down-FIR + decimation
    T* yp = y + plane * (int64_t)g.oh * g.ow;
    const int tx = tid % TOX, ty = tid / TOX;
    const int ox = ox0 + tx, oy = oy0 + ty;
    if (ox >= g.ow || oy >= g.oh) return;
    float acc = 0.f;
    const float* m = s_mid + (ty * g.down) * g.tmw + tx * g.down;
    for (int ky = 0; ky < g.fdh; ++ky)
        for (int kx = 0; kx < g.fdw; ++kx) acc = fmaf(m[ky * g.tmw + kx], kd[ky * g.fdw + kx], acc);
    ia::Num<T>::store(yp + (int64_t)oy * g.ow + ox, acc);
}

template <class T>
int launch_flr(const void* x, const float* fu, const float* fd, const void* b, void* y, const FlrGeo& g, hipStream_t s) {
    const size_t lds = sizeof(float) * ((size_t)g.fuh * g.fuw + (size_t)g.fdh * g.fdw + (size_t)g.tih * g.tiw + (size_t)g.tmh * g.tmw);
    if (lds > 160 * 1024) return ia::fail(IA_ERR_UNSUPPORTED, "ia_filtered_lrelu: tile needs %zu bytes of LDS", lds);
    auto k = filtered_lrelu_kernel<T>;
    if (const int rs = ia::reserve_lds((const void*)k, (size_t)(lds), "filtered_lrelu")) return rs;
    const dim3 grid(((g.ow + TOX - 1) / TOX) * ((g.oh + TOY - 1) / TOY), g.n * g.c);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, (const T*)x, fu, fd, (const T*)b, (T*)y, g);
    return ia::check_launch("ia_filtered_lrelu");
}

}  // namespace

extern "C" int ia_filtered_lrelu(const void* x, const float* fu, const float* fd, const void* b, void* y, int dtype,
                                 int n, int c, int in_h, int in_w, int out_h, int out_w, int fu_h, int fu_w, int fd_h, int fd_w,
                                 int up, int down, int px0, int px1, int py0, int py1, float gain, float slope, float clamp,
                                 int flip_filter, void* stream) {
    IA_REQUIRE(x && y, "null pointer argument");
    IA_REQUIRE(n > 0 && c > 0 && in_h > 0 && in_w > 0, "x has zero size");
    IA_REQUIRE(up >= 1 && down >= 1, "up / down factors must be at least 1");
    IA_REQUIRE(fu_h >= 1 && fu_w >= 1 && fd_h >= 1 && fd_w >= 1, "filters must be at least 1x1 (pass NULL + 1x1 for identity)");
    IA_REQUIRE(gain > 0.f && slope >= 0.f, "gain must be positive and slope non-negative");
    IA_REQUIRE((int64_t)n * c <= 65535, "too many planes for one launch");
    FlrGeo g;
    g.n = n; g.c = c; g.ih = in_h; g.iw = in_w; g.up = up; g.down = down; g.px0 = px0; g.py0 = py0;
    g.fuh = fu_h; g.fuw = fu_w; g.fdh = fd_h; g.fdw = fd_w;
    g.mh = in_h * up + py0 + py1 - (fu_h - 1);
    g.mw = in_w * up + px0 + px1 - (fu_w - 1);
    g.oh = (g.mh - (fd_h - 1) + (down - 1)) / down;     // (filtered_lrelu.py:142-143)
    g.ow = (g.mw - (fd_w - 1) + (down - 1)) / down;
    IA_REQUIRE(g.mh >= 1 && g.mw >= 1 && g.oh >= 1 && g.ow >= 1, "output would be empty");
    IA_REQUIRE(g.oh == out_h && g.ow == out_w, "out size must be %d x %d for these parameters (got %d x %d)", g.oh, g.ow, out_h, out_w);
    IA_REQUIRE((int64_t)n * c * in_h * in_w <= INT32_MAX && (int64_t)n * c * g.oh * g.ow <= INT32_MAX, "tensor is too large");
    g.tmh = (TOY - 1) * down + fd_h; g.tmw = (TOX - 1) * down + fd_w;
    g.tih = (g.tmh + fu_h - 1 + up - 1) / up + 1; g.tiw = (g.tmw + fu_w - 1 + up - 1) / up + 1;
    g.gain_up = (float)(up * up); g.gain = gain; g.slope = slope; g.clamp = clamp; g.flip = flip_filter;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case IA_F32: return launch_flr<float>(x, fu, fd, b, y, g, s);
        case IA_F16: return launch_flr<__half>(x, fu, fd, b, y, g, s);
        case IA_F64: return ia::fail(IA_ERR_UNSUPPORTED, "ia_filtered_lrelu: float64 runs through the unfused composition");
        default: return ia::fail(IA_ERR_INVALID_ARG, "unsupported dtype %d", dtype);
    }
}

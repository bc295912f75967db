// ia_upsample_bilinear_add: y = bilinear_resize(x, align_corners = True) + addend.
//
// Replaces `_upsample_add` of the e4e feature pyramid (encoder_inversion/models/e4e.py:48-65 in the reference: F.interpolate(x, size,
// mode='bilinear', align_corners=True) + y), two calls per encode on 512-channel maps of 16^2 -> 32^2 -> 64^2.  The ATen kernel for it
// takes 158 us per call on these shapes (one thread per output pixel looping over the channels); here one thread = one output element,
// the source taps of a row are neighbours in memory.  Same source-index arithmetic as ATen (area_pixel_compute_source_index with
// align_corners: src = dst * (in - 1) / (out - 1), computed in float) and the same order of the four products.
#include "ia_common.h"

namespace {

__global__ __launch_bounds__(256) void upsample_bilinear_add_kernel(const float* __restrict__ x, const float* __restrict__ addend, float* __restrict__ y,
                                                                   int H, int W, int OH, int OW, float sy, float sx, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
    const int64_t plane = i / ((int64_t)OW * OH);
    const float fy = sy * oy, fx = sx * ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 < H - 1 ? y0 + 1 : y0, x1 = x0 < W - 1 ? x0 + 1 : x0;
    const float ly1 = fy - y0, ly0 = 1.f - ly1, lx1 = fx - x0, lx0 = 1.f - lx1;
    const float* p = x + plane * (int64_t)H * W;
    const float v = ly0 * (lx0 * p[y0 * W + x0] + lx1 * p[y0 * W + x1]) + ly1 * (lx0 * p[y1 * W + x0] + lx1 * p[y1 * W + x1]);
    y[i] = addend ? v + addend[i] : v;
}

}  // namespace

extern "C" int ia_upsample_bilinear_add(const float* x, const float* addend, float* y, int planes, int H, int W, int OH, int OW, void* stream) {
    IA_REQUIRE(x && y, "x and y must be device pointers");
    IA_REQUIRE(planes > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "empty tensor");
    IA_REQUIRE((int64_t)planes * OH * OW <= INT32_MAX && (int64_t)planes * H * W <= INT32_MAX, "tensor is too large");
    const int64_t total = (int64_t)planes * OH * OW;
    const float sy = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f, sx = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    hipLaunchKernelGGL(upsample_bilinear_add_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, addend, y, H, W, OH, OW,
                       sy, sx, total);
    return ia::check_launch("ia_upsample_bilinear_add");
}

// ia_fill_mouth: GPU flood fill that closes the mouth hole of the rasterised face mask.
//
// Replaces fill_mouth(images, blur_mouth_edge=False) of the reference
// (training_avatar_texture/volumetric_rendering/renderer.py:716-741), which copies the mask to the host and calls
// cv2.floodFill(img*255, seed (0,0), newVal 255, loDiff 0, upDiff 254, FLOODFILL_FIXED_RANGE) per frame -- a device
// sync twice per frame.  Semantics kept: a pixel is "passable" when seed <= v <= seed + 254 (v = alpha*255,
// seed = v at (0,0)); the 4-connected passable region containing (0,0) is filled with 255;
//   mouth = (255 - filled) / 255.
// One workgroup per image; the label image lives in LDS (1 byte / pixel, rows padded by 4 bytes so that per-row
// sweeps hit distinct banks).  Propagation is by alternating full-line sweeps (left/right per row, up/down per
// column), each of which carries the fill along a whole line in one pass; convex-ish masks converge in 2-3 rounds.
#include "ia_common.h"

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void fill_mouth_kernel(const float* __restrict__ alpha, float* __restrict__ mouth, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) unsigned char st[];   // 0 wall, 1 passable, 2 reached
    __shared__ int changed;
    const int S = W + 4;
    const float* a = alpha + (int64_t)blockIdx.x * H * W;
    float* m = mouth + (int64_t)blockIdx.x * H * W;
    const float seed = a[0] * 255.f;
    for (int i = threadIdx.x; i < H * W; i += kThreads) {
        const int y = i / W, x = i - y * W;
        const float v = a[i] * 255.f;
        st[y * S + x] = (v >= seed && v <= seed + 254.f) ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) st[0] = 2;
    __syncthreads();
    for (int round = 0; round < H + W; ++round) {
        if (threadIdx.x == 0) changed = 0;
        __syncthreads();
        int local = 0;
        for (int y = threadIdx.x; y < H; y += kThreads) {
            unsigned char* row = st + y * S;
            bool reach = false;
            for (int x = 0; x < W; ++x) { const unsigned char s = row[x]; if (s == 2) reach = true; else if (s == 1 && reach) { row[x] = 2; local = 1; } else if (s == 0) reach = false; }
            reach = false;
            for (int x = W - 1; x >= 0; --x) { const unsigned char s = row[x]; if (s == 2) reach = true; else if (s == 1 && reach) { row[x] = 2; local = 1; } else if (s == 0) reach = false; }
        }
        __syncthreads();
        for (int x = threadIdx.x; x < W; x += kThreads) {
            unsigned char* col = st + x;
            bool reach = false;
            for (int y = 0; y < H; ++y) { const unsigned char s = col[y * S]; if (s == 2) reach = true; else if (s == 1 && reach) { col[y * S] = 2; local = 1; } else if (s == 0) reach = false; }
            reach = false;
            for (int y = H - 1; y >= 0; --y) { const unsigned char s = col[y * S]; if (s == 2) reach = true; else if (s == 1 && reach) { col[y * S] = 2; local = 1; } else if (s == 0) reach = false; }
        }
        if (local) changed = 1;
        __syncthreads();
        if (!changed) break;
        __syncthreads();
    }
    for (int i = threadIdx.x; i < H * W; i += kThreads) {
        const int y = i / W, x = i - y * W;
        m[i] = st[y * S + x] == 2 ? 0.f : (255.f - a[i] * 255.f) / 255.f;
    }
}

}  // namespace

extern "C" int ia_fill_mouth(const float* alpha, float* mouth, int B, int H, int W, void* stream) {
    IA_REQUIRE(alpha && mouth, "null pointer argument");
    IA_REQUIRE(B > 0 && H > 0 && W > 0, "empty tensor");
    const size_t lds = (size_t)H * (W + 4);
    if (lds > 150 * 1024) return ia::fail(IA_ERR_UNSUPPORTED, "mask %dx%d does not fit the LDS label image", H, W);
    (void)hipFuncSetAttribute((const void*)fill_mouth_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(fill_mouth_kernel, dim3(B), dim3(kThreads), lds, (hipStream_t)stream, alpha, mouth, H, W);
    return ia::check_launch("ia_fill_mouth");
}

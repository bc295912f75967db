// ia_layout_grid_u8: the output side of the generator path -- float image batch -> uint8 picture grid.
//
// Replaces layout_grid of the reference's scripts (reenact_avatar_next3d.py:117-131; the same helper is in eval_seq.py and
// gen_videos scripts):   img = (img * 127.5 + 128).clamp(0, 255).to(uint8)   [:123]
//                        reshape(grid_h, grid_w, C, H, W).permute(2, 0, 3, 1, 4).reshape(C, grid_h*H, grid_w*W)  [:124-126]
//                        permute(1, 2, 0)  (chw_to_hwc)   [:127-128]
// in ONE pass over the frames: each thread converts 4 horizontally adjacent pixels of all channels and writes them as packed
// bytes.  The arithmetic is the reference's, operation for operation: one multiply, one add (no fma: the library is built with
// -ffp-contract=off), clamp, truncation toward zero.  HBM-bound: 4 B read + 1 B written per element.
// With grid_w = 1, grid_h = B the output is the batch of HWC uint8 frames [B, H, W, C]: the form handed to the video writer and
// the form all-gathered between GPUs (a quarter of the fp32 bytes).
#include "ia_common.h"

namespace {

__device__ __forceinline__ unsigned to_u8(float v) {
    float t = v * 127.5f;
    t = t + 128.f;
    t = fminf(fmaxf(t, 0.f), 255.f);      // NaN -> 0 (fmaxf returns the non-NaN operand), as torch.clamp + cast give on the CPU path
    return (unsigned)(int)t;
}

template <int C>
__global__ __launch_bounds__(256) void layout_grid_u8_kernel(const float* __restrict__ img, uint8_t* __restrict__ out, int B, int H, int W,
                                                            int grid_w, int hwc) {
    const int wq = W / 4;                                         // quads per row
    const int64_t total = (int64_t)B * H * wq;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t OW = (int64_t)grid_w * W;
    const int grid_h = B / grid_w;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int q = (int)(i % wq);
        const int y = (int)((i / wq) % H);
        const int b = (int)(i / ((int64_t)wq * H));
        const int gy = b / grid_w, gx = b - gy * grid_w;
        const int64_t oy = (int64_t)gy * H + y, ox = (int64_t)gx * W + 4 * q;
        unsigned px[C][4];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(img + (((int64_t)b * C + c) * H + y) * W + 4 * q);
            px[c][0] = to_u8(v.x); px[c][1] = to_u8(v.y); px[c][2] = to_u8(v.z); px[c][3] = to_u8(v.w);
        }
        if (hwc) {
            uint8_t* dst = out + (oy * OW + ox) * C;              // 4 pixels x C bytes, contiguous
            if constexpr (C == 3) {                                // 12 bytes as three aligned words (ox % 4 == 0 => offset % 12 == 0)
                unsigned w0 = px[0][0] | (px[1][0] << 8) | (px[2][0] << 16) | (px[0][1] << 24);
                unsigned w1 = px[1][1] | (px[2][1] << 8) | (px[0][2] << 16) | (px[1][2] << 24);
                unsigned w2 = px[2][2] | (px[0][3] << 8) | (px[1][3] << 16) | (px[2][3] << 24);
                unsigned* d32 = reinterpret_cast<unsigned*>(dst);
                d32[0] = w0; d32[1] = w1; d32[2] = w2;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int c = 0; c < C; ++c) dst[k * C + c] = (uint8_t)px[c][k];
            }
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                unsigned w = px[c][0] | (px[c][1] << 8) | (px[c][2] << 16) | (px[c][3] << 24);
                *reinterpret_cast<unsigned*>(out + ((int64_t)c * grid_h * H + oy) * OW + ox) = w;
            }
        }
    }
}

// Up to 8 (source, destination, bytes) segments copied by one launch: the per-frame inputs of a captured frame (latents, camera,
// UV map, jitter) go into the graph's static buffers in one ~8 us kernel instead of four back-to-back copy launches.
struct Segments { const char* src[8]; char* dst[8]; int64_t bytes[8]; int64_t first_block[9]; int n; };
__global__ __launch_bounds__(256) void stage_inputs_kernel(Segments sg) {
    int k = 0;
    while (k + 1 < sg.n && (int64_t)blockIdx.x >= sg.first_block[k + 1]) ++k;      // block-uniform
    const int64_t base = ((int64_t)blockIdx.x - sg.first_block[k]) * 4096 + (int64_t)threadIdx.x * 16;
    const char* s = sg.src[k]; char* d = sg.dst[k];
    const int64_t n = sg.bytes[k];
    const bool vec = ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0;
    if (vec && base + 16 <= n) { *reinterpret_cast<uint4*>(d + base) = *reinterpret_cast<const uint4*>(s + base); return; }
    for (int64_t i = base; i < base + 16 && i < n; ++i) d[i] = s[i];
}

}  // namespace

extern "C" int ia_stage_inputs(const void* const* src, void* const* dst, const int64_t* nbytes, int n, void* stream) {
    IA_REQUIRE(src && dst && nbytes, "null pointer argument");
    IA_REQUIRE(n >= 1 && n <= 8, "1 to 8 segments per launch (got %d)", n);
    Segments sg;
    int64_t blocks = 0;
    for (int k = 0; k < n; ++k) {
        IA_REQUIRE(src[k] && dst[k] && nbytes[k] > 0, "segment %d is empty or null", k);
        sg.src[k] = static_cast<const char*>(src[k]); sg.dst[k] = static_cast<char*>(dst[k]); sg.bytes[k] = nbytes[k];
        sg.first_block[k] = blocks;
        blocks += (nbytes[k] + 4095) / 4096;
    }
    sg.first_block[n] = blocks; sg.n = n;
    IA_REQUIRE(blocks <= INT32_MAX, "too many bytes for one launch");
    hipLaunchKernelGGL(stage_inputs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, sg);
    return ia::check_launch("ia_stage_inputs");
}

extern "C" int ia_layout_grid_u8(const float* img, uint8_t* out, int B, int C, int H, int W, int grid_w, int grid_h, int chw_to_hwc,
                                 void* stream) {
    IA_REQUIRE(img && out, "null pointer argument");
    IA_REQUIRE(B > 0 && H > 0 && W > 0, "empty tensor");
    IA_REQUIRE(grid_w > 0 && grid_h > 0 && grid_w * grid_h == B, "batch %d does not fill a %d x %d grid", B, grid_w, grid_h);
    IA_REQUIRE((int64_t)B * C * H * W <= INT32_MAX, "tensor is too large");
    if (W % 4 != 0 || !(C == 1 || C == 3 || C == 4))
        return ia::fail(IA_ERR_UNSUPPORTED, "ia_layout_grid_u8: needs W %% 4 == 0 and 1, 3 or 4 channels (got W=%d, C=%d)", W, C);
    const int64_t work = (int64_t)B * H * (W / 4);
    const dim3 grid(ia::streaming_grid(work, 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (C == 3) hipLaunchKernelGGL(layout_grid_u8_kernel<3>, grid, block, 0, s, img, out, B, H, W, grid_w, chw_to_hwc);
    else if (C == 1) hipLaunchKernelGGL(layout_grid_u8_kernel<1>, grid, block, 0, s, img, out, B, H, W, grid_w, chw_to_hwc);
    else hipLaunchKernelGGL(layout_grid_u8_kernel<4>, grid, block, 0, s, img, out, B, H, W, grid_w, chw_to_hwc);
    return ia::check_launch("ia_layout_grid_u8");
}

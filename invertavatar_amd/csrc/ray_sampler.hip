// ia_ray_sampler: camera -> per-pixel rays, batched, no host round trip.
//
// Replaces RaySampler_zxc.forward (training_avatar_texture/volumetric_rendering/ray_sampler.py:70-107): K' = K with rows 0-1
// scaled by the render resolution; d = normalize(R * K'^-1 * [i, j, 1]); o = t.  The reference loops over the batch in Python
// and calls torch.linalg.inv per frame (a LAPACK/rocSOLVER call that also breaks hipGraph capture); here the 3x3 inverse is the
// closed-form adjugate, evaluated once per thread (it is 20 flops).
#include "ia_common.h"

namespace {

__global__ __launch_bounds__(256) void ray_sampler_kernel(const float* __restrict__ cam, int cam_stride, float* __restrict__ rays_o,
                                                          float* __restrict__ rays_d, int B, int res, int normalize) {
    const int R = res * res;
    const int ray = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (ray >= R) return;
    const float* c2w = cam + (int64_t)b * cam_stride;      // 16 floats, row-major 4x4
    const float* Kp = c2w + 16;                             // 9 floats, row-major 3x3
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = Kp[i] * (i < 6 ? (float)res : 1.f);
    // inverse by cofactors
    const float c00 = k[4] * k[8] - k[5] * k[7], c01 = k[5] * k[6] - k[3] * k[8], c02 = k[3] * k[7] - k[4] * k[6];
    const float det = k[0] * c00 + k[1] * c01 + k[2] * c02, id = 1.f / det;
    const float inv[9] = {c00 * id, (k[2] * k[7] - k[1] * k[8]) * id, (k[1] * k[5] - k[2] * k[4]) * id,
                          c01 * id, (k[0] * k[8] - k[2] * k[6]) * id, (k[2] * k[3] - k[0] * k[5]) * id,
                          c02 * id, (k[1] * k[6] - k[0] * k[7]) * id, (k[0] * k[4] - k[1] * k[3]) * id};
    const float px = (float)(ray % res), py = (float)(ray / res);       // integer pixel coordinates, x fastest
    const float dx = inv[0] * px + inv[1] * py + inv[2], dy = inv[3] * px + inv[4] * py + inv[5], dz = inv[6] * px + inv[7] * py + inv[8];
    float wx = c2w[0] * dx + c2w[1] * dy + c2w[2] * dz, wy = c2w[4] * dx + c2w[5] * dy + c2w[6] * dz,
          wz = c2w[8] * dx + c2w[9] * dy + c2w[10] * dz;
    if (normalize) {
        const float n = fmaxf(sqrtf(wx * wx + wy * wy + wz * wz), 1e-12f);    // F.normalize eps
        wx /= n; wy /= n; wz /= n;
    }
    const int64_t o = ((int64_t)b * R + ray) * 3;
    rays_d[o] = wx; rays_d[o + 1] = wy; rays_d[o + 2] = wz;
    rays_o[o] = c2w[3]; rays_o[o + 1] = c2w[7]; rays_o[o + 2] = c2w[11];
}

}  // namespace

extern "C" int ia_ray_sampler(const float* cam, int cam_stride, float* rays_o, float* rays_d, int B, int resolution, int normalize,
                              void* stream) {
    IA_REQUIRE(cam && rays_o && rays_d, "null pointer argument");
    IA_REQUIRE(B > 0 && resolution > 0 && cam_stride >= 25, "bad dimensions");
    dim3 grid((resolution * resolution + 255) / 256, B);
    hipLaunchKernelGGL(ray_sampler_kernel, grid, dim3(256), 0, (hipStream_t)stream, cam, cam_stride, rays_o, rays_d, B, resolution, normalize);
    return ia::check_launch("ia_ray_sampler");
}

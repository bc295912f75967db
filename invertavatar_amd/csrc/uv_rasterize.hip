// ia_uv_rasterize: driver-side UV rasteriser -- the step immediately BEFORE the generator path.
//
// Replaces, for one drive frame, Faceverse_manager.make_driven_rendering from the rasteriser call on
// (data_preprocess/FaceVerse/renderer.py:66-82): pytorch3d MeshRasterizer (orthographic camera, 512^2, faces_per_pixel = 1,
// blur_radius = 1e-6; ortho_renderer.py:56-83) -> render_after_rasterize (volumetric_rendering/renderer.py:556-571: barycentric
// interpolation of the per-face (u, v, mask) attributes + visibility channel) -> rendering *= vis * mask -> crop
// [left 128, top 114, 256 x 256] -> channels (u, v, mask) in HWC with the mask binarised at 0.5.  Output = uvcoords_image, the
// mesh condition TriPlaneGenerator.synthesis consumes.  pytorch3d is a CUDA-only third-party dependency that is not in
// /root/reference: its published naive rasteriser (pytorch3d/renderer/mesh/rasterize_meshes.py, rasterize_meshes_python) is
// restated: NDC pixel centres with +X left / +Y up, face kept for a pixel when the centre is inside or closer than
// sqrt(blur_radius) to the triangle, nearest z wins (ties: lower face index), barycentrics clipped to [0, 1] and renormalised
// because blur_radius > 0.  Parity is pinned by this repository's own CPU restatement (oracle/uv_rasterize.py).
//
// Three launches: clear the packed (z | face) buffer of the crop window, one thread per triangle splatting into it with 64-bit
// atomicMin, one thread per pixel resolving attributes.  Inputs are the vertices AFTER batch_orth_proj and the z flip
// (renderer.py:62-64), i.e. what the reference hands to Meshes().
#include "ia_common.h"

namespace {

struct UvGeo {
    int B, V, F, S;
    int cx, cy, cw, ch;     // crop window inside the S x S raster (left, top, width, height)
    float blur;             // squared NDC distance (pytorch3d's blur_radius)
};

constexpr float kEps = 1e-8f;

__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

__device__ __forceinline__ float seg_dist2(float px, float py, float ax, float ay, float bx, float by) {
    const float dx = bx - ax, dy = by - ay, l2 = dx * dx + dy * dy;
    if (l2 <= kEps) return (px - bx) * (px - bx) + (py - by) * (py - by);
    float t = ((px - ax) * dx + (py - ay) * dy) / l2;
    t = fminf(fmaxf(t, 0.f), 1.f);
    const float qx = ax + t * dx - px, qy = ay + t * dy - py;
    return qx * qx + qy * qy;
}

// NDC of pixel centre i of an S-pixel axis; index 0 is the +1 side (pytorch3d: +X left, +Y up)
__device__ __forceinline__ float pix_ndc(int i, int S) { return 1.f - (2.f * i + 1.f) / S; }

struct Hit { bool ok; float w0, w1, w2, z; };

__device__ __forceinline__ Hit test_pixel(float px, float py, const float (&v)[3][3], float area, float blur) {
    Hit h; h.ok = false;
    float w0 = edge_fn(px, py, v[1][0], v[1][1], v[2][0], v[2][1]) / (area + kEps);
    float w1 = edge_fn(px, py, v[2][0], v[2][1], v[0][0], v[0][1]) / (area + kEps);
    float w2 = edge_fn(px, py, v[0][0], v[0][1], v[1][0], v[1][1]) / (area + kEps);
    const bool inside = w0 > 0.f && w1 > 0.f && w2 > 0.f;
    if (!inside) {
        const float d = fminf(seg_dist2(px, py, v[0][0], v[0][1], v[1][0], v[1][1]),
                              fminf(seg_dist2(px, py, v[1][0], v[1][1], v[2][0], v[2][1]), seg_dist2(px, py, v[2][0], v[2][1], v[0][0], v[0][1])));
        if (d >= blur) return h;
    }
    // blur_radius > 0: clip_barycentric_coords
    w0 = fminf(fmaxf(w0, 0.f), 1.f); w1 = fminf(fmaxf(w1, 0.f), 1.f); w2 = fminf(fmaxf(w2, 0.f), 1.f);
    const float s = fmaxf(w0 + w1 + w2, 1e-5f);
    w0 /= s; w1 /= s; w2 /= s;
    const float z = w0 * v[0][2] + w1 * v[1][2] + w2 * v[2][2];
    if (z < 0.f) return h;
    h.ok = true; h.w0 = w0; h.w1 = w1; h.w2 = w2; h.z = z;
    return h;
}

__device__ __forceinline__ void load_face(const float* verts, const int* tris, int b, int f, const UvGeo& g, float (&v)[3][3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float* p = verts + ((int64_t)b * g.V + tris[f * 3 + k]) * 3;
        // the camera of get_renderer(orthoCam=True, K=[-1,-1,0,0], T=[0,0,10]): ndc = (-x, -y), view depth = z + 10
        v[k][0] = -p[0]; v[k][1] = -p[1]; v[k][2] = p[2] + 10.f;
    }
}

__global__ __launch_bounds__(256) void uv_clear_kernel(unsigned long long* zbuf, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) zbuf[i] = ~0ull;
}

__global__ __launch_bounds__(256) void uv_splat_kernel(const float* __restrict__ verts, const int* __restrict__ tris, unsigned long long* __restrict__ zbuf, UvGeo g) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)g.B * g.F) return;
    const int b = (int)(i / g.F), f = (int)(i - (int64_t)b * g.F);
    float v[3][3];
    load_face(verts, tris, b, f, g, v);
    const float area = edge_fn(v[2][0], v[2][1], v[0][0], v[0][1], v[1][0], v[1][1]);
    if (fabsf(area) <= kEps) return;                                           // zero-area faces are skipped
    if (fmaxf(v[0][2], fmaxf(v[1][2], v[2][2])) < 0.f) return;                // entirely behind the camera
    const float r = sqrtf(g.blur);
    const float xmin = fminf(v[0][0], fminf(v[1][0], v[2][0])) - r, xmax = fmaxf(v[0][0], fmaxf(v[1][0], v[2][0])) + r;
    const float ymin = fminf(v[0][1], fminf(v[1][1], v[2][1])) - r, ymax = fmaxf(v[0][1], fmaxf(v[1][1], v[2][1])) + r;
    // pixel index i has ndc 1 - (2i+1)/S: larger ndc = smaller index
    int c0 = (int)floorf(((1.f - xmax) * g.S - 1.f) * 0.5f), c1 = (int)ceilf(((1.f - xmin) * g.S - 1.f) * 0.5f);
    int r0 = (int)floorf(((1.f - ymax) * g.S - 1.f) * 0.5f), r1 = (int)ceilf(((1.f - ymin) * g.S - 1.f) * 0.5f);
    c0 = max(c0, g.cx); c1 = min(c1, g.cx + g.cw - 1); r0 = max(r0, g.cy); r1 = min(r1, g.cy + g.ch - 1);
    for (int yy = r0; yy <= r1; ++yy)
        for (int xx = c0; xx <= c1; ++xx) {
            const Hit h = test_pixel(pix_ndc(xx, g.S), pix_ndc(yy, g.S), v, area, g.blur);
            if (!h.ok) continue;
            const unsigned long long key = ((unsigned long long)__float_as_uint(h.z) << 32) | (unsigned)f;   // z >= 0: bit order = value order
            atomicMin(&zbuf[((int64_t)b * g.ch + (yy - g.cy)) * g.cw + (xx - g.cx)], key);
        }
}

__global__ __launch_bounds__(256) void uv_resolve_kernel(const float* __restrict__ verts, const int* __restrict__ tris, const float* __restrict__ attrs,
                                                        const unsigned long long* __restrict__ zbuf, float* __restrict__ out, UvGeo g, int binarize) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)g.B * g.ch * g.cw) return;
    const int xx = (int)(i % g.cw), yy = (int)((i / g.cw) % g.ch), b = (int)(i / ((int64_t)g.cw * g.ch));
    const unsigned long long key = zbuf[i];
    float u = 0.f, vv = 0.f, m = 0.f;
    if (key != ~0ull) {
        const int f = (int)(key & 0xffffffffu);
        float v[3][3];
        load_face(verts, tris, b, f, g, v);
        const float area = edge_fn(v[2][0], v[2][1], v[0][0], v[0][1], v[1][0], v[1][1]);
        const Hit h = test_pixel(pix_ndc(g.cx + xx, g.S), pix_ndc(g.cy + yy, g.S), v, area, g.blur);
        const float* a = attrs + (int64_t)f * 9;                       // [3 vertices][u, v, mask]
        const float au = h.w0 * a[0] + h.w1 * a[3] + h.w2 * a[6];
        const float av = h.w0 * a[1] + h.w1 * a[4] + h.w2 * a[7];
        const float am = h.w0 * a[2] + h.w1 * a[5] + h.w2 * a[8];
        const float rm = 1.f * am;                                     // render_mask = vis * face_mask (:72-73)
        u = au * rm; vv = av * rm; m = am * rm;
    }
    float* o = out + i * 3;
    o[0] = u; o[1] = vv; o[2] = !binarize ? m : (m < 0.5f ? 0.f : 1.f);    // (:82; a caller that resizes first thresholds afterwards, :78-82)
}

}  // namespace

extern "C" int ia_uv_rasterize(const float* verts, const int* tris, const float* face_attrs, void* zbuf_scratch, float* uvcoords_image,
                               int B, int V, int F, int raster_size, int crop_left, int crop_top, int crop_w, int crop_h, float blur_radius,
                               int binarize_mask, void* stream) {
    IA_REQUIRE(verts && tris && face_attrs && zbuf_scratch && uvcoords_image, "null pointer argument");
    IA_REQUIRE(B > 0 && V > 0 && F > 0 && raster_size > 0, "empty input");
    IA_REQUIRE(crop_left >= 0 && crop_top >= 0 && crop_w > 0 && crop_h > 0 && crop_left + crop_w <= raster_size && crop_top + crop_h <= raster_size,
               "crop window [%d,%d,%d,%d] does not fit a %d^2 raster", crop_left, crop_top, crop_w, crop_h, raster_size);
    IA_REQUIRE(blur_radius >= 0.f, "blur_radius must be non-negative");
    UvGeo g{B, V, F, raster_size, crop_left, crop_top, crop_w, crop_h, blur_radius};
    hipStream_t s = (hipStream_t)stream;
    const int64_t npix = (int64_t)B * crop_w * crop_h, nface = (int64_t)B * F;
    auto* zb = static_cast<unsigned long long*>(zbuf_scratch);
    hipLaunchKernelGGL(uv_clear_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, zb, npix);
    hipLaunchKernelGGL(uv_splat_kernel, dim3((unsigned)((nface + 255) / 256)), dim3(256), 0, s, verts, tris, zb, g);
    hipLaunchKernelGGL(uv_resolve_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, verts, tris, face_attrs, zb, uvcoords_image, g, binarize_mask);
    return ia::check_launch("ia_uv_rasterize");
}

// ia_rasterize_level / ia_blend_planes: the glue between the three StyleGAN2 backbones and the renderer.
//
// ia_rasterize_level -- one pyramid level of TriPlaneGenerator.rasterize
// (training_avatar_texture/triplane_v20.py:328-337) in ONE pass:
//     rend = AA_resize( grid_sample(texture_k, uv) , 256 -> res )          (F.grid_sample + F.interpolate(antialias=True))
//     a    = AA_resize( alpha, 256 -> res )
//     s    = AA_resize( static_k[:, :, bbox], bbox -> res )                 (2x up-sampling of the face crop)
//     out[:, :C] = rend * a + s * (1 - a);   out[:, C] = AA_resize( upper_mouth_alpha, 256 -> res )
// The reference materialises the 256^2 x C grid_sample result (134 MB at C = 512) and three resized tensors per
// level.  Here a workgroup owns 4 horizontally adjacent output pixels: it stages the source pixels of their anti-aliasing
// footprint once (bilinear cell, fractions, AA weights; LDS), MERGES their weights per texel (a level's texture has the
// resolution of its output, so the 40 .. 640 source pixels land on about 18 texels; see "Merge windows" below) and then
// every thread (= four channels of the CHANNELS-LAST texture, so each gather is a coalesced line per wave) walks the
// merged list.  Weights follow aten's _upsample_bilinear2d_aa and grid_sampler_2d (C4) in fp32; the merge
// changes the summation order only (per-texel weight sums first), results stay within 2e-5 of aten's (tests).
//
// ia_blend_planes -- the plane blend of triplane_v20.py:119-128 fused with the layout change the renderer wants:
// AA-resize the face stitch + alpha to 128^2, paste into the bbox of plane 0, blend over the static planes and
// write the three planes channels-last [B,3,256,256,32].
#include "ia_common.h"

// Compile-time ablations (tools/, never in the product build): bit 0 = no static-crop loads, bit 1 = no texture walk
#ifndef IA_RAST_ABLATE
#define IA_RAST_ABLATE 0
#endif

namespace {

constexpr int kSrc = 256;                  // UV / alpha maps are 256 x 256 (triplane_v20.py:114,322)
constexpr int PXB = 4;                     // output pixels per workgroup (one float4 store per channel)
constexpr int kMaxScale = 8;

// aten's anti-aliased triangle filter along one axis: taps [lo, hi) and their normalised weights for output o.
__device__ __forceinline__ void aa_taps(int o, int n_in, int n_out, int& lo, int& hi, float& center, float& inv, float& total) {
    const float scale = (float)n_in / (float)n_out;
    const float support = scale >= 1.f ? scale : 1.f;
    inv = scale >= 1.f ? 1.f / scale : 1.f;
    center = scale * ((float)o + 0.5f);
    lo = max((int)(center - support + 0.5f), 0);
    hi = min((int)(center + support + 0.5f), n_in);
    total = 0.f;
    for (int j = lo; j < hi; ++j) total += fmaxf(0.f, 1.f - fabsf(((float)j - center + 0.5f) * inv));
}
__device__ __forceinline__ float aa_weight(int j, float center, float inv, float total) {
    return fmaxf(0.f, 1.f - fabsf(((float)j - center + 0.5f) * inv)) / total;
}

struct RastParams {
    const float* tex_cl;      // [B][Rt][Rt][C]  channels-last texture level
    const float* uv;          // [B][256][256][3] (u, v, mask)
    const float* upper;       // [B][256][256]    clamp(mask + upper mouth mask)
    const float* sta;         // static feature level, NCHW view: element (b,c,y,x) at b*sta_bs + c*Rs*Rs + y*Rs + x
    float* out;               // [B][C+1][res][res]
    int64_t sta_bs;
    int B, C, Rt, Rs, res;
    int by0, by1, bx0, bx1;   // crop of the static level that is resized to res
};

// Merge windows: the texels under the anti-aliasing footprint of a workgroup's 4 output pixels.  A texture level has the
// resolution of its output, so the 16 x 40 .. 4 x 10 source pixels of a footprint land on a handful of texels (about 3 x 6
// for a UV map at the scale of the image); their (AA weight x bilinear weight) products are summed PER TEXEL first (64-bit
// fixed-point LDS atomics: integer adds commute, so the sums do not depend on the order the threads arrive in) and the
// channel walk then gathers each texel once.  Two windows, because the footprints along the silhouette of the face see two
// clusters: the face's texels and the one texel the constant background UV points at.  Window A is anchored at the corner of
// the bounding box of all bilinear cells, window B at the corner of the bounding box of the cells A does not hold; when
// cells are left over after that (a seam inside a strongly stretched region) the workgroup takes the direct form.
constexpr int kWinW = 16, kWinH = 8, kWinN = kWinW * kWinH;
constexpr int kChunk = 128, kList = 4 * kChunk;      // direct form: 128 source pixels (x 4 bilinear taps) a pass
constexpr float kFix = 1099511627776.f;              // 2^40
static_assert(kList >= 2 * kWinN, "the merged list must fit the list buffers");

// S = 256 / res (8, 4, 2): sizes the footprint staging buffers
template <int S>
__global__ __launch_bounds__(256) void rasterize_level_kernel(RastParams p) {
    constexpr int kMaxRows = 2 * S, kMaxCols = (PXB + 1) * S;
    __shared__ float4 s_px[kMaxRows * kMaxCols];    // per source pixel: (fx, fy, row AA weight, packed clamped (x0, y0))
    __shared__ float4 s_wx[kMaxCols];               // column AA weight of each source column for the 4 output pixels
    __shared__ float s_alpha[kMaxRows * kMaxCols], s_upper[kMaxRows * kMaxCols];
    __shared__ unsigned long long s_acc[2 * kWinN * PXB];
    __shared__ int s_lidx[kList];                   // walk list: texel offset (pre-multiplied by C) ...
    __shared__ float4 s_lw[kList];                  // ... and its weight for each of the 4 output pixels
    __shared__ int s_bb[8], s_n[1];
    __shared__ float s_a[PXB], s_u[PXB];
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];   // partial sums of the split walk

    const int tid = threadIdx.x, nthr = blockDim.x;
    const int res = p.res;
    // Workgroup -> output tile, XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs, so id % 8 picks the
    // XCD and id / 8 the position inside that XCD's share; each XCD then owns a contiguous band of output rows and its
    // L2 only has to hold the texels under that band (a 128^2 x 256-channel level is 16 MB, a band of it 2 MB).
    const int xtiles = (res + PXB - 1) / PXB, per_img = xtiles * res, nblk = per_img * p.B;
    const int per_xcd = (nblk + 7) / 8;
    const int logical = (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8);
    if (logical >= nblk) return;
    const int b = logical / per_img, rem = logical - b * per_img;
    const int y = rem / xtiles, xt = (rem - y * xtiles) * PXB;

    // ---- AA footprint: rows of output row y, columns of output pixels xt .. xt+3
    int ylo, yhi; float yc, yinv, ytot;
    aa_taps(y, kSrc, res, ylo, yhi, yc, yinv, ytot);
    int xlo[PXB], xhi[PXB]; float xc[PXB], xinv[PXB], xtot[PXB];
#pragma unroll
    for (int k = 0; k < PXB; ++k) aa_taps(min(xt + k, res - 1), kSrc, res, xlo[k], xhi[k], xc[k], xinv[k], xtot[k]);
    const int c0 = xlo[0], ncols = xhi[PXB - 1] - c0, nrows = yhi - ylo, nsrc = nrows * ncols;

    if (tid == 0) { s_bb[0] = s_bb[1] = s_bb[4] = s_bb[5] = INT_MAX; s_bb[2] = s_bb[3] = s_bb[6] = s_bb[7] = INT_MIN; }
    for (int i = tid; i < 2 * kWinN * PXB; i += nthr) s_acc[i] = 0ull;
    for (int i = tid; i < ncols; i += nthr) {
        const int X = c0 + i;
        float w[PXB];
#pragma unroll
        for (int k = 0; k < PXB; ++k) w[k] = (X >= xlo[k] && X < xhi[k] && xt + k < res) ? aa_weight(X, xc[k], xinv[k], xtot[k]) : 0.f;
        s_wx[i] = make_float4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();
    const float* uvb = p.uv + (int64_t)b * kSrc * kSrc * 3;
    const float* upb = p.upper + (int64_t)b * kSrc * kSrc;
    const int Rt = p.Rt;
    // ---- stage the footprint: bilinear cell + fractions of every source pixel, and the bounding box of the texels they touch
    int mnx = INT_MAX, mny = INT_MAX, mxx = INT_MIN, mxy = INT_MIN;
    for (int i = tid; i < nsrc; i += nthr) {
        const int r = i / ncols, cidx = i - r * ncols;
        const int Y = ylo + r, X = c0 + cidx;
        const float wy = aa_weight(Y, yc, yinv, ytot);
        const float* px = uvb + ((int64_t)Y * kSrc + X) * 3;
        const float gx = px[0], gy = px[1];
        // grid_sampler_2d, bilinear, zeros, align_corners = False
        const float ix = (gx + 1.f) * (0.5f * (float)Rt) - 0.5f, iy = (gy + 1.f) * (0.5f * (float)Rt) - 0.5f;
        const float x0f = floorf(ix), y0f = floorf(iy);
        const float fx = ix - x0f, fy = iy - y0f;
        const int x0 = (int)fminf(fmaxf(x0f, -2.f), (float)Rt + 1.f), y0 = (int)fminf(fmaxf(y0f, -2.f), (float)Rt + 1.f);
        s_px[i] = make_float4(fx, fy, wy, __int_as_float(((y0 + 2) << 16) | (x0 + 2)));
        s_alpha[i] = px[2] * wy;
        s_upper[i] = upb[(int64_t)Y * kSrc + X] * wy;
        const int ax = max(x0, 0), bx = min(x0 + 1, Rt - 1), ay = max(y0, 0), by = min(y0 + 1, Rt - 1);
        if (ax <= bx && ay <= by) { mnx = min(mnx, ax); mxx = max(mxx, bx); mny = min(mny, ay); mxy = max(mxy, by); }
    }
    if (mnx <= mxx) { atomicMin(&s_bb[0], mnx); atomicMin(&s_bb[1], mny); atomicMax(&s_bb[2], mxx); atomicMax(&s_bb[3], mxy); }
    __syncthreads();
    // ---- resized alpha / upper-mouth alpha of the 4 output pixels (wave 0)
    if (tid < 64) {
        float a[PXB] = {0.f, 0.f, 0.f, 0.f}, u[PXB] = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < nsrc; i += 64) {
            const float4 wx = s_wx[i % ncols];
            const float av = s_alpha[i], uvv = s_upper[i];
            a[0] = fmaf(av, wx.x, a[0]); a[1] = fmaf(av, wx.y, a[1]); a[2] = fmaf(av, wx.z, a[2]); a[3] = fmaf(av, wx.w, a[3]);
            u[0] = fmaf(uvv, wx.x, u[0]); u[1] = fmaf(uvv, wx.y, u[1]); u[2] = fmaf(uvv, wx.z, u[2]); u[3] = fmaf(uvv, wx.w, u[3]);
        }
#pragma unroll
        for (int k = 0; k < PXB; ++k)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { a[k] += __shfl_xor(a[k], off); u[k] += __shfl_xor(u[k], off); }
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < PXB; ++k) { s_a[k] = a[k]; s_u[k] = u[k]; }
        }
    }
    // ---- merged form: per-texel weight sums over the windows, then an ordered compaction into the walk list
    const int ax0 = s_bb[0], ay0 = s_bb[1];
    const bool none = s_bb[2] < s_bb[0];                                   // every bilinear cell lies in the zero padding
    const bool one = none || (s_bb[2] - ax0 < kWinW && s_bb[3] - ay0 < kWinH);
    // clamped extent of a staged pixel's bilinear cell, and whether window A holds all of it
    auto cell = [&](float packed, int& x0, int& y0) { const int pk = __float_as_int(packed); x0 = (pk & 0xffff) - 2; y0 = (pk >> 16) - 2; };
    auto in_a = [&](int x0, int y0) {
        const int lx = max(x0, 0), hx = min(x0 + 1, Rt - 1), ly = max(y0, 0), hy = min(y0 + 1, Rt - 1);
        return lx > hx || ly > hy || (hx - ax0 < kWinW && hy - ay0 < kWinH);
    };
    if (!one) {                                                            // (uniform) second window: bounding box of what A leaves
        int bnx = INT_MAX, bny = INT_MAX, bxx = INT_MIN, bxy = INT_MIN;
        for (int i = tid; i < nsrc; i += nthr) {
            int x0, y0;
            cell(s_px[i].w, x0, y0);
            if (!in_a(x0, y0)) { bnx = min(bnx, max(x0, 0)); bxx = max(bxx, min(x0 + 1, Rt - 1)); bny = min(bny, max(y0, 0)); bxy = max(bxy, min(y0 + 1, Rt - 1)); }
        }
        if (bnx <= bxx) { atomicMin(&s_bb[4], bnx); atomicMin(&s_bb[5], bny); atomicMax(&s_bb[6], bxx); atomicMax(&s_bb[7], bxy); }
        __syncthreads();
    }
    const int bx0 = s_bb[4], by0 = s_bb[5];
    const bool merged = one || (s_bb[6] - bx0 < kWinW && s_bb[7] - by0 < kWinH);
    int nlist = 0;
    if (merged) {
        if (!none) {
            for (int i = tid; i < nsrc; i += nthr) {
                const float4 q = s_px[i];
                const float4 wxv = s_wx[i % ncols];
                const float wx[PXB] = {wxv.x, wxv.y, wxv.z, wxv.w};
                int x0, y0;
                cell(q.w, x0, y0);
                const bool a = one || in_a(x0, y0);
                unsigned long long* win = s_acc + (a ? 0 : kWinN * PXB);
                const int ox = a ? ax0 : bx0, oy = a ? ay0 : by0;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int xi = x0 + (t & 1), yi = y0 + (t >> 1);
                    if (xi < 0 || xi >= Rt || yi < 0 || yi >= Rt) continue;
                    const float w = ((t & 1) ? q.x : 1.f - q.x) * ((t >> 1) ? q.y : 1.f - q.y) * q.z;
                    unsigned long long* slot = win + ((yi - oy) * kWinW + (xi - ox)) * PXB;
#pragma unroll
                    for (int k = 0; k < PXB; ++k) {
                        const unsigned long long v = (unsigned long long)(w * wx[k] * kFix);
                        if (v) atomicAdd(slot + k, v);
                    }
                }
            }
        }
        __syncthreads();
        if (tid < 64) {                                                    // wave 0: ordered compaction, 64 window slots a step
            int cnt = 0;
            for (int base = 0; base < (one ? kWinN : 2 * kWinN); base += 64) {
                const int e = base + tid;
                const unsigned long long a0 = s_acc[e * PXB], a1 = s_acc[e * PXB + 1], a2 = s_acc[e * PXB + 2], a3 = s_acc[e * PXB + 3];
                const bool nz = (a0 | a1 | a2 | a3) != 0ull;
                const unsigned long long m = __ballot(nz);
                if (nz) {
                    const int pos = cnt + __popcll(m & ((1ull << tid) - 1ull));
                    const int slot = e % kWinN, ox = e < kWinN ? ax0 : bx0, oy = e < kWinN ? ay0 : by0;
                    s_lidx[pos] = ((oy + slot / kWinW) * Rt + ox + slot % kWinW) * p.C;
                    s_lw[pos] = make_float4((float)a0 * (1.f / kFix), (float)a1 * (1.f / kFix), (float)a2 * (1.f / kFix), (float)a3 * (1.f / kFix));
                }
                cnt += __popcll(m);
            }
            if (tid == 0) s_n[0] = cnt;
        }
        __syncthreads();
        nlist = s_n[0];
    }

    // ---- static crop: 2-tap (per axis) AA up-sampling weights, identical for every channel
    const int crop_h = p.by1 - p.by0, crop_w = p.bx1 - p.bx0;
    int sylo, syhi; float syc, syinv, sytot;
    aa_taps(y, crop_h, res, sylo, syhi, syc, syinv, sytot);
    int sxlo[PXB], sxhi[PXB]; float sxc[PXB], sxinv[PXB], sxtot[PXB];
#pragma unroll
    for (int k = 0; k < PXB; ++k) aa_taps(min(xt + k, res - 1), crop_w, res, sxlo[k], sxhi[k], sxc[k], sxinv[k], sxtot[k]);

    const float* texb = p.tex_cl + (int64_t)b * Rt * Rt * p.C;
    const int64_t rr = (int64_t)res * res;
    float* outb = p.out + (int64_t)b * (p.C + 1) * rr + (int64_t)y * res + xt;
    const bool vec_ok = (xt + PXB <= res) && (res % 4 == 0);
    // each thread owns 4 consecutive channels (one 16-byte gather per texel) when C % 4 == 0, else one channel; when the
    // level has fewer channel groups than the workgroup has threads, the threads split the walk list nsplit ways (every
    // nsplit-th entry) and the partial sums are added through LDS in split order.
    const int CV = (p.C % 4 == 0) ? 4 : 1;
    const int ncq = (p.C + CV - 1) / CV;
    const int lanes = min(ncq, nthr);                                      // channel groups walked concurrently
    const int nsplit = min(8, nthr / lanes);
    float* s_red = reinterpret_cast<float*>(s_dyn);                        // [nsplit][lanes][16]

    // add entries first, first + step, ... (< n) of the walk list for channels c .. c+CV-1
    auto walk = [&](int c, int first, int step, int n, float (&acc)[4][PXB]) {
        const float* tc = texb + c;
        if (IA_RAST_ABLATE & 2) return;
        if (CV == 4) {
#pragma unroll 8
            for (int e = first; e < n; e += step) {
                const float4 t = *(const float4*)(tc + s_lidx[e]);
                const float4 w = s_lw[e];
                acc[0][0] = fmaf(t.x, w.x, acc[0][0]); acc[0][1] = fmaf(t.x, w.y, acc[0][1]); acc[0][2] = fmaf(t.x, w.z, acc[0][2]); acc[0][3] = fmaf(t.x, w.w, acc[0][3]);
                acc[1][0] = fmaf(t.y, w.x, acc[1][0]); acc[1][1] = fmaf(t.y, w.y, acc[1][1]); acc[1][2] = fmaf(t.y, w.z, acc[1][2]); acc[1][3] = fmaf(t.y, w.w, acc[1][3]);
                acc[2][0] = fmaf(t.z, w.x, acc[2][0]); acc[2][1] = fmaf(t.z, w.y, acc[2][1]); acc[2][2] = fmaf(t.z, w.z, acc[2][2]); acc[2][3] = fmaf(t.z, w.w, acc[2][3]);
                acc[3][0] = fmaf(t.w, w.x, acc[3][0]); acc[3][1] = fmaf(t.w, w.y, acc[3][1]); acc[3][2] = fmaf(t.w, w.z, acc[3][2]); acc[3][3] = fmaf(t.w, w.w, acc[3][3]);
            }
        } else {
            for (int e = first; e < n; e += step) {
                const float t = tc[s_lidx[e]];
                const float4 w = s_lw[e];
                acc[0][0] = fmaf(t, w.x, acc[0][0]); acc[0][1] = fmaf(t, w.y, acc[0][1]); acc[0][2] = fmaf(t, w.z, acc[0][2]); acc[0][3] = fmaf(t, w.w, acc[0][3]);
            }
        }
    };
    // Static crop: the (<= kSRows x kSCols) source window of the 4 output pixels and its separable AA weights are the same
    // for every channel, so they are tabulated once (wide windows fall back to the direct double loop).
    constexpr int kSRows = 3, kSCols = 6;
    const int s_c0 = sxlo[0], s_nc = sxhi[PXB - 1] - sxlo[0], s_nr = syhi - sylo;
    const bool s_tab = s_nr <= kSRows && s_nc <= kSCols && s_nc > 0;
    float swy[kSRows], swx[PXB][kSCols];
#pragma unroll
    for (int r = 0; r < kSRows; ++r) swy[r] = (s_tab && r < s_nr) ? aa_weight(sylo + r, syc, syinv, sytot) : 0.f;
#pragma unroll
    for (int k = 0; k < PXB; ++k)
#pragma unroll
        for (int i = 0; i < kSCols; ++i) {
            const int X = s_c0 + i;
            swx[k][i] = (s_tab && X >= sxlo[k] && X < sxhi[k]) ? aa_weight(X, sxc[k], sxinv[k], sxtot[k]) : 0.f;
        }
    // blend with the resized static crop and store channels c .. c+CV-1 of the 4 output pixels
    auto finish = [&](int c, const float (&acc)[4][PXB]) {
        for (int j = 0; j < CV; ++j) {
            const float* sc = p.sta + (int64_t)b * p.sta_bs + (int64_t)(c + j) * p.Rs * p.Rs;
            float o[PXB];
            if (s_tab) {
                float sv[PXB] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < kSRows; ++r) {
                    if (r < s_nr) {
                        const float* row = sc + (int64_t)(p.by0 + sylo + r) * p.Rs + p.bx0 + s_c0;
                        float h[PXB] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int i = 0; i < kSCols; ++i) {
                            if (i < s_nc) {
                                const float v = (IA_RAST_ABLATE & 1) ? 0.f : row[i];
#pragma unroll
                                for (int k = 0; k < PXB; ++k) h[k] = fmaf(v, swx[k][i], h[k]);
                            }
                        }
#pragma unroll
                        for (int k = 0; k < PXB; ++k) sv[k] = fmaf(h[k], swy[r], sv[k]);
                    }
                }
#pragma unroll
                for (int k = 0; k < PXB; ++k) { const float a = s_a[k]; o[k] = acc[j][k] * a + sv[k] * (1.f - a); }
            } else {
#pragma unroll
                for (int k = 0; k < PXB; ++k) {
                    float sv = 0.f;
                    for (int jj = sylo; jj < syhi; ++jj) {
                        const float wj = aa_weight(jj, syc, syinv, sytot);
                        for (int i = sxlo[k]; i < sxhi[k]; ++i)
                            sv = fmaf(sc[(int64_t)(p.by0 + jj) * p.Rs + p.bx0 + i], wj * aa_weight(i, sxc[k], sxinv[k], sxtot[k]), sv);
                    }
                    const float a = s_a[k];
                    o[k] = acc[j][k] * a + sv * (1.f - a);
                }
            }
            float* dst = outb + (int64_t)(c + j) * rr;
            if (vec_ok) *(float4*)dst = make_float4(o[0], o[1], o[2], o[3]);
            else for (int k = 0; k < PXB && xt + k < res; ++k) dst[k] = o[k];
        }
    };

    const int sp = tid / lanes, cq = tid - sp * lanes;
    const bool walker = sp < nsplit;
    for (int cbase = 0; cbase < ncq; cbase += lanes) {                     // one pass unless C > 4 * workgroup size
        const int mycq = cbase + cq;
        const bool active = walker && mycq < ncq;
        float acc[4][PXB];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < PXB; ++k) acc[j][k] = 0.f;
        if (merged) {
            if (active) walk(mycq * CV, sp, nsplit, nlist, acc);
        } else {
            // direct form: the footprint touches more texels than the window holds (a seam or the silhouette of the UV map):
            // every (source pixel, bilinear tap) becomes its own list entry, 128 source pixels a pass
            for (int base = 0; base < nsrc; base += kChunk) {
                const int cnt = min(kChunk, nsrc - base);
                __syncthreads();
                for (int i = tid; i < cnt; i += nthr) {
                    const float4 q = s_px[base + i];
                    const float4 wxv = s_wx[(base + i) % ncols];
                    const int pk = __float_as_int(q.w), x0 = (pk & 0xffff) - 2, y0 = (pk >> 16) - 2;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int xi = x0 + (t & 1), yi = y0 + (t >> 1);
                        const bool ok = xi >= 0 && xi < Rt && yi >= 0 && yi < Rt;
                        const float w = ok ? ((t & 1) ? q.x : 1.f - q.x) * ((t >> 1) ? q.y : 1.f - q.y) * q.z : 0.f;
                        s_lidx[4 * i + t] = ok ? (yi * Rt + xi) * p.C : 0;
                        s_lw[4 * i + t] = make_float4(w * wxv.x, w * wxv.y, w * wxv.z, w * wxv.w);
                    }
                }
                __syncthreads();
                if (active) walk(mycq * CV, sp, nsplit, 4 * cnt, acc);
            }
        }
        if (nsplit > 1) {
            __syncthreads();                                               // s_red of the previous pass has been read
            if (active) {
                float* mine = s_red + ((int64_t)sp * lanes + cq) * 16;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < PXB; ++k) mine[j * PXB + k] = acc[j][k];
            }
            __syncthreads();
            if (active && sp == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < PXB; ++k) acc[j][k] = 0.f;
                for (int q = 0; q < nsplit; ++q) {
                    const float* part = s_red + ((int64_t)q * lanes + cq) * 16;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int k = 0; k < PXB; ++k) acc[j][k] += part[j * PXB + k];
                }
                finish(mycq * CV, acc);
            }
        } else if (active) {
            finish(mycq * CV, acc);
        }
    }
    if (tid < PXB && xt + tid < res) outb[(int64_t)p.C * rr + tid] = s_u[tid];
}

struct BlendParams {
    const float* stitch;     // [B][32][256][256]
    const float* alpha;      // [B][256][256]      full (mouth-filled) alpha
    const float* sta;        // static planes: element (b, plane, c, y, x) at b*sta_bs + (plane*32 + c)*65536 + y*256 + x
    float* planes_cl;        // [B][3][256][256][32]
    int64_t sta_bs;
    int B, y0, y1, x0, x1;   // bbox of the 128^2 paste (triplane_v20.py:114)
};

// One thread per (b, plane, channel group of 8, y, x).  Plane 0 inside the bbox: AA 256->128 resize of stitch and alpha (16 taps per
// channel); four channel groups per pixel so that the 128^2 bbox pixels -- 16x the work of the others -- are spread over four times
// as many workgroups (with 32 channels per thread the bbox workgroups alone set the kernel's duration: 40 us for 58 MB).
constexpr int kBlendCh = 8;
__global__ __launch_bounds__(256) void blend_planes_kernel(BlendParams p) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int cg = blockIdx.z % 4, plane = (blockIdx.z / 4) % 3, b = blockIdx.z / 12;
    const int c0 = cg * kBlendCh;
    const float* st = p.sta + (int64_t)b * p.sta_bs + ((int64_t)plane * 32 + c0) * 65536 + (int64_t)y * 256 + x;
    float v[kBlendCh];
#pragma unroll
    for (int c = 0; c < kBlendCh; ++c) v[c] = st[(int64_t)c * 65536];
    if (plane == 0 && y >= p.y0 && y < p.y1 && x >= p.x0 && x < p.x1) {
        const int oy = y - p.y0, ox = x - p.x0, n_out = p.y1 - p.y0;       // 128
        int ylo, yhi, xlo, xhi; float yc, yinv, ytot, xc, xinv, xtot;
        aa_taps(oy, 256, n_out, ylo, yhi, yc, yinv, ytot);
        aa_taps(ox, 256, n_out, xlo, xhi, xc, xinv, xtot);
        float a = 0.f, s[kBlendCh];
#pragma unroll
        for (int c = 0; c < kBlendCh; ++c) s[c] = 0.f;
        const float* sb = p.stitch + ((int64_t)b * 32 + c0) * 65536;
        const float* ab = p.alpha + (int64_t)b * 65536;
        // aten resizes separably (rows of the horizontal pass feed the vertical pass); the two-pass order is kept
        for (int j = ylo; j < yhi; ++j) {
            const float wy = aa_weight(j, yc, yinv, ytot);
            float ra = 0.f, rs[kBlendCh];
#pragma unroll
            for (int c = 0; c < kBlendCh; ++c) rs[c] = 0.f;
            for (int i = xlo; i < xhi; ++i) {
                const float wx = aa_weight(i, xc, xinv, xtot);
                ra = fmaf(ab[j * 256 + i], wx, ra);
#pragma unroll
                for (int c = 0; c < kBlendCh; ++c) rs[c] = fmaf(sb[(int64_t)c * 65536 + j * 256 + i], wx, rs[c]);
            }
            a = fmaf(ra, wy, a);
#pragma unroll
            for (int c = 0; c < kBlendCh; ++c) s[c] = fmaf(rs[c], wy, s[c]);
        }
#pragma unroll
        for (int c = 0; c < kBlendCh; ++c) v[c] = s[c] * a + v[c] * (1.f - a);
    }
    float4* dst = (float4*)(p.planes_cl + ((((int64_t)b * 3 + plane) * 256 + y) * 256 + x) * 32 + c0);
#pragma unroll
    for (int c = 0; c < kBlendCh / 4; ++c) dst[c] = make_float4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
}

}  // namespace

extern "C" int ia_rasterize_level(const float* tex_cl, const float* uv, const float* upper_alpha, const float* sta, int64_t sta_batch_stride,
                                  float* out, int B, int C, int tex_res, int sta_res, int res, int by0, int by1, int bx0, int bx1,
                                  void* stream) {
    IA_REQUIRE(tex_cl && uv && upper_alpha && sta && out, "null pointer argument");
    IA_REQUIRE(B > 0 && C > 0 && tex_res > 0 && sta_res > 0 && res > 0, "empty tensor");
    IA_REQUIRE(by0 >= 0 && by1 > by0 && by1 <= sta_res && bx0 >= 0 && bx1 > bx0 && bx1 <= sta_res, "bbox outside the static level");
    if (res > kSrc || kSrc % res != 0 || kSrc / res > kMaxScale || kSrc / res < 2)
        return ia::fail(IA_ERR_UNSUPPORTED, "rasterize level resolution %d: supported are 32, 64, 128 (256 -> res by 8, 4, 2)", res);
    RastParams p{tex_cl, uv, upper_alpha, sta, out, sta_batch_stride, B, C, tex_res, sta_res, res, by0, by1, bx0, bx1};
    const int nblk = ((res + PXB - 1) / PXB) * res * B;
    dim3 grid(((nblk + 7) / 8) * 8);
    // one thread per group of 4 channels (64 .. 256 threads); spare threads split the walk list up to 8 ways
    const int ncq = C % 4 == 0 ? C / 4 : C;
    const int threads = ncq <= 64 ? 64 : ncq <= 128 ? 128 : 256;
    const int lanes = ncq < threads ? ncq : threads;
    const int nsplit = threads / lanes < 8 ? threads / lanes : 8;
    const size_t red_bytes = nsplit > 1 ? (size_t)nsplit * lanes * 16 * sizeof(float) : 0;
    const hipStream_t s = (hipStream_t)stream;
    switch (kSrc / res) {
        case 8: hipLaunchKernelGGL(rasterize_level_kernel<8>, grid, dim3(threads), red_bytes, s, p); break;
        case 4: hipLaunchKernelGGL(rasterize_level_kernel<4>, grid, dim3(threads), red_bytes, s, p); break;
        default: hipLaunchKernelGGL(rasterize_level_kernel<2>, grid, dim3(threads), red_bytes, s, p); break;
    }
    return ia::check_launch("ia_rasterize_level");
}

// y[b,c] = cond[b,c] * a + x[b,c] * (1 - a), a = cond[b,C]: the paste of a rasterised condition over the features / skip image
// (networks_stylegan2_new.py:537-540), same operation order as the reference's four elementwise kernels.
__global__ __launch_bounds__(256) void cond_blend_kernel(const float* __restrict__ cond, const float* __restrict__ x, float* __restrict__ y,
                                                         int C, int64_t hw4, int64_t total4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int64_t pix4 = i % hw4, bc = i / hw4, b = bc / C, c = bc - b * C;
    const float4 cv = reinterpret_cast<const float4*>(cond)[(b * (C + 1) + c) * hw4 + pix4];
    const float4 av = reinterpret_cast<const float4*>(cond)[(b * (C + 1) + C) * hw4 + pix4];
    const float4 xv = reinterpret_cast<const float4*>(x)[i];
    float4 o;
    o.x = cv.x * av.x + xv.x * (1.f - av.x); o.y = cv.y * av.y + xv.y * (1.f - av.y);
    o.z = cv.z * av.z + xv.z * (1.f - av.z); o.w = cv.w * av.w + xv.w * (1.f - av.w);
    reinterpret_cast<float4*>(y)[i] = o;
}

// The same blend for a feature map whose only consumer is the next block's first convolution (networks_stylegan2_new.py:539-540
// then :448 of the next block): the result is written in SPLIT format (ia_act_split), multiplied by that layer's styles, and the
// fp32 tensor is never materialised.  One thread = one pixel of one 8-channel group.
typedef _Float16 h16x8_cb __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void cond_blend_split_kernel(const float* __restrict__ cond, const float* __restrict__ x,
                                                               const float* __restrict__ styles_next, h16x8_cb* __restrict__ ys, int B, int C, int64_t hw) {
    const int C8 = C / 8;
    const int64_t total = (int64_t)B * C8 * hw, stride = (int64_t)gridDim.x * blockDim.x;
    ia::SatWatch watch;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t pix = i % hw;
        const int c8 = (int)((i / hw) % C8), b = (int)(i / (hw * C8));
        const float a = cond[((int64_t)b * (C + 1) + C) * hw + pix];
        h16x8_cb hi, lo;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            const int c = c8 * 8 + cc;
            const float cv = cond[((int64_t)b * (C + 1) + c) * hw + pix], xv = x[((int64_t)b * C + c) * hw + pix];
            float v = cv * a + xv * (1.f - a);
            if (styles_next) v *= styles_next[b * C + c];
            _Float16 h, l;
            ia::split_f16(v, h, l, watch);
            hi[cc] = h; lo[cc] = l;
        }
        ys[((int64_t)(b * 2) * C8 + c8) * hw + pix] = hi;
        ys[((int64_t)(b * 2 + 1) * C8 + c8) * hw + pix] = lo;
    }
    watch.report();
}

// The same, four consecutive pixels per thread (H*W % 4 == 0): 16-byte loads of both inputs -- a quarter of the load instructions of
// the one-pixel form, which at 1.5 TB/s was bound by issuing 4-byte loads, not by HBM.
__global__ __launch_bounds__(256) void cond_blend_split4_kernel(const float* __restrict__ cond, const float* __restrict__ x,
                                                                const float* __restrict__ styles_next, h16x8_cb* __restrict__ ys, int B, int C, int64_t hw) {
    const int C8 = C / 8;
    const int64_t hw4 = hw / 4, total = (int64_t)B * C8 * hw4, stride = (int64_t)gridDim.x * blockDim.x;
    ia::SatWatch watch;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t p4 = i % hw4;
        const int c8 = (int)((i / hw4) % C8), b = (int)(i / (hw4 * C8));
        const float4 a4 = reinterpret_cast<const float4*>(cond + ((int64_t)b * (C + 1) + C) * hw)[p4];
        const float a[4] = {a4.x, a4.y, a4.z, a4.w};
        float4 cv[8], xv[8];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            const int c = c8 * 8 + cc;
            cv[cc] = reinterpret_cast<const float4*>(cond + ((int64_t)b * (C + 1) + c) * hw)[p4];
            xv[cc] = reinterpret_cast<const float4*>(x + ((int64_t)b * C + c) * hw)[p4];
        }
        h16x8_cb hi[4], lo[4];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            const float sn = styles_next ? styles_next[b * C + c8 * 8 + cc] : 1.f;
            const float c4[4] = {cv[cc].x, cv[cc].y, cv[cc].z, cv[cc].w}, x4[4] = {xv[cc].x, xv[cc].y, xv[cc].z, xv[cc].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v = c4[k] * a[k] + x4[k] * (1.f - a[k]);
                if (styles_next) v *= sn;
                _Float16 h, l;
                ia::split_f16(v, h, l, watch);
                hi[k][cc] = h; lo[k][cc] = l;
            }
        }
        h16x8_cb* dh = ys + ((int64_t)(b * 2) * C8 + c8) * hw + 4 * p4;
        h16x8_cb* dl = ys + ((int64_t)(b * 2 + 1) * C8 + c8) * hw + 4 * p4;
#pragma unroll
        for (int k = 0; k < 4; ++k) { dh[k] = hi[k]; dl[k] = lo[k]; }
    }
    watch.report();
}

extern "C" int ia_cond_blend_split(const float* cond, const float* x, const float* styles_next, void* ys, int B, int C, int H, int W, void* stream) {
    IA_REQUIRE(cond && x && ys, "null pointer argument");
    IA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "empty tensor");
    IA_REQUIRE(C % 8 == 0, "the split format stores channels in groups of 8 (C = %d)", C);
    IA_REQUIRE((int64_t)B * (C + 1) * H * W <= INT32_MAX, "tensor is too large");
    const int64_t work = (int64_t)B * (C / 8) * H * W;
    if (((int64_t)H * W) % 4 == 0 && ((reinterpret_cast<uintptr_t>(cond) | reinterpret_cast<uintptr_t>(x)) & 15) == 0)
        hipLaunchKernelGGL(cond_blend_split4_kernel, dim3(ia::streaming_grid(work / 4, 256)), dim3(256), 0, (hipStream_t)stream, cond, x, styles_next,
                           static_cast<h16x8_cb*>(ys), B, C, (int64_t)H * W);
    else
        hipLaunchKernelGGL(cond_blend_split_kernel, dim3(ia::streaming_grid(work, 256)), dim3(256), 0, (hipStream_t)stream, cond, x, styles_next,
                           static_cast<h16x8_cb*>(ys), B, C, (int64_t)H * W);
    return ia::check_launch("ia_cond_blend_split");
}

// [B][C][HW] -> [B][HW][C]: 64 channels x 64 pixels per workgroup through LDS (reads and writes in 256-byte runs)
__global__ __launch_bounds__(256) void channels_last_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int64_t HW) {
    __shared__ float tile[64][65];
    const int t = threadIdx.x, lo = t & 63, hi4 = t >> 6;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64, b = blockIdx.z;
    const float* xb = x + (int64_t)b * C * HW;
    float* yb = y + (int64_t)b * C * HW;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int ch = hi4 + 4 * j;
        tile[ch][lo] = (c0 + ch < C && p0 + lo < HW) ? xb[(int64_t)(c0 + ch) * HW + p0 + lo] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int px = hi4 + 4 * j;
        if (c0 + lo < C && p0 + px < HW) yb[(p0 + px) * C + c0 + lo] = tile[lo][px];
    }
}

extern "C" int ia_channels_last(const float* x, float* y, int B, int C, int H, int W, void* stream) {
    IA_REQUIRE(x && y, "null pointer argument");
    IA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && B <= 65535, "empty tensor");
    const int64_t HW = (int64_t)H * W;
    hipLaunchKernelGGL(channels_last_kernel, dim3((unsigned)((HW + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       x, y, C, HW);
    return ia::check_launch("ia_channels_last");
}

extern "C" int ia_cond_blend(const float* cond, const float* x, float* y, int B, int C, int H, int W, void* stream) {
    IA_REQUIRE(cond && x && y, "null pointer argument");
    IA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "empty tensor");
    IA_REQUIRE(((int64_t)H * W) % 4 == 0, "H*W must be a multiple of 4");
    const int64_t hw4 = (int64_t)H * W / 4, total4 = (int64_t)B * C * hw4;
    hipLaunchKernelGGL(cond_blend_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cond, x, y, C, hw4, total4);
    return ia::check_launch("ia_cond_blend");
}

extern "C" int ia_blend_planes(const float* stitch, const float* full_alpha, const float* static_planes, int64_t sta_batch_stride,
                               float* planes_cl, int B, int y0, int y1, int x0, int x1, void* stream) {
    IA_REQUIRE(stitch && full_alpha && static_planes && planes_cl, "null pointer argument");
    IA_REQUIRE(B > 0, "empty tensor");
    IA_REQUIRE(y0 >= 0 && y1 <= 256 && x0 >= 0 && x1 <= 256 && y1 - y0 == x1 - x0 && y1 > y0, "bbox must be a square inside 256^2");
    BlendParams p{stitch, full_alpha, static_planes, planes_cl, sta_batch_stride, B, y0, y1, x0, x1};
    hipLaunchKernelGGL(blend_planes_kernel, dim3(4, 64, 12 * B), dim3(256), 0, (hipStream_t)stream, p);
    return ia::check_launch("ia_blend_planes");
}

// ia_conv1x1: the ToRGB layer -- a modulated 1x1 convolution without demodulation, + bias, clamp, + skip image
// (training/networks_stylegan2.py:340-362 ToRGBLayer.forward; the skip add of SynthesisBlock.forward :457) -- as a
// STREAMING kernel.  A ToRGB layer reads C_in x H x W floats once and writes 3 .. 96 channels: 33 MB at 256^2 x 128
// channels for 0.4 - 1.6 GFLOP, so HBM (or, for 96 output channels, the fp32 MFMA) bounds it, not the tile machinery of the
// 3x3 kernels: no LDS staging, no stream-K slabs, no fix-up launch.
//
// One wave owns 32*V consecutive pixels (V = 4 or 2) and one block of 32 output channels (blockIdx.z; layers with 33 .. 96
// output channels read their activations up to three times, from L2 after the first -- one wave holding all 96 channels
// runs at one wave per SIMD and measured 75 us at 256^2 x 128 -> 96 where three independent blocks take a third).  v_mfma_f32_32x32x2_f32
// with A = weights [32 channels out x 2 channels in], B = activations [2 channels in x 32 pixels]: lane l loads V
// consecutive pixels of input channel k + l/32 as one 16- or 8-byte load (a wave reads two 128*V-byte runs) and feeds
// component j to MFMA j, so MFMA j computes pixels {V*n + j}; in the accumulators a lane then holds V CONSECUTIVE pixels
// of an output channel and stores them as one vector.  Weights come from the packed [C_in][C_out] array (coalesced 128-byte
// runs, L2-resident) and are multiplied by the styles in registers, (w * s) * x like the reference's fused_modconv path.
// KS waves of a workgroup split the input channels (small images have too few pixel tiles to fill the machine) and add
// their accumulators through LDS in wave order.
#include "ia_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct C1Params {
    const float* x;          // [B][I][P]
    const float* wk;         // [I][O]
    const float* styles;     // [B][I] or null
    const float* bias;       // [O] or null
    const float* residual;   // [B][O][P] or null (added after the clamp)
    float* y;                // [B][O][P]
    int I, O;
    int64_t P;
    float clamp;             // < 0: none
};

template <int V> struct PixVec;
template <> struct PixVec<4> { using type = float4; };
template <> struct PixVec<2> { using type = float2; };

template <int V> __device__ __forceinline__ float comp(const typename PixVec<V>::type& v, int j);
template <> __device__ __forceinline__ float comp<4>(const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }
template <> __device__ __forceinline__ float comp<2>(const float2& v, int j) { return j == 0 ? v.x : v.y; }

template <int V, int KS>
__global__ __launch_bounds__(64 * KS) void conv1x1_kernel(C1Params p) {
    using vec = typename PixVec<V>::type;
    constexpr int NB = 1;
    constexpr int U = 8;                                       // k-steps (channel pairs) in flight per wave
    __shared__ float s_red[KS > 1 ? NB * V * 16 * 64 : 1];
    const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.y, o0 = blockIdx.z * 32;
    const int64_t pix = (int64_t)blockIdx.x * (32 * V) + V * l31;
    const bool pvalid = pix < p.P;
    const int kw = p.I / KS, k_begin = ks * kw, k_end = k_begin + kw;     // this wave's input channels
    const float* xp = p.x + ((int64_t)b * p.I + half) * p.P + (pvalid ? pix : 0);
    const float* sp = p.styles ? p.styles + (int64_t)b * p.I + half : nullptr;
    const float* wp = p.wk + (int64_t)half * p.O + o0 + l31;
    bool ovalid[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) ovalid[blk] = o0 + blk * 32 + l31 < p.O;

    f32x16 acc[NB][V];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int j = 0; j < V; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[blk][j][r] = 0.f;

    vec xn[U];
    float wn[U][NB], sn[U];
    auto load_block = [&](int k0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 2 * u;
            xn[u] = pvalid ? *reinterpret_cast<const vec*>(xp + (int64_t)k * p.P) : vec{};
            sn[u] = sp ? sp[k] : 1.f;
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) wn[u][blk] = ovalid[blk] ? wp[(int64_t)k * p.O + blk * 32] : 0.f;
        }
    };
    load_block(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += 2 * U) {
        vec xc[U];
        float wc[U][NB];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            xc[u] = xn[u];
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) wc[u][blk] = wn[u][blk] * sn[u];
        }
        if (k0 + 2 * U < k_end) load_block(k0 + 2 * U);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int blk = 0; blk < NB; ++blk)
#pragma unroll
                for (int j = 0; j < V; ++j)
                    acc[blk][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[u][blk], comp<V>(xc[u], j), acc[blk][j], 0, 0, 0);
    }

    if constexpr (KS > 1) {       // waves 1 .. KS-1 hand their sums to wave 0, one after the other (fixed order)
        for (int w = 1; w < KS; ++w) {
            if (ks == w) {
#pragma unroll
                for (int blk = 0; blk < NB; ++blk)
#pragma unroll
                    for (int j = 0; j < V; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) s_red[((blk * V + j) * 16 + r) * 64 + lane] = acc[blk][j][r];
            }
            __syncthreads();
            if (ks == 0) {
#pragma unroll
                for (int blk = 0; blk < NB; ++blk)
#pragma unroll
                    for (int j = 0; j < V; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[blk][j][r] += s_red[((blk * V + j) * 16 + r) * 64 + lane];
            }
            __syncthreads();
        }
        if (ks != 0) return;
    }
    if (!pvalid) return;
    // C/D map of the 32x32 MFMA: row (output channel) = (r&3) + 8*(r>>2) + 4*half, column (pixel group) = l31
    float* yb = p.y + (int64_t)b * p.O * p.P + pix;
    const float* rb = p.residual ? p.residual + (int64_t)b * p.O * p.P + pix : nullptr;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (o >= p.O) continue;
            const float bo = p.bias ? p.bias[o] : 0.f;
            float v[V];
#pragma unroll
            for (int j = 0; j < V; ++j) {
                v[j] = acc[blk][j][r] + bo;
                if (p.clamp >= 0.f) v[j] = fminf(fmaxf(v[j], -p.clamp), p.clamp);
            }
            if (rb) {
                const vec rv = *reinterpret_cast<const vec*>(rb + (int64_t)o * p.P);
#pragma unroll
                for (int j = 0; j < V; ++j) v[j] += comp<V>(rv, j);
            }
            vec out;
            if constexpr (V == 4) out = make_float4(v[0], v[1], v[2], v[3]);
            else out = make_float2(v[0], v[1]);
            *reinterpret_cast<vec*>(yb + (int64_t)o * p.P) = out;
        }
}


// ---- ia_torgb: the same layer with every load of a wave in flight at once, and the skip image's up-sampling in the epilogue -------------
//
// conv1x1_kernel keeps 8 channel pairs (8 KB) in flight per wave and walks its channels in 8 - 32 dependent load -> MFMA rounds: at
// one frame per call the 22 ToRGB launches of a frame are latency chains (18 - 40 us each for 0.5 .. 33 MB).  Here a wave owns 32
// pixels x 128 input channels (64 channel pairs): all 64 activation loads, 64 weight loads and the styles are issued before the first
// MFMA -- one memory round trip -- and the input channels of a 256 / 512-channel layer are split over KS = 2 / 4 waves that meet in LDS
// (fixed order).  A workgroup is always 8 waves: 8 / KS pixel fragments.  The epilogue adds either a residual image or
// upsample2d(skip) -- the 2x up-sampling of the previous block's image with the 4x4 resample filter (upfirdn2d.upsample2d :341-350:
// zero insertion, padding [2,1,2,1], gain 4) evaluated at the output pixel: of the 16 taps the 2 x 2 that fall on real samples, in the
// order and with the fused multiply-adds of ia_upfirdn2d (the other 12 multiply zeros there), so the image is bit-identical to the
// two-launch route's -- and the separate upfirdn2d launch per block disappears.
struct TParams {
    const float* x;          // [B][I][P]
    const float* wk;         // [I][O]
    const float* styles;     // [B][I] or null
    const float* bias;       // [O] or null
    const float* residual;   // [B][O][P] or null (added after the clamp)
    const float* skip;       // [B][O][H/2][W/2] or null: up-sampled 2x and added after the clamp
    const float* filt;       // [4][4] resample filter of the skip up-sampling
    float* y;                // [B][O][P]
    int I, O, H, W;
    int64_t P;
    float clamp;             // < 0: none
};

constexpr int kTU = 32;      // channel pairs per register block (two blocks are in flight)

template <int KS>
__global__ __launch_bounds__(512) void torgb_kernel(TParams p) {
    constexpr int NG = 8 / KS;                                   // pixel fragments per workgroup
    __shared__ float s_red[KS > 1 ? 8 * 16 * 64 : 1];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l31 = lane & 31, half = lane >> 5;
    const int grp = wave / KS, ks = wave - grp * KS;
    const int b = blockIdx.y, o0 = blockIdx.z * 32;
    const int64_t pix = ((int64_t)blockIdx.x * NG + grp) * 32 + l31;
    const bool pvalid = pix < p.P, ovalid = o0 + l31 < p.O;
    const int kw = p.I / KS, k_begin = ks * kw;                  // this wave's input channels: 64 or 128 of them
    const float* xp = p.x + ((int64_t)b * p.I + k_begin + half) * p.P + (pvalid ? pix : 0);
    const float* wp = p.wk + (int64_t)(k_begin + half) * p.O + o0 + (ovalid ? l31 : 0);
    // styles of the wave's channels: lane l holds those of channels l and 64 + l; a pair's two values are read back with v_readlane
    float sv0 = 1.f, sv1 = 1.f;
    if (p.styles) {
        const float* sp = p.styles + (int64_t)b * p.I + k_begin;
        sv0 = sp[lane];
        if (kw > 64) sv1 = sp[64 + lane];
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    float xa[kTU], wa[kTU], xb[kTU], wb[kTU];
    auto load_block = [&](float (&xv)[kTU], float (&wv)[kTU], int k0) {
#pragma unroll
        for (int u = 0; u < kTU; ++u) {
            const int k = k0 + 2 * u;                            // this lane's channel: k_begin + k + half
            xv[u] = pvalid ? xp[(int64_t)k * p.P] : 0.f;
            wv[u] = ovalid ? wp[(int64_t)k * p.O] : 0.f;
        }
    };
    auto mma_block = [&](const float (&xv)[kTU], const float (&wv)[kTU], float sv) {
#pragma unroll
        for (int u = 0; u < kTU; ++u) {
            const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sv), 2 * u));
            const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sv), 2 * u + 1));
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u] * (half ? s1 : s0), xv[u], acc, 0, 0, 0);      // (w * s) * x, the reference's order
        }
    };
    load_block(xa, wa, 0);
    if (kw > 64) load_block(xb, wb, 64);

    // The epilogue's operands do not depend on the products: their loads go out now, behind the activations, and land under the MFMAs.
    // C/D map of the 32x32 MFMA: row (output channel) = (r&3) + 8*(r>>2) + 4*half, column (pixel) = l31.  With KS > 1 the KS waves
    // of a pixel fragment leave their sums in LDS and wave ks finishes registers [ks * 16/KS, (ks+1) * 16/KS), adding in wave order.
    constexpr int NR = 16 / KS;
    // the 2 x 2 real taps of the up-sampling at this pixel: rows my + t with weights ky[t], columns mx + u with kx[u]
    int soff[4];
    float sk[4];
    const int Ws = p.W >> 1, Hs = p.H >> 1;
    if (p.skip) {
        const int64_t pp = pvalid ? pix : 0;
        const int oy = (int)(pp / p.W), ox = (int)(pp - (int64_t)oy * p.W);
        const int my = ((oy + (oy & 1)) >> 1) - 1, mx = ((ox + (ox & 1)) >> 1) - 1;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int iy = (oy & 1) + 2 * t, ix = (ox & 1) + 2 * u;          // tap index in the 4x4 filter (before the flip)
                const bool in = my + t >= 0 && my + t < Hs && mx + u >= 0 && mx + u < Ws;
                soff[2 * t + u] = in ? (my + t) * Ws + mx + u : 0;
                sk[2 * t + u] = in ? p.filt[(3 - iy) * 4 + (3 - ix)] * 4.f : 0.f;
            }
    }
    float e_bias[NR], e_res[NR], e_tap[NR][4];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int r = ks * NR + j;
        const int o = min(o0 + (r & 3) + 8 * (r >> 2) + 4 * half, p.O - 1);
        e_bias[j] = p.bias ? p.bias[o] : 0.f;
        e_res[j] = (p.residual && pvalid) ? p.residual[((int64_t)b * p.O + o) * p.P + pix] : 0.f;
        if (p.skip) {
            const float* sb = p.skip + ((int64_t)b * p.O + o) * Hs * Ws;
#pragma unroll
            for (int q = 0; q < 4; ++q) e_tap[j][q] = sb[soff[q]];
        }
    }

    mma_block(xa, wa, sv0);
    if (kw > 64) mma_block(xb, wb, sv1);

    float v[NR];
    if constexpr (KS > 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s_red[(wave * 16 + r) * 64 + lane] = acc[r];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            float s = s_red[((grp * KS) * 16 + ks * NR + j) * 64 + lane];
#pragma unroll
            for (int w = 1; w < KS; ++w) s += s_red[((grp * KS + w) * 16 + ks * NR + j) * 64 + lane];
            v[j] = s;
        }
    } else {
#pragma unroll
        for (int j = 0; j < NR; ++j) v[j] = acc[ks * NR + j];
    }
    if (!pvalid) return;
    float* yb = p.y + (int64_t)b * p.O * p.P + pix;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int r = ks * NR + j;
        const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (o >= p.O) continue;
        float s = v[j] + e_bias[j];
        if (p.clamp >= 0.f) s = fminf(fmaxf(s, -p.clamp), p.clamp);
        if (p.residual) s += e_res[j];
        if (p.skip) {
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) a = fmaf(e_tap[j][q], sk[q], a);
            s += a;
        }
        yb[(int64_t)o * p.P] = s;
    }
}

// ---- ToRGB proper: up to four output channels on a large image (the SR head's 256 -> 3 @256^2: 67 MB of activations for 0.1 GFLOP).
// torgb_kernel pads the three channels to a 32-row MFMA fragment and fetches a pixel fragment with 4-byte loads, 32 pixels x 2 channels
// per instruction: 0.33 TB/s on that layer (205 us, r03 frame trace).  This is a streaming kernel instead: a lane owns FOUR consecutive
// pixels and reads them as one 16-byte load per input channel (a wave: 1 KB contiguous per channel), the products are plain FMAs against
// (weight x style) rows kept in LDS (broadcast reads), the input channels are split over the KS waves of a workgroup -- every wave has
// 2 x 8 loads in flight -- and the partial sums meet in LDS in wave order.  Wave o then finishes output channel o: bias, clamp,
// residual or up-sampled skip image with the arithmetic of torgb_kernel (the up-sampled image stays bit-identical to ia_upfirdn2d's).
constexpr int kFewO = 4, kFewU = 8;
template <int KS>
__global__ __launch_bounds__(64 * KS) void torgb_few_kernel(TParams p) {
    static_assert(KS >= kFewO, "wave o finishes output channel o");
    __shared__ float4 s_w[1024];                     // (w[k][0..3] * s[k]) per input channel
    __shared__ float4 s_red[KS][kFewO][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int64_t pix0 = ((int64_t)blockIdx.x * 64 + lane) * 4;
    for (int k = tid; k < p.I; k += 64 * KS) {
        const float st = p.styles ? p.styles[(int64_t)b * p.I + k] : 1.f;
        float w4[kFewO];
#pragma unroll
        for (int o = 0; o < kFewO; ++o) w4[o] = o < p.O ? p.wk[(int64_t)k * p.O + o] * st : 0.f;      // (w * s), the reference's order
        s_w[k] = make_float4(w4[0], w4[1], w4[2], w4[3]);
    }
    __syncthreads();
    const int kw = p.I / KS, k0 = wave * kw;
    const float* xp = p.x + ((int64_t)b * p.I + k0) * p.P + pix0;
    float acc[kFewO][4];
#pragma unroll
    for (int o = 0; o < kFewO; ++o)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[o][j] = 0.f;
    float4 xa[kFewU], xb[kFewU];
    auto load_block = [&](float4 (&xv)[kFewU], int k) {
#pragma unroll
        for (int u = 0; u < kFewU; ++u) xv[u] = *reinterpret_cast<const float4*>(xp + (int64_t)(k + u) * p.P);
    };
    auto fma_block = [&](const float4 (&xv)[kFewU], int k) {
#pragma unroll
        for (int u = 0; u < kFewU; ++u) {
            const float4 w = s_w[k0 + k + u];
            const float wv[kFewO] = {w.x, w.y, w.z, w.w}, xj[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
#pragma unroll
            for (int o = 0; o < kFewO; ++o)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[o][j] = fmaf(wv[o], xj[j], acc[o][j]);
        }
    };
    load_block(xa, 0);
    for (int k = 0; k < kw; k += 2 * kFewU) {              // (kw is a multiple of 16: I in {128, 256, 512, 1024}, KS = 4)
        load_block(xb, k + kFewU);
        fma_block(xa, k);
        if (k + 2 * kFewU < kw) load_block(xa, k + 2 * kFewU);
        fma_block(xb, k + kFewU);
    }
#pragma unroll
    for (int o = 0; o < kFewO; ++o) s_red[wave][o][lane] = make_float4(acc[o][0], acc[o][1], acc[o][2], acc[o][3]);
    __syncthreads();
    const int o = wave;
    if (o >= p.O) return;
    float4 t = s_red[0][o][lane];
#pragma unroll
    for (int w = 1; w < KS; ++w) { const float4 v = s_red[w][o][lane]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    const float v4[4] = {t.x, t.y, t.z, t.w};
    const float bias = p.bias ? p.bias[o] : 0.f;
    const int64_t at = ((int64_t)b * p.O + o) * p.P + pix0;
    float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.residual) r4 = *reinterpret_cast<const float4*>(p.residual + at);
    const float rj[4] = {r4.x, r4.y, r4.z, r4.w};
    const int Ws = p.W >> 1, Hs = p.H >> 1;
    const int oy = (int)(pix0 / p.W), ox0 = (int)(pix0 - (int64_t)oy * p.W);      // (W % 4 == 0: the four pixels share a row)
    const float* sb = p.skip ? p.skip + ((int64_t)b * p.O + o) * Hs * Ws : nullptr;
    float out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float sres = v4[j] + bias;
        if (p.clamp >= 0.f) sres = fminf(fmaxf(sres, -p.clamp), p.clamp);
        if (p.residual) sres += rj[j];
        if (p.skip) {                 // the 2 x 2 real taps of the up-sampling at this pixel, in ia_upfirdn2d's order (see torgb_kernel)
            const int ox = ox0 + j;
            const int my = ((oy + (oy & 1)) >> 1) - 1, mx = ((ox + (ox & 1)) >> 1) - 1;
            float a = 0.f;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int iy = (oy & 1) + 2 * tt, ix = (ox & 1) + 2 * u;
                    const bool in = my + tt >= 0 && my + tt < Hs && mx + u >= 0 && mx + u < Ws;
                    const float tap = in ? sb[(my + tt) * Ws + mx + u] : 0.f;
                    const float kk = in ? p.filt[(3 - iy) * 4 + (3 - ix)] * 4.f : 0.f;
                    a = fmaf(tap, kk, a);
                }
            sres += a;
        }
        out[j] = sres;
    }
    *reinterpret_cast<float4*>(p.y + at) = make_float4(out[0], out[1], out[2], out[3]);
}

template <int V>
void launch_ks(int ksplit, dim3 grid, hipStream_t s, const C1Params& p) {
    switch (ksplit) {
        case 4: hipLaunchKernelGGL((conv1x1_kernel<V, 4>), grid, dim3(256), 0, s, p); break;
        case 2: hipLaunchKernelGGL((conv1x1_kernel<V, 2>), grid, dim3(128), 0, s, p); break;
        default: hipLaunchKernelGGL((conv1x1_kernel<V, 1>), grid, dim3(64), 0, s, p); break;
    }
}

}  // namespace

extern "C" int ia_conv1x1(const float* x, const float* wk, const float* styles, const float* bias, const float* residual, float* y,
                          int B, int I, int O, int H, int W, float clamp, void* stream) {
    IA_REQUIRE(x && wk && y, "x, wk and y must be device pointers");
    IA_REQUIRE(B > 0 && I > 0 && O > 0 && H > 0 && W > 0, "empty tensor");
    const int64_t P = (int64_t)H * W;
    if (O > 96 || I % 32 != 0 || P % 4 != 0)
        return ia::fail(IA_ERR_UNSUPPORTED, "ia_conv1x1 covers C_out <= 96, C_in %% 32 == 0 and H*W %% 4 == 0 (got C_in %d, C_out %d, %d x %d)", I, O, H, W);
    IA_REQUIRE(B <= 65535, "batch too large for one launch");
    // pixels per wave: 128, or 64 when that leaves the machine short of waves; then split the input channels over up to 4 waves
    // of a workgroup until there are about two waves per SIMD (each wave keeps >= 16 input channels)
    const int64_t want = 2 * 4 * (int64_t)ia::kNumCU;
    const int oblocks = (O + 31) / 32;
    int v = 4;
    if (((P + 127) / 128) * B * oblocks * 4 < want) v = 2;
    const int64_t tiles = (P + 32 * v - 1) / (32 * v);
    int ksplit = 1;
    while (ksplit < 4 && tiles * B * oblocks * ksplit < want && (I / (ksplit * 2)) % 16 == 0) ksplit *= 2;
    C1Params p{x, wk, styles, bias, residual, y, I, O, P, clamp};
    const dim3 grid((unsigned)tiles, (unsigned)B, (unsigned)oblocks);
    const hipStream_t s = (hipStream_t)stream;
    if (v == 4) launch_ks<4>(ksplit, grid, s, p);
    else launch_ks<2>(ksplit, grid, s, p);
    return ia::check_launch("ia_conv1x1");
}

extern "C" int ia_torgb_supported(int I, int O, int H, int W, int with_skip) {
    if (O > 96 || !(I == 128 || I == 256 || I == 512 || I == 1024)) return 0;
    if (with_skip && ((H | W) & 1)) return 0;
    return 1;
}

extern "C" int ia_torgb(const float* x, const float* wk, const float* styles, const float* bias, const float* residual, const float* skip,
                        const float* skip_filter, float* y, int B, int I, int O, int H, int W, float clamp, void* stream) {
    IA_REQUIRE(x && wk && y, "x, wk and y must be device pointers");
    IA_REQUIRE(B > 0 && I > 0 && O > 0 && H > 0 && W > 0, "empty tensor");
    IA_REQUIRE(!(residual && skip), "either a residual image or a skip image to up-sample, not both");
    IA_REQUIRE(!skip || skip_filter, "the skip image needs its 4x4 resample filter");
    if (!ia_torgb_supported(I, O, H, W, skip != nullptr))
        return ia::fail(IA_ERR_UNSUPPORTED, "ia_torgb covers C_out <= 96, C_in 128 / 256 / 512 / 1024 and even H, W with a skip image (got C_in %d, C_out %d, %d x %d)", I, O, H, W);
    IA_REQUIRE(B <= 65535, "batch too large for one launch");
    const int64_t P = (int64_t)H * W;
    // 128 input channels per wave; 512-channel layers on small images (up to 64^2: few pixel fragments) split over all 8 waves instead
    int ks = I / 128;
    if (P <= 4096 && I == 512) ks = 8;
    TParams p{x, wk, styles, bias, residual, skip, skip_filter, y, I, O, H, W, P, clamp};
    const hipStream_t s_ = (hipStream_t)stream;
    if (O <= kFewO && P >= 16384 && P % 256 == 0 && W % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
        (!residual || (reinterpret_cast<uintptr_t>(residual) & 15) == 0)) {      // ToRGB proper on a large image: the streaming form
        hipLaunchKernelGGL((torgb_few_kernel<4>), dim3((unsigned)(P / 256), (unsigned)B), dim3(256), 0, s_, p);
        return ia::check_launch("ia_torgb");
    }
    const int ng = 8 / ks;
    const dim3 grid((unsigned)((P + 32 * ng - 1) / (32 * ng)), (unsigned)B, (unsigned)((O + 31) / 32));
    const hipStream_t s = (hipStream_t)stream;
    switch (ks) {
        case 8: hipLaunchKernelGGL((torgb_kernel<8>), grid, dim3(512), 0, s, p); break;
        case 4: hipLaunchKernelGGL((torgb_kernel<4>), grid, dim3(512), 0, s, p); break;
        case 2: hipLaunchKernelGGL((torgb_kernel<2>), grid, dim3(512), 0, s, p); break;
        default: hipLaunchKernelGGL((torgb_kernel<1>), grid, dim3(512), 0, s, p); break;
    }
    return ia::check_launch("ia_torgb");
}

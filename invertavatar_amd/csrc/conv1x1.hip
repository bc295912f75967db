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
#include <cstdlib>

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
constexpr int kWideMinPixels = 16384;     // torgb_wide_kernel from 128^2 up
#ifndef IA_TORGB_F16X3
#define IA_TORGB_F16X3 1      // 64 / 96 output channels on the fp16 pipe (three products per term); 0: fp32 MFMAs for every block count
#endif

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

// ---- ia_torgb on the large images (128^2, 256^2): torgb_kernel with the activations read ONCE for all output channels (r06).
// torgb_kernel gives each block of 32 output channels its own workgroup: the 96-channel layers of the texture network read their
// activations three times and launch three times the waves, and every lane fetches its 64 .. 128 weights from global memory beside
// its 64 .. 128 activations.  Here the workgroup first stages (weight x style) for ALL output channels in LDS (C_in x 32 * blocks
// values, 16 .. 96 KB; the activation loads are already in flight), then a wave walks the channel blocks over the activations it holds in
// registers.
//   F16 = false (one block of 32 channels): fp32 MFMAs; the A operand of each is one conflict-free ds_read_b32.  Products, k order, the
//     KS-wave sum and the epilogue are torgb_kernel's: bit-identical results.
//   F16 = true (64 / 96 channels, where 128 .. 192 dependent fp32 MFMAs per wave were the launch): the fp32 products are formed from fp16
//     hi / lo pairs on v_mfma_f32_32x32x16_f16 like the 3x3 layers' (hi x hi into one accumulator, hi x lo' + lo' x hi into a second one
//     that enters with 2^-11; lo x lo ~ 2^-22 dropped): 24 MFMAs of 8 passes per block instead of 64 of 16.  A lane loads the eight
//     channels 16 kb + 8 half + j of its pixel -- the B fragment of k-block kb -- and splits them in registers (hi = fp16(x), lo' =
//     fp16((x - hi) * 2^11), round to nearest); the weights are staged as (w x style x 2^10) split the same way, 16-byte
//     A fragments [octet][channel][8].  Activations outside the fp16 range raise the library's range-watch word like every other
//     producer of fp16 pairs (ia_split_saturation_poll).
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 pk2 __attribute__((ext_vector_type(2)));
constexpr float kWideWScale = 1024.f, kWideLoScale = 2048.f;

__device__ __forceinline__ void wide_split8(const float (&v)[8], h8& hi, h8& lo) {
    union { pk2 p[4]; h8 v; } uh, ul;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // (a high part that would be an fp16 denormal is flushed by the MFMA: such a value rides entirely in the scaled low part, like ia::split_f16)
        const float h0 = fabsf(v[2 * j]) < 6.103515625e-5f ? 0.f : v[2 * j], h1 = fabsf(v[2 * j + 1]) < 6.103515625e-5f ? 0.f : v[2 * j + 1];
        uh.p[j] = pk2{(__fp16)h0, (__fp16)h1};                   // (round to nearest: residuals <= 2^-12 of the value, so lo x lo <= 2^-24 .. 2^-22)
        const float r0 = (v[2 * j] - (float)uh.p[j][0]) * kWideLoScale, r1 = (v[2 * j + 1] - (float)uh.p[j][1]) * kWideLoScale;
        ul.p[j] = pk2{(__fp16)r0, (__fp16)r1};                   // (round to nearest: truncation here would bias every term the same way)
    }
    hi = uh.v;
    lo = ul.v;
}

template <int KS, int EPI, int OB, bool F16>      // EPI: 0 bias + clamp only, 1 + residual image, 2 + up-sampled skip image; OB blocks of 32 output channels
__global__ __launch_bounds__(512) void torgb_wide_kernel(TParams p) {
    constexpr int NG = 8 / KS, NR = 16 / KS;
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    constexpr int OP = OB * 32;                                  // (compile-time: every weight read is one base register + an immediate)
    float* s_w = s_dyn;                                          // fp32: [I][OP] wk * style, zero beyond O; fp16 pairs: hi [I/8][OP][8], then lo
    float* s_red = s_dyn + (int64_t)p.I * OP;                    // [8][16][64] (KS > 1)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, half = lane >> 5;
    const int grp = wave / KS, ks = wave - grp * KS;
    const int b = blockIdx.y;
    const int64_t pix = ((int64_t)blockIdx.x * NG + grp) * 32 + l31;
    const bool pvalid = pix < p.P;
    const int kw = p.I / KS, k_begin = ks * kw;                  // this wave's input channels: 128 of them
    // fp32: register u of block a / b is channel 2u + half (+ 64); fp16 pairs: register 8 kb + j is channel 16 kb + 8 half + j
    const float* xp = p.x + ((int64_t)b * p.I + k_begin + (F16 ? 8 * half : half)) * p.P + (pvalid ? pix : 0);
    float xa[kTU], xb[kTU];
#pragma unroll
    for (int u = 0; u < kTU; ++u) xa[u] = pvalid ? xp[(int64_t)(F16 ? 16 * (u >> 3) + (u & 7) : 2 * u) * p.P] : 0.f;
#pragma unroll
    for (int u = 0; u < kTU; ++u) xb[u] = pvalid ? xp[(int64_t)(64 + (F16 ? 16 * (u >> 3) + (u & 7) : 2 * u)) * p.P] : 0.f;

    // (w * s), the reference's order, for every output channel
    {
        const float* sp = p.styles ? p.styles + (int64_t)b * p.I : nullptr;
        if constexpr (!F16) {                                    // 512 threads walk the [I][OP] image, OP / 4 float4 columns per row
            const int cols = OP >> 2, total = p.I * cols;
            const bool vec_ok = (p.O & 3) == 0;
            for (int e = tid; e < total; e += 512) {
                const int k = e / cols, o = (e - k * cols) * 4;
                const float st = sp ? sp[k] : 1.f;
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (vec_ok) {
                    if (o < p.O) w = *reinterpret_cast<const float4*>(p.wk + (int64_t)k * p.O + o);
                } else {
                    const float* wr = p.wk + (int64_t)k * p.O;
                    w.x = o < p.O ? wr[o] : 0.f; w.y = o + 1 < p.O ? wr[o + 1] : 0.f; w.z = o + 2 < p.O ? wr[o + 2] : 0.f; w.w = o + 3 < p.O ? wr[o + 3] : 0.f;
                }
                *reinterpret_cast<float4*>(s_w + (int64_t)k * OP + o) = make_float4(w.x * st, w.y * st, w.z * st, w.w * st);
            }
        } else {                                                 // one (octet of input channels, output channel) per step: a 16-byte A fragment
            h8* s_hi = reinterpret_cast<h8*>(s_w);
            h8* s_lo = s_hi + (p.I >> 3) * OP;
            const int total = (p.I >> 3) * OP;
            for (int e = tid; e < total; e += 512) {
                const int oct = e / OP, o = e - oct * OP;
                float w8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = oct * 8 + j;
                    const float w = o < p.O ? p.wk[(int64_t)k * p.O + o] : 0.f;
                    w8[j] = (w * (sp ? sp[k] : 1.f)) * kWideWScale;
                    w8[j] = fminf(fmaxf(w8[j], -65504.f), 65504.f);
                }
                h8 hi, lo;
                wide_split8(w8, hi, lo);
                s_hi[e] = hi;
                s_lo[e] = lo;
            }
        }
    }

    // the 2 x 2 real taps of the up-sampling at this pixel (see torgb_kernel)
    int soff[4];
    float sk[4];
    const int Ws = p.W >> 1, Hs = p.H >> 1;
    if constexpr (EPI == 2) {
        const int64_t pp = pvalid ? pix : 0;
        const int oy = (int)(pp / p.W), ox = (int)(pp - (int64_t)oy * p.W);
        const int my = ((oy + (oy & 1)) >> 1) - 1, mx = ((ox + (ox & 1)) >> 1) - 1;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int iy = (oy & 1) + 2 * t, ix = (ox & 1) + 2 * u;
                const bool in = my + t >= 0 && my + t < Hs && mx + u >= 0 && mx + u < Ws;
                soff[2 * t + u] = in ? (my + t) * Ws + mx + u : 0;
                sk[2 * t + u] = in ? p.filt[(3 - iy) * 4 + (3 - ix)] * 4.f : 0.f;
            }
    }

    // fp16 pairs: the wave's activations become B fragments (8 k-blocks x hi, lo') as they land
    h8 xh[F16 ? 8 : 1], xl[F16 ? 8 : 1];
    if constexpr (F16) {
        float amax = 0.f;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            float v8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v8[j] = kb < 4 ? xa[8 * kb + j] : xb[8 * (kb - 4) + j];
                amax = fmaxf(amax, fabsf(v8[j]));
            }
            wide_split8(v8, xh[kb], xl[kb]);
        }
        if (!(amax <= 65504.f)) atomicOr(&ia_tu_saturated, 1u);
    }
    __syncthreads();

    float* yb = p.y + (int64_t)b * p.O * p.P + (pvalid ? pix : 0);
    for (int ob = 0; ob < OB; ++ob) {
        const int o0 = ob * 32;
        // the epilogue's operands of this block go out ahead of its MFMAs
        float e_bias[NR], e_add[NR], e_tap[EPI == 2 ? NR : 1][4];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int r = ks * NR + j;
            const int o = min(o0 + (r & 3) + 8 * (r >> 2) + 4 * half, p.O - 1);
            e_bias[j] = p.bias ? p.bias[o] : 0.f;
            e_add[j] = 0.f;
            if constexpr (EPI == 1) e_add[j] = pvalid ? p.residual[((int64_t)b * p.O + o) * p.P + pix] : 0.f;
            if constexpr (EPI == 2) {
                const float* sb = p.skip + ((int64_t)b * p.O + o) * Hs * Ws;
#pragma unroll
                for (int q = 0; q < 4; ++q) e_tap[j][q] = sb[soff[q]];
            }
        }
        auto fold_taps = [&]() {                                 // four taps become one value per row
            if constexpr (EPI == 2) {
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    float a = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) a = fmaf(e_tap[j][q], sk[q], a);
                    e_add[j] = a;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if constexpr (!F16) {
            const float* wo = s_w + (int64_t)(k_begin + half) * OP + l31 + o0;
            // weights: eight ds_reads one group of eight MFMAs (512 cycles) ahead, pinned so that the scheduler does not hoist all 64
            constexpr int WG8 = 8;
            float wq[2][WG8];
#pragma unroll
            for (int u = 0; u < WG8; ++u) wq[0][u] = wo[(2 * u) * OP];
#pragma unroll
            for (int g = 0; g < 2 * kTU / WG8; ++g) {
                if (g + 1 < 2 * kTU / WG8) {
#pragma unroll
                    for (int u = 0; u < WG8; ++u) wq[(g + 1) & 1][u] = wo[(2 * ((g + 1) * WG8 + u)) * OP];
                }
#pragma unroll
                for (int u = 0; u < WG8; ++u) {
                    const int uu = g * WG8 + u;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[g & 1][u], uu < kTU ? xa[uu] : xb[uu - kTU], acc, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (g == 2 * kTU / WG8 / 2 - 1) fold_taps();     // half way: the taps have landed
            }
        } else {
            const h8* wh = reinterpret_cast<const h8*>(s_w) + (int64_t)((k_begin >> 3) + half) * OP + l31 + o0;
            const h8* wlo = wh + (p.I >> 3) * OP;
            f32x16 acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
            h8 qh[2], ql[2];
            qh[0] = wh[0];
            ql[0] = wlo[0];
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                if (kb + 1 < 8) {
                    qh[(kb + 1) & 1] = wh[(2 * (kb + 1)) * OP];
                    ql[(kb + 1) & 1] = wlo[(2 * (kb + 1)) * OP];
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[kb & 1], xh[kb], acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[kb & 1], xl[kb], acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ql[kb & 1], xh[kb], acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kb == 3) fold_taps();
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = (acc[r] + acc1[r] * (1.f / kWideLoScale)) * (1.f / kWideWScale);
        }

        float v[NR];
        if constexpr (KS > 1) {
            if (ob) __syncthreads();                             // the previous block's sums have been read
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
            for (int j = 0; j < NR; ++j) v[j] = acc[j];
        }
        if (pvalid) {
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int r = ks * NR + j;
                const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (o >= p.O) continue;
                float s = v[j] + e_bias[j];
                if (p.clamp >= 0.f) s = fminf(fmaxf(s, -p.clamp), p.clamp);
                if constexpr (EPI != 0) s += e_add[j];
                yb[(int64_t)o * p.P] = s;
            }
        }
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
    const hipStream_t s = (hipStream_t)stream;
    // large images, 128 channels per wave: the activations are read once for all output channels (weights x styles in LDS)
    static const int wide_min_p = getenv("IA_TORGB_WIDE_P") ? atoi(getenv("IA_TORGB_WIDE_P")) : kWideMinPixels;
    const size_t wide_lds = ((size_t)I * (((O + 31) / 32) * 32) + (ks > 1 ? 8 * 16 * 64 : 0)) * sizeof(float);
    if (P >= wide_min_p && I / ks == 128 && ks <= 2 && wide_lds <= 144 * 1024) {
        const dim3 wgrid((unsigned)((P + 32 * ng - 1) / (32 * ng)), (unsigned)B);
        const int epi = skip ? 2 : residual ? 1 : 0, ob = (O + 31) / 32;
        auto go = [&](auto kern) -> int {
            if (const int st = ia::reserve_lds((const void*)kern, wide_lds, "ia_torgb")) return st;
            hipLaunchKernelGGL(kern, wgrid, dim3(512), wide_lds, s, p);
            return ia::check_launch("ia_torgb");
        };
#define IA_WIDE_OB(KS_, EPI_) (ob == 1 ? go(torgb_wide_kernel<KS_, EPI_, 1, false>) : ob == 2 ? go(torgb_wide_kernel<KS_, EPI_, 2, IA_TORGB_F16X3 != 0>) : go(torgb_wide_kernel<KS_, EPI_, 3, IA_TORGB_F16X3 != 0>))
#define IA_WIDE_EPI(KS_) (epi == 2 ? IA_WIDE_OB(KS_, 2) : epi == 1 ? IA_WIDE_OB(KS_, 1) : IA_WIDE_OB(KS_, 0))
        return ks == 2 ? IA_WIDE_EPI(2) : IA_WIDE_EPI(1);
#undef IA_WIDE_EPI
#undef IA_WIDE_OB
    }
    const dim3 grid((unsigned)((P + 32 * ng - 1) / (32 * ng)), (unsigned)B, (unsigned)((O + 31) / 32));
    switch (ks) {
        case 8: hipLaunchKernelGGL((torgb_kernel<8>), grid, dim3(512), 0, s, p); break;
        case 4: hipLaunchKernelGGL((torgb_kernel<4>), grid, dim3(512), 0, s, p); break;
        case 2: hipLaunchKernelGGL((torgb_kernel<2>), grid, dim3(512), 0, s, p); break;
        default: hipLaunchKernelGGL((torgb_kernel<1>), grid, dim3(512), 0, s, p); break;
    }
    return ia::check_launch("ia_torgb");
}

// ia_conv2d_mfma_sx / ia_act_split: the large 3x3 convolutions of the StyleGAN2 stack on activations that are ALREADY stored as
// fp16 hi/lo pairs, with both operands DMA'd straight into LDS.
//
// Same arithmetic as ia_conv2d_mfma_s (conv_mfma.hip, HM = 2): every fp32 product a*b is taken as
// a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation, a = packed weights (hi, lo of w * 2^e),
// b = style-scaled activation (hi, lo * 2^11): the same products as that path (the stride-1 kernels add them in another order since
// they pair the odd tap of a chunk with the next chunk's, see PAIR below).  What differs is where the split
// happens and how the operands reach LDS:
//
//   * the PRODUCER of an activation (the FIR tail of an up-sampling layer, the epilogue of the previous convolution, or the
//     stand-alone ia_act_split) multiplies it by the consumer's style and stores the pair as two planes of 8-channel, 16-byte
//     units:  xs[b][plane][c/8][y][x][c%8]  (4 bytes per element, the size of the fp32 tensor it replaces).  The consumer's
//     staging then needs no arithmetic at all;
//   * both operands are fetched with buffer_load_dwordx4 ... lds (LDS-DMA): no staging registers, no VALU conversion, no
//     ds_write -- in the register-staged kernel those cost as much time as the MFMAs (DESIGN.md 6.1 ablation).  Zero padding
//     comes from the buffer bounds check (out-of-range lanes write zeros to LDS).  Two LDS stages: the DMA of chunk k+1 is in
//     flight while chunk k is multiplied; one barrier per chunk.
//
// Tiles, the tile window, stream-K scheduling, slabs and the fix-up kernel are those of conv_mfma.hip (conv_common.h).
// Replaces, like ia_conv2d_mfma, modulated_conv2d -> conv2d_resample -> conv2d / conv_transpose2d (+ bias_act) of the reference
// (training/networks_stylegan2.py:34-91, torch_utils/ops/conv2d_resample.py:114-136).
#include "conv_common.h"
#include "lds_dma.h"
#include "conv_small.h"
#include <cstddef>
#ifndef IA_EPI_RELOAD
#define IA_EPI_RELOAD 1      // 0: the epilogue reads the by-value kernel argument (r02 - r05; A/B builds)
#endif
#include <cstdlib>

// Compile-time ablations for tools/ablate_conv_split.sh (never set in the product build): 1 = no DMA after the first chunk of a
// segment, 2 = no MFMAs (operand reads kept alive), 3 = no operand reads (MFMAs on stale registers), 4 = no output stores.
#ifndef IA_ABLATE
#define IA_ABLATE 0
#endif
#ifndef IA_PAIR_TAPS
#define IA_PAIR_TAPS 1
#endif
#ifndef IA_ANTIPHASE
#define IA_ANTIPHASE 1      // 0: every tile on the in-step K loop (A/B builds in tools/)
#endif
#define IA_AB_NODMA (IA_ABLATE == 1 || IA_ABLATE >= 5)
#define IA_AB_NOREAD (IA_ABLATE == 3 || IA_ABLATE >= 5)
#define IA_AB_NOSTORE (IA_ABLATE == 4 || IA_ABLATE >= 5)

namespace {

constexpr size_t kLdsBytes = 160 * 1024;     // per CU on gfx950

// fp32 NCHW (times an optional per-(batch, channel) style) -> split planes.  One thread = one pixel of one 8-channel group.
__global__ __launch_bounds__(256) void act_split_kernel(const float* __restrict__ x, const float* __restrict__ styles, const float* __restrict__ shift,
                                                       h16x8* __restrict__ out, int B, int C, int64_t HW, int planes) {
    const int C8 = C / 8;
    const int64_t total = (int64_t)B * C8 * HW, stride = (int64_t)gridDim.x * blockDim.x;
    ia::SatWatch watch;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t pix = i % HW;
        const int c8 = (int)((i / HW) % C8), b = (int)(i / (HW * C8));
        h16x8 hi, lo;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            const int c = c8 * 8 + cc;
            float v = x[((int64_t)b * C + c) * HW + pix];
            if (styles) v *= styles[b * C + c];
            if (shift) v += shift[b * C + c];
            if (planes == 2) { _Float16 h, l; ia::split_f16(v, h, l, watch); hi[cc] = h; lo[cc] = l; }
            else hi[cc] = ia::round_f16(v, watch);
        }
        out[((int64_t)(b * planes) * C8 + c8) * HW + pix] = hi;
        if (planes == 2) out[((int64_t)(b * 2 + 1) * C8 + c8) * HW + pix] = lo;
    }
    watch.report();
}

// The same, four consecutive pixels per thread (H*W % 4 == 0): 16-byte loads, a quarter of the load instructions.
__global__ __launch_bounds__(256) void act_split4_kernel(const float* __restrict__ x, const float* __restrict__ styles, const float* __restrict__ shift,
                                                        h16x8* __restrict__ out, int B, int C, int64_t HW, int planes) {
    const int C8 = C / 8;
    const int64_t HW4 = HW / 4, total = (int64_t)B * C8 * HW4, stride = (int64_t)gridDim.x * blockDim.x;
    ia::SatWatch watch;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t p4 = i % HW4;
        const int c8 = (int)((i / HW4) % C8), b = (int)(i / (HW4 * C8));
        float4 xv[8];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) xv[cc] = reinterpret_cast<const float4*>(x + ((int64_t)b * C + c8 * 8 + cc) * HW)[p4];
        h16x8 hi[4], lo[4];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            const int c = c8 * 8 + cc;
            const float st = styles ? styles[b * C + c] : 1.f, sh = shift ? shift[b * C + c] : 0.f;
            const float v4[4] = {xv[cc].x, xv[cc].y, xv[cc].z, xv[cc].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v = v4[k];
                if (styles) v *= st;
                if (shift) v += sh;
                if (planes == 2) { _Float16 h, l; ia::split_f16(v, h, l, watch); hi[k][cc] = h; lo[k][cc] = l; }
                else hi[k][cc] = ia::round_f16(v, watch);
            }
        }
        h16x8* dh = out + ((int64_t)(b * planes) * C8 + c8) * HW + 4 * p4;
#pragma unroll
        for (int k = 0; k < 4; ++k) dh[k] = hi[k];
        if (planes == 2) {
            h16x8* dl = out + ((int64_t)(b * 2 + 1) * C8 + c8) * HW + 4 * p4;
#pragma unroll
            for (int k = 0; k < 4; ++k) dl[k] = lo[k];
        }
    }
    watch.report();
}

// Train-mode BatchNorm2d folded into the split staging (ia_bn_train_split), pass 1: per (channel, chunk) sum and sum of squares of the
// channel's B * HW values, in double (the variance is taken as E[x^2] - E[x]^2 of those sums).
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, double* __restrict__ partials, int B, int C, int64_t HW, int J) {
    __shared__ double red[2][4];
    const int c = blockIdx.x, j = blockIdx.y;
    const int64_t n = (int64_t)B * HW, per = (n + J - 1) / J, e0 = j * per, e1 = e0 + per < n ? e0 + per : n;
    double s = 0.0, q = 0.0;
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) {
        const int64_t b = e / HW, p = e - b * HW;
        const double v = (double)x[((int64_t)b * C + c) * HW + p];
        s += v; q = fma(v, v, q);
    }
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[((int64_t)c * J + j) * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partials[((int64_t)c * J + j) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// Pass 2: one workgroup = one 8-channel group of one batch element and one pixel chunk.  Its first eight threads finish the statistics
// of their channels (fixed order over the J partials), the workgroup of chunk 0 / batch 0 also moves the running statistics the way
// torch.nn.BatchNorm2d does in train mode (momentum lerp, unbiased variance), then the block streams split((x - mean) * a + bias).
template <int PX>      // pixels per thread: 4 (HW % 4 == 0, 16-byte loads) or 1
__global__ __launch_bounds__(256) void bn_split_kernel(const float* __restrict__ x, const double* __restrict__ partials, const float* __restrict__ weight,
                                                      const float* __restrict__ bias, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                      long long* __restrict__ num_batches, h16x8* __restrict__ out, int B, int C, int64_t HW, int J,
                                                      float eps, float momentum, int planes) {
    __shared__ float sc_a[8], sc_c[8];
    const int c8 = blockIdx.x, b = blockIdx.z, C8 = C / 8;
    if (threadIdx.x < 8) {
        const int c = c8 * 8 + threadIdx.x;
        double s = 0.0, q = 0.0;
        for (int j = 0; j < J; ++j) { s += partials[((int64_t)c * J + j) * 2]; q += partials[((int64_t)c * J + j) * 2 + 1]; }
        const double n = (double)B * (double)HW, mean = s / n;
        double var = q / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float a = (weight ? weight[c] : 1.f) * rsqrtf((float)var + eps);
        sc_a[threadIdx.x] = a;
        sc_c[threadIdx.x] = (bias ? bias[c] : 0.f) - (float)mean * a;
        if (blockIdx.y == 0 && b == 0 && running_mean) {
            running_mean[c] += momentum * ((float)mean - running_mean[c]);
            running_var[c] += momentum * ((float)(var * (n / (n > 1.0 ? n - 1.0 : 1.0))) - running_var[c]);
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && b == 0 && num_batches) *num_batches += 1;
    __syncthreads();
    float a[8], cc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = sc_a[k]; cc[k] = sc_c[k]; }
    ia::SatWatch watch;
    const int64_t items = HW / PX;
    const float* xb = x + ((int64_t)b * C + c8 * 8) * HW;
    h16x8* dh = out + ((int64_t)(b * planes) * C8 + c8) * HW;
    h16x8* dl = out + ((int64_t)(b * 2 + 1) * C8 + c8) * HW;
    for (int64_t it = (int64_t)blockIdx.y * 256 + threadIdx.x; it < items; it += (int64_t)gridDim.y * 256) {
        float v[8][PX];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if constexpr (PX == 4) {
                const float4 t = reinterpret_cast<const float4*>(xb + k * HW)[it];
                v[k][0] = t.x; v[k][1] = t.y; v[k][2] = t.z; v[k][3] = t.w;
            } else {
                v[k][0] = xb[k * HW + it];
            }
        }
#pragma unroll
        for (int px = 0; px < PX; ++px) {
            h16x8 hi, lo;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float t = fmaf(v[k][px], a[k], cc[k]);
                if (planes == 2) { _Float16 h, l; ia::split_f16(t, h, l, watch); hi[k] = h; lo[k] = l; }
                else hi[k] = ia::round_f16(t, watch);
            }
            dh[it * PX + px] = hi;
            if (planes == 2) dl[it * PX + px] = lo;
        }
    }
    watch.report();
}

// Elements of the hi plane that sit on the fp16 maximum: values the split clamped (see ia_split_saturation_count).
__global__ __launch_bounds__(256) void split_saturation_kernel(const unsigned short* __restrict__ xs, int64_t per_batch_hi, int64_t batch_stride,
                                                              int B, unsigned int* __restrict__ count) {
    unsigned int mine = 0;
    const int64_t total = (int64_t)B * per_batch_hi, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t b = i / per_batch_hi, e = i - b * per_batch_hi;
        mine += ((xs[b * batch_stride + e] & 0x7fffu) == 0x7bffu) ? 1u : 0u;      // |hi| == 65504
    }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(count, mine);                  // (integer adds commute: the count is deterministic)
}

// Accumulator tile -> fp32 NCHW (y, optional) and/or split planes (e.ys, optional); stride-1 form.
// The per-channel terms of the epilogue (demodulation coefficient, bias, the consumer's style) are the same for every pixel of the tile:
// they are staged once per tile in LDS (`lds_f`, the K loop's stage memory, free by now) and read back as float4 per register quad.
// Reading them from global memory per element -- 3 loads for each of a lane's 64 values -- made the epilogue of a tile a chain of
// ~200 L1 round trips: 31 us of a 106 us layer with a short K loop (1024 <- 32 channels @128^2), 10 - 20 us on the large layers.
template <int FO, int FP, int WO, int WP>
__device__ __forceinline__ void store_tile_dual(const f32x16 (&acc)[1][FO][FP], float* __restrict__ y, const Geo& g, const Epi& e,
                                                int b, int o0, int p0, int tid, float* lds_f) {
    constexpr int BO = 32 * FO * WO, NTHREADS = WO * WP * 64;
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int wo = wave / WP, wp = wave % WP;
    const int npts = g.GH * g.GW;
    const int64_t ohw = (int64_t)g.OH * g.OW;
    const float ns = e.noise ? (e.noise_strength ? *e.noise_strength : 1.f) : 0.f;
    const int OR = e.d2s ? g.O / 4 : g.O;                      // real channels
    const int NB = OR / 32;
    float* c_dm = lds_f;                                         // [BO] demodulation (1 if none)
    float* c_bs = lds_f + BO;                                    // [BO] bias (0 if none)
    float* c_sn = lds_f + 2 * BO;                                // [BO] styles of the consumer (1 if none)
    float* c_sl = lds_f + 3 * BO;                                // [BO] negative slope of the activation (1 = linear)
    ia::SatWatch watch;
    const bool lrelu = e.act == IA_ACT_LRELU;
    __syncthreads();                                             // the K loop's last operand reads are done
    for (int t = tid; t < BO; t += NTHREADS) {
        // real channel of tile row t (depth-to-space: rows are groups of 32 = (row phase, block of 32 real channels, column phase))
        const int o = e.d2s ? ((((o0 >> 5) + (t >> 5)) >> 1) % NB) * 32 + (t & 31) : o0 + t;
        const bool ok = o < OR;
        c_dm[t] = (e.demod && ok) ? e.demod[b * OR + o] : 1.f;
        c_bs[t] = (e.bias && ok) ? e.bias[o] : 0.f;
        c_sn[t] = (e.styles_next && ok) ? e.styles_next[b * OR + o] : 1.f;
        c_sl[t] = !lrelu ? 1.f : (e.alpha_vec && ok) ? e.alpha_vec[o] : e.alpha;
    }
    __syncthreads();
    // The options of the epilogue as VALUES, not branches: a lane finishes 64 values here, and with a handful of uniform branches per
    // value (noise? activation? clamp? second output? channel in range?) the loop was ~57 vector instructions and 7 branches per value
    // (r05 ISA) -- a quarter of a 128-channel layer's time.  x * 1, fma(0, 0, x) and a clamp at infinity are exact.
    const float clamp = e.clamp >= 0.f ? e.clamp : INFINITY, gain = e.gain;
    const int OW2 = 2 * g.GW;
    const int64_t ohw_out = e.d2s ? 4 * ohw : ohw;
    float* yb = y ? y + ((int64_t)b * OR) * ohw_out : nullptr;
    const float* rb = e.residual ? e.residual + ((int64_t)b * OR) * ohw_out : nullptr;
#pragma unroll
    for (int fp = 0; fp < FP; ++fp) {
        const int p = p0 + (wp * FP + fp) * 32 + l31;
        if (p >= npts) continue;
        const int r = e.d2s ? p / g.GW : 0, c = p - r * g.GW;
#pragma unroll
        for (int fo = 0; fo < FO; ++fo) {
            // output pixel and first real channel of this fragment
            const int grp = (o0 >> 5) + wo * FO + fo;
            const int64_t pix = e.d2s ? (int64_t)(2 * r + ((grp >> 1) / NB)) * OW2 + 2 * c + (grp & 1) : (int64_t)p;
            const int o_frag = e.d2s ? ((grp >> 1) % NB) * 32 : o0 + (wo * FO + fo) * 32;
            const float nz = e.noise ? e.noise[pix] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {                      // register quad q: channels 8q + 4*half + (0..3) of the fragment
                const int trow = (wo * FO + fo) * 32 + 8 * q + 4 * half, o_first = o_frag + 8 * q + 4 * half;
                const float4 dm4 = *reinterpret_cast<const float4*>(c_dm + trow), bs4 = *reinterpret_cast<const float4*>(c_bs + trow);
                const float4 sn4 = *reinterpret_cast<const float4*>(c_sn + trow), sl4 = *reinterpret_cast<const float4*>(c_sl + trow);
                const float dm[4] = {dm4.x, dm4.y, dm4.z, dm4.w}, bs[4] = {bs4.x, bs4.y, bs4.z, bs4.w}, sn[4] = {sn4.x, sn4.y, sn4.z, sn4.w};
                const float sl[4] = {sl4.x, sl4.y, sl4.z, sl4.w};
                const bool inside = o_first + 3 < OR;            // (channel counts are multiples of 8: a quad is inside or outside as a whole)
                const int64_t at = (int64_t)o_first * ohw_out + pix;
                float res[4] = {0.f, 0.f, 0.f, 0.f};
                if (rb && inside) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) res[k] = rb[at + k * ohw_out];
                }
                float v[4], t[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float a = acc[0][fo][fp][4 * q + k] * dm[k];                       // (the order of epilogue(): demod, noise, bias, ...)
                    a = fmaf(nz, ns, a);
                    a += bs[k];
                    a = a > 0.f ? a : a * sl[k];
                    a *= gain;
                    { const float c = fminf(fmaxf(a, -clamp), clamp); a = (a != a) ? a : c; }      // NaN stays NaN (fmaxf would return -clamp), as torch.clamp / bias_act.cu:144 keep it
                    a += res[k];
                    v[k] = inside ? a : 0.f;
                    t[k] = v[k] * sn[k];
                }
                if (yb && inside) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) yb[at + k * ohw_out] = v[k];
                }
                if (e.ys && inside) split_store4(e.ys, nullptr, e.ys_planes, b, OR, ohw_out, o_first, pix, t, watch);
            }
        }
    }
    watch.report();
}

// store_tile_dual + the ToRGB layer that consumes the tile (Epi::rgb_*): the tile holds every output channel of its pixels, so the
// 1x1 convolution over them is a reduction inside the workgroup -- per lane over its 16*FO channels, across the two half-waves
// by a shuffle, across the WO channel waves through LDS (fixed order).  `lds_f`: the K loop's stage memory, free by now.
template <int FO, int FP, int WO, int WP>
__device__ __forceinline__ void store_tile_rgb(const f32x16 (&acc)[1][FO][FP], float* __restrict__ y, const Geo& g, const Epi& e,
                                               int b, int o0, int p0, int tid, float* lds_f) {
    constexpr int BO = 32 * FO * WO, NTHREADS = WO * WP * 64;
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int wo = wave / WP, wp = wave % WP;
    const int npts = g.GH * g.GW;
    const int64_t ohw = (int64_t)g.OH * g.OW;
    const float ns = e.noise ? (e.noise_strength ? *e.noise_strength : 1.f) : 0.f;
    float* yb = y ? y + ((int64_t)b * g.O) * ohw : nullptr;
    float* ws = lds_f;                              // [kMaxRgb][BO]: ToRGB weight x style of the tile's channels
    float* red = lds_f + kMaxRgb * BO;              // [WO][WP][FP][kMaxRgb][32]
    float* c_dm = red + WO * WP * FP * kMaxRgb * 32; // [BO] per-channel epilogue terms, staged once per tile (see store_tile_dual)
    float* c_bs = c_dm + BO;
    ia::SatWatchNow watch;                          // (no register for a flag here: 256 VGPRs)
    __syncthreads();                                // the K loop's last operand reads are done
    for (int i = tid; i < kMaxRgb * BO; i += NTHREADS) {
        const int c = i / BO, o = o0 + i - c * BO;
        ws[i] = (c < e.rgb_n && o < g.O) ? e.rgb_w[o * e.rgb_n + c] * (e.rgb_styles ? e.rgb_styles[b * g.O + o] : 1.f) : 0.f;
    }
    for (int t = tid; t < BO; t += NTHREADS) {
        const int o = o0 + t;
        c_dm[t] = (e.demod && o < g.O) ? e.demod[b * g.O + o] : 1.f;
        c_bs[t] = (e.bias && o < g.O) ? e.bias[o] : 0.f;
    }
    __syncthreads();
    const bool lrelu = e.act == IA_ACT_LRELU;
    int pp[FP];
    bool valid[FP];
    float part[FP][kMaxRgb];
#pragma unroll
    for (int fp = 0; fp < FP; ++fp) {
        const int pr = p0 + (wp * FP + fp) * 32 + l31;
        valid[fp] = pr < npts;
        pp[fp] = valid[fp] ? pr : npts - 1;
#pragma unroll
        for (int c = 0; c < kMaxRgb; ++c) part[fp][c] = 0.f;
    }
    float nz[FP];
#pragma unroll
    for (int fp = 0; fp < FP; ++fp) nz[fp] = e.noise ? e.noise[pp[fp]] : 0.f;
    // channel-major: the ToRGB weights of a channel are read once and meet that channel's value at every pixel of the lane
#pragma unroll
    for (int fo = 0; fo < FO; ++fo)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ol = (wo * FO + fo) * 32 + 8 * q + 4 * half, o_first = o0 + ol;
            float v[FP][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int o = o_first + k;
                float wc[kMaxRgb];
#pragma unroll
                for (int c = 0; c < kMaxRgb; ++c) wc[c] = ws[c * BO + ol + k];
#pragma unroll
                for (int fp = 0; fp < FP; ++fp) {
                    // (this epilogue sits at the 256-register limit: the option tests stay branches here -- as values, store_tile_dual's
                    //  form, the longer straight-line regions spilled 7 - 19 registers)
                    float a = acc[0][fo][fp][4 * q + k] * c_dm[ol + k];              // (the order of epilogue(): demod, noise, bias, ...)
                    if (e.noise) a = fmaf(nz[fp], ns, a);
                    a += c_bs[ol + k];
                    if (lrelu) a = a > 0.f ? a : a * (e.alpha_vec ? e.alpha_vec[min(o, g.O - 1)] : e.alpha);
                    a *= e.gain;
                    if (e.clamp >= 0.f) a = fminf(fmaxf(a, -e.clamp), e.clamp);
                    v[fp][k] = o < g.O ? a : 0.f;
                    if (yb && valid[fp] && o < g.O) yb[(int64_t)o * ohw + pp[fp]] = v[fp][k];
#pragma unroll
                    for (int c = 0; c < kMaxRgb; ++c) part[fp][c] = fmaf(v[fp][k], wc[c], part[fp][c]);
                }
            }
#pragma unroll
            for (int fp = 0; fp < FP; ++fp)
                if (e.ys && valid[fp] && o_first + 3 < g.O) split_store4<false>(e.ys, e.styles_next, e.ys_planes, b, g.O, ohw, o_first, pp[fp], v[fp], watch);
            __builtin_amdgcn_sched_barrier(0);      // keep the next quad's weight / noise / bias loads from being hoisted over this one
        }
#pragma unroll
    for (int fp = 0; fp < FP; ++fp)
#pragma unroll
        for (int c = 0; c < kMaxRgb; ++c) {
            part[fp][c] += __shfl_xor(part[fp][c], 32);
            if (half == 0) red[((((wo * WP + wp) * FP + fp) * kMaxRgb) + c) * 32 + l31] = part[fp][c];
        }
    watch.report();
    __syncthreads();
    if (wo != 0 || half != 0) return;
#pragma unroll
    for (int fp = 0; fp < FP; ++fp) {
        const int p = p0 + (wp * FP + fp) * 32 + l31;
        if (p >= npts) continue;
        for (int c = 0; c < e.rgb_n; ++c) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WO; ++w) s += red[((((w * WP + wp) * FP + fp) * kMaxRgb) + c) * 32 + l31];
            if (e.rgb_bias) s += e.rgb_bias[c];
            if (e.rgb_clamp >= 0.f) s = fminf(fmaxf(s, -e.rgb_clamp), e.rgb_clamp);
            const int64_t at = ((int64_t)b * e.rgb_n + c) * ohw + p;
            if (e.rgb_res) s += e.rgb_res[at];
            e.rgb_out[at] = s;
        }
    }
}

// FO x FP fragments (32 channels x 32 points) per wave, WO x WP waves; 8 input channels per K chunk; JP = patch DMA
// instructions per wave per chunk (host-chosen from the worst window of the launch); SK as in conv_mfma_kernel.
// NP = operand planes: 2 = hi / lo pairs, three products per k-step (fp32-equivalent, the arithmetic of ia_conv2d_mfma_s);
// 1 = one fp16 plane, one product (fp16 operands / fp32 accumulation, the arithmetic of ia_conv2d_mfma_h: the reference's fp16
// blocks) -- the fp16-STORAGE form of the SR head: activations travel between its convolutions as 2 bytes per element.
template <int NP, bool TR, int FO, int FP, int WO, int WP, int JP, bool SK, bool RGB = false, bool APH = false>
__global__ __launch_bounds__(WO * WP * 64, WO * WP == 4 ? 2 : WO * WP / 4) void conv_split_kernel(const h16x8* __restrict__ xs, const h16x8* __restrict__ wk,
                                                                                      float* __restrict__ y, float* __restrict__ slabs, Geo g, Epi e) {
    constexpr int KS = 3, NT = 9, NTP = NT + 1;
    constexpr int NPH = TR ? 4 : 1;
    constexpr int BO = 32 * FO * WO, BP = 32 * FP * WP, NWAVES = WO * WP, NTHREADS = NWAVES * 64;
    constexpr int PAD = TR ? 0 : KS / 2;
    constexpr int NACC = NPH * FO * FP * 16;
    // (pairs are formed inside a segment, so a stream-K range of any length works; the transposed tiles are not MFMA-bound and have no
    // registers to spare for the kept operands: measured no gain on 256 -> 128 @256^2)
    constexpr bool PAIR = !TR && IA_ABLATE == 0 && IA_PAIR_TAPS;
    constexpr int WSLOTS = NP * NTP * BO;             // 16-byte slots of the weight region of a stage: [plane][tap][BO], then one all-zero row per plane
    constexpr int WG = (NP * NT * BO + 63) / 64;       // weight DMA instructions per chunk (64 slots each), spread over the waves
    constexpr int JW = (WG + NWAVES - 1) / NWAVES;
    static_assert(WG * 64 <= WSLOTS, "a DMA instruction fills 64 consecutive slots of the [plane][tap][BO] rows (a last partial one ends in the zero rows)");
    // LDS row of (plane, tap): the DMA'd rows are contiguous (a 64-slot piece may span two rows of a 32-channel tile), the zero taps follow
    auto wrow = [](int pl, int tap) { return tap == kZeroTap ? NP * NT + pl : pl * NT + tap; };
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wo = wave / WP, wp = wave % WP;
    const int b = blockIdx.y;
    int worker = blockIdx.x;
    if (!SK && g.xcd_bands) {
        // Whole-tile launches: workgroups go to the 8 XCDs round-robin, and a tile's input window overlaps its neighbours' (a stride-1
        // tile is one or two image rows + halo rows).  Give XCD x the contiguous band of tiles [start(x), start(x+1)) instead of every
        // eighth tile, so that a halo row is fetched into ONE L2 instead of three (the tile order inside the band follows the dispatch
        // order, so the bands advance together).
        const int T = gridDim.x, per = T / ia::kNumXCD, rem = T - per * ia::kNumXCD;
        const int x = worker % ia::kNumXCD, j = worker / ia::kNumXCD;
        worker = x * per + (x < rem ? x : rem) + j;
    }
    const int npts = g.GH * g.GW;
    const int cap = g.patch_cap;                       // patch positions reserved per plane (multiple of 64)
    const int PG = NP * cap / 64;                      // patch DMA instructions per chunk
    const int stage_bytes = (WSLOTS + NP * cap) * 16;
    const int64_t U = (int64_t)(g.T - g.T_dp) * g.C;
    const int64_t u_begin = SK ? range_begin(worker, U, g.G) : (int64_t)worker * g.C;
    const int64_t u_end = SK ? range_begin(worker + 1, U, g.G) : u_begin + g.C;
    const int first_tile = (int)(u_begin / g.C);
    const int tile_base = SK ? g.T_dp : 0;
    const int HW = g.H * g.W;
    const int plane_bytes = (g.I / 8) * HW * 16;
    const u32x4 rs_x = buffer_rsrc(xs + (int64_t)b * NP * (g.I / 8) * HW, (unsigned)(NP * plane_bytes));
    const u32x4 rs_w = buffer_rsrc(wk, (unsigned)(NP * NT * (g.I / 8) * g.O * 16));
    const unsigned lds_base = (unsigned)(unsigned long long)(lds_char*)lds;

    // the all-zero tap of every plane, in every stage of the ring (the DMA never writes there)
    constexpr int NS = 2;                              // LDS stages of the DMA ring: chunk ch + 1 is in flight under chunk ch (deeper rings measured no gain, r03)
    // The 128-channel x 256-point stride-1 tile (12 MFMAs per k-step and wave): the two wave groups of the workgroup run in antiphase (see
    // the K loop).  The narrow whole-tile families (3 or 6 MFMAs per k-step) take it on request (APH: the host asks for layers of
    // >= 128^2 points; same-box 256 -> 256 @128^2 88 -> 72 us with it, but 512 -> 512 @64^2 79 -> 94 us); the transposed tiles never
    // (8-wave transposed tile 121 -> 601 us).
    constexpr bool PP = IA_ANTIPHASE && NWAVES == 8 && !TR && ((FO == 2 && FP == 2) || APH);
    const int grp = wave >> 2;
    for (int i = tid; i < NS * NP * BO; i += NTHREADS) {
        const int stg = i / (NP * BO), r = i - stg * NP * BO, pl = r / BO, o = r - pl * BO;
        *reinterpret_cast<float4*>(reinterpret_cast<char*>(lds) + stg * stage_bytes + ((NP * NT + pl) * BO + o) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    }

  for (int64_t u = u_begin; u < u_end;) {
    // ---- one segment: tile `tile`, K chunks [c_lo, c_hi)
    const int tile_l = (int)(u / g.C), c_lo = (int)(u - (int64_t)tile_l * g.C);
    const int c_hi = (int)min((int64_t)g.C, (int64_t)c_lo + (u_end - u));
    u += c_hi - c_lo;
    const int tile = tile_base + tile_l;
    const int o0 = (tile % g.TO) * BO;
    const int p0 = (tile / g.TO) * BP;
    const int p_last = min(p0 + BP, npts) - 1;
    const int SD = TR ? 1 : g.stride;                  // stride-2 convolution: point (r, c) sits at window position (2r, 2c)
    const Window win = tile_window(p0, p_last, g.GW, PAD, TR, SD);
    const int PW = win.PW, PSZ = win.PSZ, seg1_off = win.nr[0] * PW;
    const int r_split = (win.nr[1] > 0) ? p_last / g.GW : (1 << 30);
    const float inv_pw = 1.0f / (float)PW;

    int bpos[FP];
#pragma unroll
    for (int fp = 0; fp < FP; ++fp) {
        const int p = min(p0 + (wp * FP + fp) * 32 + l31, p_last);
        const int r = p / g.GW, c = p - r * g.GW;
        const int sg = (r == r_split) ? 1 : 0;
        bpos[fp] = sg * seg1_off + (SD * r - PAD - win.r0[sg]) * PW + (SD * c - PAD - win.c0[sg]);
    }
    int toff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ky = t / KS, kx = t % KS;
        toff[t] = TR ? -((ky >> 1) * PW + (kx >> 1)) : ky * PW + kx;
    }

    f32x16 acc[NPH][FO][FP];
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
        for (int fo = 0; fo < FO; ++fo)
#pragma unroll
            for (int fp = 0; fp < FP; ++fp)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ph][fo][fp][r] = 0.f;

    // ---- DMA plan of this tile: per-lane source byte offsets (chunk 0), fixed for the whole K loop; a chunk adds an SGPR offset.
    // Weight instruction j of this wave covers load slots [(j*NWAVES + wave)*64, +64) of the [plane][tap][BO] slab of a chunk.
    constexpr int kOutside = 0x7ffffff0;
    int w_voff[JW], p_voff[JP];
#pragma unroll
    for (int j = 0; j < JW; ++j) {
        const int e_ = (j * NWAVES + wave) * 64 + lane;                      // (slots past the last row: zeros into the zero rows)
        const int row = e_ / BO, o = e_ - row * BO;                         // row = plane*NT + tap
        w_voff[j] = e_ < NP * NT * BO ? ((row * (g.I / 8)) * g.O + min(o0 + o, g.O - 1)) * 16 : kOutside;
    }
#pragma unroll
    for (int j = 0; j < JP; ++j) {
        const int q = (j * NWAVES + wave) * 64 + lane;
        const int pl = q >= cap ? 1 : 0, pp = q - pl * cap;
        const int sg = (pp >= seg1_off && win.nr[1] > 0) ? 1 : 0;
        const int qq = pp - sg * seg1_off;
        const int pr = (int)(((float)qq + 0.5f) * inv_pw), pc = qq - pr * PW;
        const int iy = win.r0[sg] + pr, ix = win.c0[sg] + pc;
        const bool ok = pp < PSZ && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
        p_voff[j] = ok ? pl * plane_bytes + (iy * g.W + ix) * 16 : kOutside;
    }
    // every DMA of one chunk into one LDS stage.  (The voffset argument goes through a plain local: hipcc 7.2 silently drops the HOST
    // stub of a kernel template that passes an element of a template-sized array straight to the raw_ptr_buffer_load_lds builtin
    // -- the .so then fails to load with an undefined kernel symbol.)
    // Slice `sl` of `nsl` of the chunk's pieces (piece q of this wave's list -- weights first, then patch -- belongs to slice q % nsl);
    // nsl = 1 issues the whole chunk.
#define IA_ISSUE_DMA_SLICE(chunk, stage, sl, nsl)                                                                                       \
    do {                                                                                                                               \
        const unsigned st_ = lds_base + (stage) * stage_bytes;                                                                         \
        const int wso_ = (chunk) * g.O * 16, pso_ = (chunk) * HW * 16;                                                                 \
        _Pragma("unroll") for (int j = 0; j < JW; ++j) {                                                                               \
            const int gidx = j * NWAVES + wave;                                                                                        \
            if (j % (nsl) == (sl) && gidx < WG) {                                                                                      \
                const int vo_ = w_voff[j];                                                                                             \
                dma_piece(rs_w, st_ + gidx * 64 * 16, vo_, wso_);                                                                      \
            }                                                                                                                          \
        }                                                                                                                              \
        _Pragma("unroll") for (int j = 0; j < JP; ++j) {                                                                               \
            const int gidx = j * NWAVES + wave;                                                                                        \
            const int vo_ = p_voff[j];                                                                                                 \
            if ((JW + j) % (nsl) == (sl) && gidx < PG) dma_piece(rs_x, st_ + (WSLOTS + gidx * 64) * 16, vo_, pso_);                    \
        }                                                                                                                              \
    } while (0)
#define IA_ISSUE_DMA(chunk, stage) IA_ISSUE_DMA_SLICE(chunk, stage, 0, 1)

    // DMA instructions this wave issues per chunk (wave-uniform): what one chunk adds to its vmcnt
    int n_dma = 0;
#pragma unroll
    for (int j = 0; j < JW; ++j) n_dma += (j * NWAVES + wave < WG) ? 1 : 0;
#pragma unroll
    for (int j = 0; j < JP; ++j) n_dma += (j * NWAVES + wave < PG) ? 1 : 0;
    // The previous segment's epilogue loads / stores are drained with a wait the compiler can SEE (vmcnt(0), gfx9 encoding): its
    // wait-count pass cannot look inside the inline-assembly DMA, so anything it still believes pending when it meets a register
    // re-use in the K loop would be waited for with vmcnt(0) there -- draining the ring on every chunk.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();                         // the previous segment's readers (and the zero-tap stores) are done
    // PAIR (whole-tile launches): the odd tap out of a chunk's nine (tap 8; tap 4 in the transposed form) does not meet an all-zero tap
    // in its k-step -- 10 % of the MFMAs multiplying zeros -- but the same tap of the NEXT chunk: lanes 0-31 read their operands at the
    // even chunk of a pair and keep them, lanes 32-63 read theirs at the odd chunk into the same registers (exec-masked LDS reads),
    // and the k-step runs once per pair: 9 k-steps per 16 input channels instead of 10.  Only the order of the fp32 additions changes.
    h16x8 a_hold[NP * FO], b_hold[NP * FP];
#pragma unroll
    for (int q = 0; q < NP * FO; ++q) a_hold[q] = h16x8{};
#pragma unroll
    for (int q = 0; q < NP * FP; ++q) b_hold[q] = h16x8{};
    if constexpr (PP) {
        // ---- K loop of the 8-wave tiles: two wave groups in antiphase (r04; the structure of up_rows_kernel, csrc/conv_up.hip, where the
        // segment trace and the ablations are).  Waves w and w + 4 share a SIMD; group 1 (waves 4-7) runs ONE barrier interval behind
        // group 0, so in every interval one wave of a SIMD is in a LOAD segment (a k-step's operand reads + its share of the refill DMA)
        // while the other is in the COMPUTE segment of its k-step (12 MFMAs): the matrix pipe always has a wave feeding it.  With all
        // eight waves in step (r02 / r03) both waves of a SIMD issued DMA and waited for their reads together, then queued for the pipe
        // together -- the MFMAs' time was added to everything else, not overlapped.
        //   * two LDS stages: chunk ch + 1 is DMA'd into the other stage during the first two LOAD segments of chunk ch (every wave
        //     waits for its operand reads -- lgkmcnt(0) -- in front of the barrier that ends a LOAD segment, so the lagging group's reads
        //     of chunk ch - 1 have returned before the leading group's first refill piece is issued), and waited for (vmcnt(0): nothing
        //     younger is in flight) in the chunk's last LOAD segment, in front of the barrier after which the leading group reads it;
        //   * every wave executes the same number of barriers (group 0 one more at the end).
        auto pp_barrier = []() { asm volatile("s_barrier" ::: "memory"); };
        auto mma_step = [&](const h16x8 (&a_use)[NP * FO], const h16x8 (&b_use)[NP * FP], int ph_) {
#if IA_ABLATE == 2
#pragma unroll
            for (int q = 0; q < NP * FO; ++q) asm volatile("" ::"v"(a_use[q]));
#pragma unroll
            for (int q = 0; q < NP * FP; ++q) asm volatile("" ::"v"(b_use[q]));
            return;
#endif
            __builtin_amdgcn_s_setprio(1);
            if constexpr (NP == 2) {
#pragma unroll
                for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                    for (int fp = 0; fp < FP; ++fp)      // lo * hi
                        acc[ph_][fo][fp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_use[(NP - 1) * FO + fo], b_use[fp], acc[ph_][fo][fp], 0, 0, 0);
                h16x8 a_sc[FO];                        // weight high parts at 2^-11: they meet the activation's low parts (scaled by 2^11)
#pragma unroll
                for (int fo = 0; fo < FO; ++fo) a_sc[fo] = a_use[fo] * (_Float16)(1.0f / kLoScale);
#pragma unroll
                for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                    for (int fp = 0; fp < FP; ++fp)      // (hi * 2^-11) * (lo * 2^11)
                        acc[ph_][fo][fp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_sc[fo], b_use[(NP - 1) * FP + fp], acc[ph_][fo][fp], 0, 0, 0);
            }
#pragma unroll
            for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                for (int fp = 0; fp < FP; ++fp)          // hi * hi
                    acc[ph_][fo][fp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_use[fo], b_use[fp], acc[ph_][fo][fp], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        if (c_lo < c_hi) IA_ISSUE_DMA(c_lo, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pp_barrier();
        if (grp) pp_barrier();
        int cur = 0;
        for (int ch = c_lo; ch < c_hi; ++ch) {
            const bool fill = !IA_AB_NODMA && ch + 1 < c_hi;
            const h16x8* wh = reinterpret_cast<const h16x8*>(reinterpret_cast<const char*>(lds) + cur * stage_bytes);
            const h16x8* ph = wh + WSLOTS;
            const int par = (ch - c_lo) & 1;                       // position of this chunk in its pair
            const bool run_odd_tap = !PAIR || par == 1 || ch == c_hi - 1;
            const int last_s = (PAIR && !run_odd_tap) ? kPairs - 2 : kPairs - 1;
#pragma unroll
            for (int s = 0; s < kPairs; ++s) {
                const bool hold_step = PAIR && s == kPairs - 1;      // the odd tap of a pair of chunks: operands kept in a_hold / b_hold
                if (hold_step && !run_odd_tap) break;
                // LOAD segment
                h16x8 a_use[NP * FO], b_use[NP * FP];
                if (!hold_step) {
                    const int tap = half ? pair_t1(TR, s) : pair_t0(TR, s);
                    const int tof = pair_t1(TR, s) == kZeroTap ? (half ? 0 : toff[pair_t0(TR, s)]) : (half ? toff[pair_t1(TR, s)] : toff[pair_t0(TR, s)]);
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                        for (int fo = 0; fo < FO; ++fo) a_use[pl * FO + fo] = wh[wrow(pl, tap) * BO + (wo * FO + fo) * 32 + l31];
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                        for (int fp = 0; fp < FP; ++fp) b_use[pl * FP + fp] = ph[pl * cap + bpos[fp] + tof];
                }
                if (PAIR && s == kPairs - 2) {                       // this chunk's odd tap, into this chunk's half of the lanes
                    constexpr int tap8 = pair_t0(TR, kPairs - 1);
                    if (half == par) {
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                            for (int fo = 0; fo < FO; ++fo) a_hold[pl * FO + fo] = wh[wrow(pl, tap8) * BO + (wo * FO + fo) * 32 + l31];
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                            for (int fp = 0; fp < FP; ++fp) b_hold[pl * FP + fp] = ph[pl * cap + bpos[fp] + toff[tap8]];
                    } else if (par == 0 && ch == c_hi - 1) {         // a last chunk without a partner: its upper lanes multiply zeros
#pragma unroll
                        for (int q = 0; q < NP * FO; ++q) a_hold[q] = h16x8{};
#pragma unroll
                        for (int q = 0; q < NP * FP; ++q) b_hold[q] = h16x8{};
                    }
                }
                if (fill && s < 2) IA_ISSUE_DMA_SLICE(ch + 1, cur ^ 1, s, 2);
                if (fill && s == last_s) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // chunk ch + 1 has landed (this wave's pieces)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads have RETURNED before the barrier (the other group may refill their stage behind it)
                __builtin_amdgcn_sched_barrier(0);
                pp_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // COMPUTE segment
                if (hold_step) mma_step(a_hold, b_hold, pair_phase(TR, s));
                else mma_step(a_use, b_use, pair_phase(TR, s));
                __builtin_amdgcn_sched_barrier(0);
                pp_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            cur ^= 1;
        }
        if (!grp) pp_barrier();
    } else {
    int issued = c_lo;                       // next chunk to issue; chunk c lives in stage (c - c_lo) % NS
    for (int k = 0; k < NS - 1 && issued < c_hi; ++k, ++issued) IA_ISSUE_DMA(issued, k);
    int cur = 0;
    for (int ch = c_lo; ch < c_hi; ++ch) {
        wait_vmcnt((issued - ch - 1) * n_dma);                // this wave's DMAs of chunk `ch` have landed (younger chunks stay in flight) ...
        __builtin_amdgcn_s_barrier();                         // ... and everybody's; the stage of chunk ch - 1 has no readers left
        asm volatile("" ::: "memory");                        // (s_barrier alone is no compiler fence: keep the operand reads below it)
        // The chunk that goes into the stage just freed (4-wave tiles: two workgroups per CU overlap each other's phases).
        const bool fill = !IA_AB_NODMA && issued < c_hi;
        const int fill_chunk = issued, fill_stage = cur == 0 ? NS - 1 : cur - 1;
        if (fill) {
            IA_ISSUE_DMA(fill_chunk, fill_stage);
            ++issued;
        }
        const h16x8* wh = reinterpret_cast<const h16x8*>(reinterpret_cast<const char*>(lds) + cur * stage_bytes);
        const h16x8* ph = wh + WSLOTS;
        // five k-steps: the 8 channels of a pair of taps (lanes 0-31 the first tap, lanes 32-63 the second).  Operand reads run one
        // k-step ahead of the MFMAs that consume them.
        h16x8 a_buf[2][NP * FO], b_buf[2][NP * FP];
        auto load_ops = [&](int s, h16x8 (&a)[NP * FO], h16x8 (&bv)[NP * FP]) {
            const int tap = half ? pair_t1(TR, s) : pair_t0(TR, s);
            const int tof = pair_t1(TR, s) == kZeroTap ? (half ? 0 : toff[pair_t0(TR, s)]) : (half ? toff[pair_t1(TR, s)] : toff[pair_t0(TR, s)]);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int fo = 0; fo < FO; ++fo) a[pl * FO + fo] = wh[wrow(pl, tap) * BO + (wo * FO + fo) * 32 + l31];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int fp = 0; fp < FP; ++fp) bv[pl * FP + fp] = ph[pl * cap + bpos[fp] + tof];
        };
        const int par = (ch - c_lo) & 1;                       // position of this chunk in its pair
        const bool run_odd_tap = !PAIR || par == 1 || ch == c_hi - 1;
        auto load_hold = [&]() {                                 // the odd tap's operands of this chunk, into this chunk's half of the lanes
            constexpr int tap8 = pair_t0(TR, kPairs - 1);
            if (half == par) {
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                    for (int fo = 0; fo < FO; ++fo) a_hold[pl * FO + fo] = wh[wrow(pl, tap8) * BO + (wo * FO + fo) * 32 + l31];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                    for (int fp = 0; fp < FP; ++fp) b_hold[pl * FP + fp] = ph[pl * cap + bpos[fp] + toff[tap8]];
            } else if (par == 0 && ch == c_hi - 1) {             // a last chunk without a partner: its upper lanes multiply zeros
#pragma unroll
                for (int q = 0; q < NP * FO; ++q) a_hold[q] = h16x8{};
#pragma unroll
                for (int q = 0; q < NP * FP; ++q) b_hold[q] = h16x8{};
            }
        };
        if (!IA_AB_NOREAD || ch == c_lo) load_ops(0, a_buf[0], b_buf[0]);
#pragma unroll
        for (int s = 0; s < kPairs; ++s) {
            const int c_ = s & 1;
            // the NEXT k-step's operand reads are issued before this k-step's MFMAs and pinned there (sched_barrier): their LDS
            // latency then hides under 12 x 32 MFMA cycles; left to itself the scheduler sinks them to just before their use
            if constexpr (PAIR) {
                if (s + 2 < kPairs) load_ops(s + 1, a_buf[(s + 1) & 1], b_buf[(s + 1) & 1]);
                else if (s + 2 == kPairs) load_hold();
            } else if ((!IA_AB_NOREAD || ch == c_lo) && s + 1 < kPairs) load_ops(s + 1, a_buf[(s + 1) & 1], b_buf[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (PAIR && s == kPairs - 1 && !run_odd_tap) {        // the even chunk of a pair: its odd tap waits for the partner's
                continue;
            }
            const h16x8 (&a_use)[NP * FO] = (PAIR && s == kPairs - 1) ? a_hold : a_buf[c_];
            const h16x8 (&b_use)[NP * FP] = (PAIR && s == kPairs - 1) ? b_hold : b_buf[c_];
#if IA_ABLATE == 2
#pragma unroll
            for (int q = 0; q < NP * FO; ++q) asm volatile("" ::"v"(a_use[q]));
#pragma unroll
            for (int q = 0; q < NP * FP; ++q) asm volatile("" ::"v"(b_use[q]));
            continue;
#endif
            const int ph_ = pair_phase(TR, s);
            if constexpr (NP == 2) {
                h16x8 a_sc[FO];                        // weight high parts at 2^-11: they meet the activation's low parts (scaled by 2^11)
#pragma unroll
                for (int fo = 0; fo < FO; ++fo) a_sc[fo] = IA_ABLATE >= 6 ? a_use[fo] : a_use[fo] * (_Float16)(1.0f / kLoScale);
                // three products per fragment pair, product-major: consecutive MFMAs write different accumulators
#pragma unroll
                for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                    for (int fp = 0; fp < FP; ++fp)      // lo * hi
                        acc[ph_][fo][fp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_use[(NP - 1) * FO + fo], b_use[fp], acc[ph_][fo][fp], 0, 0, 0);
#pragma unroll
                for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                    for (int fp = 0; fp < FP; ++fp)      // (hi * 2^-11) * (lo * 2^11)
                        acc[ph_][fo][fp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_sc[fo], b_use[(NP - 1) * FP + fp], acc[ph_][fo][fp], 0, 0, 0);
            }
#pragma unroll
            for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                for (int fp = 0; fp < FP; ++fp)          // hi * hi
                    acc[ph_][fo][fp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_use[fo], b_use[fp], acc[ph_][fo][fp], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // (the kept operands were read from the stage that is refilled after the next barrier: they are in registers before it)
        if constexpr (PAIR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        cur = cur + 1 == NS ? 0 : cur + 1;
    }
    }

#pragma unroll
    for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
        for (int fo = 0; fo < FO; ++fo)
#pragma unroll
            for (int fp = 0; fp < FP; ++fp)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ph][fo][fp][r] *= g.acc_scale;   // back from the scale of the packed weights (exact; 1 for NP = 1)
    if (IA_AB_NOSTORE) {
        float sink = 0.f;
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
            for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                for (int fp = 0; fp < FP; ++fp)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sink += acc[ph][fo][fp][r];
        if (sink == 123.456f) y[0] = sink;
    } else if (!SK || (c_lo == 0 && c_hi == g.C)) {
        if constexpr (TR) store_tile<TR, FO, FP, WO, WP>(acc, y, g, e, b, o0, p0, tid);      // (demodulation only: by value; the reload measured 4 - 7 % slower here)
        else {
#if IA_EPI_RELOAD
            // the epilogue descriptor comes back from the kernarg segment here (see reload_kernarg): nothing of `e` is live in the K loop.
            // r06, same box: 37 - 56 fewer SGPR spills per stride-1 instantiation (main tile 96 -> 59, antiphase narrow tile 56 -> 17), layer
            // times unchanged within 0.5 % (profiles/r06_epilogue_descriptor_reload.txt): the spills were never on the critical path.
            struct KArgs { const h16x8* xs; const h16x8* wk; float* y; float* slabs; Geo g; Epi e; };
            static_assert(sizeof(Epi) % 16 == 0 && offsetof(KArgs, e) % 8 == 0, "Epi travels as whole 16-byte groups");
            const Epi e_now = reload_kernarg<Epi, offsetof(KArgs, e)>();
#else
            const Epi& e_now = e;
#endif
            if constexpr (RGB) store_tile_rgb<FO, FP, WO, WP>(acc, y, g, e_now, b, o0, p0, tid, lds);
            else store_tile_dual<FO, FP, WO, WP>(acc, y, g, e_now, b, o0, p0, tid, lds);
        }
    } else if constexpr (SK) {
        const int slot = (tile_l == first_tile) ? 0 : 1;
        float4* slab = reinterpret_cast<float4*>(slabs + (((int64_t)b * g.G + worker) * 2 + slot) * ((int64_t)NACC * NTHREADS)) + tid;
#pragma unroll
        for (int q = 0; q < NACC / 4; ++q) {
            const int fr = q >> 2, r0 = (q & 3) * 4;
            const f32x16& a = acc[fr / (FP * FO)][(fr / FP) % FO][fr % FP];
            slab[(int64_t)q * NTHREADS] = make_float4(a[r0], a[r0 + 1], a[r0 + 2], a[r0 + 3]);
        }
    }
  }   // segments of this worker
#undef IA_ISSUE_DMA
#undef IA_ISSUE_DMA_SLICE
}

template <int NP, bool TR, int FO, int FP, int WO, int WP, int JP, bool WHOLE = false, bool APH = false>
int launch_jp(const h16x8* xs, const h16x8* wk, float* y, float* scratch, const Geo& g_in, const Epi& e, hipStream_t s) {
    constexpr int BO = 32 * FO * WO, NWAVES = WO * WP;
    const size_t stage = (size_t)(NP * 10 * BO + NP * g_in.patch_cap) * 16;
    if (2 * stage > kLdsBytes) return ia::fail(IA_ERR_UNSUPPORTED, "conv tile needs %zu bytes of LDS", 2 * stage);
    // Two LDS stages.  (r03: with the ring really running ahead -- no compiler-inserted drain, see dma_piece -- 3 .. 6 stages changed no
    // layer by more than the run-to-run noise, while the 4-wave tiles lose 35-55 % when a deeper ring takes the LDS of the second workgroup
    // of their CU; the experiment switches IA_RING_STAGES / IA_DMA_SPREAD were removed in r04.)
    Geo g = g_in;
    constexpr int ns = 2;
    g.xcd_bands = (g.B == 1 || g.T_dp % ia::kNumXCD == 0) ? 1 : 0;      // (the linear workgroup id of batch element b starts at b * T_dp)
    const size_t lds = stage * ns;
    int st = IA_OK;
    if (g.T_dp > 0) {
        if constexpr (!TR && !WHOLE) {
            if (e.rgb_out) {      // (the fused ToRGB epilogue is its own instantiation: it costs the plain kernel registers otherwise)
                auto k = conv_split_kernel<NP, TR, FO, FP, WO, WP, JP, false, true>;
                if (const int rs = ia::reserve_lds((const void*)k, (size_t)(lds), "conv_split")) return rs;
                hipLaunchKernelGGL(k, dim3(g.T_dp, g.B), dim3(WO * WP * 64), lds, s, xs, wk, y, scratch, g, e);
                return ia::check_launch("ia_conv2d_mfma_sx_rgb");
            }
        }
        auto k = conv_split_kernel<NP, TR, FO, FP, WO, WP, JP, false, false, APH>;
        if (const int rs = ia::reserve_lds((const void*)k, (size_t)(lds), "conv_split")) return rs;
        hipLaunchKernelGGL(k, dim3(g.T_dp, g.B), dim3(WO * WP * 64), lds, s, xs, wk, y, scratch, g, e);
        st = ia::check_launch("ia_conv2d_mfma_sx");
    }
    if constexpr (WHOLE) {
        if (g.T > g.T_dp) return ia::fail(IA_ERR_UNSUPPORTED, "this tile family runs whole tiles only");
    } else if (st == IA_OK && g.T > g.T_dp) {
        auto k = conv_split_kernel<NP, TR, FO, FP, WO, WP, JP, true>;
        if (const int rs = ia::reserve_lds((const void*)k, (size_t)(lds), "conv_split")) return rs;
        hipLaunchKernelGGL(k, dim3(g.G, g.B), dim3(WO * WP * 64), lds, s, xs, wk, y, scratch, g, e);
        st = ia::check_launch("ia_conv2d_mfma_sx(stream-K)");
        const int64_t U = (int64_t)(g.T - g.T_dp) * g.C;
        const bool whole_tiles = U % g.G == 0 && (U / g.G) % g.C == 0;
        if (st == IA_OK && !whole_tiles) {
            constexpr int NACC = (TR ? 4 : 1) * FO * FP * 16;
            hipLaunchKernelGGL((conv_fixup_kernel<TR, FO, FP, WO, WP>), dim3(g.T - g.T_dp, g.B, NACC / (TR ? 8 : 4)), dim3(WO * WP * 64), 0, s,
                               scratch, y, g, e);
            st = ia::check_launch("ia_conv2d_mfma_sx(fix-up)");
        }
    }
    return st;
}

template <int NP, bool TR, int FO, int FP, int WO, int WP, bool WHOLE = false, bool APH = false>
int launch_sx(const h16x8* xs, const h16x8* wk, float* y, float* scratch, const Geo& g_in, const Epi& e, hipStream_t s) {
    constexpr int BP = 32 * FP * WP, NWAVES = WO * WP;
    Geo g = g_in;
    const int npts = g.GH * g.GW;
    int worst = 0;
    for (int q0 = 0; q0 < npts; q0 += BP) {
        const int q1 = (q0 + BP < npts ? q0 + BP : npts) - 1;
        const Window w = tile_window(q0, q1, g.GW, TR ? 0 : 1, TR, g.stride);
        if (w.PSZ > worst) worst = w.PSZ;
    }
    // stride 1: the planner's budget (it chose this tile family knowing that its windows fit); stride 2: whatever the two LDS stages hold
    // beside the weight rows (launch_jp checks) and eight patch DMA instructions per wave reach
    if (worst > (g.stride == 1 ? kPatchFloats : 8 * NWAVES * 64 / NP))
        return ia::fail(IA_ERR_UNSUPPORTED, "conv tile patch of %d positions exceeds the LDS budget", worst);
    g.patch_cap = (worst + 63) & ~63;
    const int per_wave = (NP * g.patch_cap / 64 + NWAVES - 1) / NWAVES;
    if (per_wave <= 2) return launch_jp<NP, TR, FO, FP, WO, WP, 2, WHOLE, APH>(xs, wk, y, scratch, g, e, s);
    if (per_wave <= 4) return launch_jp<NP, TR, FO, FP, WO, WP, 4, WHOLE, APH>(xs, wk, y, scratch, g, e, s);
    return launch_jp<NP, TR, FO, FP, WO, WP, 8, WHOLE, APH>(xs, wk, y, scratch, g, e, s);
}

}  // namespace

extern "C" int ia_act_split(const float* x, const float* styles, const float* shift, void* xs, int planes, int B, int C, int H, int W, void* stream) {
    IA_REQUIRE(planes == 1 || planes == 2, "planes: 2 = hi / lo pair, 1 = one fp16 plane");
    IA_REQUIRE(x && xs, "null pointer argument");
    IA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "empty tensor");
    IA_REQUIRE(C % 8 == 0, "the split format stores channels in groups of 8 (C = %d)", C);
    IA_REQUIRE((int64_t)B * C * H * W <= INT32_MAX, "tensor is too large");
    const int64_t work = (int64_t)B * (C / 8) * H * W;
    if (((int64_t)H * W) % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
        hipLaunchKernelGGL(act_split4_kernel, dim3(ia::streaming_grid(work / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, styles, shift,
                           static_cast<h16x8*>(xs), B, C, (int64_t)H * W, planes);
    else
        hipLaunchKernelGGL(act_split_kernel, dim3(ia::streaming_grid(work, 256)), dim3(256), 0, (hipStream_t)stream, x, styles, shift,
                           static_cast<h16x8*>(xs), B, C, (int64_t)H * W, planes);
    return ia::check_launch("ia_act_split");
}

extern "C" int ia_bn_train_split(const float* x, const float* weight, const float* bias, float* running_mean, float* running_var,
                                 long long* num_batches_tracked, double* partials, int chunks, void* xs, int planes, int B, int C, int H, int W,
                                 float eps, float momentum, void* stream) {
    IA_REQUIRE(planes == 1 || planes == 2, "planes: 2 = hi / lo pair, 1 = one fp16 plane");
    IA_REQUIRE(x && xs && partials, "null pointer argument");
    IA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "empty tensor");
    IA_REQUIRE(C % 8 == 0, "the split format stores channels in groups of 8 (C = %d)", C);
    IA_REQUIRE((int64_t)B * H * W > 1, "batch statistics need more than one value per channel");
    IA_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "running_mean and running_var come together");
    IA_REQUIRE(chunks >= 1 && chunks <= 1024, "chunks: 1 .. 1024 partial sums per channel");
    IA_REQUIRE((int64_t)B * C * H * W <= INT32_MAX && B <= 65535, "tensor is too large");
    const int64_t HW = (int64_t)H * W;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_partial_kernel, dim3((unsigned)C, (unsigned)chunks), dim3(256), 0, s, x, partials, B, C, HW, chunks);
    int st = ia::check_launch("ia_bn_train_split(statistics)");
    if (st != IA_OK) return st;
    const bool quad = HW % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const int64_t items = quad ? HW / 4 : HW;
    int gy = (int)((items + 255) / 256);
    const int want = (4 * ia::kNumCU + (C / 8) * B - 1) / ((C / 8) * B);       // ~4 workgroups per CU over the whole launch
    if (gy > want) gy = want < 1 ? 1 : want;
    const dim3 grid((unsigned)(C / 8), (unsigned)gy, (unsigned)B);
    if (quad)
        hipLaunchKernelGGL(bn_split_kernel<4>, grid, dim3(256), 0, s, x, partials, weight, bias, running_mean, running_var, num_batches_tracked,
                           static_cast<h16x8*>(xs), B, C, HW, chunks, eps, momentum, planes);
    else
        hipLaunchKernelGGL(bn_split_kernel<1>, grid, dim3(256), 0, s, x, partials, weight, bias, running_mean, running_var, num_batches_tracked,
                           static_cast<h16x8*>(xs), B, C, HW, chunks, eps, momentum, planes);
    return ia::check_launch("ia_bn_train_split");
}

extern "C" int ia_split_saturation_count(const void* xs, int planes, int B, int C, int H, int W, unsigned int* count, void* stream) {
    IA_REQUIRE(xs && count, "null pointer argument");
    IA_REQUIRE((planes == 1 || planes == 2) && B > 0 && C > 0 && C % 8 == 0 && H > 0 && W > 0, "not a split tensor shape");
    const int64_t per_plane = (int64_t)C * H * W;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(count, 0, sizeof(unsigned int), s) != hipSuccess) return ia::fail(IA_ERR_LAUNCH, "hipMemsetAsync failed");
    hipLaunchKernelGGL(split_saturation_kernel, dim3(ia::streaming_grid((int64_t)B * per_plane, 256)), dim3(256), 0, s,
                       static_cast<const unsigned short*>(xs), per_plane, (int64_t)planes * per_plane, B, count);
    return ia::check_launch("ia_split_saturation_count");
}

// (make_plan / tile selection live in conv_mfma.hip: both forms of a layer share tiles, worker counts and slab sizes)
int ia_conv2d_plan_tiles(int B, int I, int O, int H, int W, int ksize, int transposed, int form, int stride, int* bo, int* bp, int* waves, int* T,
                         int* TO, int* C, int* T_dp, int* slab_floats);

struct RgbArgs { const float* w; const float* styles; const float* bias; const float* res; float* out; int n; float clamp; };

static int conv_sx_impl(const void* xs, int planes, const void* wk_split, int wk_exp, const float* demod, const float* noise,
                        const float* noise_strength, const float* bias, const float* residual, float* y, void* ys, int ys_planes,
                        const float* styles_next, float* scratch, size_t scratch_bytes, int B, int I, int O, int H, int W,
                        int transposed, int act, float alpha, float gain, float clamp, int ksplit, void* stream, const RgbArgs& rgb,
                        const float* prelu_alpha = nullptr, int d2s = 0, int stride = 1) {
    IA_REQUIRE(xs && wk_split && (y || ys || rgb.out), "xs, wk and at least one output must be device pointers");
    IA_REQUIRE(stride == 1 || (stride == 2 && !transposed && !d2s && !rgb.out && noise == nullptr), "stride 2: the plain 3x3 convolution with padding 1");
    IA_REQUIRE(B > 0 && I > 0 && O > 0 && H > 0 && W > 0, "empty tensor");
    IA_REQUIRE(I % 8 == 0 && O % 8 == 0, "the split form needs I %% 8 == 0 and O %% 8 == 0");
    IA_REQUIRE(wk_exp >= -14 && wk_exp <= 30, "wk_exp is the power of two the weights were scaled by at pack time");
    IA_REQUIRE((planes == 1 && wk_exp == 0) || planes == 2, "planes: 2 = hi / lo pairs (weights from pack_conv_weight_split), 1 = fp16 operands (wk_exp 0)");
    IA_REQUIRE(!ys || ys_planes == 1 || ys_planes == 2, "ys_planes must be 1 or 2");
    IA_REQUIRE(act == IA_ACT_LINEAR || act == IA_ACT_LRELU, "conv epilogue supports linear and lrelu");
    IA_REQUIRE(ksplit >= 0, "worker count must be >= 0");
    IA_REQUIRE(!transposed || (y && !ys && noise == nullptr && bias == nullptr && residual == nullptr && act == IA_ACT_LINEAR),
               "the transposed form writes the fp32 (2H+1)x(2W+1) image only; FIR + bias_act (+ split) follow in ia_fir_tail_split");
    Geo g;
    g.B = B; g.I = I; g.O = O; g.H = H; g.W = W;
    g.GH = transposed ? H + 1 : H; g.GW = transposed ? W + 1 : W;
    g.OH = transposed ? 2 * H + 1 : H; g.OW = transposed ? 2 * W + 1 : W;
    g.stride = stride;
    if (stride == 2) { g.GH = g.OH = (H - 1) / 2 + 1; g.GW = g.OW = (W - 1) / 2 + 1; }
    IA_REQUIRE((int64_t)B * O * g.OH * g.OW <= INT32_MAX && (int64_t)B * I * H * W <= INT32_MAX, "tensor is too large");
    IA_REQUIRE(!d2s || (!transposed && O % 32 == 0 && residual == nullptr && !rgb.out && !prelu_alpha),
               "the depth-to-space store takes a stride-1 layer of 4 x O/4 channels without residual / fused ToRGB");
    int bo, bp, waves, slab_floats;
    const int st_plan = ia_conv2d_plan_tiles(B, I, O, stride == 2 ? g.GH : H, stride == 2 ? g.GW : W, 3, transposed, 3, stride, &bo, &bp, &waves, &g.T, &g.TO,
                                             &g.C, &g.T_dp, &slab_floats);
    if (st_plan != IA_OK) return st_plan;
    const bool wide = waves == 8;
    IA_REQUIRE(wide || (transposed && bo == 64), "the split form covers 3x3 layers on the two-stage tiles (large stride-1 layers, stride-2 transposed)");
    const bool narrow = wide && bp == 256 && (bo == 32 || (bo == 64 && !transposed));      // whole-tile families of the mid-sized layers
    IA_REQUIRE(!narrow || g.T_dp == g.T, "the narrow tile families run whole tiles only");
    if (rgb.out && narrow) return ia::fail(IA_ERR_UNSUPPORTED, "the fused ToRGB needs the 128-channel tile");
    if (rgb.out && (transposed || g.TO != 1 || g.T_dp != g.T))
        return ia::fail(IA_ERR_UNSUPPORTED, "the fused ToRGB needs a stride-1 layer whose tiles hold every output channel and run in whole rounds "
                        "(O %d, %d channel tiles, %d of %d tiles in whole rounds)", O, g.TO, g.T_dp, g.T);
    g.G = 0;
    const bool small = g.T_dp < g.T && conv_small_shape(B, I, O, H, W, 3, transposed, stride) && !rgb.out && !d2s;
    if (g.T_dp < g.T && !small) {
        IA_REQUIRE(ksplit >= 1, "this layer has stream-K tiles: pass the worker count from ia_conv2d_plan");
        const int64_t Ur = (int64_t)(g.T - g.T_dp) * g.C;
        g.G = (int)(ksplit > Ur ? Ur : ksplit);
        const size_t need = (size_t)B * g.G * 2 * slab_floats * sizeof(float);
        const bool whole_tiles = Ur % g.G == 0 && (Ur / g.G) % g.C == 0;
        IA_REQUIRE(whole_tiles || (scratch && scratch_bytes >= need), "stream-K needs %zu bytes of scratch, got %zu", need, scratch_bytes);
    }
    g.patch_cap = 0;
    g.acc_scale = ldexpf(1.f, -wk_exp);
    Epi e{demod, noise, noise_strength, bias, residual, act, alpha, gain, clamp, ys, styles_next, ys_planes};
    e.alpha_vec = prelu_alpha;
    e.d2s = d2s;
    if (d2s && (bo != 128 || O % 128 != 0 || g.T_dp != g.T))
        return ia::fail(IA_ERR_UNSUPPORTED, "depth-to-space store: needs whole 128-channel tiles (4 x %d channels, %d-channel tiles, %d of %d tiles whole)",
                        O / 4, bo, g.T_dp, g.T);
    if (rgb.out) {
        IA_REQUIRE(rgb.w && rgb.n >= 1 && rgb.n <= kMaxRgb, "the fused ToRGB takes 1 .. %d output channels and its packed weight", kMaxRgb);
        e.rgb_w = rgb.w; e.rgb_styles = rgb.styles; e.rgb_bias = rgb.bias; e.rgb_res = rgb.res; e.rgb_out = rgb.out; e.rgb_n = rgb.n; e.rgb_clamp = rgb.clamp;
    }
    hipStream_t s = (hipStream_t)stream;
    if (small) return conv_small_launch(xs, planes, wk_split, y, g, e, transposed != 0, s);      // low-resolution layers: K split inside the workgroup, no slabs, no fix-up (conv_small.h)
    const h16x8* x8 = static_cast<const h16x8*>(xs);
    const h16x8* w8 = static_cast<const h16x8*>(wk_split);
    if (planes == 1) {
        if (narrow && bo == 32 && !transposed) return launch_sx<1, false, 1, 1, 1, 8, true>(x8, w8, y, scratch, g, e, s);
        IA_REQUIRE(!narrow, "one-plane operands: the narrow tile family covers the 32-channel stride-1 tile only");
        if (transposed && bp == 256) return launch_sx<1, true, 1, 2, 2, 4>(x8, w8, y, scratch, g, e, s);
        if (transposed && bp == 128) return launch_sx<1, true, 1, 2, 2, 2>(x8, w8, y, scratch, g, e, s);
        if (transposed) return launch_sx<1, true, 1, 1, 2, 2>(x8, w8, y, scratch, g, e, s);
        return launch_sx<1, false, 2, 2, 2, 4>(x8, w8, y, scratch, g, e, s);
    }
    if (narrow && bo == 32 && transposed) return launch_sx<2, true, 1, 1, 1, 8, true>(x8, w8, y, scratch, g, e, s);
    const bool aph = (int64_t)g.GH * g.GW >= 128 * 128;      // narrow tiles: antiphase wave groups from 128^2 points (see conv_split_kernel)
    if (narrow && bo == 32 && aph) return launch_sx<2, false, 1, 1, 1, 8, true, true>(x8, w8, y, scratch, g, e, s);
    if (narrow && bo == 32) return launch_sx<2, false, 1, 1, 1, 8, true>(x8, w8, y, scratch, g, e, s);
    if (narrow && aph) return launch_sx<2, false, 1, 2, 2, 4, true, true>(x8, w8, y, scratch, g, e, s);
    if (narrow) return launch_sx<2, false, 1, 2, 2, 4, true>(x8, w8, y, scratch, g, e, s);
    if (transposed && bp == 256) return launch_sx<2, true, 1, 2, 2, 4>(x8, w8, y, scratch, g, e, s);
    if (transposed && bp == 128) return launch_sx<2, true, 1, 2, 2, 2>(x8, w8, y, scratch, g, e, s);
    if (transposed) return launch_sx<2, true, 1, 1, 2, 2>(x8, w8, y, scratch, g, e, s);
    // (16 waves of 32ch x 64pt on the same 128 x 256 tile -- four waves per SIMD to overlap DMA issue, operand reads and MFMAs of
    // different waves -- measured 240 vs 245 us on 256->256 @256^2 and slower on the smaller layers: not kept)
    return launch_sx<2, false, 2, 2, 2, 4>(x8, w8, y, scratch, g, e, s);
}

extern "C" int ia_conv2d_mfma_sx(const void* xs, int planes, const void* wk_split, int wk_exp, const float* demod, const float* noise,
                                 const float* noise_strength, const float* bias, const float* residual, float* y, void* ys, int ys_planes,
                                 const float* styles_next, float* scratch, size_t scratch_bytes, int B, int I, int O, int H, int W,
                                 int transposed, int act, float alpha, const float* prelu_alpha, float gain, float clamp, int ksplit, void* stream) {
    IA_REQUIRE(!prelu_alpha || act == IA_ACT_LRELU, "prelu_alpha are the per-channel slopes of IA_ACT_LRELU");
    return conv_sx_impl(xs, planes, wk_split, wk_exp, demod, noise, noise_strength, bias, residual, y, ys, ys_planes, styles_next, scratch, scratch_bytes,
                        B, I, O, H, W, transposed, act, alpha, gain, clamp, ksplit, stream, RgbArgs{nullptr, nullptr, nullptr, nullptr, nullptr, 0, -1.f},
                        prelu_alpha);
}

extern "C" int ia_conv2d_down_sx(const void* xs, int planes, const void* wk_split, int wk_exp, const float* demod, const float* bias, const float* residual,
                                 float* y, void* ys, int ys_planes, const float* styles_next, float* scratch, size_t scratch_bytes, int B, int I, int O,
                                 int H, int W, int act, float alpha, const float* prelu_alpha, float gain, float clamp, int ksplit, void* stream) {
    IA_REQUIRE(!prelu_alpha || act == IA_ACT_LRELU, "prelu_alpha are the per-channel slopes of IA_ACT_LRELU");
    IA_REQUIRE(H >= 1 && W >= 1 && ((H - 1) / 2 + 1) >= 8 && ((W - 1) / 2 + 1) >= 8, "the stride-2 form takes outputs from 8^2 up");
    return conv_sx_impl(xs, planes, wk_split, wk_exp, demod, nullptr, nullptr, bias, residual, y, ys, ys_planes, styles_next, scratch, scratch_bytes,
                        B, I, O, H, W, 0, act, alpha, gain, clamp, ksplit, stream, RgbArgs{nullptr, nullptr, nullptr, nullptr, nullptr, 0, -1.f},
                        prelu_alpha, 0, 2);
}

extern "C" int ia_conv2d_mfma_sx_rgb(const void* xs, int planes, const void* wk_split, int wk_exp, const float* demod, const float* noise,
                                     const float* noise_strength, const float* bias, float* y, void* ys, int ys_planes, const float* styles_next,
                                     const float* rgb_wk, const float* rgb_styles, const float* rgb_bias, const float* rgb_residual, float* rgb_out,
                                     int rgb_channels, float rgb_clamp, int B, int I, int O, int H, int W, int act, float alpha, float gain, float clamp,
                                     void* stream) {
    IA_REQUIRE(rgb_out && rgb_wk, "rgb_out and rgb_wk must be device pointers");
    return conv_sx_impl(xs, planes, wk_split, wk_exp, demod, noise, noise_strength, bias, nullptr, y, ys, ys_planes, styles_next, nullptr, 0,
                        B, I, O, H, W, 0, act, alpha, gain, clamp, 0, stream, RgbArgs{rgb_wk, rgb_styles, rgb_bias, rgb_residual, rgb_out, rgb_channels, rgb_clamp});
}

extern "C" int ia_upconv2d_fir_sx(const void* xs, int planes, const void* wk_split, int wk_exp, const float* demod, const float* noise,
                                  const float* noise_strength, const float* bias, float* y, void* ys, int ys_planes, const float* styles_next,
                                  int B, int I, int O, int H, int W, int act, float alpha, float gain, float clamp, void* stream) {
    IA_REQUIRE(O > 0 && (int64_t)4 * O <= INT32_MAX / 4, "empty tensor");
    return conv_sx_impl(xs, planes, wk_split, wk_exp, demod, noise, noise_strength, bias, nullptr, y, ys, ys_planes, styles_next, nullptr, 0,
                        B, I, 4 * O, H, W, 0, act, alpha, gain, clamp, 0, stream, RgbArgs{nullptr, nullptr, nullptr, nullptr, nullptr, 0, -1.f},
                        nullptr, 1);
}

// Pieces shared by the MFMA convolution kernels (conv_mfma.hip: operands staged through registers from fp32 activations;
// conv_split.hip: operands DMA'd into LDS from pre-split fp16 hi/lo activations): geometry, the tile window, the epilogue,
// the accumulator-tile store and the stream-K fix-up kernel.
#pragma once
#include "ia_common.h"
#ifndef IA_TR_SIMPLE_EPI
#define IA_TR_SIMPLE_EPI 1      // 0: the transposed store evaluates the generic epilogue per element (A/B builds in tools/)
#endif
#include <type_traits>
#include <cstdlib>

namespace {


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

constexpr int kPatchFloats = 1024;  // per-channel LDS patch capacity (floats)
// The low parts of the activations' hi/lo fp16 split are scaled by 2^11 so that they are normal fp16 numbers (the MFMA
// flushes fp16 denormals); the weight's high part is multiplied by 2^-11 where it meets one (see HM = 2 below).
constexpr float kLoScale = 2048.f;
// fp16 forms: one k-step of v_mfma_f32_32x32x16_f16 = 8 channels of TWO taps (lanes 0-31 carry the first tap of the pair,
// lanes 32-63 the second).  The nine taps make five pairs; the odd tap out pairs with an all-zero tap (index 9).  In the
// transposed form both taps of a pair must feed the same output phase: phase 0 owns taps {0,2,6,8}, phase 1 {1,7},
// phase 2 {3,5}, phase 3 {4}.
constexpr int kPairs = 5, kZeroTap = 9;
__host__ __device__ constexpr int pair_t0(bool tr, int s) { return tr ? (s == 0 ? 0 : s == 1 ? 6 : s == 2 ? 1 : s == 3 ? 3 : 4) : 2 * s; }
__host__ __device__ constexpr int pair_t1(bool tr, int s) { return tr ? (s == 0 ? 2 : s == 1 ? 8 : s == 2 ? 7 : s == 3 ? 5 : kZeroTap) : (s == 4 ? kZeroTap : 2 * s + 1); }
__host__ __device__ constexpr int pair_phase(bool tr, int s) { return tr ? (s < 2 ? 0 : s - 1) : 0; }

// Low-resolution 3x3 layers of the split-DMA form (at most kSmallMaxPoints points; 17^2 in the transposed form) whose tile plan would be stream-K: the K
// range is dealt to the waves of a workgroup instead (csrc/conv_small.h) -- no scratch, no fix-up launch.  The planner (conv_mfma.hip)
// and the dispatcher (conv_split.hip) both ask here.  IA_CONV_SMALL = 0: A/B builds that keep those layers on the stream-K tiles.
#ifndef IA_CONV_SMALL
#define IA_CONV_SMALL 1
#endif
// Above 16^2 only launches of at most 1.5 rounds: the 512-workgroup launch of a backbone's 512 -> 512 @32^2 layer is faster alone (42.9 -> 33.4 us)
// but costs the FRAME 1.5 % -- it fills every CU twice while the other networks' streams wait (same-box three-way A/B, frames/s: up to 32^2
// 385.4, up to 16^2 391.1, never 387.9; one-shot inversion 28.8 / 29.9 / 30.1 ms: its 32^2 layers are 256 - 384 workgroups and keep the form).
constexpr int kSmallMaxWgsAbove256 = 384;
constexpr int kSmallMaxPoints = 1024;      // (32^2: 512 -> 512 42.9 -> 33.4 us, 256 -> 256 23.2 -> 11.7, 384 -> 384 33.2 -> 25.4, same box; r06)
// ... as long as its workgroups (32 channels x 32 points each, all of K) stay within two rounds of the machine.  Measured r06, one box,
// graph replay, us, against the stream-K tiles + fix-up (channel tiles on blockIdx.x: one L2 per weight slice): 512 -> 512 @8^2 24.9 -> 15.0,
// @16^2 25.5 -> 15.8, 8 frames @8^2 41.6 -> 16.6, 256 -> 256 @16^2 19.8 -> 10.5, 1024 -> 1024 @16^2 (256 workgroups) 41.6 -> 28.0,
// 1024 -> 512 @16^2 30.9 -> 23.8, 4 frames 512 -> 512 @16^2 (512 workgroups) 35.2 -> 30.9; transposed @8^2 21.8 -> 15.3, @16^2 38.3 -> 15.8,
// 8 frames @8^2 (384 workgroups) 37.1 -> 29.5; but 8 frames @16^2 (1024 workgroups) 44.7 -> 59.3: every workgroup streams its own copy
// of the operands through its CU's 64 B/clk vector-memory path (profiles/r06_conv_small.txt).
__host__ inline bool conv_small_shape(int B, int I, int O, int H, int W, int ksize, int transposed, int stride) {
    const int npts = transposed ? (H + 1) * (W + 1) : H * W;
    const int64_t wgs = (int64_t)B * ((npts + 31) / 32) * ((O + 31) / 32);
    // (transposed: the four-phase form of the 8^2 / 16^2 up-sampling layers -- 81 / 289 points -- which ran on a 64-channel x 64-point
    // stream-K tile at 0.05 of the matrix pipe with 0.29 LDS bank conflicts, r05 PMC)
    // (IA_CONV_SMALL_WGS: experiment override of the limit, read per call -- the sweep of profiles/r06_conv_small.txt)
    const char* ew = getenv("IA_CONV_SMALL_WGS");
    const int64_t max_wgs = ew ? atoll(ew) : 2 * (int64_t)ia::kNumCU;
    const char* ep = getenv("IA_CONV_SMALL_PTS");
    const int max_pts = ep ? atoi(ep) : kSmallMaxPoints;
    constexpr int k_weight = 1;
    (void)I;
    return IA_CONV_SMALL && ksize == 3 && stride == 1 && npts <= (transposed ? 17 * 17 : max_pts) && wgs * k_weight <= (npts > 256 && !ew ? kSmallMaxWgsAbove256 : max_wgs);
}

struct Geo {
    int B, I, O, H, W;     // input
    int GH, GW;            // point grid: conv H x W, transposed (H+1) x (W+1)
    int OH, OW;            // output image
    int G;                 // stream-K workers per batch element (0: none)
    int C;                 // K chunks per tile
    int TO, T;             // out-channel tiles, tiles in total (point tiles x out-channel tiles)
    int T_dp;              // tiles [0, T_dp) run one per workgroup, tiles [T_dp, T) are stream-K
    int patch_cap;         // floats per channel reserved for the patch in LDS
    int xcd_bands = 0;     // conv_split.hip: 1 = whole-tile launches give each XCD a contiguous band of tiles
    int stride = 1;        // conv_split.hip, stride-1 tile families: 2 = the point grid is every second pixel of the padded input (3x3 stride-2 convolution)
    float acc_scale;       // fp16-pair form: 2^-wk_exp, takes the accumulators back from the scale of the packed weights
};

// unit range of worker w: [range_begin(w), range_begin(w+1)) over U = (T - T_dp)*C units
__host__ __device__ inline int64_t range_begin(int w, int64_t U, int G) { return ((int64_t)w * U) / G; }

struct Epi {
    const float* demod;           // [B,O] or null
    const float* noise;           // [OH*OW] or null
    const float* noise_strength;  // device scalar (may be null => 1)
    const float* bias;            // [O] or null
    const float* residual;        // [B,O,OH,OW] or null, added after the clamp
    int act;                      // IA_ACT_LINEAR or IA_ACT_LRELU
    float alpha, gain, clamp;
    // optional second output (conv_split.hip): the result as fp16 hi/lo planes [B][2][O/8][OH*OW][8], pre-multiplied by the styles
    // of the layer that consumes it (styles_next [B,O] or null) -- the input format of ia_conv2d_mfma_sx
    void* ys;
    const float* styles_next;
    int ys_planes;                // 2: hi / lo planes (fp32-equivalent consumers), 1: one fp16 plane (fp16-operand consumers)
    // optional third output (conv_split.hip, stride-1 tiles that hold every output channel): the ToRGB layer that consumes this
    // result, evaluated in the epilogue:  rgb_out[b][c][pix] = clamp(sum_o (rgb_w[o][c] * rgb_styles[b][o]) * v[o] + rgb_bias[c]) + rgb_res
    const float* rgb_w = nullptr;       // [O][rgb_n] (ia_conv2d_mfma's ksize-1 packing)
    const float* rgb_styles = nullptr;  // [B][O] or null
    const float* rgb_bias = nullptr;    // [rgb_n] or null
    const float* rgb_res = nullptr;     // [B][rgb_n][OH*OW] or null, added after the clamp
    float* rgb_out = nullptr;           // [B][rgb_n][OH*OW]
    float rgb_clamp = -1.f;
    int rgb_n = 0;
    // optional per-output-channel negative slopes of the leaky ReLU ([O]; PReLU of the inversion encoders): replaces `alpha`
    const float* alpha_vec = nullptr;
    // depth-to-space store (conv_split.hip, ia_upconv2d_fir_sx): the layer's O channels are 4 output PHASES x O/4 real channels
    // (phase-major), and point (r, c) of phase (py, px) is pixel (2r + py, 2c + px) of a 2H x 2W image; demod / bias / noise /
    // styles_next are indexed by the real channel and the output pixel
    int d2s = 0;
};
// A by-value kernel argument re-read from the kernarg segment WHERE IT IS USED (s_load through the kernarg pointer the dispatch leaves in
// s[0:1]): the epilogue descriptor of the convolution kernels is 36 dwords that the compiler otherwise loads at kernel entry and keeps --
// spilled to VGPR lanes (v_writelane / v_readlane + s_nop) -- across the whole K loop (r05: 60 - 142 SGPR spills in every
// conv_split_kernel instantiation, 220 v_readlane + 140 s_nop in the epilogue).  OFF = offsetof(the kernel's argument list, the argument).
// Each s_load is waited for inside its own asm statement: scalar loads may return out of order, and the compiler is free to overlap the
// registers of outputs it considers dead.
typedef unsigned int ia_u32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int ia_u32x4 __attribute__((ext_vector_type(4)));
template <class T, int OFF>
__device__ __forceinline__ T reload_kernarg() {
    static_assert(sizeof(T) % 16 == 0 && OFF % 4 == 0, "whole 16-byte groups at a dword offset");
    constexpr int ND = sizeof(T) / 4;
    typedef __attribute__((address_space(4))) const char kchar;
    kchar* ka = (kchar*)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned int d[ND];
    constexpr int N16 = ND / 16;
#pragma unroll
    for (int i = 0; i < N16; ++i) {
        ia_u32x16 v;
        asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ka), "n"(OFF + 64 * i));
#pragma unroll
        for (int j = 0; j < 16; ++j) d[16 * i + j] = v[j];
    }
#pragma unroll
    for (int i = 16 * N16; i < ND; i += 4) {
        ia_u32x4 v;
        asm volatile("s_load_dwordx4 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ka), "n"(OFF + 4 * i));
#pragma unroll
        for (int j = 0; j < 4; ++j) d[i + j] = v[j];
    }
    T out;
    __builtin_memcpy(&out, d, sizeof(T));
    return out;
}

constexpr int kMaxRgb = 3;      // (ToRGB proper: three colours; a fourth accumulator per point fragment spilled in the 256-register epilogue)

typedef _Float16 h16x4_t __attribute__((ext_vector_type(4)));

// Channels o .. o+3 (o % 4 == 0) of output pixel `pix` into the split tensor: 8 bytes per plane; the lane that holds channels
// o+4 .. o+7 (or o-4 .. o-1) of the same pixel writes the other half of the 16-byte unit.
template <bool STRAIGHT = true, class Watch = ia::SatWatch>
__device__ __forceinline__ void split_store4(void* ys, const float* sn, int planes, int b, int O, int64_t ohw, int o, int64_t pix, const float (&v)[4], Watch& watch) {
    h16x4_t hi, lo;
    if constexpr (STRAIGHT) {
        h16x4_t h1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {   // straight-line: both forms of the high part are computed, the plane count selects (one branch below, not four)
            const float t = sn ? v[k] * sn[b * O + o + k] : v[k];
            _Float16 h, l;
            ia::split_f16(t, h, l, watch);
            hi[k] = h; lo[k] = l;
            h1[k] = (_Float16)fminf(fmaxf(t, -65504.f), 65504.f);      // (round_f16's value; the watch has seen t)
        }
        if (planes != 2) hi = h1;
    } else {                            // (the fused-ToRGB epilogue sits at the register limit: the plane test per value keeps its live ranges short)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t = sn ? v[k] * sn[b * O + o + k] : v[k];
            if (planes == 2) { _Float16 h, l; ia::split_f16(t, h, l, watch); hi[k] = h; lo[k] = l; }
            else hi[k] = ia::round_f16(t, watch);
        }
    }
    char* base = static_cast<char*>(ys);
    const int64_t slot = ((int64_t)(b * planes) * (O / 8) + (o >> 3)) * ohw + pix;
    *reinterpret_cast<h16x4_t*>(base + slot * 16 + (o & 4) * 2) = hi;
    if (planes == 2) *reinterpret_cast<h16x4_t*>(base + (slot + (int64_t)(O / 8) * ohw) * 16 + (o & 4) * 2) = lo;
}

// Input window of one tile (a contiguous range [p0, p_last] of the row-major point grid), as one or two row
// segments that share a row stride PW.  Three shapes:
//   single : the tile lies in one grid row            -> rows x (its columns + halo)
//   split  : two grid rows of a WIDE image            -> segment 0 = tail of the first row, segment 1 = head of the second
//   full   : anything else (narrow images, >= 3 rows) -> all needed rows x full width
// Halo: stride-1 conv reads (r + ky - PAD, c + kx - PAD); the transposed form reads (r - ky/2, c - kx/2).
struct Window {
    int r0[2], nr[2], c0[2];   // first input row, row count, first input column of each segment
    int PW, PSZ;               // shared row stride, floats per channel
};

__host__ __device__ inline Window tile_window(int p0, int p_last, int GW, int pad, bool tr, int stride = 1) {
    // `stride` (stride-1 form only): grid point (r, c) reads input (stride * r + ky - pad, stride * c + kx - pad)
    const int s = tr ? 1 : stride;
    const int up = tr ? 1 : pad, dn = tr ? 0 : pad, lf = tr ? 1 : pad, rt = tr ? 0 : pad;
    const int r_first = p0 / GW, r_last = p_last / GW;
    const int c_first = p0 - r_first * GW, c_last = p_last - r_last * GW;
    const int nrows = up + dn + 1;
    Window w;
    w.nr[1] = 0; w.r0[1] = 0; w.c0[1] = 0;
    if (r_first == r_last) {
        w.r0[0] = s * r_first - up; w.nr[0] = nrows; w.c0[0] = s * c_first - lf; w.PW = (s * c_last + rt) - w.c0[0] + 1;
    } else {
        const int w0 = (s * (GW - 1) + rt) - (s * c_first - lf) + 1, w1 = (s * c_last + rt) - (0 - lf) + 1;
        const int pw_split = w0 > w1 ? w0 : w1, pw_full = s * (GW - 1) + 1 + lf + rt;
        const int sz_split = 2 * nrows * pw_split, sz_full = (s * (r_last - r_first) + nrows) * pw_full;
        if (r_last == r_first + 1 && sz_split < sz_full) {
            w.r0[0] = s * r_first - up; w.nr[0] = nrows; w.c0[0] = s * c_first - lf;
            w.r0[1] = s * r_last - up;  w.nr[1] = nrows; w.c0[1] = -lf;
            w.PW = pw_split;
        } else {
            w.r0[0] = s * r_first - up; w.nr[0] = s * (r_last - r_first) + nrows; w.c0[0] = -lf; w.PW = pw_full;
        }
    }
    w.PSZ = (w.nr[0] + w.nr[1]) * w.PW;
    return w;
}

__device__ __forceinline__ float epilogue(float v, int b, int o, int64_t pix, int64_t ohw, const Geo& g, const Epi& e, float ns) {
    if (e.demod) v *= e.demod[b * g.O + o];
    if (e.noise) v = fmaf(e.noise[pix], ns, v);
    if (e.bias) v += e.bias[o];
    if (e.act == IA_ACT_LRELU) v = v > 0.f ? v : v * (e.alpha_vec ? e.alpha_vec[o] : e.alpha);
    v *= e.gain;
    if (e.clamp >= 0.f) v = fminf(fmaxf(v, -e.clamp), e.clamp);
    if (e.residual) v += e.residual[((int64_t)b * g.O + o) * ohw + pix];
    return v;
}

// DB (two LDS stages) is used by the transposed 64ch x 64pt x 4-phase tile, which does a quarter of the MFMAs of the conv tile
// per staged chunk, and by the 8-wave 128ch x 256pt tile, which is alone on its CU: one barrier per chunk, the next chunk is
// committed to the other stage at the top of an iteration, and the loads of the chunk after that are spread over the MFMA
// steps.  DB kernels require 16-byte aligned weight rows (O % 4 == 0); the host falls back to the single-stage form otherwise.
__host__ __device__ constexpr bool db_family(bool tr, int fo, int fp, int wo, int wp) { return (tr && wo == 2) || wo * wp == 8; }

// Accumulator tile -> global memory.  C/D map of the 32x32 MFMA: row(channel) = (r&3) + 8*(r>>2) + 4*half, col(point) = l31.
template <bool TR, int FO, int FP, int WO, int WP>
__device__ __forceinline__ void store_tile(const f32x16 (&acc)[TR ? 4 : 1][FO][FP], float* __restrict__ y, const Geo& g, const Epi& e,
                                           int b, int o0, int p0, int tid) {
    constexpr int NPH = TR ? 4 : 1;
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int wo = wave / WP, wp = wave % WP;
    const int npts = g.GH * g.GW;
    const int64_t ohw = (int64_t)g.OH * g.OW;
    const float ns = e.noise ? (e.noise_strength ? *e.noise_strength : 1.f) : 0.f;
    float* yb = y + ((int64_t)b * g.O) * ohw;
#pragma unroll
    for (int fp = 0; fp < FP; ++fp) {
        const int p = p0 + (wp * FP + fp) * 32 + l31;
        if (p >= npts) continue;
        const int pr = p / g.GW, pc = p - pr * g.GW;
        if constexpr (TR) {
            // the two horizontal phases of a point are adjacent output pixels: one 8-byte store per (row phase, channel), so
            // that a half-wave writes 64 consecutive floats instead of every other one twice
            const int ox = 2 * pc;
            // The transposed form's epilogue is the demodulation alone (FIR, noise, bias, activation follow in the FIR tail): its
            // coefficients -- 16 per fragment, the same for all four phases -- are read once, ahead of the stores (read per element
            // they are re-loaded after every store: the compiler cannot prove that y does not alias them).
            // (the quad-wise stores below skip a channel quad as a whole: only for channel counts that are multiples of 4 -- a ragged count
            // takes the per-element loop with its `o >= g.O` test)
            const bool simple = IA_TR_SIMPLE_EPI && !e.noise && !e.bias && !e.residual && e.act == IA_ACT_LINEAR && e.clamp < 0.f && e.gain == 1.f && (g.O & 3) == 0;
            float dmv[FO][16];
#pragma unroll
            for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = o0 + (wo * FO + fo) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    dmv[fo][r] = (simple && e.demod && o < g.O) ? e.demod[b * g.O + o] : 1.f;
                }
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const int oy = 2 * pr + py;
                if (oy >= g.OH) continue;
                const int64_t pix = (int64_t)oy * g.OW + ox;
                const bool pair = ox + 1 < g.OW;
                if (simple) {
                    // the frame's case, straight-line: the option tests (`simple`, last column, channel range) sit around the
                    // element loops, not inside them (r05 ISA: ~1400 branches in this store)
                    float* dst0 = yb + pix;
                    if (pair) {
#pragma unroll
                        for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int o_first = o0 + (wo * FO + fo) * 32 + 8 * q + 4 * half;
                                if (o_first + 3 >= g.O) continue;         // (O % 4 == 0 here: the quad is inside or outside as a whole)
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    const int r = 4 * q + k;
                                    const float2 v = make_float2(acc[2 * py][fo][fp][r] * dmv[fo][r], acc[2 * py + 1][fo][fp][r] * dmv[fo][r]);
                                    __builtin_memcpy(dst0 + (int64_t)(o_first + k) * ohw, &v, 8);     // (rows of odd width: 4-byte aligned only)
                                }
                            }
                    } else {
#pragma unroll
                        for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int o_first = o0 + (wo * FO + fo) * 32 + 8 * q + 4 * half;
                                if (o_first + 3 >= g.O) continue;
#pragma unroll
                                for (int k = 0; k < 4; ++k) dst0[(int64_t)(o_first + k) * ohw] = acc[2 * py][fo][fp][4 * q + k] * dmv[fo][4 * q + k];
                            }
                    }
                    continue;
                }
#pragma unroll
                for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int o = o0 + (wo * FO + fo) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (o >= g.O) continue;
                        float* dst = yb + (int64_t)o * ohw + pix;
                        const float v0 = epilogue(acc[2 * py][fo][fp][r], b, o, pix, ohw, g, e, ns);
                        if (pair) {
                            const float v1 = epilogue(acc[2 * py + 1][fo][fp][r], b, o, pix + 1, ohw, g, e, ns);
                            __builtin_memcpy(dst, &(const float2&)make_float2(v0, v1), 8);
                        } else dst[0] = v0;
                    }
            }
            continue;
        }
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            int64_t pix;
            if (TR) {
                const int oy = 2 * pr + (ph >> 1), ox = 2 * pc + (ph & 1);
                if (oy >= g.OH || ox >= g.OW) continue;
                pix = (int64_t)oy * g.OW + ox;
            } else pix = p;
#pragma unroll
            for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = o0 + (wo * FO + fo) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (o >= g.O) continue;
                    yb[(int64_t)o * ohw + pix] = epilogue(acc[ph][fo][fp][r], b, o, pix, ohw, g, e, ns);
                }
        }
    }
}

// Fix-up: stream-K tiles that were split between workers.  grid = (stream-K tile, batch, accumulator quad): a workgroup adds one
// register quad (float4; a pair of them in the transposed form) of every thread position over the tile's slabs in worker order and stores it through the epilogue.
// Loads of up to kFixBatch workers are in flight together (the adds keep the worker order).
constexpr int kFixBatch = 8;
template <bool TR, int FO, int FP, int WO, int WP>
__global__ __launch_bounds__(WO * WP * 64) void conv_fixup_kernel(const float* __restrict__ slabs, float* __restrict__ y, Geo g, Epi e) {
    constexpr int NPH = TR ? 4 : 1;
    constexpr int BO = 32 * FO * WO, BP = 32 * FP * WP, NTHREADS = WO * WP * 64;
    constexpr int NACC = NPH * FO * FP * 16;
    // (transposed form: blockIdx.z enumerates quads of ROW phases; the thread sums the quads of both horizontal phases and
    // stores them as 8-byte pairs of adjacent output pixels, like store_tile)
    constexpr int NQ = TR ? 2 : 1;
    const int tile_l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int zq = blockIdx.z, rq = zq & 3, frp = zq >> 2, fp = frp % FP, fo = (frp / FP) % FO, pyh = frp / (FP * FO);
    int q[NQ];
#pragma unroll
    for (int h = 0; h < NQ; ++h) q[h] = ((((TR ? 2 * pyh + h : 0) * FO + fo) * FP + fp) << 2) + rq;
    const int64_t U = (int64_t)(g.T - g.T_dp) * g.C, t_begin = (int64_t)tile_l * g.C, t_end = t_begin + g.C;
    int w_first = (int)((t_begin * g.G) / U);
    while (w_first > 0 && range_begin(w_first, U, g.G) > t_begin) --w_first;
    while (range_begin(w_first + 1, U, g.G) <= t_begin) ++w_first;
    int w_last = w_first;
    while (range_begin(w_last + 1, U, g.G) < t_end) ++w_last;
    if (w_first == w_last) return;                       // the tile was finished by a single worker
    const int64_t slab4 = (int64_t)NACC * NTHREADS / 4;
    const float4* base = reinterpret_cast<const float4*>(slabs) + ((int64_t)b * g.G) * 2 * slab4 + tid;
    // only the first worker can hold this tile in its trailing slot (1); every later worker starts inside the tile (slot 0)
    const int slot_first = (tile_l == (int)(range_begin(w_first, U, g.G) / g.C)) ? 0 : 1;
    constexpr int FB = kFixBatch / NQ;
    float4 acc[NQ];
#pragma unroll
    for (int h = 0; h < NQ; ++h) acc[h] = base[((int64_t)w_first * 2 + slot_first) * slab4 + (int64_t)q[h] * NTHREADS];
    for (int w = w_first + 1; w <= w_last; w += FB) {
        float4 v[FB][NQ];
#pragma unroll
        for (int j = 0; j < FB; ++j)
#pragma unroll
            for (int h = 0; h < NQ; ++h) v[j][h] = base[((int64_t)min(w + j, w_last) * 2) * slab4 + (int64_t)q[h] * NTHREADS];
#pragma unroll
        for (int j = 0; j < FB; ++j)
            if (w + j <= w_last) {
#pragma unroll
                for (int h = 0; h < NQ; ++h) { acc[h].x += v[j][h].x; acc[h].y += v[j][h].y; acc[h].z += v[j][h].z; acc[h].w += v[j][h].w; }
            }
    }
    // decode (register index, thread) -> (channel, point) exactly as the MFMA kernel lays its accumulators out
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int wo = wave / WP, wp = wave % WP;
    const int tile = g.T_dp + tile_l;
    const int o0 = (tile % g.TO) * BO, p0 = (tile / g.TO) * BP;
    const int npts = g.GH * g.GW;
    const int64_t ohw = (int64_t)g.OH * g.OW;
    const float ns = e.noise ? (e.noise_strength ? *e.noise_strength : 1.f) : 0.f;
    float* yb = y + ((int64_t)b * g.O) * ohw;
    const int p = p0 + (wp * FP + fp) * 32 + l31;
    if (p >= npts) return;
    const int pr = p / g.GW, pc = p - pr * g.GW;
    int64_t pix = p;
    bool pair = false;
    if (TR) {
        const int oy = 2 * pr + pyh, ox = 2 * pc;
        if (oy >= g.OH) return;
        pix = (int64_t)oy * g.OW + ox;
        pair = ox + 1 < g.OW;
    }
    const float v0[4] = {acc[0].x, acc[0].y, acc[0].z, acc[0].w};
    const float v1[4] = {acc[NQ - 1].x, acc[NQ - 1].y, acc[NQ - 1].z, acc[NQ - 1].w};
    float outv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = rq * 4 + k;
        const int o = o0 + (wo * FO + fo) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (o >= g.O) continue;
        const float a0 = epilogue(v0[k], b, o, pix, ohw, g, e, ns);
        outv[k] = a0;
        if (!y) continue;
        float* dst = yb + (int64_t)o * ohw + pix;
        if (TR && pair) {
            const float a1 = epilogue(v1[k], b, o, pix + 1, ohw, g, e, ns);
            __builtin_memcpy(dst, &(const float2&)make_float2(a0, a1), 8);
        } else dst[0] = a0;
    }
    if constexpr (!TR) {
        const int o_first = o0 + (wo * FO + fo) * 32 + 8 * rq + 4 * half;      // channels (rq*4 + k): k + 8*rq + 4*half
        if (e.ys && o_first + 3 < g.O) {
            ia::SatWatch watch;
            split_store4(e.ys, e.styles_next, e.ys_planes, b, g.O, ohw, o_first, pix, outv, watch);
            watch.report();
        }
    }
}

}  // namespace

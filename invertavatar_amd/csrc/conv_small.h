// conv_small_kernel (included by conv_split.hip, whose entry points dispatch to it): the 3x3 convolutions of the LOW-RESOLUTION layers (8^2 .. 32^2 images: 512 -> 512 channels on 64 .. 1024 points in
// the three backbones of a frame, the ConvGRU cells and residual units of the inversion encoders) on split-format (fp16 hi / lo)
// operands -- the arithmetic of conv_split_kernel (csrc/conv_split.hip), another decomposition.
//
// Why: these layers are three orders of magnitude smaller than the machine.  On the 128-channel x 256-point tile
// of the stride-1 family they were cut ALONG K between 64 stream-K workgroups (mfma_pipe_util 0.05 - 0.12, LDS conflicts 0.10 - 0.29),
// every workgroup wrote a 128 KB accumulator slab of a tile that is three quarters padding at 8^2, and a second launch
// (conv_fixup_kernel) summed the slabs: 17 + 10 us for 0.3 GFLOP, twelve of those launch pairs per frame.
//
// Here the split along K stays INSIDE a workgroup:
//   * a workgroup owns 32 output channels x (32 * FP) points and ALL of K; its eight waves take the input-channel octets round-robin
//     (wave w: octets w, w + 8, ...), each accumulating the whole 32 x 32FP tile over its share of K;
//   * no operand is shared between the waves of a workgroup (different K), and a wave's A / B fragments are exactly what a lane loads:
//     16 bytes per lane and plane straight from global memory into the MFMA operand registers (raw buffer loads: a tap outside the
//     image, the zero tap of the odd pair and channels past O read zeros through the descriptor's range check) -- no LDS staging,
//     no barriers in the K loop; five k-steps (one input-channel octet) are in flight per wave ahead of the one being multiplied;
//   * the eight partial tiles meet in LDS (64 KB at FP = 2), are summed in wave order (deterministic) and leave through the same
//     epilogue as every other convolution of the family (demodulation, noise, bias, leaky ReLU / PReLU, gain, clamp, residual;
//     fp32 and / or split-format output).
// One launch, no scratch, no fix-up.  Weights are read once per point tile (L2 serves the repeats), the activations of a layer are
// 128 - 512 KB and live in L2.
#pragma once
#include "conv_common.h"
#include "lds_dma.h"

namespace {

constexpr int kSmallWaves = 8;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t small_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ h16x8 load16(__amdgpu_buffer_rsrc_t r, int voffset, int soffset) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voffset, soffset, 0);
    return __builtin_bit_cast(h16x8, v);
}

// NP = operand planes (2: hi / lo pairs, three products; 1: one fp16 plane, one product); FP = point fragments (32 points each) per workgroup.
// TR: the stride-2 transposed convolution of an up-sampling layer in its four-phase form (conv_common.h: point (r, c) of the (H+1) x (W+1)
// grid reads input (r - ky/2, c - kx/2) and owns output pixels (2r + py, 2c + px); the tap pairs of a k-step feed ONE phase) -- four
// accumulator fragments per wave, one octet's five k-steps in flight (the registers the second octet of the stride-1 ring takes), the
// epilogue is the demodulation alone (FIR + noise + bias + activation follow in ia_fir_tail_split) into the (2H+1) x (2W+1) fp32 image.
template <int NP, int FP, bool TR>
__global__ __launch_bounds__(kSmallWaves * 64) void conv_small_kernel(const h16x8* __restrict__ xs, const h16x8* __restrict__ wk,
                                                                       float* __restrict__ y, Geo g, Epi e) {
    constexpr int NT = 9, NPH = TR ? 4 : 1, NQ = NPH * FP * 4, RING = TR ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) float part[];      // [kSmallWaves][NQ][64 lanes][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    // blockIdx.x = the channel tile: workgroups go to the 8 XCDs round-robin by linear id, so (with a multiple of 8 channel tiles) ALL point
    // tiles of a channel tile land on one XCD and its 590 KB weight slice is fetched into ONE L2 (point tiles on x: every XCD fetched every
    // slice -- PMC r06: 47 MB per launch for 9.4 MB of weights)
    const int b = blockIdx.z, o0 = blockIdx.x * 32, p0 = blockIdx.y * (32 * FP);
    const int HW = g.H * g.W, npts = g.GH * g.GW, I8 = g.I / 8;
    const int plane_bytes = I8 * HW * 16, wplane_bytes = NT * I8 * g.O * 16;
    constexpr int kOutside = 0x7ffffff0;

    const char* xb = reinterpret_cast<const char*>(xs + (int64_t)b * NP * I8 * HW);
    const __amdgpu_buffer_rsrc_t rx0 = small_rsrc(xb, (unsigned)plane_bytes), rx1 = small_rsrc(xb + (NP - 1) * (int64_t)plane_bytes, (unsigned)plane_bytes);
    const char* wb = reinterpret_cast<const char*>(wk);
    const __amdgpu_buffer_rsrc_t rw0 = small_rsrc(wb, (unsigned)wplane_bytes), rw1 = small_rsrc(wb + (NP - 1) * (int64_t)wplane_bytes, (unsigned)wplane_bytes);

    // per-lane byte offsets of the operands of k-step s at octet 0: lanes 0-31 carry the first tap of pair s, lanes 32-63 the second
    // (pair 4 = tap 8 | the all-zero tap)
    int a_off[kPairs], b_off[FP][kPairs];
    const int o_ld = o0 + l31;
#pragma unroll
    for (int s = 0; s < kPairs; ++s) {
        const int tap = half ? pair_t1(TR, s) : pair_t0(TR, s);
        a_off[s] = (tap == kZeroTap || o_ld >= g.O) ? kOutside : ((tap * I8) * g.O + o_ld) * 16;
        const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
        for (int fp = 0; fp < FP; ++fp) {
            const int p = p0 + fp * 32 + l31;
            const int r = p / g.GW, c = p - r * g.GW;
            const int iy = TR ? r - (ky >> 1) : r + ky - 1, ix = TR ? c - (kx >> 1) : c + kx - 1;
            const bool ok = tap != kZeroTap && p < npts && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
            b_off[fp][s] = ok ? (iy * g.W + ix) * 16 : kOutside;
        }
    }

    f32x16 acc[NPH][FP];
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
        for (int fp = 0; fp < FP; ++fp)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ph][fp][r] = 0.f;

    // ring of TWO octets' k-steps (2 x 5 slots of 4 loads): nine slots are in flight ahead of the one being multiplied.
    // `live` = false: the same instructions with every lane outside the buffers (zeros, no memory traffic) -- the last rounds refill
    // nothing, and a loop body without branches around its loads lets the compiler count them (s_waitcnt vmcnt(36) in the steady
    // state: loads return in order); the sched_barriers pin "multiply slot, refill slot" so that the count is the program's.
    h16x8 ra[RING][kPairs][NP], rb[RING][kPairs][NP][FP];
    auto load_step = [&](int par, int c8, int s, bool live) {
        const int wso = c8 * g.O * 16, pso = c8 * HW * 16;
        const int dead = live ? 0 : kOutside;      // (a scalar OR-ed into the lane offsets: an offset >= kOutside is outside every buffer)
        const int ao = a_off[s] | dead;
        ra[par][s][0] = load16(rw0, ao, wso);
        if constexpr (NP == 2) ra[par][s][1] = load16(rw1, ao, wso);
#pragma unroll
        for (int fp = 0; fp < FP; ++fp) {
            const int bo = b_off[fp][s] | dead;
            rb[par][s][0][fp] = load16(rx0, bo, pso);
            if constexpr (NP == 2) rb[par][s][1][fp] = load16(rx1, bo, pso);
        }
    };
#pragma unroll
    for (int par = 0; par < RING; ++par) {
        const int c = wave + par * kSmallWaves;
        const bool live = c < I8;
#pragma unroll
        for (int s = 0; s < kPairs; ++s) {
            load_step(par, live ? c : 0, s, live);
            __builtin_amdgcn_sched_barrier(0);      // (the ring is filled in the order the loop consumes it: the loop's counted waits hold from its first round)
        }
    }
    for (int c8 = wave; c8 < I8; c8 += RING * kSmallWaves) {
#pragma unroll
        for (int par = 0; par < RING; ++par) {
            const int nxt = c8 + (par + RING) * kSmallWaves;
            const bool more = nxt < I8;
            const int nxt_c = more ? nxt : 0;
#pragma unroll
            for (int s = 0; s < kPairs; ++s) {
                const int ph = TR ? (s < 2 ? 0 : s - 1) : 0;      // pair_phase(TR, s): a constant once the loop is unrolled
                const h16x8 a_hi = ra[par][s][0];
                h16x8 b_hi[FP];
#pragma unroll
                for (int fp = 0; fp < FP; ++fp) b_hi[fp] = rb[par][s][0][fp];
                if constexpr (NP == 2) {
                    const h16x8 a_lo = ra[par][s][1];
                    const h16x8 a_sc = a_hi * (_Float16)(1.0f / kLoScale);      // weight high parts at 2^-11: they meet the activations' low parts (scaled by 2^11)
#pragma unroll
                    for (int fp = 0; fp < FP; ++fp) acc[ph][fp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi[fp], acc[ph][fp], 0, 0, 0);                 // lo * hi
#pragma unroll
                    for (int fp = 0; fp < FP; ++fp) acc[ph][fp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_sc, rb[par][s][NP - 1][fp], acc[ph][fp], 0, 0, 0);   // (hi * 2^-11) * (lo * 2^11)
                }
#pragma unroll
                for (int fp = 0; fp < FP; ++fp) acc[ph][fp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi[fp], acc[ph][fp], 0, 0, 0);                     // hi * hi
                __builtin_amdgcn_sched_barrier(0);
                load_step(par, nxt_c, s, more);                             // refill the slot (the MFMAs above have read its registers)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- the eight partial tiles meet in LDS: part[wave][quad q = (ph * FP + fp) * 4 + rq][lane] = registers 4 rq .. 4 rq + 3 of fragment (ph, fp)
    float4* pw = reinterpret_cast<float4*>(part) + (wave * NQ) * 64 + lane;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
        for (int fp = 0; fp < FP; ++fp)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                pw[((ph * FP + fp) * 4 + rq) * 64] = make_float4(acc[ph][fp][4 * rq], acc[ph][fp][4 * rq + 1], acc[ph][fp][4 * rq + 2], acc[ph][fp][4 * rq + 3]);
    __syncthreads();
    const int64_t ohw = (int64_t)g.OH * g.OW;
    const float ns = e.noise ? (e.noise_strength ? *e.noise_strength : 1.f) : 0.f;
    for (int q = wave; q < NQ; q += kSmallWaves) {
        const float4* pr = reinterpret_cast<const float4*>(part) + q * 64 + lane;
        float4 v = pr[0];
#pragma unroll
        for (int w = 1; w < kSmallWaves; ++w) {      // wave order: the same bits on every launch
            const float4 u = pr[(w * NQ) * 64];
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        const int frag = q >> 2, rq = q & 3, fp = frag % FP, ph = frag / FP;
        const int p = p0 + fp * 32 + l31;
        if (p >= npts) continue;
        // channels of registers 4 rq + k: (r & 3) + 8 * (r >> 2) + 4 * half = 8 rq + 4 half + k (the C/D map of the 32 x 32 MFMA)
        const int o_first = o0 + 8 * rq + 4 * half;
        const float vin[4] = {v.x * g.acc_scale, v.y * g.acc_scale, v.z * g.acc_scale, v.w * g.acc_scale};
        if constexpr (TR) {
            const int pr_ = p / g.GW, pc = p - pr_ * g.GW;
            const int oy = 2 * pr_ + (ph >> 1), ox = 2 * pc + (ph & 1);
            if (oy >= g.OH || ox >= g.OW) continue;
            const int64_t pix = (int64_t)oy * g.OW + ox;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int o = o_first + k;
                if (o < g.O) y[((int64_t)b * g.O + o) * ohw + pix] = epilogue(vin[k], b, o, pix, ohw, g, e, ns);
            }
        } else {
            float outv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int o = o_first + k;
                if (o >= g.O) continue;
                outv[k] = epilogue(vin[k], b, o, p, ohw, g, e, ns);
                if (y) y[((int64_t)b * g.O + o) * ohw + p] = outv[k];
            }
            if (e.ys && o_first + 3 < g.O) {
                ia::SatWatch watch;
                split_store4(e.ys, e.styles_next, e.ys_planes, b, g.O, ohw, o_first, p, outv, watch);
                watch.report();
            }
        }
    }
}

template <int NP, bool TR>
int conv_small_launch_np(const h16x8* x8, const h16x8* w8, float* y, const Geo& g, const Epi& e, hipStream_t s) {
    constexpr int FP = 1;
    const size_t lds = (size_t)kSmallWaves * (TR ? 4 : 1) * FP * 4 * 64 * 4 * sizeof(float);
    const int npts = g.GH * g.GW;
    auto k = conv_small_kernel<NP, FP, TR>;
    if (const int rs = ia::reserve_lds((const void*)k, lds, "conv_small")) return rs;
    hipLaunchKernelGGL(k, dim3((g.O + 31) / 32, (npts + 32 * FP - 1) / (32 * FP), g.B), dim3(kSmallWaves * 64), lds, s, x8, w8, y, g, e);
    return ia::check_launch("ia_conv2d_mfma_sx(small)");
}

inline int conv_small_launch(const void* xs, int planes, const void* wk_split, float* y, const Geo& g, const Epi& e, bool transposed, hipStream_t s) {
    const h16x8* x8 = static_cast<const h16x8*>(xs);
    const h16x8* w8 = static_cast<const h16x8*>(wk_split);
    if (transposed) return planes == 2 ? conv_small_launch_np<2, true>(x8, w8, y, g, e, s) : conv_small_launch_np<1, true>(x8, w8, y, g, e, s);
    return planes == 2 ? conv_small_launch_np<2, false>(x8, w8, y, g, e, s) : conv_small_launch_np<1, false>(x8, w8, y, g, e, s);
}

}  // namespace

// ia_render_rays: the whole importance renderer for one ray per wavefront, fused.
//
//   coarse depths (stratified + injected jitter) -> tri-plane gather -> OSG decoder (density only)
//   -> ray march weights -> smoothed inverse-CDF importance resampling -> merge (stable, coarse first)
//   -> tri-plane gather + full decoder on the 96 merged samples -> composite -> rgb[32], depth, weight sum
//
// Replaces ImportanceRenderer_bsMotion.forward (training_avatar_texture/volumetric_rendering/renderer.py:
// 309-351 with sample_stratified :384-408, run_model :353-363, sample_from_planes :85-97,
// sample_importance/sample_pdf :410-469, unify_samples :372-382), OSGDecoder.forward
// (training_avatar_texture/triplane_v20.py:426-438) and MipRayMarcher2.run_forward
// (volumetric_rendering/ray_marcher.py:25-57).  No [B, R*S, 32] intermediate ever reaches HBM: the only
// traffic is the planes (read through L2), rays/jitter in and 34 floats per ray out.
//
// Work decomposition (wave64, one ray per wave, 16 samples x 4 lane-quarters per step):
//   lane = s + 16*q : s = sample within the group of 16, q = quarter.  The decoder runs on
//   v_mfma_f32_16x16x4_f32 as D[unit, sample] = W[unit, k] * X[k, sample]; with that orientation
//     - the tri-plane gather runs in its own lane role (lane = 4 * sample + chunk: four consecutive lanes read 64 contiguous
//       bytes of a texel, 32-bit texel offsets on a scalar plane base) and hands the blended features to the decoder role
//       through the sample's 128-byte colour slot in LDS: decoder lane (s, q) reads channels 8q..8q+7 -- that IS the B operand,
//     - layer-1 results land as 16 hidden units per lane which ARE the B operand of layer 2 (k permuted
//       identically on the weight side), so no cross-lane shuffle sits between gather, layer 1 and layer 2,
//     - the density row (1 of 33 outputs) is a 16-term VALU dot product + 2 butterfly adds instead of a
//       second, 94%-empty MFMA tile.
//   fp32 MFMA == fmaf chain bitwise (exact fp32); weights (pre-scaled by the FullyConnectedLayer gains) live in
//   LDS in fragment order, staged once per persistent workgroup.
//   Every sample is decoded exactly ONCE: the coarse pass evaluates density AND colours of its 48 samples and parks them in LDS
//   (33 floats per sample, 6.3 KB per ray), the fine pass does the same for the 48 importance samples, and the compositing pass
//   walks the merged order reading both from LDS.  (The first version decoded the coarse samples twice -- density only, then
//   everything again in merged order: a third of the tri-plane gathers and of layer 1, the two largest legs of the kernel.)
//
// Numerical contract: linspace bit rule of CPU torch; cumulative products / sums accumulate
// in fp64 and round each prefix to fp32 like CPU torch.cumprod / cumsum; searchsorted(right=True) semantics and
// the 47-bin / 45-weight quirk of sample_importance are preserved.
#include "ia_common.h"

// Activation constants are folded into the staged weights (r05): layer 1 weights and bias carry log2(e), so softplus is
// log2(1 + exp2(acc)) in units of ln 2, which the density row of layer 2 carries; the colour rows of layer 2 are staged NEGATED with
// bias -log2(e) b, so sigmoid is rcp(1 + exp2(acc)); both accumulators start from the bias instead of zero (10 -> 5 and 6 -> 4 vector
// instructions per activation).  Cross-row sums and broadcasts are lane swaps / DPP (v_permlane16/32_swap, row_share), no ds_bpermute.
// IA_RENDER_TRACE (tools/trace_render.py): workgroup 0 stamps s_memtime at the phase boundaries of its first rays into the
// dbg_sigma_coarse buffer (a profiling build: that debug output is not written).
#ifndef IA_RENDER_TRACE
#define IA_RENDER_TRACE 0
#endif
#if IA_RENDER_TRACE
#define IA_RSTAMP(slot) do { if (blockIdx.x == 0 && lane == 0 && tr_ray < 8 && p.dbg_sigma_coarse) \
    reinterpret_cast<unsigned long long*>(p.dbg_sigma_coarse)[(wave * 8 + tr_ray) * 32 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define IA_RSTAMP(slot) do { } while (0)
#endif

// IA_RENDER_F16X3 (r06): both decoder layers form their fp32 products from fp16 hi / lo pairs on v_mfma_f32_16x16x32_f16 (three products per
// k-step, lo * lo dropped: the arithmetic of the convolutions, csrc/conv_split.hip) instead of v_mfma_f32_16x16x4_f32: 24 matrix
// instructions of ~17 cycles per group of 16 samples instead of 64 of 32 cycles -- the fp32 pipe was 35 % of the kernel's issue time and did
// not overlap its vector work (profiles/r05_render_rays_variants.txt).  0 builds the exact-fp32 decoder of r02 - r05 (A/B builds).
#ifndef IA_RENDER_F16X3
#define IA_RENDER_F16X3 1
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8r __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2r __attribute__((ext_vector_type(2)));

// Scales of the fp16 pair form.  Activations are split at 2^8 (|v| < 255: plane features and softplus outputs; their low parts stay
// normal fp16 numbers down to |v| ~ 5e-4, below that the dropped bits are < 2.4e-7 absolute) and weights at 2^12 (|w| < 16), both
// low parts UNscaled, so the three products of a k-step share one accumulator at 2^20; it starts from bias * 2^20 and one multiply by
// 2^-20 brings the sums back.  Power-of-two scalings: exact.
constexpr float kActScale = 256.f, kWgtScale = 4096.f, kAccScale = kActScale * kWgtScale;

// (hi, lo) of two scaled values: hi = fp16(v) towards zero (v_cvt_pkrtz: any rounding of hi is exact as long as lo is its residual),
// lo = fp16(v - hi).
__device__ __forceinline__ void split2(float v0, float v1, h16x2r& hi, h16x2r& lo) {
    hi = __builtin_bit_cast(h16x2r, __builtin_amdgcn_cvt_pkrtz(v0, v1));
    lo = __builtin_bit_cast(h16x2r, __builtin_amdgcn_cvt_pkrtz(v0 - (float)hi[0], v1 - (float)hi[1]));
}

constexpr int NS = 48;            // coarse samples == importance samples (depth_resolution[_importance])
constexpr int NM = 2 * NS;        // merged
constexpr int WAVES = 8;          // rays per workgroup pass: one workgroup per CU, two waves per SIMD, one LDS image of the weights
constexpr int kEarlyPlanes = 1;   // planes whose texels are prefetched one sample group ahead (the third is loaded at use: registers)

struct Params {
    const float* planes;          // [B][3][PH][PW][32] channels-last fp32
    const float* rays_o;          // [B][R][3]
    const float* rays_d;          // [B][R][3]
    const float* jitter;          // [B][R][48]
    const float* u_imp;           // [B][R][48] sorted uniform draws of the importance pass, or null: linspace(0, 1, 48)
    const float* dist;            // device scalar: batch mean of |ray origin| (renderer.py:311); [B] with dist_per_frame
    const float* limits;          // ImportanceRenderer (renderer.py:129-139): per-ray [near, far] of the 'auto' box limits [B][R][2], or null
    double range_start, range_end;  // ... its fixed rendering_options['ray_start'/'ray_end'] (python doubles) when range_fixed
    int range_fixed, flip_z;
    const float* w0; const float* b0; const float* w1; const float* b1;   // decoder parameters, reference layout
    float w0_gain, w1_gain, b_gain;
    float box_scale;              // 2 / box_warp
    int B, R, PH, PW;
    int white_back, channel_major, dist_per_frame;
    float* rgb;                   // [B][R][32], or [B][32][R] when channel_major
    // optional second copy of rgb in the operand format of the convolution that reads it (the SR head's first layer): fp16 hi / lo planes
    // [B][planes][4][R][8] of rgb * split_styles[b][channel] -- what ia_act_split would make of the [B][32][R] image (null: not written)
    void* split_out;
    const float* split_styles;    // [B][32] or null (unscaled)
    int split_planes;
    float* depth;                 // [B][R]   un-clamped (may be +inf), see ia_render_finalize
    float* wsum;                  // [B][R]
    float* minmax;                // [gridDim.x][2] per-workgroup min / max of all sample depths; dist_per_frame: [gridDim.x * WAVES][B][2]
    // optional stage outputs for parity tests (null in production)
    float* dbg_z_fine;            // [B][R][48]
    int* dbg_inds;                // [B][R][48]
    int* dbg_order;               // [B][R][96]
    float* dbg_w_coarse;          // [B][R][47]
    float* dbg_sigma_coarse;      // [B][R][48]
};

// LDS image (floats)
constexpr int A1_OFF = 0;                       // [4 tiles][8 ksteps][64 lanes]
constexpr int A2_OFF = A1_OFF + 4 * 8 * 64;     // [2 tiles][16 ksteps][64 lanes]
constexpr int WS_OFF = A2_OFF + 2 * 16 * 64;    // [16 ksteps][4 quarters]   density row of layer 2
constexpr int B0_OFF = WS_OFF + 64;             // [64]
constexpr int B1_OFF = B0_OFF + 64;             // [33] (+pad)
constexpr int B1C_OFF = B1_OFF + 40;            // [32] colour biases of layer 2 as the accumulators start from them, 16-byte aligned
constexpr int SCR_OFF = B1C_OFF + 32;           // per-wave scratch
constexpr int COL_OFF = 10 * NS;                // colours of the 96 samples, [sample][quarter][8] (coarse 0..47, fine 48..95)
constexpr int CS = 36;                          // floats per colour slot: 32 + 4 of padding (slots 128 bytes apart put the 16 samples of a group on two banks)
constexpr int SGF_OFF = COL_OFF + NM * CS;      // densities of the fine samples [48]
constexpr int SRC_OFF = SGF_OFF + NS;           // merged slot -> sample index [96] (int)
constexpr int SCR = SRC_OFF + NM;               // tc, sc, wc, av, pdf, cdf, bins, tf (8 x 48) + tm (96) + the above
constexpr int LDS_FLOATS = SCR_OFF + WAVES * SCR;

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// softplus = log(1 + exp(x)) on the hardware transcendentals: v_exp_f32(x * log2e) (what __expf is) and v_log_f32(.) * ln 2.
// __logf would add an extended-precision multiplication by ln 2 (two fmas and an add per call) and a range fix-up for denormal and
// infinite arguments; the argument here is 1 + exp(x) in [1, 1 + e^20] and v_log_f32 is good to 1 ulp, so the plain product stays
// within 1.5 ulp of the true logarithm -- the noise level of the CPU reference's own log1p(exp(x)).
__device__ __forceinline__ float exp_raw(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float log_of_normal(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309417f; }
// Branch-free on purpose (a bit select, not ?:): the 16 softplus of a lane must stay ONE basic block so that their exp / log
// chains interleave; as 16 conditional blocks each one exposes its own transcendental and LDS latencies.
__device__ __forceinline__ float softplus_fast(float x) {
    const float l = log_of_normal(1.f + exp_raw(x));
    const unsigned big = x > 20.f ? 0xffffffffu : 0u;
    return __uint_as_float((__float_as_uint(x) & big) | (__float_as_uint(l) & ~big));
}

// The same function in base 2: x2 = x * log2(e) in, softplus(x) / ln 2 out.  log2(1 + 2^x2) >= x2 always, and equals x2 to fp32
// precision from x2 = 25 up, so max(., x2) is the large-argument branch (torch's threshold 20 = 28.9 here: both sides agree to an
// ulp long before); the min keeps 2^x2 finite.  Five instructions, no constants.
__device__ __forceinline__ float softplus2_fast(float x2) {
    const float l = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(fminf(x2, 126.f)));
    return fmaxf(l, x2);
}

// x + (value of the lane 16 / 32 rows away): the xor-16 / xor-32 butterfly steps on the VALU (gfx950 lane swaps)
__device__ __forceinline__ float add_across_rows(float x) {
    const unsigned u = __float_as_uint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);       // {rows 0,0,2,2 ; rows 1,1,3,3}
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const unsigned v = __float_as_uint(s);
    const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);       // {lower half twice ; upper half twice}
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

constexpr float kMeanScale = (1.f / 3.f) * (IA_RENDER_F16X3 ? kActScale : 1.f);      // fl(1/3) times a power of two: the mean's bits, shifted

// CPU torch.linspace bit rule.
__device__ __forceinline__ float linspace_at(float s, float e, float step, int k, int n) {
    return (k < n / 2) ? __fmaf_rn(step, (float)k, s) : __fmaf_rn(-step, (float)(n - 1 - k), e);
}

// Bilinear, zero-padded, align_corners=False gather of channels [8q, 8q+8) of the three planes at one point,
// averaged over the planes (sample_from_planes + the mean of OSGDecoder.forward).
__device__ __forceinline__ void gather_features(const float* __restrict__ planes_b, int PH, int PW, int q,
                                                float x, float y, float z, float (&f)[8]) {
    float acc[3][8];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const float gx = (p == 2) ? z : x;
        const float gy = (p == 0) ? y : (p == 1 ? z : x);
        const float ix = (gx + 1.f) * (0.5f * (float)PW) - 0.5f;
        const float iy = (gy + 1.f) * (0.5f * (float)PH) - 0.5f;
        const float x0f = floorf(ix), y0f = floorf(iy);
        const float fx = ix - x0f, fy = iy - y0f;
        // clamp before the int conversion so far-away points cannot overflow; they are masked below anyway
        const int x0 = (int)fminf(fmaxf(x0f, -2.f), (float)PW + 1.f), y0 = (int)fminf(fmaxf(y0f, -2.f), (float)PH + 1.f);
        const float* pl = planes_b + (int64_t)p * PH * PW * 32 + 8 * q;
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[p][c] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int xi = x0 + (t & 1), yi = y0 + (t >> 1);
            const float wgt = ((t & 1) ? fx : 1.f - fx) * ((t >> 1) ? fy : 1.f - fy);
            if (xi >= 0 && xi < PW && yi >= 0 && yi < PH) {
                const float4* src = (const float4*)(pl + ((int64_t)yi * PW + xi) * 32);
                const float4 a = src[0], b = src[1];
                acc[p][0] = fmaf(a.x, wgt, acc[p][0]); acc[p][1] = fmaf(a.y, wgt, acc[p][1]);
                acc[p][2] = fmaf(a.z, wgt, acc[p][2]); acc[p][3] = fmaf(a.w, wgt, acc[p][3]);
                acc[p][4] = fmaf(b.x, wgt, acc[p][4]); acc[p][5] = fmaf(b.y, wgt, acc[p][5]);
                acc[p][6] = fmaf(b.z, wgt, acc[p][6]); acc[p][7] = fmaf(b.w, wgt, acc[p][7]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) f[c] = (acc[0][c] + acc[1][c] + acc[2][c]) * (1.f / 3.f);   // mean over planes (<= 1 ulp from x / 3)
}

// The same gather in two halves, so that the loads of the NEXT group of samples fly under the decoder of the current one:
// gather_issue computes the 12 texel addresses / bilinear weights and issues the 24 float4 loads (taps outside the plane read
// a clamped, valid texel with weight 0: fmaf(v, 0, acc) == acc, so the sum is the bits of gather_features), gather_reduce
// is the accumulation in the same order.
template <int P0, int P1>
__device__ __forceinline__ void gather_issue(const float* __restrict__ planes_b, int PH, int PW, int c, float x, float y, float z,
                                             float4 (&raw)[24], float (&wgt)[12]) {
    // planes_b is wave-uniform (scalar base); the texel offsets are 32-bit byte offsets (the host checks 3*PH*PW*128 < 2^32)
    const char* base = reinterpret_cast<const char*>(planes_b);
#pragma unroll
    for (int p = P0; p < P1; ++p) {
        const float gx = (p == 2) ? z : x;
        const float gy = (p == 0) ? y : (p == 1 ? z : x);
        const float ix = (gx + 1.f) * (0.5f * (float)PW) - 0.5f;
        const float iy = (gy + 1.f) * (0.5f * (float)PH) - 0.5f;
        const float x0f = floorf(ix), y0f = floorf(iy);
        const float fx = ix - x0f, fy = iy - y0f;
        const int x0 = (int)fminf(fmaxf(x0f, -2.f), (float)PW + 1.f), y0 = (int)fminf(fmaxf(y0f, -2.f), (float)PH + 1.f);
        const int x1 = x0 + 1, y1 = y0 + 1;
        // per-axis weights, zero outside the plane: the products are the four bilinear weights (0 for a tap outside)
        const float wx0 = (unsigned)x0 < (unsigned)PW ? 1.f - fx : 0.f, wx1 = (unsigned)x1 < (unsigned)PW ? fx : 0.f;
        const float wy0 = (unsigned)y0 < (unsigned)PH ? 1.f - fy : 0.f, wy1 = (unsigned)y1 < (unsigned)PH ? fy : 0.f;
        wgt[p * 4 + 0] = wx0 * wy0; wgt[p * 4 + 1] = wx1 * wy0; wgt[p * 4 + 2] = wx0 * wy1; wgt[p * 4 + 3] = wx1 * wy1;
        const unsigned xc0 = (unsigned)min(max(x0, 0), PW - 1), xc1 = (unsigned)min(max(x1, 0), PW - 1);
        const unsigned r0 = __umul24((unsigned)min(max(y0, 0), PH - 1), (unsigned)PW), r1 = __umul24((unsigned)min(max(y1, 0), PH - 1), (unsigned)PW);
        const unsigned po = (unsigned)p * (unsigned)(PH * PW) * 128u + 16u * (unsigned)c;     // plane + this lane's 16-byte chunk
        const unsigned off[4] = {((r0 + xc0) << 7) + po, ((r0 + xc1) << 7) + po, ((r1 + xc0) << 7) + po, ((r1 + xc1) << 7) + po};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            raw[(p * 4 + t) * 2] = *reinterpret_cast<const float4*>(base + off[t]);             // channels 4c .. 4c+3 (chunk c of the 128-byte texel)
            raw[(p * 4 + t) * 2 + 1] = *reinterpret_cast<const float4*>(base + off[t] + 64u);   // channels 16+4c .. 16+4c+3 (chunk 4 + c)
        }
    }
}

__device__ __forceinline__ void gather_reduce(const float4 (&raw)[24], const float (&wgt)[12], float (&f)[8]) {
    float acc[3][8];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[p][c] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 a = raw[(p * 4 + t) * 2], b = raw[(p * 4 + t) * 2 + 1];
            const float w = wgt[p * 4 + t];
            acc[p][0] = fmaf(a.x, w, acc[p][0]); acc[p][1] = fmaf(a.y, w, acc[p][1]);
            acc[p][2] = fmaf(a.z, w, acc[p][2]); acc[p][3] = fmaf(a.w, w, acc[p][3]);
            acc[p][4] = fmaf(b.x, w, acc[p][4]); acc[p][5] = fmaf(b.y, w, acc[p][5]);
            acc[p][6] = fmaf(b.z, w, acc[p][6]); acc[p][7] = fmaf(b.w, w, acc[p][7]);
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) f[c] = (acc[0][c] + acc[1][c] + acc[2][c]) * kMeanScale;      // mean over the planes (x kActScale in the fp16 pair form: exact)
}

// Gather layout -> MFMA operand layout, through the sample's colour slot in LDS (the colours overwrite it afterwards).
// In : f = channels {4c..4c+3, 16+4c..16+4c+3} of sample gj (what gather_reduce leaves in lane 4*gj + c).
// Out: f = channels 8q..8q+7 of sample s (lane s + 16q): the B operand of layer 1, k-slot q, k-step t <-> channel 8q + t.
__device__ __forceinline__ void features_to_operand(float* group_slots, int gj, int gc, int s, int q, float (&f)[8]) {
    float4* w = reinterpret_cast<float4*>(group_slots + gj * CS + 4 * gc);
    w[0] = make_float4(f[0], f[1], f[2], f[3]);
    w[4] = make_float4(f[4], f[5], f[6], f[7]);
    wave_sync();
    const float4* r = reinterpret_cast<const float4*>(group_slots + s * CS + 8 * q);
    const float4 a = r[0], b = r[1];
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// Layer 1 (32 -> 64, softplus) on MFMA.  In: f[8] = channels 8q..8q+7 of this lane's sample (x kActScale in the fp16 pair form).
// Out: h[T][r] = hidden unit 16T + 4q + r of this lane's sample.
__device__ __forceinline__ void decoder_hidden(const float* __restrict__ lds, int lane, int q, const float (&f)[8], f32x4 (&h)[4]) {
#pragma unroll
    for (int T = 0; T < 4; ++T) h[T] = *reinterpret_cast<const f32x4*>(lds + B0_OFF + 16 * T + 4 * q);      // accumulate onto the bias
#if IA_RENDER_F16X3
    // one k-step of 32: this lane's eight channels ARE its B fragment (k = 8q + j); A fragments [plane][tile][lane] staged in that order
    h16x8r f_hi, f_lo;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        h16x2r a, b;
        split2(__builtin_amdgcn_fmed3f(f[j], -65000.f, 65000.f), __builtin_amdgcn_fmed3f(f[j + 1], -65000.f, 65000.f), a, b);
        f_hi[j] = a[0]; f_hi[j + 1] = a[1]; f_lo[j] = b[0]; f_lo[j + 1] = b[1];
    }
    const h16x8r* a1 = reinterpret_cast<const h16x8r*>(lds + A1_OFF);
#pragma unroll
    for (int T = 0; T < 4; ++T) h[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[(4 + T) * 64 + lane], f_hi, h[T], 0, 0, 0);      // lo * hi
#pragma unroll
    for (int T = 0; T < 4; ++T) h[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[T * 64 + lane], f_lo, h[T], 0, 0, 0);            // hi * lo
#pragma unroll
    for (int T = 0; T < 4; ++T) h[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[T * 64 + lane], f_hi, h[T], 0, 0, 0);            // hi * hi
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            h[T][r] = softplus2_fast(h[T][r] * (1.f / kAccScale));
#else
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int T = 0; T < 4; ++T)
            h[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(lds[A1_OFF + (T * 8 + t) * 64 + lane], f[t], h[T], 0, 0, 0);
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            h[T][r] = softplus2_fast(h[T][r]);
#endif
}

// Density: output row 0 of layer 2.  Each lane owns 16 of the 64 hidden units; butterfly over the 4 quarters.
__device__ __forceinline__ float decoder_sigma(const float* __restrict__ lds, int q, const f32x4 (&h)[4]) {
    float part = 0.f;
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) part = fmaf(h[T][r], lds[WS_OFF + (T * 4 + r) * 4 + q], part);
    return add_across_rows(part) + lds[B1_OFF];
}


// Colours: output rows 1..32 of layer 2 on MFMA.  c[U][r] = channel 16U + 4q + r of this lane's sample.
__device__ __forceinline__ void decoder_rgb(const float* __restrict__ lds, int lane, int q, const f32x4 (&h)[4], f32x4 (&c)[2]) {
    c[0] = *reinterpret_cast<const f32x4*>(lds + B1C_OFF + 4 * q);
    c[1] = *reinterpret_cast<const f32x4*>(lds + B1C_OFF + 16 + 4 * q);
#if IA_RENDER_F16X3
    // two k-steps of 32 hidden units: step t2 takes this lane's h[2 t2][0..3], h[2 t2 + 1][0..3] (k = 8q + j <-> unit 16 (2 t2 + (j >> 2)) + 4q + (j & 3),
    // the weight side is staged with the same permutation); A fragments [plane][tile U][step t2][lane]
    const h16x8r* a2 = reinterpret_cast<const h16x8r*>(lds + A2_OFF);
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
        h16x8r b_hi, b_lo;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            h16x2r a, b;
            split2(h[2 * t2 + (j >> 2)][j & 3] * kActScale, h[2 * t2 + (j >> 2)][(j & 3) + 1] * kActScale, a, b);
            b_hi[j] = a[0]; b_hi[j + 1] = a[1]; b_lo[j] = b[0]; b_lo[j + 1] = b[1];
        }
#pragma unroll
        for (int U = 0; U < 2; ++U) c[U] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[((2 + U) * 2 + t2) * 64 + lane], b_hi, c[U], 0, 0, 0);   // lo * hi
#pragma unroll
        for (int U = 0; U < 2; ++U) c[U] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[(U * 2 + t2) * 64 + lane], b_lo, c[U], 0, 0, 0);         // hi * lo
#pragma unroll
        for (int U = 0; U < 2; ++U) c[U] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[(U * 2 + t2) * 64 + lane], b_hi, c[U], 0, 0, 0);         // hi * hi
    }
#pragma unroll
    for (int U = 0; U < 2; ++U)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            c[U][r] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(c[U][r] * (1.f / kAccScale))) * 1.002f - 0.001f;      // c = -log2(e) * logit
        }
#else
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int U = 0; U < 2; ++U)
                c[U] = __builtin_amdgcn_mfma_f32_16x16x4f32(lds[A2_OFF + (U * 16 + T * 4 + r) * 64 + lane], h[T][r], c[U], 0, 0, 0);
#pragma unroll
    for (int U = 0; U < 2; ++U)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            c[U][r] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(c[U][r])) * 1.002f - 0.001f;      // c = -log2(e) * logit
        }
#endif
}


// Data-parallel-primitive moves inside a row of 16 lanes (the 16 samples of a group live in one DPP row per quarter): register
// moves on the VALU instead of ds_bpermute round trips through the LDS crossbar.
//   row_ror<N>: lane s receives lane (s - N) mod 16 of its row (row_ror1: lane 0 receives lane 15);
//   row_shr<N>: lane s >= N receives lane s - N, lanes below N keep their own value.
template <int N>
__device__ __forceinline__ float row_ror(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x120 + N, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_ror1(float x) { return row_ror<1>(x); }
template <int N>
__device__ __forceinline__ double row_shr(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x110 + N, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x110 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double row_last(double x) {      // lane 15 of the caller's row of 16, in every lane of that row (DPP row_share:15)
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x15f, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x15f, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double lane_value(double x, int l) {      // wave-uniform copy of lane l's value (v_readlane, no LDS)
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}

// Smoothed inverse-CDF importance resampling of one ray, all in per-wave LDS scratch.
//   in : tc[48] coarse depths, wc[47] coarse weights        out: tf[48] fine depths, returns ind for lane < 48
__device__ __forceinline__ int importance_resample(float* scr, int lane, const float* u_row = nullptr) {
    float* tc = scr; float* wc = scr + 2 * NS; float* av = scr + 3 * NS; float* pdf = scr + 4 * NS;
    float* cdf = scr + 5 * NS; float* bins = scr + 6 * NS; float* tf = scr + 7 * NS;
    // max_pool1d(2,1,pad 1) then avg_pool1d(2,1), + 0.01 (renderer.py:421-423); bins = depth midpoints (:425)
    if (lane < NS - 1) {
        const float wl = lane > 0 ? wc[lane - 1] : -INFINITY, wm = wc[lane], wr = lane + 1 < NS - 1 ? wc[lane + 1] : -INFINITY;
        av[lane] = (fmaxf(wl, wm) + fmaxf(wm, wr)) * 0.5f + 0.01f;
        bins[lane] = 0.5f * (tc[lane] + tc[lane + 1]);
    }
    wave_sync();
    // pdf over the 45 interior weights (renderer.py:426,443-444).  The normaliser reproduces the summation ORDER of
    // CPU torch.sum over a contiguous row (ATen SumKernel vectorized_inner_sum: 8-lane vectors, 4-way ILP cascade,
    // scalar tail first, then the 8 lane partials) because the last importance sample (u == 1) compares against
    // cdf[45] ~= 1 and flips with a 1-ulp change of the total.
    if (lane < 8) {
        float part = (av[1 + lane] + 1e-5f) + (av[33 + lane] + 1e-5f);
        part += av[9 + lane] + 1e-5f;
        part += av[17 + lane] + 1e-5f;
        part += av[25 + lane] + 1e-5f;
        pdf[lane] = part;                                   // scratch use of pdf[0..7]
    }
    wave_sync();
    float total = 0.f;
    for (int j = 41; j <= NS - 3; ++j) total += av[j] + 1e-5f;
    for (int j = 0; j < 8; ++j) total += pdf[j];
    wave_sync();
    if (lane < NS - 3) pdf[lane] = (av[lane + 1] + 1e-5f) / total;
    wave_sync();
    // cdf[0] = 0, cdf[j] = fl32(sum_{i<j} pdf[i]) accumulated in fp64 (CPU torch.cumsum, C10).  The pdf entries are fp32 numbers
    // in [1e-4, 1] (every smoothed weight is >= 0.01 of a total <= 46 * 1.02) with a sum of 1, so every partial sum needs at most
    // 24 + 14 significant bits: the fp64 additions are EXACT and a parallel scan gives the bits of the sequential loop.
    {
        const double mine = lane < NS - 3 ? (double)pdf[lane] : 0.0;
        double incl = mine;
        { const double u = row_shr<1>(incl); if ((lane & 15) >= 1) incl += u; }
        { const double u = row_shr<2>(incl); if ((lane & 15) >= 2) incl += u; }
        { const double u = row_shr<4>(incl); if ((lane & 15) >= 4) incl += u; }
        { const double u = row_shr<8>(incl); if ((lane & 15) >= 8) incl += u; }
        const double r0 = lane_value(incl, 15), r1 = lane_value(incl, 31);  // totals of rows 0 and 1 (lanes >= 48 hold no cdf entry)
        const int row = lane >> 4;
        const double before = row == 0 ? 0.0 : (row == 1 ? r0 : r0 + r1);
        if (lane < NS - 2) cdf[lane] = (float)(before + incl - mine);
    }
    wave_sync();
    int ind = 0;
    if (lane < NS) {
        const float ustep = 1.0f / (float)(NS - 1);
        // evaluation: the deterministic grid linspace(0, 1, 48) (renderer.py:450); otherwise the caller's uniform draws (:453), which
        // it hands over sorted so that the fine depths come out sorted like the grid's
        const float u = u_row ? u_row[lane] : linspace_at(0.f, 1.f, ustep, lane, NS);
        // searchsorted(right=True) on the non-decreasing cdf[0..45]: number of entries <= u, by bisection
        int lo = 0, hi = NS - 2;                      // the count lies in [lo, hi]
#pragma unroll
        for (int it = 0; it < 6; ++it) {              // 2^6 > 47 candidates
            const int mid = (lo + hi) >> 1;
            const bool active = lo < hi, le = cdf[mid] <= u;     // (mid <= 46 < NS: always a readable slot)
            lo = (active && le) ? mid + 1 : lo;
            hi = (active && !le) ? mid : hi;
        }
        ind = lo;
        const int below = max(ind - 1, 0), above = min(ind, NS - 3);
        const float c0 = cdf[below], c1 = cdf[above], b0 = bins[below], b1 = bins[above];
        float denom = c1 - c0;
        if (denom < 1e-5f) denom = 1.f;
        tf[lane] = b0 + (u - c0) / denom * (b1 - b0);
    }
    wave_sync();
    return ind;
}

// Stable merge of two ascending 48-lists into tm[96]; returns this lane's (coarse, fine) destination slots.
__device__ __forceinline__ void merge_sorted(float* scr, int lane, int& pos_c, int& pos_f) {
    float* tc = scr; float* tf = scr + 7 * NS; float* tm = scr + 8 * NS;
    pos_c = pos_f = 0;
    if (lane < NS) {
        const float a = tc[lane], b = tf[lane];
        // both lists ascend: nc = #{tf < a} (lower bound), nf = #{tc <= b} (upper bound), by bisection over 49 candidates each
        int lc = 0, hc = NS, lf = 0, hf = NS;
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int mc = (lc + hc) >> 1, mf = (lf + hf) >> 1;
            const bool ac = lc < hc, af = lf < hf;
            const bool c_lt = tf[min(mc, NS - 1)] < a, f_le = tc[min(mf, NS - 1)] <= b;
            lc = (ac && c_lt) ? mc + 1 : lc; hc = (ac && !c_lt) ? mc : hc;
            lf = (af && f_le) ? mf + 1 : lf; hf = (af && !f_le) ? mf : hf;
        }
        pos_c = lane + lc; pos_f = lane + lf;
        tm[pos_c] = a; tm[pos_f] = b;
    }
    wave_sync();
}

// SQ: square planes (every tri-plane generator's case).  The gather then sees PH == PW at compile time and the per-axis work of a
// coordinate that two planes share (x: planes 0, 1 as column and plane 2 as row; z: plane 1 as row, plane 2 as column) is done once.
// BOX: the ImportanceRenderer form (ia_render_rays_box: per-ray limits or a fixed range, flip_z); its scalars stay out of the generator's kernel.
template <bool SQ, bool BOX>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void render_rays_kernel(Params p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = lane & 15, q = lane >> 4;       // decoder (MFMA operand) role of the lane: sample s of the group, k-slot q
    const int gj = lane >> 2, gc = lane & 3;      // gather role: sample gj of the group, 16-byte chunks gc and 4 + gc of its texels

    // layer 1 produces log2(e) * (W0 f + b0); the hidden value is softplus / ln 2, so the density row carries ln 2;
    // the colour rows need -log2(e) * ln 2 = -1: an exact negation
    constexpr float kIn = 1.44269504088896340736f, kHid = 0.693147180559945309417f, kRgb = -1.f;
    // ---- stage decoder weights in MFMA-fragment order
#if IA_RENDER_F16X3
    {
        // fp16 pair form: element (fragment, lane, j) -> halves hi / lo of the weight x kWgtScale, planes [hi | lo] of [fragment][lane][8]
        _Float16* a1 = reinterpret_cast<_Float16*>(lds + A1_OFF);
        for (int e = tid; e < 4 * 64 * 8; e += WAVES * 64) {
            const int j = e & 7, l = (e >> 3) & 63, T = e >> 9;
            const float w = p.w0[(16 * T + (l & 15)) * 32 + 8 * (l >> 4) + j] * p.w0_gain * kIn * kWgtScale;
            const _Float16 hi = (_Float16)w;
            a1[(T * 64 + l) * 8 + j] = hi;
            a1[((4 + T) * 64 + l) * 8 + j] = (_Float16)(w - (float)hi);
        }
        _Float16* a2 = reinterpret_cast<_Float16*>(lds + A2_OFF);
        for (int e = tid; e < 4 * 64 * 8; e += WAVES * 64) {
            const int j = e & 7, l = (e >> 3) & 63, t2 = (e >> 9) & 1, U = e >> 10;
            const float w = p.w1[(1 + 16 * U + (l & 15)) * 64 + 16 * (2 * t2 + (j >> 2)) + 4 * (l >> 4) + (j & 3)] * p.w1_gain * kRgb * kWgtScale;
            const _Float16 hi = (_Float16)w;
            a2[((U * 2 + t2) * 64 + l) * 8 + j] = hi;
            a2[(((2 + U) * 2 + t2) * 64 + l) * 8 + j] = (_Float16)(w - (float)hi);
        }
    }
    if (tid < 64) {
        const int k = tid >> 2, qq = tid & 3;
        lds[WS_OFF + tid] = p.w1[16 * (k >> 2) + 4 * qq + (k & 3)] * p.w1_gain * kHid;
        lds[B0_OFF + tid] = p.b0[tid] * p.b_gain * kIn * kAccScale;
    }
#else
    for (int e = tid; e < 4 * 8 * 64; e += WAVES * 64) {
        const int l = e & 63, t = (e >> 6) & 7, T = e >> 9;
        lds[A1_OFF + e] = p.w0[(16 * T + (l & 15)) * 32 + 8 * (l >> 4) + t] * p.w0_gain * kIn;
    }
    for (int e = tid; e < 2 * 16 * 64; e += WAVES * 64) {
        const int l = e & 63, k = (e >> 6) & 15, U = e >> 10;     // k = T*4 + r
        lds[A2_OFF + e] = p.w1[(1 + 16 * U + (l & 15)) * 64 + 16 * (k >> 2) + 4 * (l >> 4) + (k & 3)] * p.w1_gain * kRgb;
    }
    if (tid < 64) {
        const int k = tid >> 2, qq = tid & 3;
        lds[WS_OFF + tid] = p.w1[16 * (k >> 2) + 4 * qq + (k & 3)] * p.w1_gain * kHid;
        lds[B0_OFF + tid] = p.b0[tid] * p.b_gain * kIn;
    }
#endif
    if (tid < 33) lds[B1_OFF + tid] = p.b1[tid] * p.b_gain;
    if (tid < 32) lds[B1C_OFF + tid] = p.b1[1 + tid] * p.b_gain * -1.44269504088896340736f * (IA_RENDER_F16X3 ? kAccScale : 1.f);
    __syncthreads();

    float* scr = lds + SCR_OFF + wave * SCR;
    float* tc = scr; float* sc = scr + NS; float* wc = scr + 2 * NS; float* tm = scr + 8 * NS;

    float blk_min = INFINITY, blk_max = -INFINITY;
    // dist_per_frame: the depth clamp is every frame's own sample range (the script renders those frames one call each), kept per wave
    // and frame in minmax[((workgroup * WAVES + wave) * B + b) * 2]: a wave meets its frames in ascending order and writes a frame's
    // range when it leaves it (lane 0 wrote the empty range to all its slots first: same lane, program order)
    int cur_b = -1;
    float* mm_wave = p.minmax + (int64_t)(blockIdx.x * WAVES + wave) * p.B * 2;
    if (p.dist_per_frame && lane == 0)
        for (int i = 0; i < p.B; ++i) { mm_wave[2 * i] = INFINITY; mm_wave[2 * i + 1] = -INFINITY; }

    const int nrays = p.B * p.R;
    const int PHs = SQ ? p.PW : p.PH;
    const float z_scale = (BOX && p.flip_z) ? -p.box_scale : p.box_scale;      // run_model's `sample_coordinates[..., -1] *= -1` (renderer.py:196-197): exact
#if IA_RENDER_TRACE
    int tr_ray = -1;
#endif
    for (int ray0 = blockIdx.x * WAVES; ray0 < nrays; ray0 += gridDim.x * WAVES) {
        const int ray = __builtin_amdgcn_readfirstlane(ray0 + wave);      // wave-uniform: ray constants and the plane base are scalars
        if (ray >= nrays) continue;
#if IA_RENDER_TRACE
        ++tr_ray;
#endif
        IA_RSTAMP(0);
        const int b = ray / p.R;
        if (p.dist_per_frame && b != cur_b) {
            if (cur_b >= 0 && lane == 0) { mm_wave[2 * cur_b] = blk_min; mm_wave[2 * cur_b + 1] = blk_max; }
            blk_min = INFINITY; blk_max = -INFINITY; cur_b = b;
        }
        // depth range: renderer.py:311-313,404-406 (python doubles, fp32 tensors); one value for the batch, or one per frame when the
        // caller renders several single-frame calls of the script as one batch (wave-uniform scalar work either way)
        // ImportanceRenderer (renderer.py:129-139): a fixed [ray_start, ray_end] takes the same torch.linspace route with the caller's doubles
        double r_lo, r_hi;
        if (BOX && p.range_fixed) { r_lo = p.range_start; r_hi = p.range_end; }
        else if (BOX) { r_lo = 0.0; r_hi = 0.0; }
        else { const double dist = (double)p.dist[p.dist_per_frame ? b : 0]; r_lo = dist - 0.45; r_hi = dist + 0.6; }
        const float t_start = (float)r_lo, t_end = (float)r_hi;
        const float t_step = (t_end - t_start) / (float)(NS - 1);
        const float t_delta = (float)((r_hi - r_lo) / (double)(NS - 1));
        const float* planes_b = p.planes + (int64_t)b * 3 * p.PH * p.PW * 32;
        const float ox = p.rays_o[ray * 3 + 0], oy = p.rays_o[ray * 3 + 1], oz = p.rays_o[ray * 3 + 2];
        const float dx = p.rays_d[ray * 3 + 0], dy = p.rays_d[ray * 3 + 1], dz = p.rays_d[ray * 3 + 2];

        // ---- coarse pass: density only
        if (BOX && p.limits) {      // (kernel-uniform) per-ray box limits: math_utils.linspace (math_utils.py:101-118) = start + (k / 47) * (stop - start)
            // in fp32 tensor arithmetic, noise scaled by the fp32 tensor (end - start) / 47 (renderer.py:236-238)
            const float lo = p.limits[(int64_t)ray * 2], hi = p.limits[(int64_t)ray * 2 + 1];
            if (lane < NS) tc[lane] = (lo + ((float)lane / (float)(NS - 1)) * (hi - lo)) + p.jitter[(int64_t)ray * NS + lane] * ((hi - lo) / (float)(NS - 1));
        } else if (lane < NS) tc[lane] = linspace_at(t_start, t_end, t_step, lane, NS) + p.jitter[(int64_t)ray * NS + lane] * t_delta;
        wave_sync();
        float4 raw[24]; float wgt[12];
        {
            const float t = tc[gj];
            gather_issue<0, kEarlyPlanes>(planes_b, PHs, p.PW, gc, (ox + t * dx) * p.box_scale, (oy + t * dy) * p.box_scale, (oz + t * dz) * z_scale, raw, wgt);
        }
#pragma unroll 1
        for (int g = 0; g < NS / 16; ++g) {
            float f[8]; f32x4 h[4];
            asm volatile("" ::: "memory");   // decoder weights are re-read from LDS every group: their registers hold the prefetched texels
            {
                const float t = tc[16 * g + gj];
                gather_issue<kEarlyPlanes, 3>(planes_b, PHs, p.PW, gc, (ox + t * dx) * p.box_scale, (oy + t * dy) * p.box_scale, (oz + t * dz) * z_scale, raw, wgt);
            }
            gather_reduce(raw, wgt, f);
            features_to_operand(scr + COL_OFF + (16 * g) * CS, gj, gc, s, q, f);
            IA_RSTAMP(1 + 4 * g);
            if (g + 1 < NS / 16) {      // next group's loads fly under this group's decoder
                const float t = tc[16 * (g + 1) + gj];
                gather_issue<0, kEarlyPlanes>(planes_b, PHs, p.PW, gc, (ox + t * dx) * p.box_scale, (oy + t * dy) * p.box_scale, (oz + t * dz) * z_scale, raw, wgt);
            }
            decoder_hidden(lds, lane, q, f, h);
            IA_RSTAMP(2 + 4 * g);
            const float sg = decoder_sigma(lds, q, h);
            if (q == 0) sc[16 * g + s] = sg;
            IA_RSTAMP(3 + 4 * g);
            f32x4 col[2];
            decoder_rgb(lds, lane, q, h, col);
            float4* dst = reinterpret_cast<float4*>(scr + COL_OFF + (16 * g + s) * CS + 8 * q);
            dst[0] = make_float4(col[0][0], col[0][1], col[0][2], col[0][3]);
            dst[1] = make_float4(col[1][0], col[1][1], col[1][2], col[1][3]);
            IA_RSTAMP(4 + 4 * g);
        }
        wave_sync();
        // ---- coarse ray march: weights only (ray_marcher.py:26-42)
        {
            float alpha = 0.f;
            double fac = 1.0;
            if (lane < NS - 1) {
                const float delta = tc[lane + 1] - tc[lane];
                const float dm = softplus_fast((sc[lane] + sc[lane + 1]) * 0.5f - 1.f);
                alpha = 1.f - __expf(-(dm * delta));
                fac = (double)(1.f - alpha + 1e-10f);
            }
            double incl = fac;
            // transmittance = exclusive fp64 prefix product: scan inside the rows of 16 lanes (DPP), then the totals of the rows before
            { const double u = row_shr<1>(incl); if ((lane & 15) >= 1) incl *= u; }
            { const double u = row_shr<2>(incl); if ((lane & 15) >= 2) incl *= u; }
            { const double u = row_shr<4>(incl); if ((lane & 15) >= 4) incl *= u; }
            { const double u = row_shr<8>(incl); if ((lane & 15) >= 8) incl *= u; }
            const double r0 = lane_value(incl, 15), r1 = lane_value(incl, 31);
            const double up1 = row_shr<1>(incl);
            const int row = lane >> 4;
            const double before = row == 0 ? 1.0 : (row == 1 ? r0 : r0 * r1);
            const float trans = (float)((lane & 15) == 0 ? before : before * up1);
            if (lane < NS - 1) {
                wc[lane] = alpha * trans;
                if (p.dbg_w_coarse) p.dbg_w_coarse[(int64_t)ray * (NS - 1) + lane] = alpha * trans;
            }
            if (!IA_RENDER_TRACE && p.dbg_sigma_coarse && lane < NS) p.dbg_sigma_coarse[(int64_t)ray * NS + lane] = sc[lane];
        }
        wave_sync();
        IA_RSTAMP(13);
        // ---- importance resampling + merge
        const int ind = importance_resample(scr, lane, p.u_imp ? p.u_imp + (int64_t)ray * NS : nullptr);
        IA_RSTAMP(14);
        int pos_c, pos_f;
        merge_sorted(scr, lane, pos_c, pos_f);
        IA_RSTAMP(15);
        if (lane < NS) {
            if (p.dbg_z_fine) p.dbg_z_fine[(int64_t)ray * NS + lane] = scr[7 * NS + lane];
            if (p.dbg_inds) p.dbg_inds[(int64_t)ray * NS + lane] = ind;
            if (p.dbg_order) { p.dbg_order[(int64_t)ray * NM + pos_c] = lane; p.dbg_order[(int64_t)ray * NM + pos_f] = NS + lane; }
        }
        blk_min = fminf(blk_min, tm[0]);
        blk_max = fmaxf(blk_max, tm[NM - 1]);
        int* src = reinterpret_cast<int*>(scr + SRC_OFF);
        if (lane < NS) { src[pos_c] = lane; src[pos_f] = NS + lane; }

        // ---- fine pass: the 48 importance samples, decoded once (density + colours) into LDS
        {
            const float* tf = scr + 7 * NS;
            {
                const float t = tf[gj];
                gather_issue<0, kEarlyPlanes>(planes_b, PHs, p.PW, gc, (ox + t * dx) * p.box_scale, (oy + t * dy) * p.box_scale, (oz + t * dz) * z_scale, raw, wgt);
            }
#pragma unroll 1
            for (int g = 0; g < NS / 16; ++g) {
                const float t = tf[16 * g + gj];
                float f[8]; f32x4 h[4], col[2];
                asm volatile("" ::: "memory");
                gather_issue<kEarlyPlanes, 3>(planes_b, PHs, p.PW, gc, (ox + t * dx) * p.box_scale, (oy + t * dy) * p.box_scale, (oz + t * dz) * z_scale, raw, wgt);
                gather_reduce(raw, wgt, f);
                features_to_operand(scr + COL_OFF + (NS + 16 * g) * CS, gj, gc, s, q, f);
                IA_RSTAMP(16 + 4 * g);
                if (g + 1 < NS / 16) {
                    const float tn = tf[16 * (g + 1) + gj];
                    gather_issue<0, kEarlyPlanes>(planes_b, PHs, p.PW, gc, (ox + tn * dx) * p.box_scale, (oy + tn * dy) * p.box_scale, (oz + tn * dz) * z_scale, raw, wgt);
                }
                decoder_hidden(lds, lane, q, f, h);
                IA_RSTAMP(17 + 4 * g);
                const float sg = decoder_sigma(lds, q, h);
                if (q == 0) scr[SGF_OFF + 16 * g + s] = sg;
                IA_RSTAMP(18 + 4 * g);
                decoder_rgb(lds, lane, q, h, col);
                float4* dst = reinterpret_cast<float4*>(scr + COL_OFF + (NS + 16 * g + s) * CS + 8 * q);
                dst[0] = make_float4(col[0][0], col[0][1], col[0][2], col[0][3]);
                dst[1] = make_float4(col[1][0], col[1][1], col[1][2], col[1][3]);
                IA_RSTAMP(19 + 4 * g);
            }
        }
        wave_sync();
        IA_RSTAMP(28);

        // ---- compositing over the 96 merged samples, 16 at a time; lane s composites the interval that ENDS at its sample.
        // Same arithmetic in the same order as when the colours came straight from the decoder: only their source changed.
        float acc_c[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc_c[c] = 0.f;
        float acc_w = 0.f, acc_z = 0.f;
        double carry_T = 1.0;
        float prev_t = 0.f, prev_sg = 0.f, prev_c[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) prev_c[c] = 0.f;
#pragma unroll 1
        for (int g = 0; g < NM / 16; ++g) {
            const float t = tm[16 * g + s];
            const int si = src[16 * g + s];
            const float sg = si < NS ? sc[si] : scr[SGF_OFF + si - NS];
            float cur_c[8];
            {
                const float4* cs = reinterpret_cast<const float4*>(scr + COL_OFF + si * CS + 8 * q);
                const float4 c0 = cs[0], c1 = cs[1];
                cur_c[0] = c0.x; cur_c[1] = c0.y; cur_c[2] = c0.z; cur_c[3] = c0.w;
                cur_c[4] = c1.x; cur_c[5] = c1.y; cur_c[6] = c1.z; cur_c[7] = c1.w;
            }
            // neighbour (previous sample) values: lane s-1 of the same quarter; lane 0 takes what the rotation delivered to it in
            // the previous group (= that group's lane 15)
            const float rot_t = row_ror1(t), rot_sg = row_ror1(sg);
            float rot_c[8], nb_c[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) rot_c[c] = row_ror1(cur_c[c]);
            const float nb_t = (s == 0) ? prev_t : rot_t, nb_sg = (s == 0) ? prev_sg : rot_sg;
#pragma unroll
            for (int c = 0; c < 8; ++c) nb_c[c] = (s == 0) ? prev_c[c] : rot_c[c];
            const bool has_interval = (g > 0) || (s > 0);
            float alpha = 0.f;
            double fac = 1.0;
            if (has_interval) {
                const float dm = softplus_fast((nb_sg + sg) * 0.5f - 1.f);
                alpha = 1.f - __expf(-(dm * (t - nb_t)));
                fac = (double)(1.f - alpha + 1e-10f);
            }
            double incl = fac;
            { const double u = row_shr<1>(incl); if (s >= 1) incl *= u; }
            { const double u = row_shr<2>(incl); if (s >= 2) incl *= u; }
            { const double u = row_shr<4>(incl); if (s >= 4) incl *= u; }
            { const double u = row_shr<8>(incl); if (s >= 8) incl *= u; }
            const double up1 = row_shr<1>(incl);
            const double excl = (s == 0) ? 1.0 : up1;
            const float trans = (float)(carry_T * excl);
            carry_T *= row_last(incl);
            const float wgt_ = alpha * trans;
#pragma unroll
            for (int c = 0; c < 8; ++c) acc_c[c] = fmaf(wgt_, (nb_c[c] + cur_c[c]) * 0.5f, acc_c[c]);
            acc_w += wgt_;
            acc_z = fmaf(wgt_, (nb_t + t) * 0.5f, acc_z);
            prev_t = rot_t; prev_sg = rot_sg;
#pragma unroll
            for (int c = 0; c < 8; ++c) prev_c[c] = rot_c[c];
        }
        // reduce the 16 lanes of each quarter: rotations by 8, 4, 2, 1 add the same pairs as the xor butterfly (after the step of
        // distance d the partial sums are periodic with period d), so every lane ends with the butterfly's bits
#define IA_ROW_SUM_STEP(N)                                                            \
        {                                                                             \
            _Pragma("unroll") for (int c = 0; c < 8; ++c) acc_c[c] += row_ror<N>(acc_c[c]); \
            acc_w += row_ror<N>(acc_w);                                               \
            acc_z += row_ror<N>(acc_z);                                               \
        }
        IA_RSTAMP(29);
        IA_ROW_SUM_STEP(8) IA_ROW_SUM_STEP(4) IA_ROW_SUM_STEP(2) IA_ROW_SUM_STEP(1)
#undef IA_ROW_SUM_STEP
        if (s == 0) {
            const int64_t ch_stride = p.channel_major ? p.R : 1;
            float* out = p.channel_major ? p.rgb + (int64_t)b * 32 * p.R + (ray - b * p.R) : p.rgb + (int64_t)ray * 32;
            float fin[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float v = acc_c[c];
                if (p.white_back) v = v + 1.f - acc_w;
                fin[c] = v * 2.f - 1.f;
                out[(16 * (c >> 2) + 4 * q + (c & 3)) * ch_stride] = fin[c];
            }
            if (p.split_out) {      // (kernel-uniform) the same values in the split format, multiplied by the consumer's styles: see Params
                typedef _Float16 h16x4r __attribute__((ext_vector_type(4)));
                ia::SatWatch watch;
                char* base = static_cast<char*>(p.split_out);
                const int64_t pix = ray - b * p.R;
#pragma unroll
                for (int U = 0; U < 2; ++U) {      // this lane's channels 16 U + 4 q .. + 3: half of the 16-byte unit of octet 2 U + (q >> 1)
                    h16x4r hi, lo;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int ch = 16 * U + 4 * q + k;
                        const float t = p.split_styles ? fin[4 * U + k] * p.split_styles[b * 32 + ch] : fin[4 * U + k];
                        _Float16 h, l;
                        if (p.split_planes == 2) { ia::split_f16(t, h, l, watch); hi[k] = h; lo[k] = l; }
                        else hi[k] = ia::round_f16(t, watch);
                    }
                    const int64_t slot = ((int64_t)(b * p.split_planes) * 4 + 2 * U + (q >> 1)) * p.R + pix;
                    *reinterpret_cast<h16x4r*>(base + slot * 16 + (q & 1) * 8) = hi;
                    if (p.split_planes == 2) *reinterpret_cast<h16x4r*>(base + (slot + (int64_t)4 * p.R) * 16 + (q & 1) * 8) = lo;
                }
                watch.report();
            }
            if (q == 0) {
                float dpt = acc_z / acc_w;
                if (dpt != dpt) dpt = INFINITY;             // nan_to_num(nan -> +inf), ray_marcher.py:49
                p.depth[ray] = dpt;
                p.wsum[ray] = acc_w;
            }
        }
        wave_sync();
        IA_RSTAMP(30);
    }
    if (p.dist_per_frame) {      // (kernel-uniform)
        if (cur_b >= 0 && lane == 0) { mm_wave[2 * cur_b] = blk_min; mm_wave[2 * cur_b + 1] = blk_max; }
        return;
    }
    // per-workgroup depth range for the batch-global clamp (ray_marcher.py:50)
    __syncthreads();
    float* red = lds + SCR_OFF;
    if (lane == 0) { red[2 * wave] = blk_min; red[2 * wave + 1] = blk_max; }
    __syncthreads();
    if (tid == 0) {
        float mn = red[0], mx = red[1];
        for (int w = 1; w < WAVES; ++w) { mn = fminf(mn, red[2 * w]); mx = fmaxf(mx, red[2 * w + 1]); }
        p.minmax[2 * blockIdx.x] = mn; p.minmax[2 * blockIdx.x + 1] = mx;
    }
}

// depth = clamp(depth, min over all sample depths, max over all sample depths)
__global__ __launch_bounds__(256) void render_finalize_kernel(float* depth, const float* minmax, int nblocks, int n) {
    __shared__ float s_mn[256], s_mx[256];
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < nblocks; i += 256) { mn = fminf(mn, minmax[2 * i]); mx = fmaxf(mx, minmax[2 * i + 1]); }
    s_mn[threadIdx.x] = mn; s_mx[threadIdx.x] = mx;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) { s_mn[threadIdx.x] = fminf(s_mn[threadIdx.x], s_mn[threadIdx.x + off]); s_mx[threadIdx.x] = fmaxf(s_mx[threadIdx.x], s_mx[threadIdx.x + off]); }
        __syncthreads();
    }
    mn = s_mn[0]; mx = s_mx[0];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) depth[i] = fminf(fmaxf(depth[i], mn), mx);
}

// ... per frame (IA_RENDER_DIST_PER_FRAME): blockIdx.y = frame, its range from the per-wave slots the render kernel filled
__global__ __launch_bounds__(256) void render_finalize_frames_kernel(float* depth, const float* minmax, int nslots, int R, int B) {
    __shared__ float s_mn[256], s_mx[256];
    const int b = blockIdx.y;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < nslots; i += 256) { mn = fminf(mn, minmax[((int64_t)i * B + b) * 2]); mx = fmaxf(mx, minmax[((int64_t)i * B + b) * 2 + 1]); }
    s_mn[threadIdx.x] = mn; s_mx[threadIdx.x] = mx;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) { s_mn[threadIdx.x] = fminf(s_mn[threadIdx.x], s_mn[threadIdx.x + off]); s_mx[threadIdx.x] = fmaxf(s_mx[threadIdx.x], s_mx[threadIdx.x + off]); }
        __syncthreads();
    }
    mn = s_mn[0]; mx = s_mx[0];
    float* d = depth + (int64_t)b * R;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < R; i += gridDim.x * 256) d[i] = fminf(fmaxf(d[i], mn), mx);
}

// Stage kernel for parity tests: importance resampling + merge order from GIVEN coarse depths/weights.
__global__ __launch_bounds__(64) void importance_stage_kernel(const float* z_coarse, const float* w_coarse, float* z_fine, int* inds,
                                                              int* order, int nrays) {
    __shared__ float scr[SCR];
    const int lane = threadIdx.x;
    for (int ray = blockIdx.x; ray < nrays; ray += gridDim.x) {
        if (lane < NS) scr[lane] = z_coarse[(int64_t)ray * NS + lane];
        if (lane < NS - 1) scr[2 * NS + lane] = w_coarse[(int64_t)ray * (NS - 1) + lane];
        wave_sync();
        const int ind = importance_resample(scr, lane);
        int pos_c, pos_f;
        merge_sorted(scr, lane, pos_c, pos_f);
        if (lane < NS) {
            z_fine[(int64_t)ray * NS + lane] = scr[7 * NS + lane];
            inds[(int64_t)ray * NS + lane] = ind;
            order[(int64_t)ray * NM + pos_c] = lane;
            order[(int64_t)ray * NM + pos_f] = NS + lane;
        }
        wave_sync();
    }
}

}  // namespace

extern "C" int ia_render_rays_grid(int B, int R) {
    const int64_t quads = ((int64_t)B * R + WAVES - 1) / WAVES;
    const int64_t cap = (int64_t)ia::kNumCU;            // one resident workgroup per CU (146 KB of LDS)
    return (int)(quads < cap ? quads : cap);
}

namespace {

int launch_render(const float* planes_cl, const float* rays_o, const float* rays_d, const float* jitter,
                  const float* u_importance, const float* dist, const float* limits, bool range_fixed, double range_start, double range_end,
                  const float* w0, const float* b0, const float* w1, const float* b1,
                  float lr_multiplier, float box_warp, int flags,
                  int B, int R, int plane_h, int plane_w, int n_coarse, int n_importance,
                  float* rgb, float* depth, float* wsum, float* minmax_scratch,
                  float* dbg_z_fine, int* dbg_inds, int* dbg_order, float* dbg_w_coarse, float* dbg_sigma_coarse,
                  void* rgb_split, const float* rgb_split_styles, int rgb_split_planes,
                  const char* what, void* stream) {
    IA_REQUIRE(!rgb_split || rgb_split_planes == 1 || rgb_split_planes == 2, "rgb_split_planes: 2 = hi / lo pair, 1 = one fp16 plane");
    IA_REQUIRE(planes_cl && rays_o && rays_d && jitter && w0 && b0 && w1 && b1, "null input pointer");
    IA_REQUIRE(rgb && depth && wsum && minmax_scratch, "null output pointer");
    IA_REQUIRE(B > 0 && R > 0 && plane_h > 0 && plane_w > 0, "empty tensor");
    if (n_coarse != NS || n_importance != NS)
        return ia::fail(IA_ERR_UNSUPPORTED, "depth_resolution=%d / depth_resolution_importance=%d: this build is specialised for 48/48",
                        n_coarse, n_importance);
    IA_REQUIRE(box_warp > 0.f, "box_warp must be positive");
    IA_REQUIRE(plane_h < (1 << 23) && plane_w < (1 << 23) && (int64_t)3 * plane_h * plane_w * 128 < ((int64_t)1 << 32),
               "planes of one batch element must stay below 4 GiB (32-bit texel offsets)");
    Params p;
    p.planes = planes_cl; p.rays_o = rays_o; p.rays_d = rays_d; p.jitter = jitter; p.u_imp = u_importance; p.dist = dist;
    p.limits = limits; p.range_fixed = range_fixed; p.range_start = range_start; p.range_end = range_end;
    p.flip_z = (flags & IA_RENDER_FLIP_Z) != 0;
    p.w0 = w0; p.b0 = b0; p.w1 = w1; p.b1 = b1;
    p.w0_gain = lr_multiplier / sqrtf(32.f); p.w1_gain = lr_multiplier / sqrtf(64.f); p.b_gain = lr_multiplier;
    p.box_scale = 2.f / box_warp;
    p.B = B; p.R = R; p.PH = plane_h; p.PW = plane_w;
    p.white_back = (flags & IA_RENDER_WHITE_BACK) != 0; p.channel_major = (flags & IA_RENDER_RGB_CHANNEL_MAJOR) != 0;
    p.dist_per_frame = (flags & IA_RENDER_DIST_PER_FRAME) != 0;
    p.rgb = rgb; p.depth = depth; p.wsum = wsum; p.minmax = minmax_scratch;
    p.split_out = rgb_split; p.split_styles = rgb_split_styles; p.split_planes = rgb_split_planes;
    p.dbg_z_fine = dbg_z_fine; p.dbg_inds = dbg_inds; p.dbg_order = dbg_order; p.dbg_w_coarse = dbg_w_coarse;
    p.dbg_sigma_coarse = dbg_sigma_coarse;
    const int grid = ia_render_rays_grid(B, R);
    hipStream_t s = (hipStream_t)stream;
    static_assert(LDS_FLOATS * sizeof(float) <= 160 * 1024, "one workgroup must fit a CU's LDS");
    const bool box = limits != nullptr || range_fixed;
    const auto kernel = box ? (plane_h == plane_w ? render_rays_kernel<true, true> : render_rays_kernel<false, true>)
                            : (plane_h == plane_w ? render_rays_kernel<true, false> : render_rays_kernel<false, false>);
    if (const int rs = ia::reserve_lds((const void*)kernel, (size_t)(LDS_FLOATS * sizeof(float)), "render_rays")) return rs;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(WAVES * 64), LDS_FLOATS * sizeof(float), s, p);
    int st = ia::check_launch(what);
    if (st != IA_OK) return st;
    if (p.dist_per_frame)
        hipLaunchKernelGGL(render_finalize_frames_kernel, dim3(ia::streaming_grid((int64_t)R, 256), B), dim3(256), 0, s, depth, minmax_scratch,
                           grid * WAVES, R, B);
    else
        hipLaunchKernelGGL(render_finalize_kernel, dim3(ia::streaming_grid((int64_t)B * R, 256)), dim3(256), 0, s, depth, minmax_scratch, grid, B * R);
    return ia::check_launch(what);
}

__device__ __forceinline__ float nan_max(float a, float b) { return (a != a || b != b) ? NAN : fmaxf(a, b); }
__device__ __forceinline__ float nan_min(float a, float b) { return (a != a || b != b) ? NAN : fminf(a, b); }

// get_ray_limits_box (math_utils.py:46-98): slab test of every ray against the cube of side `side` centred at the origin, the three
// slabs chained axis by axis as the reference chains them; a ray that misses gets (-1, -2).  Each workgroup also leaves the range of
// its hit rays' near limits in part[2 * blockIdx.x ..] for the second kernel.
__global__ __launch_bounds__(256) void ray_limits_box_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, float bb_min, float bb_max,
                                                             float* __restrict__ limits, float* __restrict__ part, int n) {
    __shared__ float s_mn[256], s_mx[256];
    float mn = INFINITY, mx = -INFINITY;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float near_ax[3], far_ax[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {      // the bound a ray enters through is picked by the sign of 1 / d (math_utils.py:64-69)
            const float o = rays_o[(int64_t)i * 3 + a], inv = 1.0f / rays_d[(int64_t)i * 3 + a];
            const bool neg = inv < 0.f;
            near_ax[a] = ((neg ? bb_max : bb_min) - o) * inv;
            far_ax[a] = ((neg ? bb_min : bb_max) - o) * inv;
        }
        float tmin = near_ax[0], tmax = far_ax[0];
        bool miss = false;
#pragma unroll
        for (int a = 1; a < 3; ++a) {
            miss = miss || (tmin > far_ax[a]) || (near_ax[a] > tmax);
            tmin = nan_max(tmin, near_ax[a]); tmax = nan_min(tmax, far_ax[a]);      // torch.max / torch.min propagate NaN (:78-79, :89-90)
        }
        if (miss) { tmin = -1.f; tmax = -2.f; }
        limits[(int64_t)i * 2] = tmin; limits[(int64_t)i * 2 + 1] = tmax;
        if (tmax > tmin) { mn = fminf(mn, tmin); mx = fmaxf(mx, tmin); }
    }
    s_mn[threadIdx.x] = mn; s_mx[threadIdx.x] = mx;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) { s_mn[threadIdx.x] = fminf(s_mn[threadIdx.x], s_mn[threadIdx.x + off]); s_mx[threadIdx.x] = fmaxf(s_mx[threadIdx.x], s_mx[threadIdx.x + off]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = s_mn[0]; part[2 * blockIdx.x + 1] = s_mx[0]; }
}

// ImportanceRenderer.forward's repair of the rays that miss the box (renderer.py:133-136): when any ray hits, a missing ray runs from
// the smallest to the LARGEST NEAR limit of the hit rays (`ray_end[~valid] = ray_start[valid].max()`, as the reference has it).
__global__ __launch_bounds__(256) void ray_limits_patch_kernel(float* __restrict__ limits, const float* __restrict__ part, int nparts, int n) {
    __shared__ float s_mn[256], s_mx[256];
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < nparts; i += 256) { mn = fminf(mn, part[2 * i]); mx = fmaxf(mx, part[2 * i + 1]); }
    s_mn[threadIdx.x] = mn; s_mx[threadIdx.x] = mx;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) { s_mn[threadIdx.x] = fminf(s_mn[threadIdx.x], s_mn[threadIdx.x + off]); s_mx[threadIdx.x] = fmaxf(s_mx[threadIdx.x], s_mx[threadIdx.x + off]); }
        __syncthreads();
    }
    mn = s_mn[0]; mx = s_mx[0];
    if (!(mn <= mx)) return;      // no ray hits the box: limits stay (-1, -2)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        if (!(limits[(int64_t)i * 2 + 1] > limits[(int64_t)i * 2])) { limits[(int64_t)i * 2] = mn; limits[(int64_t)i * 2 + 1] = mx; }
}

}  // namespace

extern "C" int ia_render_rays(const float* planes_cl, const float* rays_o, const float* rays_d, const float* jitter,
                              const float* u_importance, const float* dist, const float* w0, const float* b0, const float* w1, const float* b1,
                              float lr_multiplier, float box_warp, int flags,
                              int B, int R, int plane_h, int plane_w, int n_coarse, int n_importance,
                              float* rgb, float* depth, float* wsum, float* minmax_scratch,
                              float* dbg_z_fine, int* dbg_inds, int* dbg_order, float* dbg_w_coarse, float* dbg_sigma_coarse,
                              void* rgb_split, const float* rgb_split_styles, int rgb_split_planes, void* stream) {
    IA_REQUIRE(dist, "null input pointer");
    IA_REQUIRE(!(flags & IA_RENDER_FLIP_Z), "IA_RENDER_FLIP_Z belongs to ia_render_rays_box");
    return launch_render(planes_cl, rays_o, rays_d, jitter, u_importance, dist, nullptr, false, 0.0, 0.0, w0, b0, w1, b1, lr_multiplier, box_warp, flags,
                         B, R, plane_h, plane_w, n_coarse, n_importance, rgb, depth, wsum, minmax_scratch,
                         dbg_z_fine, dbg_inds, dbg_order, dbg_w_coarse, dbg_sigma_coarse, rgb_split, rgb_split_styles, rgb_split_planes, "ia_render_rays", stream);
}

extern "C" int ia_render_rays_box(const float* planes_cl, const float* rays_o, const float* rays_d, const float* jitter,
                                  const float* u_importance, const float* ray_limits, double ray_start, double ray_end,
                                  const float* w0, const float* b0, const float* w1, const float* b1,
                                  float lr_multiplier, float box_warp, int flags,
                                  int B, int R, int plane_h, int plane_w, int n_coarse, int n_importance,
                                  float* rgb, float* depth, float* wsum, float* minmax_scratch,
                                  float* dbg_z_fine, int* dbg_inds, int* dbg_order, float* dbg_w_coarse, float* dbg_sigma_coarse,
                                  void* stream) {
    IA_REQUIRE(!(flags & IA_RENDER_DIST_PER_FRAME), "IA_RENDER_DIST_PER_FRAME belongs to ia_render_rays");
    IA_REQUIRE(u_importance, "ImportanceRenderer draws its importance samples (renderer.py:280): u_importance must be given");
    return launch_render(planes_cl, rays_o, rays_d, jitter, u_importance, nullptr, ray_limits, ray_limits == nullptr, ray_start, ray_end,
                         w0, b0, w1, b1, lr_multiplier, box_warp, flags,
                         B, R, plane_h, plane_w, n_coarse, n_importance, rgb, depth, wsum, minmax_scratch,
                         dbg_z_fine, dbg_inds, dbg_order, dbg_w_coarse, dbg_sigma_coarse, nullptr, nullptr, 2, "ia_render_rays_box", stream);
}

extern "C" int ia_ray_limits_box_parts(int n_rays) {
    const int64_t g = ((int64_t)n_rays + 255) / 256;
    return (int)(g < 1024 ? (g < 1 ? 1 : g) : 1024);
}

extern "C" int ia_ray_limits_box(const float* rays_o, const float* rays_d, double box_side_length, int n_rays, int repair_misses,
                                 float* ray_limits, float* part_scratch, void* stream) {
    IA_REQUIRE(rays_o && rays_d && ray_limits && part_scratch, "null pointer");
    IA_REQUIRE(n_rays > 0, "empty tensor");
    const int parts = ia_ray_limits_box_parts(n_rays);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ray_limits_box_kernel, dim3(parts), dim3(256), 0, s, rays_o, rays_d,
                       (float)(-1 * (box_side_length / 2)), (float)(1 * (box_side_length / 2)), ray_limits, part_scratch, n_rays);
    int st = ia::check_launch("ia_ray_limits_box");
    if (st != IA_OK || !repair_misses) return st;
    hipLaunchKernelGGL(ray_limits_patch_kernel, dim3(ia::streaming_grid((int64_t)n_rays, 256)), dim3(256), 0, s, ray_limits, part_scratch, parts, n_rays);
    return ia::check_launch("ia_ray_limits_box(repair)");
}

extern "C" int ia_importance_stage(const float* z_coarse, const float* w_coarse, float* z_fine, int* inds, int* order,
                                   int nrays, void* stream) {
    IA_REQUIRE(z_coarse && w_coarse && z_fine && inds && order && nrays > 0, "bad argument");
    const int grid = nrays < 4096 ? nrays : 4096;
    hipLaunchKernelGGL(importance_stage_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, z_coarse, w_coarse, z_fine, inds, order, nrays);
    return ia::check_launch("ia_importance_stage");
}

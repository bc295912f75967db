// ia_conv2d_mfma: the dense convolutions of the StyleGAN2 stack as fp32 implicit GEMMs on
// v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s peak on gfx950).
//
// Replaces, for one layer, the reference's modulated_conv2d -> conv2d_resample -> cuDNN
// conv2d / conv_transpose2d chain (training/networks_stylegan2.py:34-91,
// torch_utils/ops/conv2d_resample.py:114-136, torch_utils/ops/conv2d_gradfix.py:37-45) and, for the
// stride-1 form, the bias_act that follows it (networks_stylegan2.py:327-329).
//
// Formulation (mathematically the reference's non-fused modconv branch, networks_stylegan2.py:70-79):
//     y[b,o] = act( d[b,o] * conv(x[b] * s[b,:], w)[o] + noise + bias[o] ) * gain   (+ residual)
// so ONE weight tensor serves the whole batch: the style s scales the input patch while it is staged
// into LDS, the demodulation d scales the accumulator in the epilogue.
//
// GEMM mapping: D[o, p] = sum_k A[o, k] * B[k, p];  A = weights (rows = out channels), B = input patch
// (cols = output points), k = (tap, in-channel).  With A = weights the MFMA result registers hold, per
// lane, one POINT and 16 channels, so stores along the flattened point index are 128-byte coalesced in
// NCHW.  Points are a contiguous range of the row-major output grid, so one kernel serves every
// resolution from 4x4 to 512x512.  Both operands are read from LDS as conflict-free ds_read_b32
// (32 consecutive channels / 32 consecutive points per half-wave).
//
//   stride-1 "conv":       out[y,x]          = sum w[o,i,ky,kx] * x[i, y+ky-pad, x+kx-pad]       (correlation)
//   stride-2 "transposed": out[2m+py,2n+px] += sum w[o,i,ky,kx] * x[i, m-ky/2, n-kx/2], ky%2==py, kx%2==px
//                          (= F.conv_transpose2d(x, w^T, stride 2)); the four output
//                          phases of a point (m,n) share one input patch and live in four accumulators.
//
// Low-resolution layers have too few output tiles to fill 256 CUs, so the in-channel range can be
// split over blockIdx.z (split-K); partial sums go to caller scratch and a second kernel reduces them in
// a fixed order (deterministic) and applies the epilogue.
#include "ia_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kPatchFloats = 1024;  // per-channel LDS patch capacity (floats)

struct Geo {
    int B, I, O, H, W;     // input
    int GH, GW;            // point grid: conv H x W, transposed (H+1) x (W+1)
    int OH, OW;            // output image
    int S, ci_per_split;   // split-K
    int patch_cap;         // floats per channel reserved for the patch in each LDS buffer
};

struct Epi {
    const float* demod;           // [B,O] or null
    const float* noise;           // [OH*OW] or null
    const float* noise_strength;  // device scalar (may be null => 1)
    const float* bias;            // [O] or null
    const float* residual;        // [B,O,OH,OW] or null, added after the clamp
    int act;                      // IA_ACT_LINEAR or IA_ACT_LRELU
    float alpha, gain, clamp;
};

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

__host__ __device__ inline Window tile_window(int p0, int p_last, int GW, int pad, bool tr) {
    const int up = tr ? 1 : pad, dn = tr ? 0 : pad, lf = tr ? 1 : pad, rt = tr ? 0 : pad;
    const int r_first = p0 / GW, r_last = p_last / GW;
    const int c_first = p0 - r_first * GW, c_last = p_last - r_last * GW;
    const int nrows = up + dn + 1;
    Window w;
    w.nr[1] = 0; w.r0[1] = 0; w.c0[1] = 0;
    if (r_first == r_last) {
        w.r0[0] = r_first - up; w.nr[0] = nrows; w.c0[0] = c_first - lf; w.PW = (c_last + rt) - w.c0[0] + 1;
    } else {
        const int w0 = (GW - 1 + rt) - (c_first - lf) + 1, w1 = (c_last + rt) - (0 - lf) + 1;
        const int pw_split = w0 > w1 ? w0 : w1, pw_full = GW + lf + rt;
        const int sz_split = 2 * nrows * pw_split, sz_full = (r_last - r_first + nrows) * pw_full;
        if (r_last == r_first + 1 && sz_split < sz_full) {
            w.r0[0] = r_first - up; w.nr[0] = nrows; w.c0[0] = c_first - lf;
            w.r0[1] = r_last - up;  w.nr[1] = nrows; w.c0[1] = -lf;
            w.PW = pw_split;
        } else {
            w.r0[0] = r_first - up; w.nr[0] = r_last - r_first + nrows; w.c0[0] = -lf; w.PW = pw_full;
        }
    }
    w.PSZ = (w.nr[0] + w.nr[1]) * w.PW;
    return w;
}

__device__ __forceinline__ float epilogue(float v, int b, int o, int64_t pix, int64_t ohw, const Geo& g, const Epi& e, float ns) {
    if (e.demod) v *= e.demod[b * g.O + o];
    if (e.noise) v = fmaf(e.noise[pix], ns, v);
    if (e.bias) v += e.bias[o];
    if (e.act == IA_ACT_LRELU) v = v > 0.f ? v : v * e.alpha;
    v *= e.gain;
    if (e.clamp >= 0.f) v = fminf(fmaxf(v, -e.clamp), e.clamp);
    if (e.residual) v += e.residual[((int64_t)b * g.O + o) * ohw + pix];
    return v;
}

// FO x FP fragments (32 channels x 32 points each) per wave, WO x WP waves per workgroup, CC in-channels per K chunk,
// NPOS patch positions staged per thread (>= ceil(worst PSZ / threads), chosen by the host).
template <int KS, bool TR, int FO, int FP, int WO, int WP, int CC, int NPOS, bool PARTIAL>
__global__ __launch_bounds__(WO * WP * 64, 2) void conv_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wk,
                                                                 const float* __restrict__ styles, float* __restrict__ y,
                                                                 Geo g, Epi e) {
    constexpr int NT = KS * KS;
    constexpr int NPH = TR ? 4 : 1;
    constexpr int BO = 32 * FO * WO, BP = 32 * FP * WP, NTHREADS = WO * WP * 64;
    constexpr int PAD = TR ? 0 : KS / 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wo = wave / WP, wp = wave % WP;
    const int o0 = blockIdx.y * BO;
    const int b = blockIdx.z / g.S, split = blockIdx.z % g.S;
    const int npts = g.GH * g.GW;
    const int p0 = blockIdx.x * BP;
    const int p_last = min(p0 + BP, npts) - 1;
    const Window win = tile_window(p0, p_last, g.GW, PAD, TR);
    const int PW = win.PW, PSZ = win.PSZ, seg1_off = win.nr[0] * PW;
    const int r_split = (win.nr[1] > 0) ? p_last / g.GW : (1 << 30);   // points in this grid row use segment 1
    const float inv_pw = 1.0f / (float)PW;

    // per-lane patch offset of each point fragment (points past the grid alias the last valid one)
    int base[FP];
#pragma unroll
    for (int fp = 0; fp < FP; ++fp) {
        const int p = min(p0 + (wp * FP + fp) * 32 + l31, p_last);
        const int r = p / g.GW, c = p - r * g.GW;
        const int sg = (r == r_split) ? 1 : 0;
        base[fp] = sg * seg1_off + (r - PAD - win.r0[sg]) * PW + (c - PAD - win.c0[sg]) + half * PSZ;
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

    const int ci_begin = split * g.ci_per_split, ci_end = min(ci_begin + g.ci_per_split, g.I);
    const float* xb = x + (int64_t)b * g.I * g.H * g.W;
    const float* sb = styles ? styles + (int64_t)b * g.I : nullptr;
    const int HW = g.H * g.W;

    // ---- staging plan, fixed for the whole K loop.
    // Patch: thread owns up to NPOS positions of the window; per position the global offset and a 0/1 mask are
    // computed once, then every chunk issues CC unconditional loads per position (no branches around loads).
    int goff[NPOS]; float gmask[NPOS];
#pragma unroll
    for (int j = 0; j < NPOS; ++j) {
        const int pp = tid + j * NTHREADS;
        const int sg = (pp >= seg1_off && win.nr[1] > 0) ? 1 : 0;
        const int qq = pp - sg * seg1_off;
        const int pr = (int)(((float)qq + 0.5f) * inv_pw), pc = qq - pr * PW;
        const int iy = win.r0[sg] + pr, ix = win.c0[sg] + pc;
        const bool ok = pp < PSZ && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
        goff[j] = ok ? iy * g.W + ix : 0;
        gmask[j] = ok ? 1.f : 0.f;
    }
    // Weights: NWV float4 per thread per chunk from the [tap][I][O] slab; rows = (tap, cc), o contiguous.
    constexpr int ROWV = BO / 4, NWV = (NT * CC * ROWV + NTHREADS - 1) / NTHREADS;
    const bool o_full = (g.O % 4) == 0 && (o0 + BO <= g.O);          // block-uniform fast path
    int w_row[NWV], w_o4[NWV];                                        // this thread's slots of the weight slab
#pragma unroll
    for (int k = 0; k < NWV; ++k) {
        const int e_ = min(tid + k * NTHREADS, NT * CC * ROWV - 1);
        w_row[k] = e_ / ROWV; w_o4[k] = (e_ - w_row[k] * ROWV) * 4;
    }
    float* w_lds = lds;                       // [NT*CC][BO]
    float* p_lds = lds + NT * CC * BO;        // [CC][PSZ]
    float pv[NPOS][CC];                       // staged patch values     (global -> registers -> LDS)
    float sv[CC];                             // style * channel-tail mask of the staged chunk
    float4 wv[NWV];                           // staged weight vectors

    // All loads of a chunk are unconditional (addresses clamped, masks applied at commit) so that they issue
    // back-to-back and one wait covers them; the chunk after next is in flight while the current one is multiplied.
    auto prefetch = [&](int ci0) {
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            const int ci = ci0 + cc;
            const float sty = sb ? sb[min(ci, g.I - 1)] : 1.f;
            sv[cc] = ci < ci_end ? sty : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NPOS; ++j)
#pragma unroll
            for (int cc = 0; cc < CC; ++cc) pv[j][cc] = xb[(int64_t)min(ci0 + cc, g.I - 1) * HW + goff[j]];
#pragma unroll
        for (int k = 0; k < NWV; ++k) {
            const int tap = w_row[k] / CC, cc = w_row[k] % CC;
            const float* src = wk + ((int64_t)tap * g.I + min(ci0 + cc, g.I - 1)) * g.O;
            const int o = o0 + w_o4[k];
            if (o_full) wv[k] = *(const float4*)(src + o);
            else {   // ragged out-channel edge: clamped scalar loads, masked
                const int last = g.O - 1;
                wv[k] = make_float4(o < g.O ? src[min(o, last)] : 0.f, o + 1 < g.O ? src[min(o + 1, last)] : 0.f,
                                    o + 2 < g.O ? src[min(o + 2, last)] : 0.f, o + 3 < g.O ? src[min(o + 3, last)] : 0.f);
            }
        }
    };
    auto commit = [&]() {   // registers -> LDS: style, zero padding and channel-tail masks folded into one multiply
#pragma unroll
        for (int j = 0; j < NPOS; ++j) {
            const int pp = tid + j * NTHREADS;
            if (pp < PSZ) {
#pragma unroll
                for (int cc = 0; cc < CC; ++cc) p_lds[cc * PSZ + pp] = pv[j][cc] * (sv[cc] * gmask[j]);
            }
        }
#pragma unroll
        for (int k = 0; k < NWV; ++k) {
            if (tid + k * NTHREADS < NT * CC * ROWV) {
                // (weights of channels past ci_end need no mask: their patch rows are zeroed through sv[])
                *(float4*)(w_lds + w_row[k] * BO + w_o4[k]) = wv[k];
            }
        }
    };

    constexpr int NSTEP_K = NT * (CC / 2);                    // k-pairs per chunk
    constexpr int KP = (FO * FP >= 4) ? 1 : (FO * FP == 2 ? 2 : 4);   // k-pairs per pipeline step: >= 4 MFMAs (256 cycles) per step
    constexpr int NSTEP = NSTEP_K / KP;
    static_assert(NSTEP_K % KP == 0, "chunk depth must be a multiple of the step depth");

    prefetch(ci_begin);
    for (int ci0 = ci_begin; ci0 < ci_end; ci0 += CC) {
        __syncthreads();                 // everyone is done reading the previous chunk
        commit();
        __syncthreads();
        if (ci0 + CC < ci_end) prefetch(ci0 + CC);   // global loads of the next chunk fly behind the MFMAs below
        // MFMA over the chunk: k-pair = channels (2cp, 2cp+1) of one tap; lane half picks the channel.  Operand reads
        // run one step ahead of the MFMAs that consume them (double-buffered registers) so the LDS latency hides
        // under the >= 256 MFMA cycles of a step.
        float a_buf[2][KP][FO], b_buf[2][KP][FP];
        auto load_ops = [&](int st, float (&a)[KP][FO], float (&bv)[KP][FP]) {
#pragma unroll
            for (int kk = 0; kk < KP; ++kk) {
                const int kp = st * KP + kk, t = kp / (CC / 2), cp = kp % (CC / 2);
#pragma unroll
                for (int fo = 0; fo < FO; ++fo) a[kk][fo] = w_lds[(t * CC + 2 * cp + half) * BO + (wo * FO + fo) * 32 + l31];
#pragma unroll
                for (int fp = 0; fp < FP; ++fp) bv[kk][fp] = p_lds[2 * cp * PSZ + base[fp] + toff[t]];
            }
        };
        load_ops(0, a_buf[0], b_buf[0]);
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            if (st + 1 < NSTEP) load_ops(st + 1, a_buf[(st + 1) & 1], b_buf[(st + 1) & 1]);
#pragma unroll
            for (int kk = 0; kk < KP; ++kk) {
                const int t = (st * KP + kk) / (CC / 2);
                const int ph = TR ? (((t / KS) & 1) * 2 + ((t % KS) & 1)) : 0;
#pragma unroll
                for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                    for (int fp = 0; fp < FP; ++fp)
                        acc[ph][fo][fp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_buf[st & 1][kk][fo], b_buf[st & 1][kk][fp],
                                                                               acc[ph][fo][fp], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, KP * (FO + FP), 0);   // next step's ds_reads first ...
            __builtin_amdgcn_sched_group_barrier(0x008, KP * FO * FP, 0);     // ... then this step's MFMAs
        }
    }

    // ---- epilogue / partial store.  C/D map: row(channel) = (r&3) + 8*(r>>2) + 4*half, col(point) = l31
    const int64_t ohw = (int64_t)g.OH * g.OW;
    const float ns = (!PARTIAL && e.noise) ? (e.noise_strength ? *e.noise_strength : 1.f) : 0.f;
    float* yb = PARTIAL ? y + ((int64_t)(split * g.B + b) * g.O) * ohw : y + ((int64_t)b * g.O) * ohw;
#pragma unroll
    for (int fp = 0; fp < FP; ++fp) {
        const int p = p0 + (wp * FP + fp) * 32 + l31;
        if (p >= npts) continue;
        const int pr = p / g.GW, pc = p - pr * g.GW;
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
                    float v = acc[ph][fo][fp][r];
                    if (!PARTIAL) v = epilogue(v, b, o, pix, ohw, g, e, ns);
                    yb[(int64_t)o * ohw + pix] = v;
                }
        }
    }
}

// Split-K second pass: y = epilogue(sum_s part[s]) in fixed order.
__global__ __launch_bounds__(256) void conv_reduce_kernel(const float* __restrict__ part, float* __restrict__ y, Geo g, Epi e) {
    const int64_t ohw = (int64_t)g.OH * g.OW, per_split = (int64_t)g.B * g.O * ohw;
    const float ns = e.noise ? (e.noise_strength ? *e.noise_strength : 1.f) : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_split; i += (int64_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        int s = 0;
        for (; s + 8 <= g.S; s += 8) {    // 8 independent loads in flight, summed in split order
            float t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = part[(s + k) * per_split + i];
#pragma unroll
            for (int k = 0; k < 8; ++k) v += t[k];
        }
        for (; s < g.S; ++s) v += part[s * per_split + i];
        const int64_t pix = i % ohw;
        const int bo = (int)(i / ohw);
        y[i] = epilogue(v, bo / g.O, bo % g.O, pix, ohw, g, e, ns);
    }
}

template <int KS, bool TR, int FO, int FP, int WO, int WP, int CC, int NPOS>
int launch_npos(const float* x, const float* wk, const float* styles, float* y, float* scratch, const Geo& g_in, const Epi& e,
                int worst, hipStream_t s) {
    constexpr int BO = 32 * FO * WO, BP = 32 * FP * WP, NT = KS * KS;
    Geo g = g_in;
    g.patch_cap = (worst + 3) & ~3;                                  // keeps the second buffer 16-byte aligned
    const int npts = g.GH * g.GW;
    dim3 grid((npts + BP - 1) / BP, (g.O + BO - 1) / BO, g.B * g.S), block(WO * WP * 64);
    const size_t lds = (size_t)(NT * CC * BO + CC * g.patch_cap) * sizeof(float);
    if (lds > 160 * 1024) return ia::fail(IA_ERR_UNSUPPORTED, "conv tile needs %zu bytes of LDS", lds);
    if (g.S > 1) {
        auto k = conv_mfma_kernel<KS, TR, FO, FP, WO, WP, CC, NPOS, true>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, grid, block, lds, s, x, wk, styles, scratch, g, e);
        const int64_t n = (int64_t)g.B * g.O * g.OH * g.OW;
        hipLaunchKernelGGL(conv_reduce_kernel, dim3(ia::streaming_grid(n, 256)), dim3(256), 0, s, scratch, y, g, e);
    } else {
        auto k = conv_mfma_kernel<KS, TR, FO, FP, WO, WP, CC, NPOS, false>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, grid, block, lds, s, x, wk, styles, y, g, e);
    }
    return ia::check_launch("ia_conv2d_mfma");
}

template <int KS, bool TR, int FO, int FP, int WO, int WP, int CC>
int launch(const float* x, const float* wk, const float* styles, float* y, float* scratch, const Geo& g, const Epi& e, hipStream_t s) {
    constexpr int BP = 32 * FP * WP, NTHREADS = WO * WP * 64;
    const int npts = g.GH * g.GW, ntiles = (npts + BP - 1) / BP;
    // largest per-channel patch any tile of this launch stages (same window function as the kernel)
    int worst = 0;
    for (int t = 0; t < ntiles; ++t) {
        const int q0 = t * BP, q1 = (q0 + BP < npts ? q0 + BP : npts) - 1;
        const Window w = tile_window(q0, q1, g.GW, TR ? 0 : KS / 2, TR);
        if (w.PSZ > worst) worst = w.PSZ;
    }
    if (worst > kPatchFloats) return ia::fail(IA_ERR_UNSUPPORTED, "conv tile patch of %d floats exceeds the LDS budget", worst);
    const int npos = (worst + NTHREADS - 1) / NTHREADS;
    if (npos <= 1) return launch_npos<KS, TR, FO, FP, WO, WP, CC, 1>(x, wk, styles, y, scratch, g, e, worst, s);
    if (npos <= 2) return launch_npos<KS, TR, FO, FP, WO, WP, CC, 2>(x, wk, styles, y, scratch, g, e, worst, s);
    return launch_npos<KS, TR, FO, FP, WO, WP, CC, 4>(x, wk, styles, y, scratch, g, e, worst, s);
}

constexpr int kChunkConv = 8, kChunkTransposed = 8;

// Tile family per layer shape: (32ch x 256pt) for narrow outputs, (128ch x 128pt) otherwise; the transposed form
// uses (64ch x 128pt x 4 phases) with 16-channel chunks so that it does as many MFMAs per staged chunk as the conv.
// Images with <= kSmallPoints output points (4x4 .. 16x16) use a (128ch x 32pt) tile: the MFMA work of a chunk is fixed
// by the tile, so a 128-point tile would spend 4 us per chunk on padding there.
constexpr int kSmallPoints = 320;
void tile_dims(int O, int npts, int transposed, int* bo, int* bp, int* cc) {
    if (npts <= kSmallPoints && O > 32) { *bo = 128; *bp = 32; *cc = kChunkConv; }
    else if (transposed) { *bo = 64; *bp = 128; *cc = kChunkTransposed; }
    else if (O <= 32) { *bo = 32; *bp = 256; *cc = kChunkConv; }
    else { *bo = 128; *bp = 128; *cc = kChunkConv; }
}

}  // namespace

extern "C" int ia_conv2d_plan(int B, int I, int O, int H, int W, int ksize, int transposed, int* h_ksplit,
                              size_t* h_scratch_bytes) {
    IA_REQUIRE(h_ksplit && h_scratch_bytes, "null output pointer");
    IA_REQUIRE(B > 0 && I > 0 && O > 0 && H > 0 && W > 0, "empty tensor");
    int bo, bp, cc;
    const int npts = transposed ? (H + 1) * (W + 1) : H * W;
    tile_dims(O, npts, transposed, &bo, &bp, &cc);
    const int64_t blocks = (int64_t)((npts + bp - 1) / bp) * ((O + bo - 1) / bo) * B;
    const int chunks = (I + cc - 1) / cc;
    int s = 1;
    // aim for >= 2 workgroups per CU, keep >= 2 chunks (16 channels x taps) of work per split
    const int min_chunks = (npts <= kSmallPoints) ? 1 : 2;   // chunks of K left per split
    while (blocks * s < 2 * ia::kNumCU && s * 2 * min_chunks <= chunks && s < 64) s *= 2;
    *h_ksplit = s;
    const int64_t oh = transposed ? 2 * H + 1 : H, ow = transposed ? 2 * W + 1 : W;
    *h_scratch_bytes = s > 1 ? (size_t)s * B * O * oh * ow * sizeof(float) : 0;
    return IA_OK;
}

extern "C" int ia_conv2d_mfma(const float* x, const float* wk, const float* styles, const float* demod,
                              const float* noise, const float* noise_strength, const float* bias, const float* residual,
                              float* y, float* scratch, size_t scratch_bytes,
                              int B, int I, int O, int H, int W, int ksize, int transposed,
                              int act, float alpha, float gain, float clamp, int ksplit, void* stream) {
    IA_REQUIRE(x && wk && y, "x, wk and y must be device pointers");
    IA_REQUIRE(B > 0 && I > 0 && O > 0 && H > 0 && W > 0, "empty tensor");
    IA_REQUIRE(ksize == 1 || ksize == 3, "kernel size must be 1 or 3");
    IA_REQUIRE(!transposed || ksize == 3, "the transposed form is 3x3 stride 2 only");
    IA_REQUIRE(act == IA_ACT_LINEAR || act == IA_ACT_LRELU, "conv epilogue supports linear and lrelu");
    IA_REQUIRE(ksplit >= 1, "ksplit must be >= 1");
    IA_REQUIRE(!transposed || (noise == nullptr && bias == nullptr && residual == nullptr && act == IA_ACT_LINEAR),
               "the transposed form only applies the demodulation; FIR + bias_act follow in ia_fir_bias_act");
    Geo g;
    g.B = B; g.I = I; g.O = O; g.H = H; g.W = W;
    g.GH = transposed ? H + 1 : H; g.GW = transposed ? W + 1 : W;
    g.OH = transposed ? 2 * H + 1 : H; g.OW = transposed ? 2 * W + 1 : W;
    IA_REQUIRE((int64_t)B * O * g.OH * g.OW <= INT32_MAX && (int64_t)B * I * H * W <= INT32_MAX, "tensor is too large");
    int bo_, bp_, cc;
    tile_dims(O, g.GH * g.GW, transposed, &bo_, &bp_, &cc);
    const int chunks = (I + cc - 1) / cc;
    if (ksplit > chunks) ksplit = chunks;
    g.S = ksplit;
    g.ci_per_split = ((chunks + ksplit - 1) / ksplit) * cc;
    if (ksplit > 1) {
        const size_t need = (size_t)ksplit * B * O * g.OH * g.OW * sizeof(float);
        IA_REQUIRE(scratch && scratch_bytes >= need, "split-K needs %zu bytes of scratch, got %zu", need, scratch_bytes);
    }
    Epi e{demod, noise, noise_strength, bias, residual, act, alpha, gain, clamp};
    hipStream_t s = (hipStream_t)stream;
    if (bp_ == 32) {   // small images: 4 waves side by side over 128 out-channels, one 32-point fragment each
        if (transposed) return launch<3, true, 1, 1, 4, 1, kChunkConv>(x, wk, styles, y, scratch, g, e, s);
        return ksize == 3 ? launch<3, false, 1, 1, 4, 1, kChunkConv>(x, wk, styles, y, scratch, g, e, s)
                          : launch<1, false, 1, 1, 4, 1, kChunkConv>(x, wk, styles, y, scratch, g, e, s);
    }
    if (transposed) return launch<3, true, 1, 2, 2, 2, kChunkTransposed>(x, wk, styles, y, scratch, g, e, s);
    if (O <= 32) {
        return ksize == 3 ? launch<3, false, 1, 2, 1, 4, kChunkConv>(x, wk, styles, y, scratch, g, e, s)
                          : launch<1, false, 1, 2, 1, 4, kChunkConv>(x, wk, styles, y, scratch, g, e, s);
    }
    return ksize == 3 ? launch<3, false, 2, 2, 2, 2, kChunkConv>(x, wk, styles, y, scratch, g, e, s)
                      : launch<1, false, 2, 2, 2, 2, kChunkConv>(x, wk, styles, y, scratch, g, e, s);
}

// d[b,o] = rsqrt(sum_i s[b,i]^2 * wsq[o,i] + 1e-8): demodulation coefficients of the modulated conv
// (training/networks_stylegan2.py:63-64), with wsq[o,i] = sum_taps w[o,i,ky,kx]^2 precomputed per layer.
__global__ __launch_bounds__(256) void demod_kernel(const float* __restrict__ styles, const float* __restrict__ wsq,
                                                    float* __restrict__ d, int B, int I, int O) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= B * O) return;
    const int b = wave / O, o = wave - b * O;
    float acc = 0.f;
    for (int i = lane; i < I; i += 64) { const float s = styles[b * I + i]; acc = fmaf(s * s, wsq[(int64_t)o * I + i], acc); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) d[wave] = 1.0f / sqrtf(acc + 1e-8f);
}

extern "C" int ia_modconv_demod(const float* styles, const float* wsq, float* demod, int B, int I, int O, void* stream) {
    IA_REQUIRE(styles && wsq && demod, "null pointer argument");
    IA_REQUIRE(B > 0 && I > 0 && O > 0, "empty tensor");
    const int waves = B * O;
    hipLaunchKernelGGL(demod_kernel, dim3((waves * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, styles, wsq, demod, B, I, O);
    return ia::check_launch("ia_modconv_demod");
}

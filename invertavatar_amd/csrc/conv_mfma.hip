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
// Scheduling.  Tiles that fill whole rounds of the machine (2 workgroups per CU) run one tile per workgroup.  The tiles
// left over -- all of them for layers with fewer tiles than slots -- are stream-K: their (tile, K-chunk) units, in
// tile-major order, are cut into G equal contiguous ranges, one per workgroup ("worker").  A worker that owns a whole
// tile applies the epilogue and stores it; a worker that owns only part of a tile's K range parks its accumulators in a
// caller-owned slab, and a fix-up kernel adds the slabs of such tiles in worker order (deterministic) and applies the
// epilogue.  The hand-off is the kernel boundary: slabs written by workgroups on one XCD are read by workgroups on
// another, and an in-kernel hand-off (write-through slabs + flags + one serial finisher per tile) measured slower than
// this parallel fix-up for every layer of the model.  This balances layers whose tile count is not a multiple of the
// machine (e.g. the (H+1)x(W+1) point grids of the transposed form: 524 equal workgroups on 512 slots ran as two rounds)
// and gives low-resolution layers, which have only a handful of tiles, a fine-grained K split.
#include "conv_common.h"
#include <cstdio>
#include <cstdlib>

namespace {

// FO x FP fragments (32 channels x 32 points each) per wave, WO x WP waves per workgroup, CC in-channels per K chunk,
// NPOS patch positions staged per thread (>= ceil(worst PSZ / threads), chosen by the host).
// SK = false: workgroup = one whole tile (tiles [0, T_dp), whole rounds of the machine);
// SK = true: stream-K ranges over tiles [T_dp, T) with slab hand-off.
// HM = 2 (DB kernels only): fp32-equivalent products from fp16 pairs.  Every operand value v is split as hi = fp16(v),
// lo = fp16(v - hi) (22 mantissa bits together) and a*b is taken as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on
// v_mfma_f32_32x32x16_f16 with fp32 accumulation: the dropped term a_lo*b_lo is 2^-22 of the product, the size of fp32's own
// rounding.  The MFMA flushes fp16 denormals, so the factors are kept normal: weights arrive scaled by 2^wk_exp (hi and lo of
// w * 2^wk_exp, pack_conv_weight_split), the low part of an activation is scaled by 2^11 at commit, and the weight's high part
// is multiplied by 2^-11 (exact, one v_pk_mul_f16 per register pair) where it meets it.  All three products then carry 2^wk_exp
// and accumulate in ONE register set, scaled back by g.acc_scale = 2^-wk_exp before the epilogue / the slab hand-off.
// `wk` = [2 (hi, lo)][tap][I/8][O][8] halves.
// HM = 1 (DB kernels only): fp16 operands on v_mfma_f32_32x32x16_f16, fp32 accumulation -- the arithmetic of the reference's
// fp16 blocks (training/networks_stylegan2.py:34-91 with x.dtype == float16; superresolution.py:209-216), activations kept
// in fp32 in memory.  `wk` then points at the fp16 weights packed [tap][I/4][O][4] (pack_conv_weight_h).
template <int KS, bool TR, int FO, int FP, int WO, int WP, int CC, int NPOS, bool SK, bool DB, int HM>
__global__ __launch_bounds__(WO * WP * 64, WO * WP == 8 ? 1 : 2) void conv_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wk,
                                                                 const float* __restrict__ styles, float* __restrict__ y,
                                                                 float* __restrict__ slabs, Geo g, Epi e) {
    constexpr int NT = KS * KS;
    constexpr int NPH = TR ? 4 : 1;
    constexpr int BO = 32 * FO * WO, BP = 32 * FP * WP, NTHREADS = WO * WP * 64;
    constexpr int PAD = TR ? 0 : KS / 2;
    constexpr int NACC = NPH * FO * FP * 16;           // accumulator registers per thread = floats per thread in a slab
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wo = wave / WP, wp = wave % WP;
    const int b = blockIdx.y, worker = blockIdx.x;
    const int npts = g.GH * g.GW;
    const int64_t U = (int64_t)(g.T - g.T_dp) * g.C;
    const int64_t u_begin = SK ? range_begin(worker, U, g.G) : (int64_t)worker * g.C;
    const int64_t u_end = SK ? range_begin(worker + 1, U, g.G) : u_begin + g.C;
    const int first_tile = (int)(u_begin / g.C);
    const int tile_base = SK ? g.T_dp : 0;

  for (int64_t u = u_begin; u < u_end;) {
    // ---- one segment: tile `tile`, K chunks [c_lo, c_hi)
    const int tile_l = (int)(u / g.C), c_lo = (int)(u - (int64_t)tile_l * g.C);
    const int c_hi = (int)min((int64_t)g.C, (int64_t)c_lo + (u_end - u));
    u += c_hi - c_lo;
    const int tile = tile_base + tile_l;
    const int o0 = (tile % g.TO) * BO;
    const int p0 = (tile / g.TO) * BP;
    const int p_last = min(p0 + BP, npts) - 1;
    const Window win = tile_window(p0, p_last, g.GW, PAD, TR);
    const int PW = win.PW, PSZ = win.PSZ, seg1_off = win.nr[0] * PW;
    const int r_split = (win.nr[1] > 0) ? p_last / g.GW : (1 << 30);   // points in this grid row use segment 1
    const float inv_pw = 1.0f / (float)PW;

    // per-lane patch offset of each point fragment (points past the grid alias the last valid one)
    int base[FP], bpos[FP];
#pragma unroll
    for (int fp = 0; fp < FP; ++fp) {
        const int p = min(p0 + (wp * FP + fp) * 32 + l31, p_last);
        const int r = p / g.GW, c = p - r * g.GW;
        const int sg = (r == r_split) ? 1 : 0;
        base[fp] = sg * seg1_off + (r - PAD - win.r0[sg]) * PW + (c - PAD - win.c0[sg]) + half * PSZ;
        bpos[fp] = base[fp] - half * PSZ;       // (fp16 forms: the lane half picks the tap of a pair, not a channel)
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

    const int ci_begin = c_lo * CC, ci_end = min(c_hi * CC, g.I);
    const float* xb = x + (int64_t)b * g.I * g.H * g.W;
    const float* sb = styles ? styles + (int64_t)b * g.I : nullptr;
    const int HW = g.H * g.W;

    // ---- staging plan, fixed for the whole K loop.
    // Patch: thread owns up to NPOS positions of the window; per position the byte offset inside this batch element's
    // channel plane is computed once (positions outside the image get an offset past the buffer, which the descriptor's
    // bounds check turns into 0.0: zero padding without a mask); every chunk then issues CC unconditional loads per
    // position with the channel offset in an SGPR.
    constexpr int kOutside = 0x7ffffff0;
    int goff[NPOS];
#pragma unroll
    for (int j = 0; j < NPOS; ++j) {
        const int pp = tid + j * NTHREADS;
        const int sg = (pp >= seg1_off && win.nr[1] > 0) ? 1 : 0;
        const int qq = pp - sg * seg1_off;
        const int pr = (int)(((float)qq + 0.5f) * inv_pw), pc = qq - pr * PW;
        const int iy = win.r0[sg] + pr, ix = win.c0[sg] + pc;
        const bool ok = pp < PSZ && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
        goff[j] = ok ? (iy * g.W + ix) * 4 : kOutside;
    }
    // Weights: NWV float4 per thread per chunk from the [tap][I][O] slab; slot e = tid + k*NTHREADS covers row
    // (tap, cc) = e / ROWV and channels 4*(e % ROWV) ..+3 of the tile, and lands at float 4*e of the LDS slab.
    // (fp16 forms: the slab of a chunk is [plane][tap][BO][8 channels] halves; 16-byte slot e = (plane*NT + tap)*BO + o.  In
    // LDS every plane has one more, all-zero, tap: slot e lands at 16-byte index e + plane*BO.)
    constexpr int NPA = HM == 2 ? 2 : 1, NPB = NPA;                   // operand planes: weights (hi, lo) at the packed scale, patch (hi, lo*2^11)
    constexpr int NTP = NT + 1;
    constexpr int ROWV = HM ? BO : BO / 4, NSLOT = HM ? NPA * NT * BO : NT * CC * ROWV, NWV = (NSLOT + NTHREADS - 1) / NTHREADS;
    static_assert(!HM || (DB && CC == 8 && NT == 9), "the fp16 MFMA forms are built for the two-stage 3x3 kernels with 8-channel chunks");
    // o_vec: every weight row is 16-byte aligned and at least one float4 long, so the slab is fetched with
    // unconditional, clamped float4 buffer loads (rows past O feed accumulator rows that are never stored; rows past
    // the end of the tensor -- channel tail of the last tap -- read as zero through the bounds check).
    const bool o_vec = (g.O % 4) == 0;
    int w_off[NWV];                                                   // byte offset of the slot at channel 0 of a chunk
#pragma unroll
    for (int k = 0; k < NWV; ++k) {
        const int e_ = min(tid + k * NTHREADS, NSLOT - 1), row = e_ / ROWV;
        if constexpr (HM != 0)     // bytes: (((plane*NT + tap)*(I/8) + octet)*O + o)*16, o clamped inside the tensor row
            w_off[k] = ((row * (g.I / 8)) * g.O + min(o0 + (e_ - row * ROWV), g.O - 1)) * 16;
        else
            w_off[k] = (((row / CC) * g.I + (row % CC)) * g.O + min(o0 + (e_ - row * ROWV) * 4, max(g.O - 4, 0))) * 4;
    }
    const int stage_floats = HM ? NPA * NTP * BO * 4 + NPB * 4 * g.patch_cap : NT * CC * BO + CC * g.patch_cap;
    float* w_lds = lds;                       // [NT*CC][BO]
    float* p_lds = lds + NT * CC * BO;        // [CC][PSZ]   (second stage, if any, stage_floats further on)
    float pv[NPOS][CC];                       // staged patch values     (global -> registers -> LDS)
    float sv[CC];                             // style * channel-tail mask of the staged chunk
    float4 wv[NWV];                           // staged weight vectors
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (int)((int64_t)g.I * HW * 4), 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wk), 0, (int)((int64_t)NT * g.I * g.O * (HM ? 2 * NPA : 4)), 0x00020000);   // bytes of all planes

    // All loads of a chunk are unconditional so that they can be issued anywhere; the chunk after next is in flight
    // while the current one is multiplied.  No 64-bit address arithmetic in the K loop: per-thread byte offsets fixed
    // for the tile (VGPR) + a per-chunk channel offset (SGPR for the patch, one v_add for the weights).
    auto fetch_styles = [&](int ci0) {
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            const int ci = ci0 + cc;
            const float sty = sb ? sb[min(ci, g.I - 1)] : 1.f;
            sv[cc] = ci < ci_end ? sty : 0.f;
        }
    };
    constexpr int NLOAD = NPOS * CC + NWV;                            // loads per thread per chunk
    auto fetch_one = [&](int idx, int ci0) {                          // idx is a compile-time constant after unrolling
        if (idx < NPOS * CC) {
            const int j = idx / CC, cc = idx % CC;
            pv[j][cc] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, goff[j], min(ci0 + cc, g.I - 1) * HW * 4, 0));
        } else {
            const int k = idx - NPOS * CC;
            // (whole-vector bit_cast: element-wise bit_casts of an ext_vector were seen to be folded to element 0)
            wv[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_off[k] + ci0 * g.O * (HM ? 2 : 4), 0, 0));
        }
    };
    auto prefetch = [&](int ci0) {
        fetch_styles(ci0);
        if (DB || o_vec) {
#pragma unroll
            for (int idx = 0; idx < NLOAD; ++idx) fetch_one(idx, ci0);
            return;
        }
#pragma unroll
        for (int idx = 0; idx < NPOS * CC; ++idx) fetch_one(idx, ci0);
#pragma unroll
        for (int k = 0; k < NWV; ++k) {   // rows that are not 16-byte aligned: clamped scalar loads, masked
            const int e_ = min(tid + k * NTHREADS, NSLOT - 1), row = e_ / ROWV;
            const float* src = wk + ((int64_t)(row / CC) * g.I + min(ci0 + row % CC, g.I - 1)) * g.O;
            const int o = o0 + (e_ - row * ROWV) * 4, last = g.O - 1;
            wv[k] = make_float4(o < g.O ? src[min(o, last)] : 0.f, o + 1 < g.O ? src[min(o + 1, last)] : 0.f,
                                o + 2 < g.O ? src[min(o + 2, last)] : 0.f, o + 3 < g.O ? src[min(o + 3, last)] : 0.f);
        }
    };
    auto commit = [&](int st_off) {   // registers -> LDS: style and channel-tail masks folded into one multiply
#pragma unroll
        for (int j = 0; j < NPOS; ++j) {
            const int pp = tid + j * NTHREADS;
            if (pp < PSZ) {
                if constexpr (HM != 0) {   // patch as [plane][PSZ][8 channels] halves: one 16-byte store per plane
                    h16x8* ph = reinterpret_cast<h16x8*>(lds + st_off + NPA * NTP * BO * 4);
                    h16x8 hi, lo;
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc) {
                        const float v = fminf(fmaxf(pv[j][cc] * sv[cc], -65504.f), 65504.f);   // fp16 range: saturate, never inf
                        // (a denormal high part would be flushed by the MFMA: below 2^-14 the value rides in the scaled low part)
                        hi[cc] = (HM == 2 && fabsf(v) < 6.103515625e-5f) ? (_Float16)0.f : (_Float16)v;
                        lo[cc] = (_Float16)((v - (float)hi[cc]) * kLoScale);
                    }
                    ph[pp] = hi;
                    if constexpr (HM == 2) ph[PSZ + pp] = lo;
                } else {
#pragma unroll
                    for (int cc = 0; cc < CC; ++cc) p_lds[st_off + cc * PSZ + pp] = pv[j][cc] * sv[cc];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NWV; ++k) {
            if (tid + k * NTHREADS < NSLOT) {
                // (weights of channels past ci_end need no mask: their patch rows are zeroed through sv[])
                const int e_ = tid + k * NTHREADS;
                *(float4*)(w_lds + st_off + (HM ? e_ + (e_ / (NT * BO)) * BO : e_) * 4) = wv[k];
            }
        }
    };

    // fp32: a k-step is a channel pair of one tap (32x32x2); fp16: the 8 channels of a pair of taps (32x32x16)
    constexpr int NSTEP_K = HM ? kPairs : NT * (CC / 2);      // k-steps per chunk
    constexpr int KP = HM ? 1 : ((FO * FP >= 4) ? 1 : (FO * FP == 2 ? 2 : 4));   // k-steps per pipeline step (fp32: >= 4 MFMAs)
    constexpr int NSTEP = NSTEP_K / KP;
    static_assert(NSTEP_K % KP == 0, "chunk depth must be a multiple of the step depth");

    prefetch(ci_begin);
    if constexpr (DB) {
        __syncthreads();                 // (previous segment's readers are done)
        if constexpr (HM != 0) {         // the all-zero tap of every plane, in both stages (never overwritten by commit)
            for (int i = tid; i < 2 * NPA * BO; i += NTHREADS) {
                const int stg = i / (NPA * BO), r = i - stg * NPA * BO, pl = r / BO, o = r - pl * BO;
                *(float4*)(lds + stg * stage_floats + ((pl * NTP + NT) * BO + o) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        commit(0);
        __syncthreads();
        if (ci_begin + CC < ci_end) prefetch(ci_begin + CC);
    }
    int st_cur = 0;                      // LDS stage the MFMAs of this chunk read (float offset)
    // DB kernels spread the global loads of the chunk after next over the first half of the MFMA steps (LPS per step)
    // instead of issuing them in one burst: ~25 loads per thread from every workgroup at once back up the request path
    // and hold the waves at VMEM issue while the MFMA pipe drains.
    constexpr int LPS = (2 * NLOAD + NSTEP - 1) / NSTEP;
    for (int ci0 = ci_begin; ci0 < ci_end; ci0 += CC) {
        if constexpr (DB) {
            // the next chunk (in registers, loaded during the previous iteration) goes to the other stage, whose readers
            // passed the barrier at the end of the previous iteration
            if (ci0 + CC < ci_end) commit(stage_floats - st_cur);
            fetch_styles(ci0 + 2 * CC);                        // (loads past the last chunk are clamped and unused)
        } else {
            __syncthreads();                 // everyone is done reading the previous chunk
            commit(0);
            __syncthreads();
            if (ci0 + CC < ci_end) prefetch(ci0 + CC);         // global loads of the next chunk fly behind the MFMAs below
        }
        // MFMA over the chunk: k-pair = channels (2cp, 2cp+1) of one tap; lane half picks the channel.  Operand reads
        // run one step ahead of the MFMAs that consume them (double-buffered registers) so the LDS latency hides
        // under the >= 256 MFMA cycles of a step.
        using op_t = std::conditional_t<HM != 0, h16x8, float>;
        constexpr int OB = 2;
        op_t a_buf[OB][KP][FO * NPA], b_buf[OB][KP][FP * NPB];
        auto load_ops = [&](int st, op_t (&a)[KP][FO * NPA], op_t (&bv)[KP][FP * NPB]) {
#pragma unroll
            for (int kk = 0; kk < KP; ++kk) {
                if constexpr (HM != 0) {
                    const int sidx = st * KP + kk;
                    const int tap = half ? pair_t1(TR, sidx) : pair_t0(TR, sidx);        // this lane half's tap of the pair
                    const int tof = pair_t1(TR, sidx) == kZeroTap ? (half ? 0 : toff[pair_t0(TR, sidx)])
                                                                  : (half ? toff[pair_t1(TR, sidx)] : toff[pair_t0(TR, sidx)]);
                    const h16x8* wh = reinterpret_cast<const h16x8*>(lds + st_cur);
                    const h16x8* ph = reinterpret_cast<const h16x8*>(lds + st_cur + NPA * NTP * BO * 4);
#pragma unroll
                    for (int pl = 0; pl < NPA; ++pl)
#pragma unroll
                        for (int fo = 0; fo < FO; ++fo) a[kk][pl * FO + fo] = wh[(pl * NTP + tap) * BO + (wo * FO + fo) * 32 + l31];
#pragma unroll
                    for (int pl = 0; pl < NPB; ++pl)
#pragma unroll
                        for (int fp = 0; fp < FP; ++fp) bv[kk][pl * FP + fp] = ph[pl * PSZ + bpos[fp] + tof];
                } else {
                    const int kp = st * KP + kk, t = kp / (CC / 2), cp = kp % (CC / 2);
#pragma unroll
                    for (int fo = 0; fo < FO; ++fo) a[kk][fo] = w_lds[st_cur + (t * CC + 2 * cp + half) * BO + (wo * FO + fo) * 32 + l31];
#pragma unroll
                    for (int fp = 0; fp < FP; ++fp) bv[kk][fp] = p_lds[st_cur + 2 * cp * PSZ + base[fp] + toff[t]];
                }
            }
        };
        load_ops(0, a_buf[0], b_buf[0]);
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            const int cur = OB == 2 ? (st & 1) : 0;
            if (OB == 2 && st + 1 < NSTEP) load_ops(st + 1, a_buf[(st + 1) & 1], b_buf[(st + 1) & 1]);
            if constexpr (DB) {
#pragma unroll
                for (int l = 0; l < LPS; ++l)
                    if (st * LPS + l < NLOAD) fetch_one(st * LPS + l, ci0 + 2 * CC);
            }
#pragma unroll
            for (int kk = 0; kk < KP; ++kk) {
                const int t = HM ? st * KP + kk : (st * KP + kk) / (CC / 2);   // fp16 forms: pair index
                const int ph = HM ? pair_phase(TR, t) : (TR ? (((t / KS) & 1) * 2 + ((t % KS) & 1)) : 0);
#pragma unroll
                for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                    for (int fp = 0; fp < FP; ++fp) {
                        if constexpr (HM == 2) {   // lo*hi, (hi*2^-11)*(lo*2^11), hi*hi: all at the scale of the packed weights
                            acc[ph][fo][fp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_buf[cur][kk][FO + fo], b_buf[cur][kk][fp],
                                                                                     acc[ph][fo][fp], 0, 0, 0);
                            acc[ph][fo][fp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_buf[cur][kk][fo] * (_Float16)(1.0f / kLoScale), b_buf[cur][kk][FP + fp],
                                                                                     acc[ph][fo][fp], 0, 0, 0);
                            acc[ph][fo][fp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_buf[cur][kk][fo], b_buf[cur][kk][fp],
                                                                                     acc[ph][fo][fp], 0, 0, 0);
                        } else if constexpr (HM == 1)
                            acc[ph][fo][fp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_buf[cur][kk][fo], b_buf[cur][kk][fp],
                                                                                     acc[ph][fo][fp], 0, 0, 0);
                        else
                            acc[ph][fo][fp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_buf[cur][kk][fo], b_buf[cur][kk][fp],
                                                                                   acc[ph][fo][fp], 0, 0, 0);
                    }
            }
            if (OB == 1 && st + 1 < NSTEP) load_ops(st + 1, a_buf[0], b_buf[0]);   // after the MFMAs that read these registers were issued
            if constexpr (OB == 2) {
                __builtin_amdgcn_sched_group_barrier(0x100, KP * (FO * NPA + FP * NPB), 0);   // next step's ds_reads first ...
                if (DB && st * LPS < NLOAD) __builtin_amdgcn_sched_group_barrier(0x020, LPS, 0);   // ... a few global loads ...
                __builtin_amdgcn_sched_group_barrier(0x008, KP * FO * FP * (HM == 2 ? 3 : 1), 0);     // ... then this step's MFMAs
            }
        }
        if constexpr (DB) {
            __syncthreads();             // publishes the stage committed above and retires the one just read
            st_cur = stage_floats - st_cur;
        }
    }

    if constexpr (HM == 2) {   // back from the scale of the packed weights (a power of two: exact)
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
            for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                for (int fp = 0; fp < FP; ++fp)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ph][fo][fp][r] *= g.acc_scale;
    }
    // ---- segment done: a whole tile is finished here, a partial K range is parked for the fix-up kernel
    if (!SK || (c_lo == 0 && c_hi == g.C)) {
        store_tile<TR, FO, FP, WO, WP>(acc, y, g, e, b, o0, p0, tid);
    } else if constexpr (SK) {
        const int slot = (tile_l == first_tile) ? 0 : 1;   // a worker has at most a leading and a trailing partial tile
        // slab layout: [NACC/4][NTHREADS] float4 (register quad q of thread t at (q*NTHREADS + t)*16 bytes)
        float4* slab = reinterpret_cast<float4*>(slabs + (((int64_t)b * g.G + worker) * 2 + slot) * ((int64_t)NACC * NTHREADS)) + tid;
#pragma unroll
        for (int q = 0; q < NACC / 4; ++q) {
            const int fr = q >> 2, r0 = (q & 3) * 4;
            const f32x16& a = acc[fr / (FP * FO)][(fr / FP) % FO][fr % FP];
            slab[(int64_t)q * NTHREADS] = make_float4(a[r0], a[r0 + 1], a[r0 + 2], a[r0 + 3]);
        }
    }
  }   // segments of this worker
}

template <int KS, bool TR, int FO, int FP, int WO, int WP, int CC, int NPOS, bool DB, int HM = 0>
int launch_npos(const float* x, const float* wk, const float* styles, float* y, float* scratch, const Geo& g_in, const Epi& e,
                int worst, hipStream_t s) {
    constexpr int BO = 32 * FO * WO, NT = KS * KS;
    Geo g = g_in;
    g.patch_cap = (worst + 3) & ~3;
    constexpr int NPA = HM == 2 ? 2 : 1, NPB = NPA;
    const size_t stage = HM ? (size_t)NPA * (NT + 1) * BO * 4 + (size_t)NPB * 4 * g.patch_cap : (size_t)NT * CC * BO + (size_t)CC * g.patch_cap;
    const size_t lds = stage * sizeof(float) * (DB ? 2 : 1);
    if (lds > 160 * 1024) return ia::fail(IA_ERR_UNSUPPORTED, "conv tile needs %zu bytes of LDS", lds);
    int st = IA_OK;
    if (g.T_dp > 0) {   // whole rounds: one tile per workgroup
        auto k = conv_mfma_kernel<KS, TR, FO, FP, WO, WP, CC, NPOS, false, DB, HM>;
        if (const int rs = ia::reserve_lds((const void*)k, (size_t)(lds), "conv_mfma")) return rs;
        hipLaunchKernelGGL(k, dim3(g.T_dp, g.B), dim3(WO * WP * 64), lds, s, x, wk, styles, y, scratch, g, e);
        st = ia::check_launch("ia_conv2d_mfma");
    }
    if (st == IA_OK && g.T > g.T_dp) {   // the rest: stream-K, then the fix-up of the tiles that were shared
        auto k = conv_mfma_kernel<KS, TR, FO, FP, WO, WP, CC, NPOS, true, DB, HM>;
        if (const int rs = ia::reserve_lds((const void*)k, (size_t)(lds), "conv_mfma")) return rs;
        hipLaunchKernelGGL(k, dim3(g.G, g.B), dim3(WO * WP * 64), lds, s, x, wk, styles, y, scratch, g, e);
        st = ia::check_launch("ia_conv2d_mfma(stream-K)");
        const int64_t U = (int64_t)(g.T - g.T_dp) * g.C;
        const bool whole_tiles = U % g.G == 0 && (U / g.G) % g.C == 0;
        if (st == IA_OK && !whole_tiles) {
            constexpr int NACC = (TR ? 4 : 1) * FO * FP * 16;
            hipLaunchKernelGGL((conv_fixup_kernel<TR, FO, FP, WO, WP>), dim3(g.T - g.T_dp, g.B, NACC / (TR ? 8 : 4)), dim3(WO * WP * 64), 0, s,
                               scratch, y, g, e);
            st = ia::check_launch("ia_conv2d_mfma(fix-up)");
        }
    }
    return st;
}

template <int KS, bool TR, int FO, int FP, int WO, int WP, int CC, int HM = 0>
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
    if constexpr (HM != 0) {
        if (npos <= 1) return launch_npos<KS, TR, FO, FP, WO, WP, CC, 1, true, HM>(x, wk, styles, y, scratch, g, e, worst, s);
        if (npos <= 2) return launch_npos<KS, TR, FO, FP, WO, WP, CC, 2, true, HM>(x, wk, styles, y, scratch, g, e, worst, s);
        return launch_npos<KS, TR, FO, FP, WO, WP, CC, 4, true, HM>(x, wk, styles, y, scratch, g, e, worst, s);
    } else {
        if constexpr (db_family(TR, FO, FP, WO, WP)) {
            if (g.O % 4 == 0) {
                if (npos <= 1) return launch_npos<KS, TR, FO, FP, WO, WP, CC, 1, true>(x, wk, styles, y, scratch, g, e, worst, s);
                if (npos <= 2) return launch_npos<KS, TR, FO, FP, WO, WP, CC, 2, true>(x, wk, styles, y, scratch, g, e, worst, s);
                return launch_npos<KS, TR, FO, FP, WO, WP, CC, 4, true>(x, wk, styles, y, scratch, g, e, worst, s);
            }
        }
        if (npos <= 1) return launch_npos<KS, TR, FO, FP, WO, WP, CC, 1, false>(x, wk, styles, y, scratch, g, e, worst, s);
        if (npos <= 2) return launch_npos<KS, TR, FO, FP, WO, WP, CC, 2, false>(x, wk, styles, y, scratch, g, e, worst, s);
        return launch_npos<KS, TR, FO, FP, WO, WP, CC, 4, false>(x, wk, styles, y, scratch, g, e, worst, s);
    }
}

constexpr int kChunkConv = 8, kChunkTransposed = 8;

// Tile family per layer shape: (32ch x 256pt) for narrow outputs, (128ch x 128pt) for mid-sized layers, (128ch x 256pt,
// 8 waves) for large 3x3 layers; the transposed form uses (64ch x 64pt x 4 phases): the small point tile gives the
// (H+1)x(W+1) grids of the 32^2..128^2 layers enough tiles that most of them run whole (a 128-point tile left every tile
// of those layers split between stream-K workers: 128 KB of accumulator slab per part).
// Images with <= kSmallPoints output points (4x4 .. 16x16) use a (128ch x 32pt) tile: the MFMA work of a chunk is fixed
// by the tile, so a 128-point tile would spend 4 us per chunk on padding there.
constexpr int kSmallPoints = 320;
// Images with >= kWidePoints points use an 8-wave (128ch x 256pt) tile for 3x3 convolutions with wide outputs: one
// workgroup per CU fetches the chunk's weight slab once instead of twice (the slab is 70 % of the staged bytes).
constexpr int kWidePoints = 1024;
constexpr int kWideTransposedPoints = 32768;   // (H+1)(W+1): the 256^2 -> 512^2 layers
constexpr int kSplitWideTransposedPoints = 4096;   // split-DMA form: the 8-wave transposed tile for 64^2 inputs
constexpr int kSplitMinPoints = 64;                // split-DMA form: smallest point grid (8^2; 9^2 transposed)

int worst_patch(int npts, int GW, int bp, int ksize, bool tr, int stride = 1) {
    int worst = 0;
    for (int q0 = 0; q0 < npts; q0 += bp) {
        const int q1 = (q0 + bp < npts ? q0 + bp : npts) - 1;
        const Window w = tile_window(q0, q1, GW, tr ? 0 : ksize / 2, tr, stride);
        if (w.PSZ > worst) worst = w.PSZ;
    }
    return worst;
}

// Patch positions per plane that the two LDS stages of the split-DMA form hold beside the weight rows of a `bo`-channel tile (two
// operand planes), and that eight patch DMA instructions per wave of an 8-wave tile reach: the budget of the stride-2 windows.
constexpr int down_patch_cap(int bo) {
    const int lds = (160 * 1024 / 2 - 2 * 10 * bo * 16) / (2 * 16) / 64 * 64;
    return lds < 2048 ? lds : 2048;
}

void tile_dims(int O, int H, int W, int ksize, int transposed, int form, int* bo, int* bp, int* cc, int* waves, int stride = 1) {
    const int npts = transposed ? (H + 1) * (W + 1) : H * W;
    *waves = 4; *cc = kChunkConv;
    // (form 3, r03: the 8^2 / 16^2 stride-1 layers and the 8^2 -> 16^2 / 16^2 -> 32^2 transposed ones run on the fp16-pair tiles too --
    //  a quarter of the matrix-pipe time of the fp32 tile's 36 MFMAs per chunk, although an 8^2 image fills a quarter of the 256-point
    //  tile; same-box frame A/B: 32^2 limit 340.4, 16^2 346.0 frames/s; on another box 16^2 327.0, 8^2 330.3, 4^2 332.2 (noise): 8^2)
    const bool sx_small = form == 3 && ksize == 3 && npts >= kSplitMinPoints;
    if (npts <= kSmallPoints && O > 32 && !sx_small) { *bo = 128; *bp = 32; }
    else if (transposed) {
        // fp16-pair form on large images: two point fragments per wave (64ch x 128pt x 4 phases) -- one accumulator set leaves
        // the registers for it, and a k-step then reads 7 operand fragments for 6 MFMAs instead of 4 for 3
        *bo = 64; *bp = (form == 2 && O % 4 == 0 && npts >= kWideTransposedPoints) ? 128 : 64; *cc = kChunkTransposed;
        // form 3 (ia_conv2d_mfma_sx, operands DMA'd into LDS): an 8-wave 64ch x 256pt x 4-phase tile, one workgroup per CU -- the
        // weight slab of a chunk (the larger part of the DMA traffic) is fetched once for 240 MFMAs instead of once per 60 / 120
        // (measured r02, B = 1: 72 -> 61 us on the 64^2 -> 128^2 layer; slower on the 32^2 / 128^2 / 256^2 inputs, whose tile counts
        // balance better over 512 four-wave slots, so only that size takes it)
        if (form == 3 && npts >= kSplitWideTransposedPoints && npts < 2 * kSplitWideTransposedPoints) { *bp = 256; *waves = 8; }
    }
    else if (O <= 32) { *bo = 32; *bp = 256; }
    else if (ksize == 3 && (O >= 128 || (O >= 64 && form == 3)) && O % 4 == 0 && (npts >= kWidePoints || sx_small)
             && worst_patch(npts, W, 256, 3, false, stride) <= (stride == 1 ? kPatchFloats : down_patch_cap(128))) {
        *bo = 128; *bp = 256; *waves = 8;
    }
    else if (stride == 2 && form == 3 && ksize == 3 && O % 32 == 0 && worst_patch(npts, W, 256, 3, false, stride) <= down_patch_cap(32)) {
        *bo = 32; *bp = 256; *waves = 8;         // stride-2 windows too large beside 128 channels of weights: the narrow whole-tile family
    }
    else { *bo = 128; *bp = 128; }
}

}  // namespace

// Shared by the planner and the entry point: tile counts and the split between whole rounds and stream-K.
struct Plan { int bo, bp, cc, waves, T, TO, C, T_dp, G, slab_floats; };
// (r03, with the 8^2 / 16^2 layers on the fp16-pair tiles and the shorter ToRGB launches: cap 128 / 64 / 48 / 32 workers = 321.8 / 326.4 /
//  326.2 / 320.0 frames/s same-box, 96 / 192 / 256 = 329.6 / 326.7 / 321.5 against 330.3 and 333.3 for 128 and 64 on another box: 64)
#ifndef IA_SMALL_LAYER_WORKERS
#define IA_SMALL_LAYER_WORKERS (ia::kNumCU / 4)      // (build-time sweeps: tools/_variants)
#endif
constexpr int kSmallLayerWorkers = IA_SMALL_LAYER_WORKERS, kSmallLayerPoints = 65 * 65;
// `stride` = 2 (form 3, stride-1 tile families): H x W is the OUTPUT grid of a 3x3 stride-2 convolution with padding 1.
static Plan make_plan(int B, int I, int O, int H, int W, int ksize, int transposed, int form, int stride = 1) {
    Plan p;
    const int npts = transposed ? (H + 1) * (W + 1) : H * W;
    tile_dims(O, H, W, ksize, transposed, form, &p.bo, &p.bp, &p.cc, &p.waves, stride);
    bool force_whole = stride == 2 && p.bo == 32 && O > 32;
    // Stride-1 3x3 layers of the split-DMA form that are smaller than the machine (512 -> 512 @64^2, 256 -> 256 @128^2 at one frame per
    // call): 32-channel x 256-point tiles, every tile whole, instead of 128 x 256 tiles cut between stream-K workers -- as soon as those
    // tiles give every CU one.  No slabs, no fix-up launch; a tile's weight rows (9 KB per chunk) are a quarter of the wide tile's.
    // Measured r03 (tools/bench_conv_layers.py, B = 1): 102.8 -> 76.2 us @64^2, 74.1 -> 67.3 us @128^2; 1024-point layers
    // (64 narrow tiles) stay on stream-K: 39 vs 64 us.
    if (form == 3 && !force_whole && !transposed && ksize == 3 && p.waves == 8 && O % 32 == 0) {
        const int64_t t_wide = (int64_t)B * ((npts + 255) / 256) * ((O + 127) / 128), t_narrow = (int64_t)B * ((npts + 255) / 256) * (O / 32);
        constexpr int min_tiles = ia::kNumCU;
        if (t_wide < ia::kNumCU && t_narrow >= min_tiles) { p.bo = 32; p.bp = 256; force_whole = true; }
    }
    p.TO = (O + p.bo - 1) / p.bo;
    p.T = ((npts + p.bp - 1) / p.bp) * p.TO;
    p.C = (I + p.cc - 1) / p.cc;
    const int slots = (p.waves == 8 ? 1 : 2) * ia::kNumCU;           // workgroups per CU of the tile family
    const int Gb = slots / B > 0 ? slots / B : 1;                    // slots of one batch element
    const int rounds = p.T / Gb, R = p.T - rounds * Gb;
    if (force_whole || R == 0 || rounds >= 8 || (rounds >= 1 && 4 * R >= 3 * Gb)) {
        p.T_dp = p.T; p.G = 0;                                       // whole (or nearly whole) rounds of whole tiles
    } else {
        p.T_dp = rounds * Gb;
        const int64_t Ur = (int64_t)R * p.C;
        // leftovers of a multi-round layer: at most an 8-way split per tile (a short tail);
        // a layer smaller than the machine: as many workers as fit, at least two chunks each
        const int per = rounds > 0 ? (p.C >= 8 ? p.C / 8 : 1) : 2;
        int64_t G = Ur / per;
        if (G > Gb) G = Gb;
        // ... but a layer smaller than the machine takes at most half of the CUs (one worker per tile if it has more tiles than
        // that): a frame runs the low-resolution layers of three networks on parallel streams, and a 512-workgroup launch of
        // 250-VGPR waves leaves no room for the other streams' kernels to be resident.  Up to 64^2 (65^2 points transposed) only:
        // measured 272 -> 280-284 frames/s at batch 1 for 0.241 -> 0.222 single-stream MFMA utilisation of the fp16-pair family;
        // capping the 128^2 layers too gives 283-285 and 0.213.  At batch >= 4 the per-element share Gb is below the cap anyway.
        if (rounds == 0 && npts <= kSmallLayerPoints) {
            constexpr int cap = kSmallLayerWorkers;
            const int64_t lim = p.T > cap ? p.T : cap;
            if (G > lim) G = lim;
        }
        if (G < 1) G = 1;
        p.G = (int)G;
    }
    const int frags = (p.bo / 32) * (p.bp / 32) / p.waves;           // fragments per wave
    p.slab_floats = (transposed ? 4 : 1) * frags * 16 * p.waves * 64;
    return p;
}

// Tile plan of a layer for the sibling translation unit (conv_split.hip): same tiles, worker counts and slab sizes for both forms.
int ia_conv2d_plan_tiles(int B, int I, int O, int H, int W, int ksize, int transposed, int form, int stride, int* bo, int* bp, int* waves, int* T,
                         int* TO, int* C, int* T_dp, int* slab_floats) {
    const Plan p = make_plan(B, I, O, H, W, ksize, transposed, form, stride);
    *bo = p.bo; *bp = p.bp; *waves = p.waves; *T = p.T; *TO = p.TO; *C = p.C; *T_dp = p.T_dp; *slab_floats = p.slab_floats;
    return IA_OK;
}

// Shapes ia_conv2d_mfma_sx takes (the single source of the rule: hipops.conv_sx_supported asks here).
extern "C" int ia_conv2d_sx_supported(int I, int O, int H, int W, int ksize, int transposed) {
    if (ksize != 3 || I % 8 || O % 8 || H < 1 || W < 1) return 0;
    const int npts = transposed ? (H + 1) * (W + 1) : H * W;
    const int side = 8;                                      // smallest image: kSplitMinPoints = 8^2 (9^2 points transposed)
    if ((H < W ? H : W) < side) return 0;
    if (transposed) return npts >= (side + 1) * (side + 1) ? 1 : 0;
    return (O >= 128 && npts >= kSplitMinPoints && W <= 512) ? 1 : 0;
}

static size_t scratch_bytes_for(int B, int G, int slab_floats) { return (size_t)B * G * 2 * slab_floats * sizeof(float); }

extern "C" int ia_conv2d_down_plan(int B, int I, int O, int H, int W, int* h_ksplit, size_t* h_scratch_bytes) {
    IA_REQUIRE(h_ksplit && h_scratch_bytes, "null output pointer");
    IA_REQUIRE(B > 0 && I > 0 && O > 0 && H > 0 && W > 0, "empty tensor");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    if (I % 8 || O % 8 || OH * OW < kSplitMinPoints || (OH < OW ? OH : OW) < 8)
        return ia::fail(IA_ERR_UNSUPPORTED, "the stride-2 form takes I %% 8 == 0, O %% 8 == 0 and outputs from 8^2 up (%d -> %d @%dx%d)", I, O, H, W);
    const Plan p = make_plan(B, I, O, OH, OW, 3, 0, 3, 2);
    const bool narrow = p.waves == 8 && p.bp == 256 && p.bo == 32;
    if (p.waves != 8 || (narrow && p.T_dp != p.T))
        return ia::fail(IA_ERR_UNSUPPORTED, "no stride-2 tile for %d -> %d @%dx%d (B %d)", I, O, H, W, B);
    *h_ksplit = p.G;
    *h_scratch_bytes = scratch_bytes_for(B, p.G, p.slab_floats);
    return IA_OK;
}

extern "C" int ia_conv2d_plan(int B, int I, int O, int H, int W, int ksize, int transposed, int form, int* h_ksplit,
                              size_t* h_scratch_bytes) {
    IA_REQUIRE(h_ksplit && h_scratch_bytes, "null output pointer");
    IA_REQUIRE(B > 0 && I > 0 && O > 0 && H > 0 && W > 0, "empty tensor");
    IA_REQUIRE(form >= 0 && form <= 3, "form: 0 = ia_conv2d_mfma, 1 = ia_conv2d_mfma_h, 2 = ia_conv2d_mfma_s, 3 = ia_conv2d_mfma_sx");
    const Plan p = make_plan(B, I, O, H, W, ksize, transposed, form);
    if (form == 3 && p.T_dp < p.T && conv_small_shape(B, I, O, H, W, ksize, transposed, 1)) {      // K split inside the workgroup (conv_small.h): no workers, no slabs
        *h_ksplit = 0;
        *h_scratch_bytes = 0;
        return IA_OK;
    }
    *h_ksplit = p.G;
    *h_scratch_bytes = scratch_bytes_for(B, p.G, p.slab_floats);
    return IA_OK;
}

static int conv2d_entry(const float* x, const void* wk_any, const float* styles, const float* demod,
                        const float* noise, const float* noise_strength, const float* bias, const float* residual,
                        float* y, float* scratch, size_t scratch_bytes,
                        int B, int I, int O, int H, int W, int ksize, int transposed,
                        int act, float alpha, float gain, float clamp, int ksplit, void* stream, int half_ops, int wk_exp = 0) {
    const float* wk = static_cast<const float*>(wk_any);
    IA_REQUIRE(x && wk && y, "x, wk and y must be device pointers");
    IA_REQUIRE(B > 0 && I > 0 && O > 0 && H > 0 && W > 0, "empty tensor");
    IA_REQUIRE(ksize == 1 || ksize == 3, "kernel size must be 1 or 3");
    IA_REQUIRE(!transposed || ksize == 3, "the transposed form is 3x3 stride 2 only");
    IA_REQUIRE(act == IA_ACT_LINEAR || act == IA_ACT_LRELU, "conv epilogue supports linear and lrelu");
    IA_REQUIRE(ksplit >= 0, "worker count must be >= 0");
    IA_REQUIRE(!transposed || (noise == nullptr && bias == nullptr && residual == nullptr && act == IA_ACT_LINEAR),
               "the transposed form only applies the demodulation; FIR + bias_act follow in ia_fir_bias_act");
    Geo g;
    g.B = B; g.I = I; g.O = O; g.H = H; g.W = W;
    g.GH = transposed ? H + 1 : H; g.GW = transposed ? W + 1 : W;
    g.OH = transposed ? 2 * H + 1 : H; g.OW = transposed ? 2 * W + 1 : W;
    IA_REQUIRE((int64_t)B * O * g.OH * g.OW <= INT32_MAX && (int64_t)B * I * H * W <= INT32_MAX, "tensor is too large");
    const Plan p = make_plan(B, I, O, H, W, ksize, transposed, half_ops);
    const int bp_ = p.bp;
    g.T = p.T; g.TO = p.TO; g.C = p.C; g.T_dp = p.T_dp; g.G = 0;
    if (p.T_dp < p.T) {
        IA_REQUIRE(ksplit >= 1, "this layer has stream-K tiles: pass the worker count from ia_conv2d_plan");
        const int64_t Ur = (int64_t)(p.T - p.T_dp) * p.C;
        g.G = (int)(ksplit > Ur ? Ur : ksplit);
        const size_t need = scratch_bytes_for(B, g.G, p.slab_floats);
        const bool whole_tiles = Ur % g.G == 0 && (Ur / g.G) % g.C == 0;
        IA_REQUIRE(whole_tiles || (scratch && scratch_bytes >= need), "stream-K needs %zu bytes of scratch, got %zu", need, scratch_bytes);
    }
    g.patch_cap = 0;
    g.acc_scale = ldexpf(1.f, -wk_exp);
    Epi e{demod, noise, noise_strength, bias, residual, act, alpha, gain, clamp, nullptr, nullptr, 2};
    hipStream_t s = (hipStream_t)stream;
    if (half_ops) {
        const bool wide = p.waves == 8;
        IA_REQUIRE(ksize == 3 && (wide || (transposed && p.bo == 64)) && I % 8 == 0 && O % 4 == 0,
                   "the fp16-operand form covers 3x3 layers on the two-stage tiles (large stride-1 layers, stride-2 transposed) with I %% 8 == 0, O %% 4 == 0");
        if (half_ops == 2) {
            if (transposed && p.bp == 128) return launch<3, true, 1, 2, 2, 2, kChunkTransposed, 2>(x, wk, styles, y, scratch, g, e, s);
            if (transposed) return launch<3, true, 1, 1, 2, 2, kChunkTransposed, 2>(x, wk, styles, y, scratch, g, e, s);
            return launch<3, false, 2, 2, 2, 4, kChunkConv, 2>(x, wk, styles, y, scratch, g, e, s);
        }
        if (transposed) return launch<3, true, 1, 1, 2, 2, kChunkTransposed, 1>(x, wk, styles, y, scratch, g, e, s);
        return launch<3, false, 2, 2, 2, 4, kChunkConv, 1>(x, wk, styles, y, scratch, g, e, s);
    }
    if (bp_ == 32) {   // small images: 4 waves side by side over 128 out-channels, one 32-point fragment each
        if (transposed) return launch<3, true, 1, 1, 4, 1, kChunkConv>(x, wk, styles, y, scratch, g, e, s);
        return ksize == 3 ? launch<3, false, 1, 1, 4, 1, kChunkConv>(x, wk, styles, y, scratch, g, e, s)
                          : launch<1, false, 1, 1, 4, 1, kChunkConv>(x, wk, styles, y, scratch, g, e, s);
    }
    if (transposed) return launch<3, true, 1, 1, 2, 2, kChunkTransposed>(x, wk, styles, y, scratch, g, e, s);
    if (O <= 32) {
        return ksize == 3 ? launch<3, false, 1, 2, 1, 4, kChunkConv>(x, wk, styles, y, scratch, g, e, s)
                          : launch<1, false, 1, 2, 1, 4, kChunkConv>(x, wk, styles, y, scratch, g, e, s);
    }
    if (p.waves == 8) return launch<3, false, 2, 2, 2, 4, kChunkConv>(x, wk, styles, y, scratch, g, e, s);
    return ksize == 3 ? launch<3, false, 2, 2, 2, 2, kChunkConv>(x, wk, styles, y, scratch, g, e, s)
                      : launch<1, false, 2, 2, 2, 2, kChunkConv>(x, wk, styles, y, scratch, g, e, s);
}

extern "C" int ia_conv2d_mfma(const float* x, const float* wk, const float* styles, const float* demod,
                              const float* noise, const float* noise_strength, const float* bias, const float* residual,
                              float* y, float* scratch, size_t scratch_bytes,
                              int B, int I, int O, int H, int W, int ksize, int transposed,
                              int act, float alpha, float gain, float clamp, int ksplit, void* stream) {
    return conv2d_entry(x, wk, styles, demod, noise, noise_strength, bias, residual, y, scratch, scratch_bytes, B, I, O, H, W, ksize,
                        transposed, act, alpha, gain, clamp, ksplit, stream, 0);
}

extern "C" int ia_conv2d_mfma_h(const float* x, const void* wk_h, const float* styles, const float* demod,
                                const float* noise, const float* noise_strength, const float* bias, const float* residual,
                                float* y, float* scratch, size_t scratch_bytes,
                                int B, int I, int O, int H, int W, int ksize, int transposed,
                                int act, float alpha, float gain, float clamp, int ksplit, void* stream) {
    return conv2d_entry(x, wk_h, styles, demod, noise, noise_strength, bias, residual, y, scratch, scratch_bytes, B, I, O, H, W, ksize,
                        transposed, act, alpha, gain, clamp, ksplit, stream, 1);
}

extern "C" int ia_conv2d_mfma_s(const float* x, const void* wk_split, int wk_exp, const float* styles, const float* demod,
                                const float* noise, const float* noise_strength, const float* bias, const float* residual,
                                float* y, float* scratch, size_t scratch_bytes,
                                int B, int I, int O, int H, int W, int ksize, int transposed,
                                int act, float alpha, float gain, float clamp, int ksplit, void* stream) {
    IA_REQUIRE(wk_exp >= -14 && wk_exp <= 30, "wk_exp is the power of two the weights were scaled by at pack time");
    return conv2d_entry(x, wk_split, styles, demod, noise, noise_strength, bias, residual, y, scratch, scratch_bytes, B, I, O, H, W, ksize,
                        transposed, act, alpha, gain, clamp, ksplit, stream, 2, wk_exp);
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

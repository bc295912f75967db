// ia_tokens_split / ia_linear_sx: the token-major linear layers of the one-shot inversion's transformer blocks
// (reference: encoder_inversion/models/mmseg/mix_transformer.py:18-116 -- Mlp.fc1 / fc2, Attention.q / kv / proj at 1 024 dims on
// 64 .. 4 096 tokens; unet_transformer.py calls them through MixVisionTransformer blocks) as fp32-equivalent GEMMs on the fp16 pipe.
//
// y[m][n] = act(sum_k x[m][k] * w[n][k] + bias[n]) (+ residual[m][n]), x fp32 [M][K] tokens, w the nn.Linear weight [N][K].
// The library route these replace runs fp32 MFMAs (rocBLAS Cijk_*_MI16x16x1: 115 - 135 TFLOP/s on 4 096 x 1 024 x 4 096, r05
// one-shot kernel table).  Here the fp32 products come from fp16 hi / lo pairs like the 3x3 convolutions' (conv_common.h): three
// v_mfma_f32_32x32x16_f16 per k-step -- hi x hi, hi x lo, lo' x (hi * 2^-11) -- into fp32 accumulators (the cross terms in their own, see the kernel), lo x lo (~2^-24) dropped.
//   * ia_tokens_split turns the token matrix into the split format once per consumer set: [2][K/8][M][8] fp16, hi = fp16(v),
//     lo' = fp16((v - hi) * 2^11) (ia::split_f16, range watch included) -- the layout of the convolutions' activations with the
//     token index in the place of the pixel, so a lane's A fragment (its token, eight consecutive k) is ONE 16-byte load.
//   * the weight is split once on the host side in the convolution weight format of a 1x1 kernel ([2][1][K/8][N][8], pre-scaled
//     by 2^wk_exp: hipops.pack_conv_weight_split) -- a lane's B fragment is one 16-byte load too.
//   * linear_split_kernel: a wave owns 32 FA tokens x 32 FB output features over all of K and loads its fragments straight into the
//     MFMA operand registers with raw buffer loads (rows past M / N read zeros through the range check): no LDS, no barriers; RING
//     k-steps are in flight ahead of the one being multiplied (branch-free loop, counted waits).  Tokens are the MFMA's rows, so a
//     lane's accumulator registers are tokens and its lane index the output feature: stores are 128-byte runs of a token row, bias
//     is one value per lane.  Four waves (2 x 2) form a workgroup and share fragments through L1 / L2 only.
//   * KS = 4 (few tokens): the four waves take the k-steps round-robin on ONE 32 FA x 32 FB tile and meet in LDS in wave order.
#include "ia_common.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

constexpr int kOutside = 0x7ffffff0;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t lin_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ h16x8 lin_load16(__amdgpu_buffer_rsrc_t r, int voffset, int soffset) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voffset, soffset, 0);
    return __builtin_bit_cast(h16x8, v);
}

// ---- tokens [M][K] fp32 -> [2][K/8][M][8] fp16 pairs.  A workgroup turns 32 tokens x 8 octets: 256-byte runs of a token row in,
// 512-byte runs of an octet plane out, the transposition through LDS.
__global__ __launch_bounds__(256) void tokens_split_kernel(const float* __restrict__ x, int64_t ld, h16x8* __restrict__ xs, int M, int K8) {
    __shared__ h16x8 s_hi[8][33], s_lo[8][33];
    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * 32, o0 = blockIdx.y * 8;
    {
        const int ml = tid >> 3, ol = tid & 7, m = m0 + ml, o = o0 + ol;
        ia::SatWatch watch;
        h16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) { hi[j] = (_Float16)0.f; lo[j] = (_Float16)0.f; }
        if (m < M && o < K8) {
            const float4* p = reinterpret_cast<const float4*>(x + (int64_t)m * ld + o * 8);
            const float4 a = p[0], b = p[1];
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                _Float16 h, l;
                ia::split_f16(v[j], h, l, watch);
                hi[j] = h;
                lo[j] = l;
            }
        }
        watch.report();
        s_hi[ol][ml] = hi;
        s_lo[ol][ml] = lo;
    }
    __syncthreads();
    const int ol = tid >> 5, ml = tid & 31, m = m0 + ml, o = o0 + ol;
    if (m < M && o < K8) {
        xs[(int64_t)o * M + m] = s_hi[ol][ml];
        xs[((int64_t)K8 + o) * M + m] = s_lo[ol][ml];
    }
}

// ---- patches of an NCHW image -> the split format of a token matrix: row m = (b, oy, ox), column k = (c, ky, kx) in the order of
// conv.weight.reshape(N, C * ks * ks); columns K .. Kp - 1 (Kp = K rounded up to 16) are zero.  One thread = one (octet, row): eight
// gathers (the image lives in L2), 16-byte stores that are contiguous over the rows of a workgroup.  blockIdx.y = the octet, so the
// (c, ky, kx) of its eight columns are scalar arithmetic.
template <int KS_>
__global__ __launch_bounds__(256) void im2col_split_kernel(const float* __restrict__ x, h16x8* __restrict__ xs, int B, int C, int H, int W, int OH, int OW,
                                                           int stride, int pad, int K8) {
    const int m = blockIdx.x * 256 + threadIdx.x, M = B * OH * OW, oct = blockIdx.y;
    if (m >= M) return;
    const int b = m / (OH * OW), r = m - b * OH * OW, oy = r / OW, ox = r - oy * OW;
    const float* xb = x + (int64_t)b * C * H * W;
    const int iy0 = oy * stride - pad, ix0 = ox * stride - pad;
    ia::SatWatch watch;
    h16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = oct * 8 + j, c = k / (KS_ * KS_), t = k - c * (KS_ * KS_), ky = t / KS_, kx = t - ky * KS_;
        const int iy = iy0 + ky, ix = ix0 + kx;
        const float v = (c < C && iy >= 0 && iy < H && ix >= 0 && ix < W) ? xb[((int64_t)c * H + iy) * W + ix] : 0.f;
        _Float16 h, l;
        ia::split_f16(v, h, l, watch);
        hi[j] = h;
        lo[j] = l;
    }
    watch.report();
    xs[(int64_t)oct * M + m] = hi;
    xs[((int64_t)K8 + oct) * M + m] = lo;
}

// ---- LayerNorm over the features of a token + the split of its result (Block.forward: norm1 -> q / kv, norm2 -> fc1: the normalised
// tokens have no other reader).  One wave per token, OPL = K / 512 octets per lane (octets l, 64 + l, ...: 2 KB runs of the row), mean and
// variance in two passes over the registers (wave reductions), then (x - mean) * rstd * gamma + beta as fp16 pairs.  A workgroup takes
// 16 consecutive tokens, four per wave, and hands them over through LDS so that the stores are 256-byte runs of an octet plane.
template <int OPL>
__global__ __launch_bounds__(256) void layernorm_split_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              h16x8* __restrict__ xs, int M, float eps) {
    constexpr int K8 = OPL * 64, K = K8 * 8;
    extern __shared__ __attribute__((aligned(16))) h16x8 s_ln[];        // [2 planes][K8][16 tokens]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.x * 16;
    float g[OPL][8], bt[OPL][8];
#pragma unroll
    for (int i = 0; i < OPL; ++i) {
        const float4* gp = reinterpret_cast<const float4*>(gamma + (i * 64 + lane) * 8);
        const float4* bp = reinterpret_cast<const float4*>(beta + (i * 64 + lane) * 8);
        const float4 a = gp[0], b = gp[1], c = bp[0], d = bp[1];
        g[i][0] = a.x; g[i][1] = a.y; g[i][2] = a.z; g[i][3] = a.w; g[i][4] = b.x; g[i][5] = b.y; g[i][6] = b.z; g[i][7] = b.w;
        bt[i][0] = c.x; bt[i][1] = c.y; bt[i][2] = c.z; bt[i][3] = c.w; bt[i][4] = d.x; bt[i][5] = d.y; bt[i][6] = d.z; bt[i][7] = d.w;
    }
    ia::SatWatch watch;
    for (int t = 0; t < 4; ++t) {
        const int ml = wave * 4 + t, m = m0 + ml;
        if (m >= M) break;
        float v[OPL][8];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < OPL; ++i) {
            const float4* p = reinterpret_cast<const float4*>(x + (int64_t)m * K + (i * 64 + lane) * 8);
            const float4 a = p[0], b = p[1];
            v[i][0] = a.x; v[i][1] = a.y; v[i][2] = a.z; v[i][3] = a.w; v[i][4] = b.x; v[i][5] = b.y; v[i][6] = b.z; v[i][7] = b.w;
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[i][j];
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
        const float mean = sum * (1.f / K);
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < OPL; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[i][j] -= mean;
                sq = fmaf(v[i][j], v[i][j], sq);
            }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sq += __shfl_xor(sq, d);
        const float rstd = rsqrtf(sq * (1.f / K) + eps);
#pragma unroll
        for (int i = 0; i < OPL; ++i) {
            h16x8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                _Float16 h, l;
                ia::split_f16(v[i][j] * rstd * g[i][j] + bt[i][j], h, l, watch);
                hi[j] = h;
                lo[j] = l;
            }
            s_ln[(i * 64 + lane) * 16 + ml] = hi;
            s_ln[(K8 + i * 64 + lane) * 16 + ml] = lo;
        }
    }
    watch.report();
    __syncthreads();
    // 256 threads = 16 octets x 16 tokens per pass
    const int ml = threadIdx.x & 15, m = m0 + ml;
    if (m < M) {
        for (int o = threadIdx.x >> 4; o < K8; o += 16) {
            xs[(int64_t)o * M + m] = s_ln[o * 16 + ml];
            xs[((int64_t)K8 + o) * M + m] = s_ln[(K8 + o) * 16 + ml];
        }
    }
}

struct NoWatch { __device__ __forceinline__ void see(float) const {} };

// ---- V of an attention layer, split ALONG THE KEYS: v [M keys][ld] (C columns) -> [2][M/8][C][8]: the B operand of P . V, whose k index is
// the key.  One thread = (key octet, column): eight reads that are contiguous over the columns of a wave, one 16-byte store per plane.
__global__ __launch_bounds__(256) void tokens_split_t_kernel(const float* __restrict__ v, int64_t ld, h16x8* __restrict__ vt, int M8, int C, int perm) {
    const int c = blockIdx.x * 256 + threadIdx.x, mo = blockIdx.y;
    if (c >= C) return;
    ia::SatWatch watch;
    h16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        _Float16 h, l;
        // perm: the keys of a 16-key step in the row order of the 32 x 32 MFMA accumulator (octet 2 t + hh = keys 16 t + {0..3, 8..11} + 4 hh):
        // the probabilities a lane of ia_attention_sx holds are then a B fragment as they stand
        const int key = perm ? 16 * (mo >> 1) + (j < 4 ? j : j + 4) + 4 * (mo & 1) : mo * 8 + j;
        ia::split_f16(v[(int64_t)key * ld + c], h, l, watch);
        hi[j] = h;
        lo[j] = l;
    }
    watch.report();
    vt[(int64_t)mo * C + c] = hi;
    vt[((int64_t)M8 + mo) * C + c] = lo;
}

// ---- softmax over the keys of one score row, written as the A operand of P . V: s [Z][N][M] fp32 (already scaled) -> [2][Z][M/8][N][8].
// One wave per row: lane l holds the octets l, 64 + l, ... of its row (two 16-byte loads each), max and sum by wave reductions, the
// probabilities leave as fp16 pairs.  The four waves of a workgroup take four consecutive rows.
template <int IT>
__global__ __launch_bounds__(256) void softmax_split_kernel(const float* __restrict__ s, h16x8* __restrict__ ps, int Z, int N, int M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= (int64_t)Z * N) return;
    const int z = (int)(row / N), n = (int)(row - (int64_t)z * N), M8 = M >> 3;
    const float4* sp = reinterpret_cast<const float4*>(s + row * M);
    float v[IT][8];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int oct = i * 64 + lane;
        const bool ok = oct < M8;
        const float4 a = ok ? sp[oct * 2] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        const float4 b = ok ? sp[oct * 2 + 1] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        v[i][0] = a.x; v[i][1] = a.y; v[i][2] = a.z; v[i][3] = a.w; v[i][4] = b.x; v[i][5] = b.y; v[i][6] = b.z; v[i][7] = b.w;
#pragma unroll
        for (int j = 0; j < 8; ++j) mx = fmaxf(mx, v[i][j]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[i][j] = expf(v[i][j] - mx);
            sum += v[i][j];
        }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
    const float inv = 1.f / sum;
    NoWatch nw;
    h16x8* hi_p = ps + ((int64_t)z * M8) * N + n;
    h16x8* lo_p = hi_p + (int64_t)Z * M8 * N;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int oct = i * 64 + lane;
        if (oct >= M8) continue;
        h16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            _Float16 h, l;
            ia::split_f16(v[i][j] * inv, h, l, nw);
            hi[j] = h;
            lo[j] = l;
        }
        hi_p[(int64_t)oct * N] = hi;
        lo_p[(int64_t)oct * N] = lo;
    }
}

struct LinParams {
    const h16x8* xs;         // [2][K8][M][8]
    const h16x8* ws;         // [2][K8][N][8], scaled by 2^wk_exp
    const float* bias;       // [N] or null
    const float* residual;   // [M][N] or null (added after the activation)
    float* y;                // [M][N]
    int M, N, K8;
    float acc_scale;         // 2^-wk_exp
    int gelu;
    // general operand placement (ia_matmul_sx; the linear layers use the dense defaults): rows of the split tensors (the octet stride is
    // rows * 16 bytes), bytes between the hi and lo planes, bytes per batch inside a plane; output row / batch strides in floats;
    // b_lo_scaled: B's low parts carry 2^11 like A's (two token matrices) instead of the weights' unscaled ones
    int a_rows, b_rows;
    int64_t a_plane, b_plane, a_batch, b_batch, y_batch, ldy;
    int b_lo_scaled;
};

__device__ __forceinline__ float lin_finish(float v, float bias, int gelu) {
    v += bias;
    if (gelu) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));      // nn.GELU(approximate='none')
    return v;
}

// FA x FB fragments of 32 tokens x 32 features per wave.  KS = 1: four waves as 2 x 2 tiles; KS = 4: four waves on one tile, k-steps round-robin.
template <int FA, int FB, int KS, int RING>
__global__ __launch_bounds__(256, 2) void linear_split_kernel(LinParams p) {      // (two workgroups per CU: 256 registers per lane, accumulators included)
    __shared__ float4 s_part[KS > 1 ? 4 * FA * FB * 4 * 64 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, half = lane >> 5;
    const int wm = KS > 1 ? 0 : wave >> 1, wn = KS > 1 ? 0 : wave & 1;
    constexpr int TM = (KS > 1 ? 1 : 2) * 32 * FA, TN = (KS > 1 ? 1 : 2) * 32 * FB;
    const int m0 = blockIdx.y * TM + wm * 32 * FA, n0 = blockIdx.x * TN + wn * 32 * FB;
    const int steps = p.K8 >> 1;                                  // one k-step = 16 input features: lanes 0-31 the even octet, 32-63 the odd one
    const int z = blockIdx.z;
    const unsigned a_bytes = (unsigned)p.K8 * (unsigned)p.a_rows * 16u, b_bytes = (unsigned)p.K8 * (unsigned)p.b_rows * 16u;
    const char* ab = reinterpret_cast<const char*>(p.xs) + z * p.a_batch;
    const char* bb = reinterpret_cast<const char*>(p.ws) + z * p.b_batch;
    const __amdgpu_buffer_rsrc_t ra0 = lin_rsrc(ab, a_bytes), ra1 = lin_rsrc(ab + p.a_plane, a_bytes);
    const __amdgpu_buffer_rsrc_t rb0 = lin_rsrc(bb, b_bytes), rb1 = lin_rsrc(bb + p.b_plane, b_bytes);
    int a_off[FA], b_off[FB];
#pragma unroll
    for (int f = 0; f < FA; ++f) {
        const int m = m0 + f * 32 + l31;
        a_off[f] = m < p.M ? (half * p.a_rows + m) * 16 : kOutside;
    }
#pragma unroll
    for (int f = 0; f < FB; ++f) {
        const int n = n0 + f * 32 + l31;
        b_off[f] = n < p.N ? (half * p.b_rows + n) * 16 : kOutside;
    }
    const int a_step = 2 * p.a_rows * 16, b_step = 2 * p.b_rows * 16;
    float* yz = p.y + z * p.y_batch;
    const float b_sc_mul = p.b_lo_scaled ? 1.f : 1.0f / 2048.0f, x_mul = p.b_lo_scaled ? 1.0f / 2048.0f : 1.f;

    // two accumulators per fragment: the MFMA's accumulate truncates, so a chain of n MFMAs drifts by ~n / 2 ulp of the SUM (measured:
    // 3e-6 on 0.6 after 192 MFMAs in one register, 1e-6 after 48) -- the cross terms (2^-11 of the sum) get their own register and
    // enter once at the end; the hi x hi chain is K / 16 long
    f32x16 acc[FA][FB], acx[FA][FB];
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acx[i][j][r] = 0.f; }

    h16x8 qa[RING][2][FA], qb[RING][2][FB];
    auto load_step = [&](int slot, int st, bool live) {
        const int dead = live ? 0 : kOutside;
        const int sa = st * a_step, sb = st * b_step;
#pragma unroll
        for (int f = 0; f < FA; ++f) {
            qa[slot][0][f] = lin_load16(ra0, a_off[f] | dead, sa);
            qa[slot][1][f] = lin_load16(ra1, a_off[f] | dead, sa);
        }
#pragma unroll
        for (int f = 0; f < FB; ++f) {
            qb[slot][0][f] = lin_load16(rb0, b_off[f] | dead, sb);
            qb[slot][1][f] = lin_load16(rb1, b_off[f] | dead, sb);
        }
    };
    const int first = KS > 1 ? wave : 0;
#pragma unroll
    for (int s = 0; s < RING; ++s) {
        const int st = first + s * KS;
        const bool live = st < steps;
        load_step(s, live ? st : 0, live);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int st0 = first; st0 < steps; st0 += RING * KS) {
#pragma unroll
        for (int s = 0; s < RING; ++s) {
            const int nxt = st0 + (s + RING) * KS;
            const bool more = nxt < steps;
            h16x8 b_sc[FB];
#pragma unroll
            for (int j = 0; j < FB; ++j) b_sc[j] = qb[s][0][j] * (_Float16)b_sc_mul;      // weight high parts at 2^-11 meet the tokens' low parts (at 2^11); two token matrices: both cross terms stay at 2^11
#pragma unroll
            for (int i = 0; i < FA; ++i)
#pragma unroll
                for (int j = 0; j < FB; ++j) acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa[s][0][i], qb[s][1][j], acx[i][j], 0, 0, 0);     // hi x lo
#pragma unroll
            for (int i = 0; i < FA; ++i)
#pragma unroll
                for (int j = 0; j < FB; ++j) acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa[s][1][i], b_sc[j], acx[i][j], 0, 0, 0);          // lo' x (hi * 2^-11)
#pragma unroll
            for (int i = 0; i < FA; ++i)
#pragma unroll
                for (int j = 0; j < FB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa[s][0][i], qb[s][0][j], acc[i][j], 0, 0, 0);     // hi x hi
            __builtin_amdgcn_sched_barrier(0);
            load_step(s, more ? nxt : 0, more);                      // refill the slot (the MFMAs above have read its registers)
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // C/D map of the 32 x 32 MFMA: row (token) = (r & 3) + 8 * (r >> 2) + 4 * half, column (feature) = l31
    if constexpr (KS > 1) {
        float4* pw = s_part + (wave * FA * FB * 4) * 64 + lane;
#pragma unroll
        for (int i = 0; i < FA; ++i)
#pragma unroll
            for (int j = 0; j < FB; ++j)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq)
                    pw[((i * FB + j) * 4 + rq) * 64] = make_float4(acc[i][j][4 * rq] + acx[i][j][4 * rq] * x_mul, acc[i][j][4 * rq + 1] + acx[i][j][4 * rq + 1] * x_mul,
                                                                   acc[i][j][4 * rq + 2] + acx[i][j][4 * rq + 2] * x_mul, acc[i][j][4 * rq + 3] + acx[i][j][4 * rq + 3] * x_mul);
        __syncthreads();
        constexpr int NQ = FA * FB * 4;
        for (int q = wave; q < NQ; q += 4) {
            const float4* pr = s_part + q * 64 + lane;
            float4 v = pr[0];
#pragma unroll
            for (int w = 1; w < 4; ++w) {                            // wave order: the same bits on every launch
                const float4 u = pr[(w * NQ) * 64];
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            const int frag = q >> 2, rq = q & 3, i = frag / FB, j = frag - i * FB;
            const int n = n0 + j * 32 + l31;
            if (n >= p.N) continue;
            const float bias = p.bias ? p.bias[n] : 0.f;
            const float vin[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int m = m0 + i * 32 + 8 * rq + 4 * half + k;
                if (m >= p.M) continue;
                float o = lin_finish(vin[k] * p.acc_scale, bias, p.gelu);
                if (p.residual) o += p.residual[(int64_t)m * p.N + n];
                yz[(int64_t)m * p.ldy + n] = o;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < FB; ++j) {
            const int n = n0 + j * 32 + l31;
            if (n >= p.N) continue;
            const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < FA; ++i) {
                float res[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    res[r] = (p.residual && m < p.M) ? p.residual[(int64_t)m * p.N + n] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (m >= p.M) continue;
                    yz[(int64_t)m * p.ldy + n] = lin_finish((acc[i][j][r] + acx[i][j][r] * x_mul) * p.acc_scale, bias, p.gelu) + res[r];
                }
                __builtin_amdgcn_sched_barrier(0);               // (one fragment's 32 accumulator registers through the VGPRs at a time)
            }
        }
    }
}

}  // namespace

extern "C" int ia_tokens_split(const float* x, int64_t ld, void* xs, int M, int K, void* stream) {
    IA_REQUIRE(x && xs, "x and xs must be device pointers");
    IA_REQUIRE(M > 0 && K > 0 && ld >= K, "empty matrix, or a row stride shorter than a row");
    if (K % 16 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0 || ld % 4 != 0)
        return ia::fail(IA_ERR_UNSUPPORTED, "ia_tokens_split needs K %% 16 == 0, a row stride %% 4 == 0 and a 16-byte aligned matrix (got K = %d, ld = %lld)", K, (long long)ld);
    IA_REQUIRE((int64_t)M * K <= (int64_t)1 << 30, "matrix too large for 32-bit plane offsets");
    const int K8 = K / 8;
    hipLaunchKernelGGL(tokens_split_kernel, dim3((unsigned)((M + 31) / 32), (unsigned)((K8 + 7) / 8)), dim3(256), 0, (hipStream_t)stream, x, ld,
                       static_cast<h16x8*>(xs), M, K8);
    return ia::check_launch("ia_tokens_split");
}

extern "C" int ia_im2col_split(const float* x, void* xs, int B, int C, int H, int W, int ksize, int stride, int pad, void* stream) {
    IA_REQUIRE(x && xs, "x and xs must be device pointers");
    IA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && stride > 0 && pad >= 0, "empty image");
    if (ksize != 7 && ksize != 3) return ia::fail(IA_ERR_UNSUPPORTED, "ia_im2col_split covers 7x7 and 3x3 patches (got %d)", ksize);
    const int OH = (H + 2 * pad - ksize) / stride + 1, OW = (W + 2 * pad - ksize) / stride + 1;
    IA_REQUIRE(OH > 0 && OW > 0, "patch larger than the padded image");
    const int64_t M = (int64_t)B * OH * OW, Kp = ((int64_t)C * ksize * ksize + 15) / 16 * 16;
    IA_REQUIRE(M * Kp <= (int64_t)1 << 30 && Kp / 8 <= 65535, "matrix too large for 32-bit plane offsets");
    const dim3 grid((unsigned)((M + 255) / 256), (unsigned)(Kp / 8));
    const hipStream_t s = (hipStream_t)stream;
    if (ksize == 7) hipLaunchKernelGGL(im2col_split_kernel<7>, grid, dim3(256), 0, s, x, static_cast<h16x8*>(xs), B, C, H, W, OH, OW, stride, pad, (int)(Kp / 8));
    else hipLaunchKernelGGL(im2col_split_kernel<3>, grid, dim3(256), 0, s, x, static_cast<h16x8*>(xs), B, C, H, W, OH, OW, stride, pad, (int)(Kp / 8));
    return ia::check_launch("ia_im2col_split");
}

extern "C" int ia_layernorm_split(const float* x, const float* gamma, const float* beta, float eps, void* xs, int M, int K, void* stream) {
    IA_REQUIRE(x && gamma && beta && xs, "x, gamma, beta and xs must be device pointers");
    IA_REQUIRE(M > 0 && K > 0, "empty matrix");
    if (!(K == 512 || K == 1024 || K == 2048) || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) != 0)
        return ia::fail(IA_ERR_UNSUPPORTED, "ia_layernorm_split covers 512, 1024 and 2048 features on 16-byte aligned arrays (got K = %d)", K);
    IA_REQUIRE((int64_t)M * K <= (int64_t)1 << 30, "matrix too large for 32-bit plane offsets");
    const size_t lds = (size_t)2 * (K / 8) * 16 * 16;
    const dim3 grid((unsigned)((M + 15) / 16));
    const hipStream_t s = (hipStream_t)stream;
    h16x8* out = static_cast<h16x8*>(xs);
    auto go = [&](auto kern) -> int {
        if (const int st = ia::reserve_lds((const void*)kern, lds, "ia_layernorm_split")) return st;
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, x, gamma, beta, out, M, eps);
        return ia::check_launch("ia_layernorm_split");
    };
    return K == 512 ? go(layernorm_split_kernel<1>) : K == 1024 ? go(layernorm_split_kernel<2>) : go(layernorm_split_kernel<4>);
}

extern "C" int ia_tokens_split_t(const float* v, int64_t ld, void* vt, int M, int C, int perm, void* stream) {
    IA_REQUIRE(v && vt, "v and vt must be device pointers");
    IA_REQUIRE(M > 0 && C > 0 && ld >= C, "empty matrix, or a row stride shorter than a row");
    if (M % 16 != 0) return ia::fail(IA_ERR_UNSUPPORTED, "ia_tokens_split_t needs M %% 16 == 0 (got %d)", M);
    IA_REQUIRE((int64_t)M * C <= (int64_t)1 << 30 && M / 8 <= 65535, "matrix too large for 32-bit plane offsets");
    hipLaunchKernelGGL(tokens_split_t_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)(M / 8)), dim3(256), 0, (hipStream_t)stream, v, ld,
                       static_cast<h16x8*>(vt), M / 8, C, perm);
    return ia::check_launch("ia_tokens_split_t");
}

extern "C" int ia_softmax_split(const float* s, void* ps, int Z, int N, int M, void* stream) {
    IA_REQUIRE(s && ps, "s and ps must be device pointers");
    IA_REQUIRE(Z > 0 && N > 0 && M > 0, "empty matrix");
    if (M % 16 != 0 || M > 4096 || (reinterpret_cast<uintptr_t>(s) & 15) != 0)
        return ia::fail(IA_ERR_UNSUPPORTED, "ia_softmax_split covers rows of M %% 16 == 0, M <= 4096 keys (got %d)", M);
    IA_REQUIRE((int64_t)Z * N * M <= (int64_t)1 << 31, "score tensor too large");
    const dim3 grid((unsigned)(((int64_t)Z * N + 3) / 4));
    const hipStream_t st = (hipStream_t)stream;
    h16x8* out = static_cast<h16x8*>(ps);
    const int it = (M / 8 + 63) / 64;
    if (it <= 1) hipLaunchKernelGGL(softmax_split_kernel<1>, grid, dim3(256), 0, st, s, out, Z, N, M);
    else if (it <= 2) hipLaunchKernelGGL(softmax_split_kernel<2>, grid, dim3(256), 0, st, s, out, Z, N, M);
    else if (it <= 4) hipLaunchKernelGGL(softmax_split_kernel<4>, grid, dim3(256), 0, st, s, out, Z, N, M);
    else hipLaunchKernelGGL(softmax_split_kernel<8>, grid, dim3(256), 0, st, s, out, Z, N, M);
    return ia::check_launch("ia_softmax_split");
}

namespace {
int lin_launch(const LinParams& p, int batch, hipStream_t s) {
    // 128 x 128 workgroup tiles (four waves of 64 x 64) from one tile per CU up; below that one 32 x 64 tile per workgroup with K over its
    // four waves (r06, tools/bench_linear.py: 4 096 x 1 024 x 1 024 39.5 vs 46.2 us, 1 024^3 31.0 vs 14.2; a 64 x 64 form of the first
    // kind won nowhere)
    static const int force = getenv("IA_LINEAR_TILE") ? atoi(getenv("IA_LINEAR_TILE")) : 0;
    const bool big = force ? force == 2 : ia::ceil_div(p.M, 128) * ia::ceil_div(p.N, 128) * batch >= ia::kNumCU;
    if (big)
        hipLaunchKernelGGL((linear_split_kernel<2, 2, 1, 3>), dim3((unsigned)ia::ceil_div(p.N, 128), (unsigned)ia::ceil_div(p.M, 128), (unsigned)batch), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((linear_split_kernel<1, 2, 4, 3>), dim3((unsigned)ia::ceil_div(p.N, 64), (unsigned)ia::ceil_div(p.M, 32), (unsigned)batch), dim3(256), 0, s, p);
    return IA_OK;
}
}  // namespace

extern "C" int ia_matmul_sx(const void* a_split, const void* b_split, float* y, int batch, int M, int N, int K, int a_rows, int64_t a_plane_bytes,
                            int64_t a_batch_bytes, int b_rows, int64_t b_plane_bytes, int64_t b_batch_bytes, int64_t y_batch_stride, int64_t y_row_stride,
                            float scale, void* stream) {
    IA_REQUIRE(a_split && b_split && y, "a_split, b_split and y must be device pointers");
    IA_REQUIRE(batch > 0 && batch <= 65535 && M > 0 && N > 0 && K > 0, "empty product");
    IA_REQUIRE(a_rows >= M && b_rows >= N && y_row_stride >= N, "rows of the split tensors / the output row stride are too short");
    if (K % 16 != 0) return ia::fail(IA_ERR_UNSUPPORTED, "ia_matmul_sx needs K %% 16 == 0 (got %d)", K);
    IA_REQUIRE((int64_t)a_rows * K <= (int64_t)1 << 30 && (int64_t)b_rows * K <= (int64_t)1 << 30, "matrix too large for 32-bit plane offsets");
    LinParams p{static_cast<const h16x8*>(a_split), static_cast<const h16x8*>(b_split), nullptr, nullptr, y, M, N, K / 8, scale, 0,
                a_rows, b_rows, a_plane_bytes, b_plane_bytes, a_batch_bytes, b_batch_bytes, y_batch_stride, y_row_stride, 1};
    lin_launch(p, batch, (hipStream_t)stream);
    return ia::check_launch("ia_matmul_sx");
}

namespace {
// partial products of a K-split linear layer [S][M * N] -> y = sum over s (in order) + bias
__global__ __launch_bounds__(256) void linear_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ y,
                                                            int S, int64_t MN, int N) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= MN) return;
    float v = part[i];
    for (int s = 1; s < S; ++s) v += part[(int64_t)s * MN + i];
    y[i] = v + (bias ? bias[i % N] : 0.f);
}
}  // namespace

extern "C" int ia_linear_splitk_plan(int M, int K, int N, int* ksplit, size_t* scratch_bytes) {
    IA_REQUIRE(ksplit && scratch_bytes, "ksplit and scratch_bytes must be host pointers");
    IA_REQUIRE(M > 0 && K > 0 && N > 0, "empty matrix");
    // few output tiles and a long K (the deepest patch embedding: 64 tokens x 50 176 x 1 024): cut K until the launch has ~2 workgroups per CU,
    // every slice a whole number of k-steps and at least 64 of them
    const int64_t tiles = ia::ceil_div(M, 32) * ia::ceil_div(N, 64);
    int s = 1;
    if (K % 16 == 0 && tiles < ia::kNumCU)
        while (s < 64 && tiles * s < 2 * ia::kNumCU && (K / 16) % (2 * s) == 0 && K / 16 / (2 * s) >= 64) s *= 2;
    *ksplit = s;
    *scratch_bytes = s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
    return IA_OK;
}

extern "C" int ia_linear_sx_splitk(const void* xs, const void* w_split, int wk_exp, const float* bias, float* y, int M, int K, int N, int ksplit,
                                   float* scratch, size_t scratch_bytes, void* stream) {
    IA_REQUIRE(xs && w_split && y, "xs, w_split and y must be device pointers");
    IA_REQUIRE(M > 0 && K > 0 && N > 0 && ksplit >= 1 && ksplit <= 65535, "empty matrix or split count");
    if (K % (16 * ksplit) != 0) return ia::fail(IA_ERR_UNSUPPORTED, "ia_linear_sx_splitk needs K %% (16 * ksplit) == 0 (got %d, %d)", K, ksplit);
    IA_REQUIRE((int64_t)M * K <= (int64_t)1 << 30 && (int64_t)N * K <= (int64_t)1 << 30, "matrix too large for 32-bit plane offsets");
    if (ksplit == 1) return ia_linear_sx(xs, w_split, wk_exp, bias, nullptr, y, M, K, N, 0, stream);
    IA_REQUIRE(scratch && scratch_bytes >= (size_t)ksplit * M * N * sizeof(float), "scratch of ksplit * M * N floats (ia_linear_splitk_plan)");
    const int k8s = K / 8 / ksplit;
    LinParams p{static_cast<const h16x8*>(xs), static_cast<const h16x8*>(w_split), nullptr, nullptr, scratch, M, N, k8s, ldexpf(1.f, -wk_exp), 0,
                M, N, (int64_t)(K / 8) * M * 16, (int64_t)(K / 8) * N * 16, (int64_t)k8s * M * 16, (int64_t)k8s * N * 16, (int64_t)M * N, N, 0};
    const hipStream_t s = (hipStream_t)stream;
    lin_launch(p, ksplit, s);
    const int64_t mn = (int64_t)M * N;
    hipLaunchKernelGGL(linear_reduce_kernel, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, s, scratch, bias, y, ksplit, mn, N);
    return ia::check_launch("ia_linear_sx_splitk");
}

extern "C" int ia_linear_sx(const void* xs, const void* w_split, int wk_exp, const float* bias, const float* residual, float* y, int M, int K, int N,
                            int act, void* stream) {
    IA_REQUIRE(xs && w_split && y, "xs, w_split and y must be device pointers");
    IA_REQUIRE(M > 0 && K > 0 && N > 0, "empty matrix");
    IA_REQUIRE(act == 0 || act == 1, "act: 0 none, 1 GELU (erf)");
    if (K % 16 != 0) return ia::fail(IA_ERR_UNSUPPORTED, "ia_linear_sx needs K %% 16 == 0 (got %d)", K);
    IA_REQUIRE((int64_t)M * K <= (int64_t)1 << 30 && (int64_t)N * K <= (int64_t)1 << 30, "matrix too large for 32-bit plane offsets");
    LinParams p{static_cast<const h16x8*>(xs), static_cast<const h16x8*>(w_split), bias, residual, y, M, N, K / 8, ldexpf(1.f, -wk_exp), act,
                M, N, (int64_t)(K / 8) * M * 16, (int64_t)(K / 8) * N * 16, 0, 0, 0, N, 0};
    lin_launch(p, 1, (hipStream_t)stream);
    return ia::check_launch("ia_linear_sx");
}

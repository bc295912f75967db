// ia_conv2d_small: the 3x3 modulated convolutions of the 4^2 .. 16^2 blocks (and the 16^2 -> 32^2 up-sampling layer) of a
// StyleGAN2 synthesis network in ONE launch each.
//
// These layers are 0.08 - 1.2 GFLOP on 16 .. 289 points: far smaller than the machine, and 9.4 MB of weights that are read once.
// The tiled kernels (conv_mfma.hip) give them 128-channel tiles, spread the K loop of each tile over up to 128 workgroups
// (stream-K), park partial accumulators in slabs and need a second launch (conv_fixup_kernel) to add them: 2 launches and
// 20 - 55 us per layer, 18 fix-up launches per frame.  Here the work is cut the other way:
//
//   * a workgroup owns a 16 channel x 16 point tile of the output (v_mfma_f32_16x16x4_f32, 4 accumulator registers per lane), so a
//     512-channel layer is 32 .. 2432 workgroups and every CU streams its own slice of the weights;
//   * the K loop (taps x input channels) of the tile is split over the 8 waves of the workgroup, which add their accumulators
//     through LDS in wave order (deterministic) -- no slabs, no second launch;
//   * operands go straight from global memory into the MFMA's registers: A = packed weights [tap][I][O] (64-byte runs per
//     k), B = x * style gathered with the tap's offset (the whole activation is <= 512 KB and lives in L2); every wave keeps all
//     loads of a tap group in flight, so a tile costs about one memory round trip + 144 MFMAs per wave.
//
// fp32 operands, fp32 accumulation: the arithmetic of ia_conv2d_mfma.  The transposed (stride-2) form evaluates the four output
// phases as separate tiles (phase (py, px) owns the taps with ky % 2 == py, kx % 2 == px: 4, 2, 2 and 1 of them, the minimal FLOP
// count) and writes the (2H+1) x (2W+1) image that ia_upfirdn2d_bias_act / ia_fir_tail_split filter, exactly like ia_conv2d_mfma.
// Replaces, for these shapes, modulated_conv2d -> conv2d_resample -> conv2d / conv_transpose2d (+ bias_act) of the reference
// (training/networks_stylegan2.py:34-91, torch_utils/ops/conv2d_resample.py:114-136).
#include "ia_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 8;            // K split of a tile
constexpr int kTile = 16;            // channels and points per tile

struct SmallParams {
    const float* x;                  // [B][I][H][W]
    const float* wk;                 // [9][I][O]
    const float* styles;             // [B][I] or null
    const float* demod;              // [B][O] or null
    const float* noise;              // [OH*OW] or null
    const float* noise_strength;     // device scalar or null (=> 1)
    const float* bias;               // [O] or null
    float* y;                        // [B][O][OH][OW]
    int I, O, H, W, OH, OW;
    int tiles0, tiles1, tiles2;      // cumulative point-tile counts of phases 0, 0..1, 0..2 (transposed form)
    int act;
    float alpha, gain, clamp;
};

// taps of output phase ph = 2 * py + px of the stride-2 transposed convolution (ky % 2 == py, kx % 2 == px)
__device__ __forceinline__ int phase_taps(int ph, int (&taps)[4]) {
    switch (ph) {
        case 0: taps[0] = 0; taps[1] = 2; taps[2] = 6; taps[3] = 8; return 4;
        case 1: taps[0] = 1; taps[1] = 7; return 2;
        case 2: taps[0] = 3; taps[1] = 5; return 2;
        default: taps[0] = 4; return 1;
    }
}

template <bool TR>
__global__ __launch_bounds__(kWaves * 64) void conv_small_kernel(SmallParams p) {
    __shared__ float s_red[kWaves][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kq = lane >> 4;                  // B column (point) / A row (channel), and the k slot of the lane
    const int o0 = blockIdx.y * kTile, b = blockIdx.z;
    const int HW = p.H * p.W;

    // ---- which points: a tile of the row-major point grid of this launch (of this phase, transposed form)
    int ph = 0, tile = blockIdx.x;
    if (TR) {
        ph = tile >= p.tiles2 ? 3 : tile >= p.tiles1 ? 2 : tile >= p.tiles0 ? 1 : 0;
        tile -= ph == 3 ? p.tiles2 : ph == 2 ? p.tiles1 : ph == 1 ? p.tiles0 : 0;
    }
    const int py = ph >> 1, px = ph & 1;
    const int GH = TR ? p.H + 1 - py : p.H, GW = TR ? p.W + 1 - px : p.W;      // points of this phase with an output pixel
    const int pt = tile * kTile + n16;
    const bool pvalid = pt < GH * GW;
    const int pr = pvalid ? pt / GW : 0, pc = pvalid ? pt - pr * GW : 0;

    // ---- taps of the tile and the lane's gather offset for each (outside the image: masked to zero)
    int taps[9], ntaps;
    if (TR) {
        int t4[4];
        ntaps = phase_taps(ph, t4);
#pragma unroll
        for (int t = 0; t < 4; ++t) taps[t] = t4[t];
    } else {
        ntaps = 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) taps[t] = t;
    }
    int xoff[9];
    bool xok[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int tp = t < ntaps ? taps[t] : 0;
        const int ky = tp / 3, kx = tp - 3 * ky;
        const int iy = TR ? pr - (ky >> 1) : pr + ky - 1, ix = TR ? pc - (kx >> 1) : pc + kx - 1;
        xok[t] = t < ntaps && pvalid && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        xoff[t] = xok[t] ? iy * p.W + ix : 0;
    }

    // ---- this wave's share of the input channels; 4 of them (one per k slot) per MFMA
    const int per_wave = p.I / kWaves, c_begin = wave * per_wave;
    const float* xb = p.x + ((int64_t)b * p.I + c_begin + kq) * HW;
    const float* sb = p.styles ? p.styles + (int64_t)b * p.I + c_begin + kq : nullptr;
    const float* wb = p.wk + (int64_t)(c_begin + kq) * p.O + o0 + n16;
    const int64_t wtap = (int64_t)p.I * p.O;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // channel groups (of 4) whose loads are in flight together.  Measured (tools/bench_small.py): 8 or 16 groups (one round trip per
    // wave instead of four) spill the operand arrays and run 3-6x slower, and so does a 16-wave workgroup (128 registers per wave).
    constexpr int G = 4;
    constexpr int NT = TR ? 4 : 9;
    for (int c = 0; c < per_wave; c += 4 * G) {
        float av[G][NT], bv[G][NT], sv[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int cc = c + 4 * g;
            const bool live = cc < per_wave;
            sv[g] = (live && sb) ? sb[cc] : 1.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bool on = live && t < ntaps;
                av[g][t] = on ? wb[(int64_t)taps[t] * wtap + (int64_t)cc * p.O] : 0.f;
                bv[g][t] = (on && xok[t]) ? xb[(int64_t)cc * HW + xoff[t]] : 0.f;
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][t], bv[g][t] * sv[g], acc, 0, 0, 0);
    }

    // ---- add the waves' accumulators in wave order; one output element per thread (threads 0 .. 255)
#pragma unroll
    for (int r = 0; r < 4; ++r) s_red[wave][r][lane] = acc[r];
    __syncthreads();
    if (tid >= 256) return;
    const int r = tid >> 6, l = tid & 63;                       // accumulator register r of lane l: channel 4 * (l / 16) + r, point l % 16
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) v += s_red[w][r][l];
    const int o = o0 + 4 * (l >> 4) + r;
    const int q = tile * kTile + (l & 15);
    if (o >= p.O || q >= GH * GW) return;
    const int qr = q / GW, qc = q - qr * GW;
    const int oy = TR ? 2 * qr + py : qr, ox = TR ? 2 * qc + px : qc;
    const int64_t pix = (int64_t)oy * p.OW + ox;
    if (p.demod) v *= p.demod[b * p.O + o];
    if (p.noise) v = fmaf(p.noise[pix], p.noise_strength ? *p.noise_strength : 1.f, v);
    if (p.bias) v += p.bias[o];
    if (p.act == IA_ACT_LRELU) v = v > 0.f ? v : v * p.alpha;
    v *= p.gain;
    if (p.clamp >= 0.f) v = fminf(fmaxf(v, -p.clamp), p.clamp);
    p.y[((int64_t)b * p.O + o) * p.OH * p.OW + pix] = v;
}

}  // namespace

extern "C" int ia_conv2d_small_supported(int I, int O, int H, int W, int transposed) {
    // K split over 8 waves in groups of 4 channels; 16-channel tiles; the activation must be small enough to be gathered from L2
    return I % (4 * kWaves) == 0 && O % kTile == 0 && H > 0 && W > 0 && (int64_t)H * W <= (transposed ? 256 : 256);
}

extern "C" int ia_conv2d_small(const float* x, const float* wk, const float* styles, const float* demod, const float* noise,
                               const float* noise_strength, const float* bias, float* y, int B, int I, int O, int H, int W,
                               int transposed, int act, float alpha, float gain, float clamp, void* stream) {
    IA_REQUIRE(x && wk && y, "x, wk and y must be device pointers");
    IA_REQUIRE(B > 0 && I > 0 && O > 0 && H > 0 && W > 0, "empty tensor");
    IA_REQUIRE(act == IA_ACT_LINEAR || act == IA_ACT_LRELU, "conv epilogue supports linear and lrelu");
    IA_REQUIRE(!transposed || (noise == nullptr && bias == nullptr && act == IA_ACT_LINEAR),
               "the transposed form only applies the demodulation; FIR + bias_act follow in ia_upfirdn2d_bias_act");
    if (!ia_conv2d_small_supported(I, O, H, W, transposed))
        return ia::fail(IA_ERR_UNSUPPORTED, "ia_conv2d_small covers I %% 32 == 0, O %% 16 == 0 and H * W <= 256 (got I %d, O %d, %d x %d)", I, O, H, W);
    IA_REQUIRE(B <= 65535 && O / kTile <= 65535, "too many tiles for one launch");
    SmallParams p;
    p.x = x; p.wk = wk; p.styles = styles; p.demod = demod; p.noise = noise; p.noise_strength = noise_strength; p.bias = bias; p.y = y;
    p.I = I; p.O = O; p.H = H; p.W = W;
    p.OH = transposed ? 2 * H + 1 : H; p.OW = transposed ? 2 * W + 1 : W;
    p.act = act; p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    auto tiles_of = [](int64_t pts) { return (int)((pts + kTile - 1) / kTile); };
    int tiles;
    if (transposed) {
        p.tiles0 = tiles_of((int64_t)(H + 1) * (W + 1));
        p.tiles1 = p.tiles0 + tiles_of((int64_t)(H + 1) * W);
        p.tiles2 = p.tiles1 + tiles_of((int64_t)H * (W + 1));
        tiles = p.tiles2 + tiles_of((int64_t)H * W);
    } else {
        p.tiles0 = p.tiles1 = p.tiles2 = 0;
        tiles = tiles_of((int64_t)H * W);
    }
    const dim3 grid((unsigned)tiles, (unsigned)(O / kTile), (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    if (transposed) hipLaunchKernelGGL(conv_small_kernel<true>, grid, dim3(kWaves * 64), 0, s, p);
    else hipLaunchKernelGGL(conv_small_kernel<false>, grid, dim3(kWaves * 64), 0, s, p);
    return ia::check_launch("ia_conv2d_small");
}

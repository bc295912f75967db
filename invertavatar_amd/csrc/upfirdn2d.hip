// ia_upfirdn2d: zero-insert up-sampling -> pad/crop -> 2-D FIR -> decimation, per channel.
//
//   out[oy,ox] = gain * sum_{ky,kx} K[ky,kx] * U[oy*down + ky, ox*down + kx]
//   U[Y,X]     = x[(Y - pad0y)/upy, (X - pad0x)/upx]  when both divisions are exact and in range, else 0
//   K          = f flipped in both axes unless `flip` (true convolution by default)
//
// (reference op definition torch_utils/ops/upfirdn2d.py:169-213, size rule
// upfirdn2d.cpp:39-40.)  Two kernels:
//   * tiled:   contiguous NCHW, down 1, square up in {1,2}, filter <= 8x8 -- the two shapes the generator hits
//              (4x4 blur after the stride-2 transposed conv; 2x up-sampling of the skip image).  A 64x16 output
//              tile per 256-thread workgroup; the input halo tile and the pre-flipped, pre-gained filter are
//              staged in LDS; rows are read/written as full 256-byte wave accesses.  Only the non-zero
//              polyphase taps are visited.
//   * generic: any strides (channels_last), any up/down, filter <= 32x32.
// HBM-bound: algorithmic traffic = (in_h*in_w + out_h*out_w) * sizeof(T) bytes per channel.
#include "ia_common.h"
#include <type_traits>

namespace {

struct Geo {
    int n, c, in_h, in_w, out_h, out_w;
    int64_t xs[4], ys[4];
    int f_h, f_w;
    int upx, upy, downx, downy, padx0, pady0;
    float gain;
};

constexpr int kMaxTaps = 32 * 32;

// Optional fused tail of a SynthesisLayer (noise + bias + lrelu * gain + clamp), applied to the FIR result
// while it is still in registers: saves the separate bias_act pass over the activation tensor
// (training/networks_stylegan2.py:318-329).
struct Tail {
    const float* noise;           // [out_h*out_w] or null
    const float* noise_strength;  // device scalar or null (=1)
    const void* bias;             // [c] of T or null
    int act;                      // IA_ACT_LINEAR / IA_ACT_LRELU
    float alpha, gain, clamp;
};

__device__ __forceinline__ int floor_div(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

// Stage K[ky][kx] = gain * f[flip ? ky : fh-1-ky][flip ? kx : fw-1-kx] into LDS.
__device__ __forceinline__ void stage_filter(float* k_lds, const float* f, int f_h, int f_w, int64_t fs0, int64_t fs1,
                                             int flip, float gain) {
    for (int i = threadIdx.x; i < f_h * f_w; i += blockDim.x) {
        int ky = i / f_w, kx = i - ky * f_w;
        int sy = flip ? ky : f_h - 1 - ky, sx = flip ? kx : f_w - 1 - kx;
        k_lds[i] = f[sy * fs0 + sx * fs1] * gain;
    }
}

template <class T>
__global__ __launch_bounds__(256) void upfirdn2d_generic(const T* __restrict__ x, const float* __restrict__ f, T* __restrict__ y,
                                                        Geo g, int64_t fs0, int64_t fs1, int flip) {
    using S = typename ia::Num<T>::compute_t;
    __shared__ float k_lds[kMaxTaps];
    stage_filter(k_lds, f, g.f_h, g.f_w, fs0, fs1, flip, g.gain);
    __syncthreads();
    const int64_t total = (int64_t)g.n * g.c * g.out_h * g.out_w;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool chan_minor = g.ys[1] == 1;  // channels_last: walk channels fastest so stores stay coalesced
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int ox, oy, ch, nb;
        int64_t r = i;
        if (chan_minor) { ch = r % g.c; r /= g.c; ox = r % g.out_w; r /= g.out_w; oy = r % g.out_h; nb = r / g.out_h; }
        else { ox = r % g.out_w; r /= g.out_w; oy = r % g.out_h; r /= g.out_h; ch = r % g.c; nb = r / g.c; }
        const int by = oy * g.downy - g.pady0, bx = ox * g.downx - g.padx0;
        // first tap whose up-sampled coordinate lands on a real sample
        int ky0 = ((-by) % g.upy + g.upy) % g.upy, kx0 = ((-bx) % g.upx + g.upx) % g.upx;
        const T* xp = x + nb * g.xs[0] + ch * g.xs[1];
        S acc = 0;
        for (int ky = ky0; ky < g.f_h; ky += g.upy) {
            int iy = (by + ky) / g.upy;
            if (by + ky < 0 || iy >= g.in_h) continue;
            for (int kx = kx0; kx < g.f_w; kx += g.upx) {
                int ix = (bx + kx) / g.upx;
                if (bx + kx < 0 || ix >= g.in_w) continue;
                acc += ia::Num<T>::load(xp + iy * g.xs[2] + ix * g.xs[3]) * (S)k_lds[ky * g.f_w + kx];
            }
        }
        ia::Num<T>::store(y + nb * g.ys[0] + ch * g.ys[1] + oy * g.ys[2] + ox * g.ys[3], acc);
    }
}

// Tiled kernel.  TW x TH outputs per workgroup; thread t -> column t % TW, rows (t / TW) * RPT .. +RPT-1.
constexpr int TW = 64, TH = 16, RPT = 4;
static_assert(TW * (TH / RPT) == 256, "tile must map onto 256 threads");

template <class T, int UP, int FS, bool TAIL>
__global__ __launch_bounds__(256) void upfirdn2d_tiled(const T* __restrict__ x, const float* __restrict__ f, T* __restrict__ y,
                                                      Geo g, int64_t fs0, int64_t fs1, int flip, Tail tail) {
    constexpr int NT = (FS + UP - 1) / UP;               // non-zero taps per axis for one output phase
    constexpr int IH = (TH + FS - 2) / UP + 2;           // input rows a tile can touch
    constexpr int IW = (TW + FS - 2) / UP + 2;
    constexpr int IWP = IW + 1;                           // +1 column: rows start on different banks
    __shared__ float k_lds[FS * FS];
    __shared__ float in_lds[IH * IWP];

    const int tiles_x = (g.out_w + TW - 1) / TW;
    const int tile = blockIdx.x;
    const int ox0 = (tile % tiles_x) * TW, oy0 = (tile / tiles_x) * TH;
    const int64_t plane = blockIdx.y;                     // n * c planes, contiguous NCHW
    const T* xp = x + plane * (int64_t)g.in_h * g.in_w;
    T* yp = y + plane * (int64_t)g.out_h * g.out_w;

    const int iy0 = floor_div(oy0 - g.pady0, UP), ix0 = floor_div(ox0 - g.padx0, UP);
    stage_filter(k_lds, f, FS, FS, fs0, fs1, flip, g.gain);
    for (int i = threadIdx.x; i < IH * IW; i += 256) {
        int r = i / IW, cidx = i - r * IW;
        int iy = iy0 + r, ix = ix0 + cidx;
        float v = 0.f;
        if (iy >= 0 && iy < g.in_h && ix >= 0 && ix < g.in_w) v = (float)ia::Num<T>::load(xp + (int64_t)iy * g.in_w + ix);
        in_lds[r * IWP + cidx] = v;
    }
    __syncthreads();

    const int tx = threadIdx.x % TW, ty = (threadIdx.x / TW) * RPT;
    const int ox = ox0 + tx;
    if (ox >= g.out_w) return;
    float t_bias = 0.f, t_ns = 0.f;
    if (TAIL) {
        if (tail.bias) t_bias = (float)ia::Num<T>::load((const T*)tail.bias + (plane % g.c));
        if (tail.noise) t_ns = tail.noise_strength ? *tail.noise_strength : 1.f;
    }
    const int bx = ox - g.padx0;
    const int kx0 = ((-bx) % UP + UP) % UP;
    const int lx = floor_div(bx + kx0, UP) - ix0;         // LDS column of the first tap
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int oy = oy0 + ty + r;
        if (oy >= g.out_h) break;
        const int by = oy - g.pady0;
        const int ky0 = ((-by) % UP + UP) % UP;
        const int ly = floor_div(by + ky0, UP) - iy0;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const int ky = ky0 + a * UP;
            if (ky >= FS) break;
#pragma unroll
            for (int bb = 0; bb < NT; ++bb) {
                const int kx = kx0 + bb * UP;
                if (kx >= FS) break;
                acc = fmaf(in_lds[(ly + a) * IWP + lx + bb], k_lds[ky * FS + kx], acc);
            }
        }
        if (TAIL) {
            if (tail.noise) acc = fmaf(tail.noise[(int64_t)oy * g.out_w + ox], t_ns, acc);
            acc += t_bias;
            if (tail.act == IA_ACT_LRELU) acc = acc > 0.f ? acc : acc * tail.alpha;
            acc *= tail.gain;
            if (tail.clamp >= 0.f) acc = fminf(fmaxf(acc, -tail.clamp), tail.clamp);
        }
        ia::Num<T>::store(yp + (int64_t)oy * g.out_w + ox, acc);
    }
}

// The same tile and the same arithmetic for the 4x4 filter at up = 1 (the FIR that follows every stride-2 transposed convolution:
// the heavy case), with the global loads of the next tile in flight while the current one is filtered: a workgroup walks
// kPipeTiles tiles through two LDS images, one barrier per tile.  (In the one-tile kernel the load phase, the LDS reads and
// the stores of a workgroup never overlap; only other workgroups of the CU fill in.)
constexpr int kPipeTiles = 4;
template <class T, bool TAIL>
__global__ __launch_bounds__(256) void upfirdn2d_tiled_pipe(const T* __restrict__ x, const float* __restrict__ f, T* __restrict__ y,
                                                           Geo g, int64_t fs0, int64_t fs1, int flip, Tail tail) {
    constexpr int FS = 4;
    constexpr int IH = TH + FS, IW = TW + FS, IWP = IW + 1, NLD = (IH * IW + 255) / 256;
    __shared__ float k_lds[FS * FS];
    __shared__ float in_lds[2][IH * IWP];

    const int tiles_x = (g.out_w + TW - 1) / TW;
    const int64_t plane = blockIdx.y;
    const T* xp = x + plane * (int64_t)g.in_h * g.in_w;
    T* yp = y + plane * (int64_t)g.out_h * g.out_w;
    stage_filter(k_lds, f, FS, FS, fs0, fs1, flip, g.gain);

    // per-thread slots of the input image of a tile (fixed for all tiles): LDS offset and (row, column) inside the image
    int l_off[NLD], l_r[NLD], l_c[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int i = threadIdx.x + j * 256;
        l_r[j] = i / IW; l_c[j] = i - l_r[j] * IW;
        l_off[j] = i < IH * IW ? l_r[j] * IWP + l_c[j] : -1;
    }
    float v[NLD];
    auto fetch = [&](int tile) {
        const int ox0 = (tile % tiles_x) * TW, oy0 = (tile / tiles_x) * TH;
        const int iy0 = oy0 - g.pady0, ix0 = ox0 - g.padx0;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int iy = iy0 + l_r[j], ix = ix0 + l_c[j];
            const bool ok = l_off[j] >= 0 && iy >= 0 && iy < g.in_h && ix >= 0 && ix < g.in_w;
            v[j] = ok ? (float)ia::Num<T>::load(xp + (int64_t)iy * g.in_w + ix) : 0.f;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NLD; ++j)
            if (l_off[j] >= 0) in_lds[buf][l_off[j]] = v[j];
    };
    const int tx = threadIdx.x % TW, ty = (threadIdx.x / TW) * RPT;
    float t_bias = 0.f, t_ns = 0.f;
    if (TAIL) {
        if (tail.bias) t_bias = (float)ia::Num<T>::load((const T*)tail.bias + (plane % g.c));
        if (tail.noise) t_ns = tail.noise_strength ? *tail.noise_strength : 1.f;
    }
    const int tile0 = blockIdx.x * kPipeTiles;
    fetch(tile0);
    commit(0);
    __syncthreads();
    float kf[FS * FS];
#pragma unroll
    for (int i = 0; i < FS * FS; ++i) kf[i] = k_lds[i];
#pragma unroll 1
    for (int t = 0; t < kPipeTiles; ++t) {
        const int tile = tile0 + t, buf = t & 1;
        if (t + 1 < kPipeTiles) fetch(tile + 1);                 // in flight under the filter below
        const int ox0 = (tile % tiles_x) * TW, oy0 = (tile / tiles_x) * TH;
        const int ox = ox0 + tx;
        if (ox < g.out_w) {
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const int oy = oy0 + ty + r;
                if (oy >= g.out_h) break;
                float acc = 0.f;
#pragma unroll
                for (int a = 0; a < FS; ++a)
#pragma unroll
                    for (int bb = 0; bb < FS; ++bb) acc = fmaf(in_lds[buf][(ty + r + a) * IWP + tx + bb], kf[a * FS + bb], acc);
                if (TAIL) {
                    if (tail.noise) acc = fmaf(tail.noise[(int64_t)oy * g.out_w + ox], t_ns, acc);
                    acc += t_bias;
                    if (tail.act == IA_ACT_LRELU) acc = acc > 0.f ? acc : acc * tail.alpha;
                    acc *= tail.gain;
                    if (tail.clamp >= 0.f) acc = fminf(fmaxf(acc, -tail.clamp), tail.clamp);
                }
                ia::Num<T>::store(yp + (int64_t)oy * g.out_w + ox, acc);
            }
        }
        if (t + 1 < kPipeTiles) commit(buf ^ 1);                 // (its last readers passed the barrier of the previous tile)
        __syncthreads();
    }
}

// FIR + tail of an up-sampling SynthesisLayer with the result stored in SPLIT format for the next convolution
// (ia_conv2d_mfma_sx): the same sums, in the same order, as upfirdn2d_tiled_pipe<float, true>, then v * styles_next[b, c] split into
// fp16 hi / lo * 2^11 (ia::split_f16) and written as 16-byte units of 8 channels.  The structure is that of the pipelined kernel
// with the pipeline running over the EIGHT CHANNELS of a channel group instead of over tiles: a workgroup filters one 64 x 16
// tile of channel k from one LDS image while the loads of channel k + 1 are in flight, keeps the 4 results per channel in
// registers, and after the eighth channel every thread holds the 8 channels of its 4 pixels: 1 KB contiguous per wave, plane
// and row.  Optionally the fp32 NCHW result is written as well (callers that still need it, e.g. the CS-SFT modulation).
typedef _Float16 h16x8_t __attribute__((ext_vector_type(8)));
#ifndef IA_FIR_DMA_MIN_PIXELS
#define IA_FIR_DMA_MIN_PIXELS 0      // (tools/: 1 << 30 = the register-staged kernel everywhere)
#endif
#ifndef IA_FIR_ABLATE
#define IA_FIR_ABLATE 0      // tools/: bit 0 = one tap instead of 16 (LDS reads + FMAs), bit 1 = no global loads
#endif
#ifndef IA_FIR_SETS
#define IA_FIR_SETS 2
#endif
constexpr int kFirSets = IA_FIR_SETS;      // channels in flight per workgroup (register sets); measured r03 on the 256^2 / 512^2 layers: 2: 40.4 / 77.2 us, 3: 42.3 / 79.1, 4: 44.5 / 77.9
#ifndef IA_FIR_WAVES
#define IA_FIR_WAVES 1      // minimum waves per SIMD asked of the register allocator (tools/).  108 VGPRs = 4 waves per SIMD; forcing 5 / 6 / 8
                            // (96 / 80 / 64 VGPRs) measured 133 / 222 / 396 us against 73 on the 512^2 layer: the loads in flight need the registers
#endif
__global__ __launch_bounds__(256, IA_FIR_WAVES) void fir_tail_split_kernel(const float* __restrict__ x, const float* __restrict__ f, float* __restrict__ y,
                                                            h16x8_t* __restrict__ ys, const float* __restrict__ styles_next, Geo g, int flip, Tail tail, int planes) {
    constexpr int FS = 4;
    constexpr int IH = TH + FS, IW = TW + FS, IWP = IW + 1, NLD = (IH * IW + 255) / 256;
    __shared__ float k_lds[FS * FS];
    __shared__ float in_lds[2][IH * IWP];
    const int tiles_x = (g.out_w + TW - 1) / TW;
    const int ox0 = (blockIdx.x % tiles_x) * TW, oy0 = (blockIdx.x / tiles_x) * TH;
    const int C8 = g.c / 8, c8 = blockIdx.y % C8, b = blockIdx.y / C8;
    const int iy0 = oy0 - g.pady0, ix0 = ox0 - g.padx0;
    stage_filter(k_lds, f, FS, FS, FS, 1, flip, g.gain);
    const int64_t in_plane = (int64_t)g.in_h * g.in_w, ohw = (int64_t)g.out_h * g.out_w;
    const float* xb = x + ((int64_t)b * g.c + c8 * 8) * in_plane;
    // per-thread slots of the input image (the same for all 8 channels): LDS offset and global offset inside a channel plane
    int l_off[NLD], g_off[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int i = threadIdx.x + j * 256, r = i / IW, c = i - r * IW;
        const int iy = iy0 + r, ix = ix0 + c;
        const bool ok = i < IH * IW && iy >= 0 && iy < g.in_h && ix >= 0 && ix < g.in_w;
        l_off[j] = i < IH * IW ? r * IWP + c : -1;
        g_off[j] = ok ? iy * g.in_w + ix : -1;
    }
    // two channels in flight per workgroup (register sets v[0], v[1]): with one, a workgroup waits out a full memory round trip per
    // channel and eight workgroups per CU do not keep enough bytes in flight for the HBM stream
    float v[kFirSets][NLD];
    auto fetch = [&](int ch, int set) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) v[set][j] = (g_off[j] >= 0 && !(IA_FIR_ABLATE & 2)) ? xb[(int64_t)ch * in_plane + g_off[j]] : 0.f;
    };
    auto commit = [&](int buf, int set) {
#pragma unroll
        for (int j = 0; j < NLD; ++j)
            if (l_off[j] >= 0) in_lds[buf][l_off[j]] = v[set][j];
    };
    const int tx = threadIdx.x % TW, ty = (threadIdx.x / TW) * RPT;
    const int ox = ox0 + tx;
    const float t_ns = tail.noise ? (tail.noise_strength ? *tail.noise_strength : 1.f) : 0.f;
    // channel c travels in register set c % kFirSets; channels ch + 1 .. ch + kFirSets stay in flight under the filter of channel ch
#pragma unroll
    for (int c = 0; c < kFirSets; ++c) fetch(c, c);
    commit(0, 0);
    fetch(kFirSets, 0);
    __syncthreads();
    float kf[FS * FS];
#pragma unroll
    for (int i = 0; i < FS * FS; ++i) kf[i] = k_lds[i];
    float nz[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int oy = oy0 + ty + r;
        nz[r] = (tail.noise && ox < g.out_w && oy < g.out_h) ? tail.noise[(int64_t)oy * g.out_w + ox] : 0.f;
    }
    h16x8_t hi[RPT], lo[RPT];
    ia::SatWatch watch;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
        const int buf = ch & 1, c = c8 * 8 + ch;
        const float t_bias = tail.bias ? ((const float*)tail.bias)[c] : 0.f;
        const float sn = styles_next ? styles_next[b * g.c + c] : 1.f;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            float acc = 0.f;
#if IA_FIR_ABLATE & 1
            acc = in_lds[buf][(ty + r + 1) * IWP + tx + 1] * kf[5];
#else
#pragma unroll
            for (int a = 0; a < FS; ++a)
#pragma unroll
                for (int bb = 0; bb < FS; ++bb) acc = fmaf(in_lds[buf][(ty + r + a) * IWP + tx + bb], kf[a * FS + bb], acc);
#endif
            if (tail.noise) acc = fmaf(nz[r], t_ns, acc);
            acc += t_bias;
            if (tail.act == IA_ACT_LRELU) acc = acc > 0.f ? acc : acc * tail.alpha;
            acc *= tail.gain;
            if (tail.clamp >= 0.f) acc = fminf(fmaxf(acc, -tail.clamp), tail.clamp);
            const int oy = oy0 + ty + r;
            if (y && ox < g.out_w && oy < g.out_h) y[((int64_t)b * g.c + c) * ohw + (int64_t)oy * g.out_w + ox] = acc;
            const float t = styles_next ? acc * sn : acc;
            if (planes == 2) { _Float16 h, l; ia::split_f16(t, h, l, watch); hi[r][ch] = h; lo[r][ch] = l; }
            else hi[r][ch] = ia::round_f16(t, watch);
        }
        if (ch + 1 < 8) commit(buf ^ 1, (ch + 1) % kFirSets);     // (its last readers passed the barrier of the previous channel)
        if (ch + 1 + kFirSets < 8) fetch(ch + 1 + kFirSets, (ch + 1) % kFirSets);
        __syncthreads();
    }
    watch.report();
    if (ox >= g.out_w) return;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int oy = oy0 + ty + r;
        if (oy >= g.out_h) break;
        const int64_t pix = (int64_t)oy * g.out_w + ox;
        ys[((int64_t)(b * planes) * C8 + c8) * ohw + pix] = hi[r];
        if (planes == 2) ys[((int64_t)(b * 2 + 1) * C8 + c8) * ohw + pix] = lo[r];
    }
}

// The same kernel with the input images DMA'd straight into LDS (r05).  The register-staged form above was issue-bound (440
// instructions per channel and 4 outputs, a branch + atomic per split value) and kept only two channel images in flight.  Here the
// images arrive as `buffer_load_dword ... lds` pieces (64 lanes x 4 bytes, lane-linear in LDS; the rows of the (2H+1)-wide
// transposed-convolution output are only 4-byte aligned, so wider pieces do not apply) through a ring of LDS slots, three channels
// in flight, and channel c is filtered as soon as its 22 pieces have landed (loads return in order: a compile-time
// `s_waitcnt vmcnt(...)` per wave, then the barrier).  No staging registers, no ds_write.  Zero padding comes from the buffer bounds
// check (out-of-range lanes carry an offset beyond the resource and write zeros).  Every wave issues the same number of pieces per
// channel (22 real ones padded to 4 x 6 with all-outside pieces that land behind the image), so the wait counts are constants.
// Same sums in the same order as fir_tail_split_kernel: bit-identical results.
typedef unsigned int u32x4_f __attribute__((ext_vector_type(4)));
constexpr unsigned kFirOutside = 0x7ffffff0u;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void fir_dma_dword(u32x4_f rsrc, unsigned lds_addr, unsigned voffset, unsigned soffset) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory", "m0");
}
#pragma clang diagnostic pop
template <int N> __device__ __forceinline__ void fir_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// TH_ x 64 output tile (RPT_ = TH_ / 4 rows per thread); the images of a tile's eight channels travel through a ring of kFirRing LDS
// slots: channels 0 .. kFirRing - 2 are issued up front, channel c + kFirRing - 1 right after the barrier in front of channel c (every
// wave has left channel c - 1 by then, whose slot it takes).  24 KB of LDS and 128 registers: four workgroups per CU (the first form
// of this kernel kept all eight images resident: 49 KB, three per CU -- 1024 workgroups of a 128-channel 256^2 layer on 768 slots).
constexpr int kFirRing = 4;
template <int TH_, int RPT_>
__global__ __launch_bounds__(256) void fir_tail_split_dma_kernel(const float* __restrict__ x, const float* __restrict__ f, h16x8_t* __restrict__ ys,
                                                                 const float* __restrict__ styles_next, Geo g, int flip, Tail tail, int planes) {
    static_assert(TH_ == 4 * RPT_, "256 threads: 64 columns x 4 row groups");
    constexpr int FS = 4;
    constexpr int IH = TH_ + FS, IW = TW + FS, NPIX = IH * IW;            // (TH_ + 4) x 68 input pixels per channel image
    constexpr int NPIECE = (NPIX + 63) / 64, PPW = (NPIECE + 3) / 4;      // pieces of 64 pixels; per wave (padded with all-outside pieces)
    constexpr int IMG = 4 * PPW * 64;                                      // floats reserved per ring slot
    static_assert((kFirRing - 1) * PPW <= 63, "the pieces in flight of a wave must fit the vmcnt counter");
    extern __shared__ __attribute__((aligned(16))) float fir_lds[];       // [kFirRing][IMG] images, then the 16 filter taps
    float* k_lds = fir_lds + kFirRing * IMG;
    const int tiles_x = (g.out_w + TW - 1) / TW;
    const int ox0 = (blockIdx.x % tiles_x) * TW, oy0 = (blockIdx.x / tiles_x) * TH_;
    const int C8 = g.c / 8, c8 = blockIdx.y % C8, b = blockIdx.y / C8;
    const int iy0 = oy0 - g.pady0, ix0 = ox0 - g.padx0;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    stage_filter(k_lds, f, FS, FS, FS, 1, flip, g.gain);
    const int64_t in_plane = (int64_t)g.in_h * g.in_w, ohw = (int64_t)g.out_h * g.out_w;
    const float* xb = x + ((int64_t)b * g.c + c8 * 8) * in_plane;
    // everything the channel loop reads besides the images is fetched first: a vector load issued between the pieces and their
    // waits would shift the counts
    const int tx = threadIdx.x % TW, ty = (threadIdx.x / TW) * RPT_;
    const int ox = ox0 + tx;
    const float t_ns = tail.noise ? (tail.noise_strength ? *tail.noise_strength : 1.f) : 0.f;
    float nz[RPT_];
#pragma unroll
    for (int r = 0; r < RPT_; ++r) {
        const int oy = oy0 + ty + r;
        nz[r] = (tail.noise && ox < g.out_w && oy < g.out_h) ? tail.noise[(int64_t)oy * g.out_w + ox] : 0.f;
    }
    float t_bias[8], sn[8];
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
        t_bias[ch] = tail.bias ? ((const float*)tail.bias)[c8 * 8 + ch] : 0.f;
        sn[ch] = styles_next ? styles_next[b * g.c + c8 * 8 + ch] : 1.f;
    }
    // this wave's pieces: piece j = wave + 4 s covers image pixels 64 j .. 64 j + 63 (row-major over IH x IW)
    unsigned voff[PPW];
#pragma unroll
    for (int s_ = 0; s_ < PPW; ++s_) {
        const int i = 64 * (wave + 4 * s_) + lane, r = i / IW, c = i - r * IW;
        const int iy = iy0 + r, ix = ix0 + c;
        const bool ok = i < NPIX && iy >= 0 && iy < g.in_h && ix >= 0 && ix < g.in_w;
        voff[s_] = ok ? (unsigned)(iy * g.in_w + ix) * 4u : kFirOutside;
    }
    u32x4_f rsrc;
    {
        const unsigned long long a = (unsigned long long)xb;
        rsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
        rsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
        rsrc[2] = __builtin_amdgcn_readfirstlane((unsigned)(8 * in_plane * 4));
        rsrc[3] = 0x00020000u;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)fir_lds;
    // noise / bias / styles must have LANDED before the first piece is issued: from there on only pieces are counted.  The empty asm
    // statements are uses the compiler has to satisfy here (its own s_waitcnt vmcnt(0) for these loads would otherwise appear at
    // their first real use, inside channel 0, and drain the ring)
#pragma unroll
    for (int r = 0; r < RPT_; ++r) asm volatile("" : "+v"(nz[r]));
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) asm volatile("" : "+v"(t_bias[ch]), "+v"(sn[ch]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned plane_bytes = (unsigned)(in_plane * 4);
#define IA_FIR_ISSUE(ch)                                                                                                       \
    _Pragma("unroll") for (int s_ = 0; s_ < PPW; ++s_)                                                                         \
        fir_dma_dword(rsrc, lds0 + (unsigned)(((ch) % kFirRing) * IMG + 64 * (wave + 4 * s_)) * 4u, voff[s_], (unsigned)(ch) * plane_bytes);
    IA_FIR_ISSUE(0) IA_FIR_ISSUE(1) IA_FIR_ISSUE(2)
    __syncthreads();                                        // (k_lds)
    float kf[FS * FS];
#pragma unroll
    for (int i = 0; i < FS * FS; ++i) kf[i] = k_lds[i];
    h16x8_t hi[RPT_], lo[RPT_];
    ia::SatWatch watch;
    // the tail's options as values instead of branches: every element of the loop below is straight-line code (selects), so the
    // LDS reads and FMAs of its rows interleave; identical results (x * 1, fma(0, 0, x), clamp at infinity are exact)
    const float slope = tail.act == IA_ACT_LRELU ? tail.alpha : 1.f, clamp = tail.clamp >= 0.f ? tail.clamp : INFINITY;
    const bool two_planes = planes == 2;
    // one channel: wait for its pieces (this wave's, then everybody's), refill the slot the previous channel left, filter, tail,
    // split.  Channels younger than `ch` in flight at its wait: ch + 1, ch + 2 (ch + 3 is issued behind the barrier).  A macro with a
    // literal channel index on purpose: hi / lo must stay registers (through a lambda, next to the asm statements' memory clobbers,
    // they went to the stack)
#define IA_FIR_CH(ch)                                                                                                          \
    {                                                                                                                          \
        fir_wait_vmcnt<PPW * ((ch) + 2 <= 7 ? 2 : 7 - (ch))>();                                                                \
        __syncthreads();                                                                                                       \
        if ((ch) + kFirRing - 1 <= 7) { IA_FIR_ISSUE((ch) + kFirRing - 1) }                                                    \
        const float* img = fir_lds + ((ch) % kFirRing) * IMG;                                                                  \
        _Pragma("unroll") for (int r = 0; r < RPT_; ++r) {                                                                     \
            float acc = 0.f;                                                                                                   \
            _Pragma("unroll") for (int a = 0; a < FS; ++a)                                                                     \
                _Pragma("unroll") for (int bb = 0; bb < FS; ++bb)                                                              \
                    acc = fmaf(img[(ty + r + a) * IW + tx + bb], kf[a * FS + bb], acc);                                        \
            acc = fmaf(nz[r], t_ns, acc);                      /* (no noise: 0 * 0) */                                         \
            acc += t_bias[ch];                                                                                                 \
            acc = acc > 0.f ? acc : acc * slope;               /* (linear: slope 1) */                                         \
            acc *= tail.gain;                                                                                                  \
            { const float c = fminf(fmaxf(acc, -clamp), clamp); acc = (acc != acc) ? acc : c; }   /* (no clamp: +-inf; NaN stays NaN) */ \
            const float t = acc * sn[ch];                      /* (no styles: 1) */                                            \
            _Float16 h, l;                                                                                                     \
            ia::split_f16(t, h, l, watch);                                                                                     \
            const _Float16 h1 = (_Float16)fminf(fmaxf(t, -65504.f), 65504.f);      /* round_f16 without a second watch */      \
            hi[r][ch] = two_planes ? h : h1;                                                                                   \
            lo[r][ch] = l;                                                                                                     \
        }                                                                                                                      \
    }
    IA_FIR_CH(0) IA_FIR_CH(1) IA_FIR_CH(2) IA_FIR_CH(3) IA_FIR_CH(4) IA_FIR_CH(5) IA_FIR_CH(6) IA_FIR_CH(7)
#undef IA_FIR_CH
#undef IA_FIR_ISSUE
    watch.report();
    if (ox >= g.out_w) return;
#pragma unroll
    for (int r = 0; r < RPT_; ++r) {
        const int oy = oy0 + ty + r;
        if (oy >= g.out_h) break;
        const int64_t pix = (int64_t)oy * g.out_w + ox;
        ys[((int64_t)(b * planes) * C8 + c8) * ohw + pix] = hi[r];
        if (planes == 2) ys[((int64_t)(b * 2 + 1) * C8 + c8) * ohw + pix] = lo[r];
    }
}

template <int TH_, int RPT_>
int launch_fir_dma(const float* x, const float* f, void* ys, const float* styles_next, const Geo& g, int flip, const Tail& tail, int planes, hipStream_t s) {
    constexpr int IMG = 4 * ((((TH_ + 4) * (TW + 4) + 63) / 64 + 3) / 4) * 64;
    const size_t lds = (size_t)(kFirRing * IMG + 16) * sizeof(float);
    const auto kernel = fir_tail_split_dma_kernel<TH_, RPT_>;
    if (const int rs = ia::reserve_lds((const void*)kernel, lds, "ia_fir_tail_split")) return rs;
    const dim3 grid(((g.out_w + TW - 1) / TW) * ((g.out_h + TH_ - 1) / TH_), g.n * (g.c / 8));
    hipLaunchKernelGGL(kernel, grid, dim3(256), lds, s, x, f, static_cast<h16x8_t*>(ys), styles_next, g, flip, tail, planes);
    return ia::check_launch("ia_fir_tail_split");
}

template <class T>
bool tiled_eligible(const Geo& g) {
    const bool nchw = g.xs[3] == 1 && g.xs[2] == g.in_w && g.xs[1] == (int64_t)g.in_h * g.in_w &&
                      g.xs[0] == (int64_t)g.c * g.in_h * g.in_w && g.ys[3] == 1 && g.ys[2] == g.out_w &&
                      g.ys[1] == (int64_t)g.out_h * g.out_w && g.ys[0] == (int64_t)g.c * g.out_h * g.out_w;
    return nchw && g.downx == 1 && g.downy == 1 && g.upx == g.upy && g.f_h == g.f_w && (int64_t)g.n * g.c <= 65535 &&
           sizeof(T) <= 4 && (g.upx == 1 || g.upx == 2) && g.f_w == 4;
}

template <class T, bool TAIL>
int launch_tiled(const void* x, const float* f, void* y, const Geo& g, int64_t fs0, int64_t fs1, int flip, const Tail& tail,
                 hipStream_t s) {
    dim3 grid(((g.out_w + TW - 1) / TW) * ((g.out_h + TH - 1) / TH), g.n * g.c);
    if (g.upx == 1 && grid.x % kPipeTiles == 0 && grid.x >= 64) {
        hipLaunchKernelGGL((upfirdn2d_tiled_pipe<T, TAIL>), dim3(grid.x / kPipeTiles, grid.y), dim3(256), 0, s, (const T*)x, f, (T*)y, g, fs0, fs1,
                           flip, tail);
        return ia::check_launch(TAIL ? "ia_upfirdn2d_bias_act" : "ia_upfirdn2d(tiled)");
    }
    if (g.upx == 1) hipLaunchKernelGGL((upfirdn2d_tiled<T, 1, 4, TAIL>), grid, dim3(256), 0, s, (const T*)x, f, (T*)y, g, fs0, fs1, flip, tail);
    else hipLaunchKernelGGL((upfirdn2d_tiled<T, 2, 4, TAIL>), grid, dim3(256), 0, s, (const T*)x, f, (T*)y, g, fs0, fs1, flip, tail);
    return ia::check_launch(TAIL ? "ia_upfirdn2d_bias_act" : "ia_upfirdn2d(tiled)");
}

template <class T>
int launch(const void* x, const float* f, void* y, const Geo& g, int64_t fs0, int64_t fs1, int flip, hipStream_t s) {
    if (tiled_eligible<T>(g)) return launch_tiled<T, false>(x, f, y, g, fs0, fs1, flip, Tail{}, s);
    const int64_t total = (int64_t)g.n * g.c * g.out_h * g.out_w;
    hipLaunchKernelGGL((upfirdn2d_generic<T>), dim3(ia::streaming_grid(total, 256)), dim3(256), 0, s,
                       (const T*)x, f, (T*)y, g, fs0, fs1, flip);
    return ia::check_launch("ia_upfirdn2d(generic)");
}

}  // namespace

extern "C" int ia_upfirdn2d(const void* x, const float* f, void* y, int dtype,
                            int n, int c, int in_h, int in_w, const int64_t* h_x_stride,
                            int f_h, int f_w, const int64_t* h_f_stride,
                            int out_h, int out_w, const int64_t* h_y_stride,
                            int upx, int upy, int downx, int downy, int padx0, int pady0,
                            int flip, float gain, void* stream) {
    IA_REQUIRE(x && f && y && h_x_stride && h_f_stride && h_y_stride, "null pointer argument");
    IA_REQUIRE(n > 0 && c > 0 && in_h > 0 && in_w > 0, "x has zero size");
    IA_REQUIRE(f_h >= 1 && f_w >= 1, "f must be at least 1x1");
    IA_REQUIRE(upx >= 1 && upy >= 1, "upsampling factor must be at least 1");
    IA_REQUIRE(downx >= 1 && downy >= 1, "downsampling factor must be at least 1");
    IA_REQUIRE(out_h >= 1 && out_w >= 1, "output must be at least 1x1");
    IA_REQUIRE((int64_t)n * c * in_h * in_w <= INT32_MAX, "x is too large");
    IA_REQUIRE((int64_t)n * c * out_h * out_w <= INT32_MAX, "output is too large");
    if (f_h * f_w > kMaxTaps) return ia::fail(IA_ERR_UNSUPPORTED, "filter %dx%d exceeds the 32x32 tap limit", f_h, f_w);
    Geo g;
    g.n = n; g.c = c; g.in_h = in_h; g.in_w = in_w; g.out_h = out_h; g.out_w = out_w;
    for (int i = 0; i < 4; ++i) { g.xs[i] = h_x_stride[i]; g.ys[i] = h_y_stride[i]; }
    g.f_h = f_h; g.f_w = f_w; g.upx = upx; g.upy = upy; g.downx = downx; g.downy = downy;
    g.padx0 = padx0; g.pady0 = pady0; g.gain = gain;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case IA_F32: return launch<float>(x, f, y, g, h_f_stride[0], h_f_stride[1], flip, s);
        case IA_F16: return launch<__half>(x, f, y, g, h_f_stride[0], h_f_stride[1], flip, s);
        case IA_F64: return launch<double>(x, f, y, g, h_f_stride[0], h_f_stride[1], flip, s);
        default: return ia::fail(IA_ERR_INVALID_ARG, "unsupported dtype %d", dtype);
    }
}

extern "C" int ia_upfirdn2d_bias_act(const void* x, const float* f, const float* noise, const float* noise_strength,
                                     const void* bias, void* y, int dtype, int n, int c, int in_h, int in_w,
                                     int f_h, int f_w, int out_h, int out_w, int up, int padx0, int pady0,
                                     int flip, float fir_gain, int act, float alpha, float act_gain, float clamp, void* stream) {
    IA_REQUIRE(x && f && y, "null pointer argument");
    IA_REQUIRE(n > 0 && c > 0 && in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0, "empty tensor");
    IA_REQUIRE(act == IA_ACT_LINEAR || act == IA_ACT_LRELU, "fused tail supports linear and lrelu");
    IA_REQUIRE((int64_t)n * c * in_h * in_w <= INT32_MAX && (int64_t)n * c * out_h * out_w <= INT32_MAX, "tensor is too large");
    Geo g;
    g.n = n; g.c = c; g.in_h = in_h; g.in_w = in_w; g.out_h = out_h; g.out_w = out_w;
    g.xs[3] = 1; g.xs[2] = in_w; g.xs[1] = (int64_t)in_h * in_w; g.xs[0] = (int64_t)c * in_h * in_w;
    g.ys[3] = 1; g.ys[2] = out_w; g.ys[1] = (int64_t)out_h * out_w; g.ys[0] = (int64_t)c * out_h * out_w;
    g.f_h = f_h; g.f_w = f_w; g.upx = g.upy = up; g.downx = g.downy = 1; g.padx0 = padx0; g.pady0 = pady0; g.gain = fir_gain;
    Tail tail{noise, noise_strength, bias, act, alpha, act_gain, clamp};
    hipStream_t s = (hipStream_t)stream;
    if (dtype == IA_F32 && tiled_eligible<float>(g)) return launch_tiled<float, true>(x, f, y, g, f_w, 1, flip, tail, s);
    if (dtype == IA_F16 && tiled_eligible<__half>(g)) return launch_tiled<__half, true>(x, f, y, g, f_w, 1, flip, tail, s);
    return ia::fail(IA_ERR_UNSUPPORTED, "ia_upfirdn2d_bias_act: needs contiguous NCHW f32/f16, up in {1,2}, 4x4 filter");
}

extern "C" int ia_fir_tail_split(const float* x, const float* f, const float* noise, const float* noise_strength, const float* bias,
                                 const float* styles_next, float* y, void* ys, int ys_planes, int n, int c, int in_h, int in_w, int out_h, int out_w,
                                 int padx0, int pady0, int flip, float fir_gain, int act, float alpha, float act_gain, float clamp, void* stream) {
    IA_REQUIRE(x && f && ys, "null pointer argument");
    IA_REQUIRE(n > 0 && c > 0 && in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0, "empty tensor");
    IA_REQUIRE(c % 8 == 0, "the split format stores channels in groups of 8 (C = %d)", c);
    IA_REQUIRE(ys_planes == 1 || ys_planes == 2, "ys_planes must be 1 or 2");
    IA_REQUIRE(act == IA_ACT_LINEAR || act == IA_ACT_LRELU, "fused tail supports linear and lrelu");
    IA_REQUIRE((int64_t)n * c * in_h * in_w <= INT32_MAX && (int64_t)n * c * out_h * out_w <= INT32_MAX, "tensor is too large");
    IA_REQUIRE((int64_t)n * (c / 8) <= 65535, "too many (batch, channel group) planes for one launch");
    Geo g;
    g.n = n; g.c = c; g.in_h = in_h; g.in_w = in_w; g.out_h = out_h; g.out_w = out_w;
    g.f_h = g.f_w = 4; g.upx = g.upy = 1; g.downx = g.downy = 1; g.padx0 = padx0; g.pady0 = pady0; g.gain = fir_gain;
    Tail tail{noise, noise_strength, bias, act, alpha, act_gain, clamp};
    const dim3 grid(((out_w + TW - 1) / TW) * ((out_h + TH - 1) / TH), n * (c / 8));
    // LDS-DMA form: split output only (a fp32 copy's stores would share the wave's vmcnt with the pieces), 8 channel planes inside one
    // buffer resource, and enough tiles that the deeper prefetch matters (below 64^2 outputs a launch is latency either way)
    if (!y && 8 * (int64_t)in_h * in_w * 4 < (int64_t)kFirOutside && (int64_t)out_h * out_w >= IA_FIR_DMA_MIN_PIXELS) {
        // (32-row tiles -- 1.20x halo instead of 1.33x, one round for 128 channels @256^2 -- measured 66.2 / 21.1 us against 57.8 / 19.1
        //  for these on the 512^2 / 256^2 layers: 192 registers, two waves per SIMD)
        return launch_fir_dma<16, 4>(x, f, ys, styles_next, g, flip, tail, ys_planes, (hipStream_t)stream);
    }
    hipLaunchKernelGGL(fir_tail_split_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, f, y, static_cast<h16x8_t*>(ys), styles_next, g, flip, tail, ys_planes);
    return ia::check_launch("ia_fir_tail_split");
}

// ia_upconv2d_rows_sx: the stride-2 transposed 3x3 convolution of an up-sampling SynthesisLayer (conv2d_resample.py:114-131 ->
// conv_transpose2d, networks_stylegan2.py:296-305) as TWO families of stride-1 tiles -- one per output ROW phase -- on the
// 128-channel x 256-point, 8-wave tile of the stride-1 kernel (conv_split.hip), for split-format (fp16 hi / lo) activations.
//
// Why (the transposed members of the fp16-pair family run at 0.07 - 0.26 of the matrix pipe): the four-phase tile of
// ia_conv2d_mfma_sx keeps FOUR accumulator sets per wave, which caps it at one fragment per wave (64 ch x 64 pt, or 64 x 256 with 8
// waves): 15 MFMAs per 20 operand reads and per 23 KB of DMA -- bound by data movement -- and a k-step in five multiplies zeros.
// Here a tile belongs to ONE row phase py of the output:
//   py = 0 (taps ky in {0, 2}: six taps, rows 2r of the output):  8 input channels per chunk, k-steps (t0|t2) (t6|t8) -> px 0, (t1|t7) -> px 1
//   py = 1 (taps ky = 1:    three taps, rows 2r + 1):            16 input channels per chunk, k-steps (t3|t5)c0 (t3|t5)c1 -> px 0, (t4 c0|t4 c1) -> px 1
// i.e. three k-steps per chunk for both, none of them padded with zeros (the same 9 products per input pixel and channel pair as the
// four-phase form: the minimum), TWO accumulator sets (the column phases, stored as 8-byte pairs of adjacent output pixels), two
// fragments x two fragments per wave: 12 MFMAs per 8 operand reads, the ratio of the stride-1 kernel.  Both phases read the SAME packed
// weights (pack_conv_weight_split: [plane][tap][I/8][O][8]) and the same split activations as ia_conv2d_mfma_sx; the DMA plan of a
// tile picks its tap rows.  The output is the (2H+1) x (2W+1) fp32 image ia_conv2d_mfma_sx(transposed) writes (demodulated), read by
// ia_fir_tail_split / ia_upfirdn2d_bias_act.
//
// Tiling.  The (H+1) x (W+1) point grid of a transposed convolution is one point wider than the image, so its row-major tiles
// straddle rows and 257^2 points leave a 1-point tile.  Here the grid is cut into the H x W interior of each row phase (aligned
// tiles: one or two whole image rows each -- for W = 256 exactly one round of 256 long and 256 short tiles), and three thin EDGE
// grids (bottom row r = H of py 0, last column c = W of either phase), whose few tiles are split `gpe`-way along K between extra
// workgroups of the same launch (partial sums to caller-owned slabs, summed in worker order by up_edge_fixup_kernel: deterministic).
// A py 1 tile has half the k-steps of a py 0 tile; layers whose tiles fit one round give a workgroup two py 1 tiles (`reps`).
#include "conv_common.h"
#include "lds_dma.h"

// Compile-time ablations for tools/ablate_conv_up.sh (never set in the product build): 1 = no DMA after a tile's first chunks,
// 2 = no MFMAs (operand reads kept alive), 3 = no output stores, 4 = no per-chunk DMA wait (races: timing only), 5 = every chunk DMAs
// the weights of chunk 0 (cache-hot), 6 = every chunk DMAs the patch of chunk 0.
#ifndef IA_UP_ABLATE
#define IA_UP_ABLATE 0
#endif

// IA_UP_TRACE (tools/trace_conv_up.sh): workgroup 0 stamps s_memtime at the segment boundaries of its first k-steps into `slabs`.
#ifndef IA_UP_TRACE
#define IA_UP_TRACE 0
#endif
#ifndef IA_UP_CHUNK_INTERVALS
#define IA_UP_CHUNK_INTERVALS 1      // 0: a k-step per barrier interval for every tile (the r04 / r05 schedule; A/B builds)
#endif
#if IA_UP_TRACE
#define IA_STAMP(slot) do { if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && tr_n < 64) \
    reinterpret_cast<unsigned long long*>(slabs)[(wave * 64 + tr_n) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define IA_STAMP(slot) do { } while (0)
#endif

namespace {

constexpr size_t kLdsBytesUp = 160 * 1024;
constexpr int kMaxSub = 5;
constexpr int kUpMaxStages = 4;

struct UpSub {
    int first_wg, n_wg;      // workgroups [first_wg, first_wg + n_wg) of a batch element serve this grid
    int n_pt;                // point tiles (x TO channel tiles = tiles)
    int tp;                  // points per point tile (the tile's capacity BP, or fewer for the one-column grids: their window is 2 wide)
    int reps;                // whole-tile grids: consecutive tiles per workgroup
    int gpe;                 // edge grids: workers per tile (0 = whole tiles)
    int edge_first;          // K-split grids: index of the grid's first tile among all K-split tiles (the fix-up's grid)
    int slab_first;          // K-split grids: index of the grid's first slab (tile lt, part j -> slab_first + lt * gpe + j)
    int GH, GW;              // points
    int r_off, c_off;        // point (r, c) of the grid is point (r_off + r, c_off + c) of the layer
    int py;                  // output row phase
};

struct UpGeo {
    int B, I, O, H, W, OH, OW, TO;
    int nsub;
    UpSub sub[kMaxSub];
    int E;                   // K-split tiles in total (edge grids, and the long interior tiles of K-deep small layers)
    int n_slabs;             // slabs per batch element
    int cap;                 // patch positions per plane in LDS (all channel groups), multiple of 128
    int stages;              // LDS stages of the DMA ring (chunks ch+1 .. ch+stages-1 are in flight under chunk ch)
    int pair;                // 1: a workgroup of grid 0 (py 0) goes on to the tile of the same index of grid 1 (py 1)
    float acc_scale;
};

struct UWin { int r0[2], nr[2], c0[2], PW, PSZ; };

// s_waitcnt vmcnt(k * N) for a wave-uniform k in 0 .. 3 (the chunks of DMA that may stay in flight) and a compile-time N <= 15
template <int N>
__device__ __forceinline__ void wait_chunks_in_flight(int k) {
    static_assert(3 * N <= 63, "vmcnt is a 6-bit counter");
    if (k <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (k == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    else if (k == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * N) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * N) : "memory");
}

// s_barrier that the compiler keeps memory operations on their side of
__device__ __forceinline__ void ia_barrier() { asm volatile("s_barrier" ::: "memory"); }

// Input window of the points [p0, p_last] of a GW-wide grid: `up` + 1 rows (r - up .. r) and columns c - 1 .. c per point; one or
// two row segments sharing a row stride (see tile_window in conv_common.h, of which this is the transposed case with a variable `up`).
__host__ __device__ inline UWin up_window(int p0, int p_last, int GW, int up) {
    const int r_first = p0 / GW, r_last = p_last / GW;
    const int c_first = p0 - r_first * GW, c_last = p_last - r_last * GW;
    const int nrows = up + 1;
    UWin w;
    w.nr[1] = 0; w.r0[1] = 0; w.c0[1] = 0;
    if (r_first == r_last) {
        w.r0[0] = r_first - up; w.nr[0] = nrows; w.c0[0] = c_first - 1; w.PW = c_last - w.c0[0] + 1;
    } else {
        const int w0 = (GW - 1) - (c_first - 1) + 1, w1 = c_last + 2;
        const int pw_split = w0 > w1 ? w0 : w1, pw_full = GW + 1;
        const int sz_split = 2 * nrows * pw_split, sz_full = (r_last - r_first + nrows) * pw_full;
        if (r_last == r_first + 1 && sz_split < sz_full) {
            w.r0[0] = r_first - up; w.nr[0] = nrows; w.c0[0] = c_first - 1;
            w.r0[1] = r_last - up;  w.nr[1] = nrows; w.c0[1] = -1;
            w.PW = pw_split;
        } else {
            w.r0[0] = r_first - up; w.nr[0] = r_last - r_first + nrows; w.c0[0] = -1; w.PW = pw_full;
        }
    }
    w.PSZ = (w.nr[0] + w.nr[1]) * w.PW;
    return w;
}

// unit u of a chunk (six (tap, channel group) rows of weights per plane): tap and channel group by row phase
__host__ __device__ inline int unit_tap(int py, int u) { return py ? 3 + u % 3 : (u < 3 ? u : u + 3); }
__host__ __device__ inline int unit_cg(int py, int u) { return py ? u / 3 : 0; }

// Accumulators (two column phases x FO x FP fragments) -> the (2H+1) x (2W+1) image: the two column phases of a point are adjacent
// pixels of output row 2r + py, stored as one 8-byte pair; the demodulation coefficients of a lane's 32 channels are read once.
template <int FP>
__device__ __forceinline__ void up_store_tile(const f32x16 (&acc)[2][2][FP], float* __restrict__ y, const float* __restrict__ demod, const UpGeo& g,
                                              const UpSub& sb, int b, int o0, int p0, int p_last, int tid, float acc_scale) {
    constexpr int FO = 2, WP = 4;
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int wo = wave / WP, wp = wave % WP;
    const int64_t ohw = (int64_t)g.OH * g.OW;
    // demodulation x the power of two that takes the accumulators back from the scale of the packed weights (exact: one multiply per
    // value instead of two)
    float dmv[FO][16];
#pragma unroll
    for (int fo = 0; fo < FO; ++fo)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + (wo * FO + fo) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            dmv[fo][r] = (demod ? demod[b * g.O + o] : 1.f) * acc_scale;
        }
    float* yb = y + ((int64_t)b * g.O + o0 + wo * FO * 32 + 4 * half) * ohw;
#pragma unroll
    for (int fp = 0; fp < FP; ++fp) {
        const int p = p0 + (wp * FP + fp) * 32 + l31;
        if (p > p_last) continue;
        const int pr = p / sb.GW, pc = p - pr * sb.GW;
        const int oy = 2 * (sb.r_off + pr) + sb.py, ox = 2 * (sb.c_off + pc);
        float* dst0 = yb + (int64_t)oy * g.OW + ox;
        // the last column has no right neighbour: that test sits around the element loop, not inside it (a divergent branch per value)
        if (ox + 1 < g.OW) {
#pragma unroll
            for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float2 v = make_float2(acc[0][fo][fp][r] * dmv[fo][r], acc[1][fo][fp][r] * dmv[fo][r]);
                    __builtin_memcpy(dst0 + (int64_t)(fo * 32 + (r & 3) + 8 * (r >> 2)) * ohw, &v, 8);     // (rows of odd width: 4-byte aligned only)
                }
        } else {
#pragma unroll
            for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                for (int r = 0; r < 16; ++r) dst0[(int64_t)(fo * 32 + (r & 3) + 8 * (r >> 2)) * ohw] = acc[0][fo][fp][r] * dmv[fo][r];
        }
    }
}

template <int FP, int JP>
__global__ __launch_bounds__(512, 2) void up_rows_kernel(const h16x8* __restrict__ xs, const h16x8* __restrict__ wk, float* __restrict__ y,
                                                         float* __restrict__ slabs, const float* __restrict__ demod, UpGeo g) {
    constexpr int FO = 2, WO = 2, WP = 4, NP = 2, NU = 6;
    constexpr int BO = 32 * FO * WO, BP = 32 * FP * WP, NWAVES = WO * WP, NTHREADS = NWAVES * 64;
    constexpr int WSLOTS = NP * NU * BO;               // 16-byte slots of the weight region of a stage: [plane][unit][BO]
    constexpr int WGI = WSLOTS / 64, JW = WGI / NWAVES; // weight DMA instructions per chunk / per wave
    static_assert(WGI % NWAVES == 0, "weight pieces divide evenly over the waves");
    constexpr int NACC = 2 * FO * FP * 16;
    constexpr bool CI = IA_UP_CHUNK_INTERVALS && FP == 1;      // whole-chunk barrier intervals (see the K loop)
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wo = wave / WP, wp = wave % WP;
    const int b = blockIdx.y;
    const int wg = blockIdx.x;
    UpSub sb = g.sub[0];       // (selected with a constant-index chain: a run-time index would move the table to scratch memory)
#pragma unroll
    for (int k = 1; k < kMaxSub; ++k)
        if (k < g.nsub && wg >= g.sub[k].first_wg) sb = g.sub[k];
    int wl = wg - sb.first_wg;
    if (!sb.gpe && g.B == 1 && (sb.first_wg & 7) == 0) {
        // whole-tile grids: workgroups go to the 8 XCDs round-robin; give XCD x a contiguous band of the grid's tiles instead of every
        // eighth one, so that the input rows two neighbouring tiles share (py 0 reads rows r - 1 and r) are fetched into ONE L2
        const int per = sb.n_wg / ia::kNumXCD, rem = sb.n_wg - per * ia::kNumXCD;
        const int x = wl % ia::kNumXCD, j = wl / ia::kNumXCD;
        wl = x * per + (x < rem ? x : rem) + j;
    }
    const int grp = wave >> 2;                                    // wave group: 1 runs one barrier interval behind 0 (see the K loop)
    const int cap = g.cap, NS = g.stages;
    const int HW = g.H * g.W;
    const int plane_bytes = (g.I / 8) * HW * 16;
    const int PG = NP * cap / 64;                                 // patch DMA instructions per chunk
    const int stage_bytes = (WSLOTS + NP * cap) * 16;
    const u32x4 rs_x = buffer_rsrc(xs + (int64_t)b * NP * (g.I / 8) * HW, (unsigned)(NP * plane_bytes));
    const u32x4 rs_w = buffer_rsrc(wk, (unsigned)(NP * 9 * (g.I / 8) * g.O * 16));
    const unsigned lds_base = (unsigned)(unsigned long long)(lds_char*)lds;

 for (int job = 0; job < 1 + g.pair; ++job) {
    if (job == 1) {               // the short tile of the same rows (grid 1), while this tile's stores drain
        if (sb.gpe || sb.py) break;
        sb = g.sub[1];
    }
    const int py = sb.py, CPC = 1 + py, C = g.I / (8 * CPC);      // channel groups per chunk, chunks per tile
    const int capg = py ? cap / 2 : cap;                          // patch positions per channel group
    int lt = sb.gpe ? wl / sb.gpe : wl * sb.reps;
    const int lt_end = sb.gpe ? lt + 1 : min(lt + sb.reps, sb.n_pt * g.TO);
    const int part = sb.gpe ? wl - lt * sb.gpe : 0;
    const int c_lo = sb.gpe ? (part * C) / sb.gpe : 0, c_hi = sb.gpe ? ((part + 1) * C) / sb.gpe : C;

    // the two operands of k-step s this lane reads: unit (0|2) (3|5) (1|4) by lane half
    int arow[3], ucg[3], udy[3], udx[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int u = half ? (s == 0 ? 2 : s == 1 ? 5 : 4) : (s == 0 ? 0 : s == 1 ? 3 : 1);
        arow[s] = u * BO;
        ucg[s] = unit_cg(py, u);
        udy[s] = py ? 0 : (u >= 3 ? 1 : 0);
        udx[s] = (u % 3 == 2) ? 1 : 0;
    }

  for (; lt < lt_end; ++lt) {
    const int o0 = (lt % g.TO) * BO;
    const int p0 = (lt / g.TO) * sb.tp;
    const int p_last = min(p0 + sb.tp, sb.GH * sb.GW) - 1;
    const UWin win = up_window(p0, p_last, sb.GW, 1 - py);
    const int PW = win.PW, PSZ = win.PSZ, seg1_off = win.nr[0] * PW;
    const int r_split = (win.nr[1] > 0) ? p_last / sb.GW : (1 << 30);
    const float inv_pw = 1.0f / (float)PW;

    int bpos[FP];
#pragma unroll
    for (int fp = 0; fp < FP; ++fp) {
        const int p = min(p0 + (wp * FP + fp) * 32 + l31, p_last);
        const int r = p / sb.GW, c = p - r * sb.GW;
        const int sg = (r == r_split) ? 1 : 0;
        bpos[fp] = sg * seg1_off + (r - win.r0[sg]) * PW + (c - win.c0[sg]);
    }
    int boff[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) boff[s] = ucg[s] * capg - (udy[s] * PW + udx[s]);

    f32x16 acc[2][FO][FP];
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int fo = 0; fo < FO; ++fo)
#pragma unroll
            for (int fp = 0; fp < FP; ++fp)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[px][fo][fp][r] = 0.f;

    // ---- DMA plan of this tile: per-lane source byte offsets of chunk 0; a chunk adds an SGPR offset
    constexpr int kOutside = 0x7ffffff0;
    int w_voff[JW], p_voff[JP];
#pragma unroll
    for (int j = 0; j < JW; ++j) {
        const int e_ = (j * NWAVES + wave) * 64 + lane;
        const int row = e_ / BO, o = e_ - row * BO;                 // row = plane * NU + unit
        const int pl = row / NU, u = row - pl * NU;
        w_voff[j] = (((pl * 9 + unit_tap(py, u)) * (g.I / 8) + unit_cg(py, u)) * g.O + o0 + o) * 16;
    }
#pragma unroll
    for (int j = 0; j < JP; ++j) {
        const int q = (j * NWAVES + wave) * 64 + lane;
        const int pl = q >= cap ? 1 : 0, pp = q - pl * cap;
        const int cg = pp >= capg ? 1 : 0, pos = pp - cg * capg;
        const int sg = (pos >= seg1_off && win.nr[1] > 0) ? 1 : 0;
        const int qq = pos - sg * seg1_off;
        const int pr = (int)(((float)qq + 0.5f) * inv_pw), pc = qq - pr * PW;
        const int iy = sb.r_off + win.r0[sg] + pr, ix = sb.c_off + win.c0[sg] + pc;
        const bool ok = j * NWAVES + wave < PG && pos < PSZ && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
        p_voff[j] = ok ? pl * plane_bytes + (cg * HW + iy * g.W + ix) * 16 : kOutside;
    }
// pieces of one chunk into one LDS stage: slice `sl` of `nsl` of this wave's JW + JP pieces (weights first, then patch; piece q belongs to
// slice q * nsl / (JW + JP)); nsl = 1 issues the whole chunk
#define IA_UP_ISSUE(chunk, stage, sl, nsl)                                                                                            \
    do {                                                                                                                               \
        const unsigned st_ = __builtin_amdgcn_readfirstlane(lds_base + (stage) * stage_bytes);                                         \
        const int wso_ = (IA_UP_ABLATE == 5 || IA_UP_ABLATE == 7) ? 0 : __builtin_amdgcn_readfirstlane((chunk) * CPC * g.O * 16);                             \
        const int pso_ = (IA_UP_ABLATE == 6 || IA_UP_ABLATE == 7) ? 0 : __builtin_amdgcn_readfirstlane((chunk) * CPC * HW * 16);                              \
        _Pragma("unroll") for (int j = 0; j < JW; ++j) {                                                                               \
            if (j * (nsl) / (JW + JP) == (sl)) {                                                                                       \
                const int vo_ = w_voff[j];                                                                                             \
                dma_piece(rs_w, st_ + (j * NWAVES + wave) * 64 * 16, vo_, wso_);                                                       \
            }                                                                                                                          \
        }                                                                                                                              \
        _Pragma("unroll") for (int j = 0; j < JP; ++j) {                                                                               \
            if ((JW + j) * (nsl) / (JW + JP) == (sl)) {                                                                                \
                const int gidx = j * NWAVES + wave;                                                                                    \
                const int vo_ = p_voff[j];                                                                                             \
                dma_piece(rs_x, gidx < PG ? st_ + (WSLOTS + gidx * 64) * 16 : dummy_lds, vo_, pso_);                                   \
            }                                                                                                                          \
        }                                                                                                                              \
    } while (0)
    // (a wave whose share of the patch pieces is short issues all-outside pieces into a dummy 1 KB region behind the ring instead: the
    //  same JW + JP instructions per chunk for every wave make the counted waits compile-time multiples -- a general s_waitcnt vmcnt(n)
    //  for a run-time n is a 64-way compare chain, measured at ~450 cycles per chunk in the LOAD segment that carries it)
    constexpr int n_dma = JW + JP;
    const unsigned dummy_lds = lds_base + NS * stage_bytes;
    // The previous tile's stores are NOT drained here: its epilogue reads no LDS and consumes its (demodulation) loads before the first
    // store, so the compiler has nothing pending that it would wait for inside the K loop; the DMA of this tile's first chunks goes out
    // behind the stores, and the counted waits below are still exact -- loads (LDS-DMA included) return in order among themselves, so
    // "at most k operations outstanding" with k younger DMAs issued implies the older chunk has landed, whatever the stores do.
    __syncthreads();
    int issued = c_lo;
    for (int k = 0; k < NS - 1 && issued < c_hi; ++k, ++issued) IA_UP_ISSUE(issued, k, 0, 1);
    wait_chunks_in_flight<n_dma>(issued - c_lo - 1);      // chunk c_lo: this wave's pieces have landed ...
    ia_barrier();                                         // ... and everybody's
    // ---- K loop, two wave groups in antiphase.  Waves w and w + 4 share a SIMD; group 1 (waves 4-7) runs ONE barrier interval behind
    // group 0, so that in every interval one wave of a SIMD is in its LOAD segment (operand reads of a k-step, a share of the DMA issue)
    // while the other is in its COMPUTE segment (the k-step's 12 MFMAs): the matrix pipe is fed by one of them at any time.  With all
    // eight waves in step (the structure of conv_split_kernel) both waves of a SIMD issue DMA / wait for their reads together and then
    // queue for the pipe together -- ablation (tools/ablate_conv_up.sh, 256 -> 128 @256^2): 152 us with, 96 us without the MFMAs, i.e.
    // nothing but the MFMAs' own 57 us was overlapped.  Every wave executes the same number of barriers (group 0 one more at the end).
    //   hazards: a stage is refilled (chunk ch + NS - 1 into the stage of chunk ch - 1) during the LOAD segments of chunk ch, a third of the
    //   pieces per k-step -- every wave waits for its operand reads (lgkmcnt(0)) in front of the barrier that ends a LOAD segment, and the
    //   leading group's first refill piece is issued behind the barrier at which the lagging group ended its last LOAD segment of chunk ch - 1; the wait for chunk ch + 1 sits in the LOAD
    //   segment of k-step 2, in front of the barrier after which the leading group starts reading it.
    if (grp) ia_barrier();
    int cur = 0;
#if IA_UP_TRACE
    int tr_n = 0;
#endif
    for (int ch = c_lo; ch < c_hi; ++ch) {
        const bool fill = IA_UP_ABLATE != 1 && issued < c_hi;
        const int fill_chunk = issued, fill_stage = cur == 0 ? NS - 1 : cur - 1;
        const h16x8* wh = reinterpret_cast<const h16x8*>(reinterpret_cast<const char*>(lds) + cur * stage_bytes);
        const h16x8* ph = wh + WSLOTS;
        if constexpr (CI) {
            // Whole-chunk intervals (r06, the 128-point tiles of the backbones' up-sampling layers): ONE LOAD segment reads the operands
            // of all three k-steps of the chunk and issues the wave's refill pieces, ONE COMPUTE segment runs their 18 MFMAs -- two
            // barriers per chunk instead of six.  With a k-step per interval these tiles spent 1 380 cycles on 6 MFMAs (192 cycles of the
            // pipe): a 360 - 500-cycle LOAD segment and ~200 cycles of barrier per interval whatever the segment holds (r05 ablation:
            // 58 us with, 46 us without the MFMAs).  Hazards as before, per chunk: the stage refilled here (chunk ch - 1's) was last read
            // by the lagging group in ITS load segment of chunk ch - 1, which ended at the barrier in front of this segment; every wave
            // waits for its own pieces of chunk ch + 1 in front of the barrier that ends its load segment of chunk ch, and the leading
            // group starts reading chunk ch + 1 two barriers later.
            h16x8 a_all[3][NP * FO], b_all[3][NP * FP];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                    for (int fo = 0; fo < FO; ++fo) a_all[s][pl * FO + fo] = wh[pl * NU * BO + arow[s] + (wo * FO + fo) * 32 + l31];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                    for (int fp = 0; fp < FP; ++fp) b_all[s][pl * FP + fp] = ph[pl * cap + bpos[fp] + boff[s]];
            }
            // (the wait goes IN FRONT of this chunk's pieces, see below: chunks ch + 2 .. issued - 1 may stay in flight)
            if (ch + 1 < c_hi && IA_UP_ABLATE != 4) wait_chunks_in_flight<n_dma>(issued - ch - 2);
            if (fill) { IA_UP_ISSUE(fill_chunk, fill_stage, 0, 1); ++issued; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            ia_barrier();
            __builtin_amdgcn_sched_barrier(0);
#if IA_UP_ABLATE == 2
#pragma unroll
            for (int s = 0; s < 3; ++s) {
#pragma unroll
                for (int q = 0; q < NP * FO; ++q) asm volatile("" ::"v"(a_all[s][q]));
#pragma unroll
                for (int q = 0; q < NP * FP; ++q) asm volatile("" ::"v"(b_all[s][q]));
            }
#else
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                constexpr int kPxC[3] = {0, 0, 1};
                const int px = kPxC[s];
#pragma unroll
                for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                    for (int fp = 0; fp < FP; ++fp)      // lo * hi
                        acc[px][fo][fp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_all[s][FO + fo], b_all[s][fp], acc[px][fo][fp], 0, 0, 0);
                h16x8 a_sc[FO];
#pragma unroll
                for (int fo = 0; fo < FO; ++fo) a_sc[fo] = a_all[s][fo] * (_Float16)(1.0f / kLoScale);
#pragma unroll
                for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                    for (int fp = 0; fp < FP; ++fp)      // (hi * 2^-11) * (lo * 2^11)
                        acc[px][fo][fp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_sc[fo], b_all[s][FP + fp], acc[px][fo][fp], 0, 0, 0);
#pragma unroll
                for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                    for (int fp = 0; fp < FP; ++fp)      // hi * hi
                        acc[px][fo][fp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_all[s][fo], b_all[s][fp], acc[px][fo][fp], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
#endif
            __builtin_amdgcn_sched_barrier(0);
            ia_barrier();
            __builtin_amdgcn_sched_barrier(0);
            cur = cur + 1 == NS ? 0 : cur + 1;
            continue;
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            // LOAD segment: the k-step's operand reads (and, at the chunk's last k-step, the wait for the next chunk's DMA)
            IA_STAMP(0);
            h16x8 a_use[NP * FO], b_use[NP * FP];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int fo = 0; fo < FO; ++fo) a_use[pl * FO + fo] = wh[pl * NU * BO + arow[s] + (wo * FO + fo) * 32 + l31];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int fp = 0; fp < FP; ++fp) b_use[pl * FP + fp] = ph[pl * cap + bpos[fp] + boff[s]];
            // this wave's share of the refill DMA, behind its reads (70 - 200 issue cycles per piece: under the partner wave's MFMAs here;
            // at the end of the COMPUTE segment they delayed the barrier the partner waits at -- measured, 620 - 720 cycle intervals)
            // (the wait for chunk ch + 1 goes IN FRONT of this k-step's pieces: an s_waitcnt vmcnt directly behind freshly issued
            //  buffer_load ... lds instructions stalled ~220 cycles although the pieces it waits for were three k-steps old -- trace builds)
            if (s == 2 && ch + 1 < c_hi && IA_UP_ABLATE != 4) {
                constexpr int kFirst2 = (2 * n_dma + 2) / 3;      // pieces of slices 0 and 1: q * 3 / n_dma < 2
                const int full = issued - ch - 2;                  // chunks beyond ch + 1 whose pieces are all out (0 or 1)
                if (fill) { if (full <= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kFirst2) : "memory");
                            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_dma + kFirst2) : "memory"); }
                else wait_chunks_in_flight<n_dma>(full);
            }
            if (fill) IA_UP_ISSUE(fill_chunk, fill_stage, s, 3);
            if (s == 2 && fill) ++issued;
            // the reads have RETURNED before the barrier: what follows it on the other wave group may refill the stage they came from
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            IA_STAMP(1);
            ia_barrier();
            IA_STAMP(2);
            __builtin_amdgcn_sched_barrier(0);
            // COMPUTE segment
            constexpr int kPx[3] = {0, 0, 1};
            const int px = kPx[s];
#if IA_UP_ABLATE == 2
#pragma unroll
            for (int q = 0; q < NP * FO; ++q) asm volatile("" ::"v"(a_use[q]));
#pragma unroll
            for (int q = 0; q < NP * FP; ++q) asm volatile("" ::"v"(b_use[q]));
#else
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                for (int fp = 0; fp < FP; ++fp)      // lo * hi
                    acc[px][fo][fp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_use[FO + fo], b_use[fp], acc[px][fo][fp], 0, 0, 0);
            h16x8 a_sc[FO];                        // weight high parts at 2^-11: they meet the activation's low parts (scaled by 2^11)
#pragma unroll
            for (int fo = 0; fo < FO; ++fo) a_sc[fo] = a_use[fo] * (_Float16)(1.0f / kLoScale);
#pragma unroll
            for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                for (int fp = 0; fp < FP; ++fp)      // (hi * 2^-11) * (lo * 2^11)
                    acc[px][fo][fp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_sc[fo], b_use[FP + fp], acc[px][fo][fp], 0, 0, 0);
#pragma unroll
            for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                for (int fp = 0; fp < FP; ++fp)      // hi * hi
                    acc[px][fo][fp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_use[fo], b_use[fp], acc[px][fo][fp], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
#endif
            __builtin_amdgcn_sched_barrier(0);
            IA_STAMP(3);
            ia_barrier();
            IA_STAMP(4);
#if IA_UP_TRACE
            ++tr_n;
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        cur = cur + 1 == NS ? 0 : cur + 1;
    }
    if (!grp) ia_barrier();
#undef IA_UP_ISSUE

    if (sb.gpe || IA_UP_ABLATE == 3) {      // (whole tiles fold the scale into their demodulation coefficients, see up_store_tile)
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
            for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                for (int fp = 0; fp < FP; ++fp)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[px][fo][fp][r] *= g.acc_scale;      // back from the scale of the packed weights (exact)
    }
    if (IA_UP_ABLATE == 3) {
        float sink = 0.f;
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
            for (int fo = 0; fo < FO; ++fo)
#pragma unroll
                for (int fp = 0; fp < FP; ++fp)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sink += acc[px][fo][fp][r];
        if (sink == 123.456f) y[0] = sink;
    } else if (!sb.gpe) {
        up_store_tile<FP>(acc, y, demod, g, sb, b, o0, p0, p_last, tid, g.acc_scale);
    } else if (!IA_UP_TRACE) {
        float4* slab = reinterpret_cast<float4*>(slabs + ((int64_t)b * g.n_slabs + sb.slab_first + lt * sb.gpe + part) * ((int64_t)NACC * NTHREADS)) + tid;
#pragma unroll
        for (int q = 0; q < NACC / 4; ++q) {
            const int fr = q >> 2, r0 = (q & 3) * 4;
            const f32x16& a = acc[fr / (FP * FO)][(fr / FP) % FO][fr % FP];
            slab[(int64_t)q * NTHREADS] = make_float4(a[r0], a[r0 + 1], a[r0 + 2], a[r0 + 3]);
        }
    }
  }
 }
}

// Edge tiles: sum the `gpe` partial accumulators of a tile in worker order and store.  grid = (edge tile, batch, FO * FP * 4 register
// quads): a thread adds one quad of both column phases over the slabs (all loads of a phase in flight together) and stores 4 pairs.
template <int FP>
__global__ __launch_bounds__(512) void up_edge_fixup_kernel(const float* __restrict__ slabs, float* __restrict__ y, const float* __restrict__ demod, UpGeo g) {
    constexpr int FO = 2, WP = 4, NTHREADS = 512, BO = 128, NACC = 2 * FO * FP * 16;
    const int e = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int zq = blockIdx.z, rq = zq & 3, fp = (zq >> 2) % FP, fo = (zq >> 2) / FP;
    UpSub sb = g.sub[0];
#pragma unroll
    for (int k = 0; k < kMaxSub; ++k)
        if (k < g.nsub && g.sub[k].gpe && e >= g.sub[k].edge_first) sb = g.sub[k];
    const int lt = e - sb.edge_first;
    const float4* base = reinterpret_cast<const float4*>(slabs + ((int64_t)b * g.n_slabs + sb.slab_first + lt * sb.gpe) * ((int64_t)NACC * NTHREADS)) + tid;
    float4 sum[2];
#pragma unroll
    for (int px = 0; px < 2; ++px) {
        const int q = (((px * FO + fo) * FP + fp) << 2) + rq;
        float4 a = base[(int64_t)q * NTHREADS];
        for (int w0 = 1; w0 < sb.gpe; w0 += 8) {             // eight partials in flight per round (one dependent load per partial took 15 - 21 us)
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = base[(int64_t)min(w0 + j, sb.gpe - 1) * (NACC * NTHREADS / 4) + (int64_t)q * NTHREADS];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (w0 + j < sb.gpe) { a.x += v[j].x; a.y += v[j].y; a.z += v[j].z; a.w += v[j].w; }      // (worker order: deterministic)
        }
        sum[px] = a;
    }
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int wo = wave / WP, wp = wave % WP;
    const int o0 = (lt % g.TO) * BO, p0 = (lt / g.TO) * sb.tp;
    const int p_last = min(p0 + sb.tp, sb.GH * sb.GW) - 1;
    const int p = p0 + (wp * FP + fp) * 32 + l31;
    if (p > p_last) return;
    const int pr = p / sb.GW, pc = p - pr * sb.GW;
    const int oy = 2 * (sb.r_off + pr) + sb.py, ox = 2 * (sb.c_off + pc);
    const int64_t ohw = (int64_t)g.OH * g.OW, pix = (int64_t)oy * g.OW + ox;
    const bool pair = ox + 1 < g.OW;
    const float v0[4] = {sum[0].x, sum[0].y, sum[0].z, sum[0].w}, v1[4] = {sum[1].x, sum[1].y, sum[1].z, sum[1].w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int o = o0 + (wo * FO + fo) * 32 + 8 * rq + 4 * half + k;
        const float dm = demod ? demod[b * g.O + o] : 1.f;
        float* dst = y + ((int64_t)b * g.O + o) * ohw + pix;
        if (pair) __builtin_memcpy(dst, &(const float2&)make_float2(v0[k] * dm, v1[k] * dm), 8);
        else dst[0] = v0[k] * dm;
    }
}

// ---- host side
struct UpPlan { UpGeo g; int fp, jp, n_wg; size_t lds, scratch; };

int plan_up(int B, int I, int O, int H, int W, UpPlan* out) {
    if (I % 16 || O % 128 || H < 16 || W < 16 || W > 1024 || H > 1024) return IA_ERR_UNSUPPORTED;
    UpPlan p{};
    UpGeo& g = p.g;
    g.B = B; g.I = I; g.O = O; g.H = H; g.W = W; g.OH = 2 * H + 1; g.OW = 2 * W + 1; g.TO = O / 128;
    if ((int64_t)B * O * g.OH * g.OW > INT32_MAX || (int64_t)B * I * H * W > INT32_MAX) return IA_ERR_UNSUPPORTED;
    // the DMA plan does 32-bit BYTE arithmetic: a frame's two split planes (4 bytes per element) and the packed weights (2 planes x 9
    // taps x I x O fp16) are each one buffer resource whose size must stay below the "outside" offset (0x7ffffff0) that yields zeros
    if (4 * (int64_t)I * H * W >= 0x7ffffff0LL || 36 * (int64_t)I * O >= 0x7ffffff0LL) return IA_ERR_UNSUPPORTED;
    p.fp = W >= 256 ? 2 : 1;
    const int BP = 128 * p.fp;
    const int npt = (int)ia::ceil_div((int64_t)H * W, BP);
    // workers per edge tile: the edge tiles only have to finish inside the time of an interior tile (they run beside them), so their K
    // range is cut just enough for that -- ~16 chunks per worker.  (The first version cut 16-way: 16 slabs of 128 ch x 128 / 256 pt per
    // edge tile were 205 MB of slab writes + reads per frame, PMC r04: more than the images these layers write.)
    const int c0 = I / 8;
    int gpe = c0 / 16 < 2 ? 2 : (c0 / 16 > 4 ? 4 : c0 / 16);
    // whole-tile grids (interior of either row phase), then the edge grids
    const bool one_round = (int64_t)B * 2 * npt * g.TO <= ia::kNumCU;
    // K-deep layers whose tiles leave most of the machine idle (512 input channels at 64^2: 64 long tiles of 192 k-steps): the long
    // (py 0) interior tiles are cut in two along K -- partial sums to slabs, finished by the fix-up launch that the edge grids need
    // anyway -- and the py 1 tiles run one per workgroup: every workgroup then has 96 k-steps (same-box 93 -> see DESIGN 4.7)
    const bool split_long = one_round && I >= 512 && (int64_t)B * 3 * npt * g.TO <= ia::kNumCU;
    constexpr int kColPts = 64;      // points per tile of the one-column grids
    // r06: a layer whose interior tiles leave CUs idle anyway (one round, no K-split interior: 256 -> 128 @128^2, 192 workgroups) runs its
    // edge tiles WHOLE beside them -- an edge tile has the K range of an interior tile and a CU of its own, so it ends with them; no slabs
    // and no up_edge_fixup launch (one launch and one graph edge less on the chain of every such layer)
    {
        const int64_t interior = (int64_t)B * g.TO * (npt + (npt + 1) / 2);
        const int64_t edges = (int64_t)B * g.TO * (ia::ceil_div(W, BP) + ia::ceil_div(H + 1, kColPts) + ia::ceil_div(H, kColPts));
        if (one_round && !split_long && interior + edges <= ia::kNumCU) gpe = 0;
    }
    const UpSub subs[kMaxSub] = {
        // first_wg n_wg n_pt tp reps gpe edge_first slab_first GH GW r_off c_off py
        {0, 0, npt, BP, 1, split_long ? 2 : 0, 0, 0, H, W, 0, 0, 0},
        {0, 0, npt, BP, (one_round && !split_long) ? 2 : 1, 0, 0, 0, H, W, 0, 0, 1},      // (not one_round: served by the workgroups of grid 0, g.pair)
        {0, 0, (int)ia::ceil_div(W, BP), BP, 1, gpe, 0, 0, 1, W, H, 0, 0},
        {0, 0, (int)ia::ceil_div(H + 1, kColPts), kColPts, 1, gpe, 0, 0, H + 1, 1, 0, W, 0},
        {0, 0, (int)ia::ceil_div(H, kColPts), kColPts, 1, gpe, 0, 0, H, 1, 0, W, 1},
    };
    g.nsub = kMaxSub;
    g.pair = one_round ? 0 : 1;
    int wg = 0, edge = 0, slabs = 0, worst0 = 0, worst1 = 0;
    for (int k = 0; k < kMaxSub; ++k) {
        UpSub s = subs[k];
        const int tiles = s.n_pt * g.TO;
        s.first_wg = wg;
        s.n_wg = s.gpe ? tiles * s.gpe : (k == 1 && !one_round) ? 0 : (int)ia::ceil_div(tiles, s.reps);
        if (s.gpe) { s.edge_first = edge; edge += tiles; s.slab_first = slabs; slabs += tiles * s.gpe; }
        wg += s.n_wg;
        g.sub[k] = s;
        const int npts = s.GH * s.GW;
        for (int q0 = 0; q0 < npts; q0 += s.tp) {
            const UWin w = up_window(q0, (q0 + s.tp < npts ? q0 + s.tp : npts) - 1, s.GW, 1 - s.py);
            int& worst = s.py ? worst1 : worst0;
            if (w.PSZ > worst) worst = w.PSZ;
        }
    }
    g.E = edge;
    g.n_slabs = slabs;
    p.n_wg = wg;
    const int need = worst0 > 2 * worst1 ? worst0 : 2 * worst1;
    g.cap = (need + 127) & ~127;
    p.jp = (int)ia::ceil_div(2 * g.cap / 64, 8);
    const size_t stage = (size_t)(2 * 6 * 128 + 2 * g.cap) * 16;
    if (3 * stage + 1024 > kLdsBytesUp || p.jp > 4) return IA_ERR_UNSUPPORTED;      // (the K loop's refill schedule needs three stages)
    // ring depth: the DMA of a chunk must land within (stages - 1) chunks of 36 - 72 MFMAs per SIMD: as deep as the LDS allows
    // (bounded by the 6-bit vmcnt: (stages - 1) x (3 + jp) instructions per wave outstanding)
    g.stages = (int)((kLdsBytesUp - 1024) / stage);
    if (g.stages > kUpMaxStages) g.stages = kUpMaxStages;

    p.lds = stage * g.stages + 1024;       // (+ the dummy target of all-outside DMA pieces)
    p.scratch = (size_t)B * g.n_slabs * (2 * 2 * p.fp * 16) * 512 * sizeof(float);
    *out = p;
    return IA_OK;
}

template <int FP, int JP>
int launch_up(const UpPlan& p, const h16x8* xs, const h16x8* wk, float* y, float* scratch, const float* demod, hipStream_t s) {
    auto k = up_rows_kernel<FP, JP>;
    if (const int rs = ia::reserve_lds((const void*)k, p.lds, "upconv_rows")) return rs;
    hipLaunchKernelGGL(k, dim3(p.n_wg, p.g.B), dim3(512), p.lds, s, xs, wk, y, scratch, demod, p.g);
    if (const int st = ia::check_launch("ia_upconv2d_rows_sx")) return st;
    if (p.g.E == 0) return IA_OK;      // every tile whole: nothing to sum
    hipLaunchKernelGGL((up_edge_fixup_kernel<FP>), dim3(p.g.E, p.g.B, 2 * FP * 4), dim3(512), 0, s, scratch, y, demod, p.g);
    return ia::check_launch("ia_upconv2d_rows_sx(edge fix-up)");
}

}  // namespace

extern "C" int ia_upconv2d_rows_plan(int B, int I, int O, int H, int W, size_t* h_scratch_bytes) {
    IA_REQUIRE(h_scratch_bytes, "null output pointer");
    IA_REQUIRE(B > 0 && I > 0 && O > 0 && H > 0 && W > 0, "empty tensor");
    UpPlan p;
    if (plan_up(B, I, O, H, W, &p) != IA_OK)
        return ia::fail(IA_ERR_UNSUPPORTED, "row-phase up-convolution: needs I %% 16 == 0, O %% 128 == 0, 16 <= H, W <= 1024 (I %d O %d %dx%d)", I, O, H, W);
    *h_scratch_bytes = p.scratch;
    return IA_OK;
}

extern "C" int ia_upconv2d_rows_sx(const void* xs, const void* wk_split, int wk_exp, const float* demod, float* y, float* scratch,
                                   size_t scratch_bytes, int B, int I, int O, int H, int W, void* stream) {
    IA_REQUIRE(xs && wk_split && y, "xs, wk and y must be device pointers");
    IA_REQUIRE(B > 0 && I > 0 && O > 0 && H > 0 && W > 0, "empty tensor");
    IA_REQUIRE(wk_exp >= -14 && wk_exp <= 30, "wk_exp is the power of two the weights were scaled by at pack time");
    UpPlan p;
    if (plan_up(B, I, O, H, W, &p) != IA_OK)
        return ia::fail(IA_ERR_UNSUPPORTED, "row-phase up-convolution: needs I %% 16 == 0, O %% 128 == 0, 16 <= H, W <= 1024 (I %d O %d %dx%d)", I, O, H, W);
    IA_REQUIRE(p.scratch == 0 || (scratch && scratch_bytes >= p.scratch), "the edge tiles need %zu bytes of scratch, got %zu", p.scratch, scratch_bytes);
    p.g.acc_scale = ldexpf(1.f, -wk_exp);
    hipStream_t s = (hipStream_t)stream;
    const h16x8* x8 = static_cast<const h16x8*>(xs);
    const h16x8* w8 = static_cast<const h16x8*>(wk_split);
    if (p.fp == 2) {
        if (p.jp <= 2) return launch_up<2, 2>(p, x8, w8, y, scratch, demod, s);
        if (p.jp == 3) return launch_up<2, 3>(p, x8, w8, y, scratch, demod, s);
        return launch_up<2, 4>(p, x8, w8, y, scratch, demod, s);
    }
    if (p.jp <= 2) return launch_up<1, 2>(p, x8, w8, y, scratch, demod, s);
    if (p.jp == 3) return launch_up<1, 3>(p, x8, w8, y, scratch, demod, s);
    return launch_up<1, 4>(p, x8, w8, y, scratch, demod, s);
}

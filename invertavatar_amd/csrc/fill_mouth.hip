// ia_fill_mouth: GPU flood fill that closes the mouth hole of the rasterised face mask.
//
// Replaces fill_mouth(images, blur_mouth_edge=False) of the reference
// (training_avatar_texture/volumetric_rendering/renderer.py:716-741), which copies the mask to the host and calls
// cv2.floodFill(img*255, seed (0,0), newVal 255, loDiff 0, upDiff 254, FLOODFILL_FIXED_RANGE) per frame -- a device
// sync twice per frame.  Semantics kept: a pixel is "passable" when seed <= v <= seed + 254 (v = alpha*255,
// seed = v at (0,0)); the 4-connected passable region containing (0,0) is filled with 255;
//   mouth = (255 - filled) / 255.
// One workgroup per image; the label image lives in LDS (1 byte / pixel, rows padded by 4 bytes so that per-row
// sweeps hit distinct banks).  Propagation is by alternating full-line sweeps (left/right per row, up/down per
// column), each of which carries the fill along a whole line in one pass; convex-ish masks converge in 2-3 rounds.
#include "ia_common.h"

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void fill_mouth_kernel(const float* __restrict__ alpha, float* __restrict__ mouth, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) unsigned char st[];   // 0 wall, 1 passable, 2 reached
    __shared__ int changed;
    const int S = W + 4;
    const float* a = alpha + (int64_t)blockIdx.x * H * W;
    float* m = mouth + (int64_t)blockIdx.x * H * W;
    const float seed = a[0] * 255.f;
    for (int i = threadIdx.x; i < H * W; i += kThreads) {
        const int y = i / W, x = i - y * W;
        const float v = a[i] * 255.f;
        st[y * S + x] = (v >= seed && v <= seed + 254.f) ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) st[0] = 2;
    __syncthreads();
    for (int round = 0; round < H + W; ++round) {
        if (threadIdx.x == 0) changed = 0;
        __syncthreads();
        int local = 0;
        for (int y = threadIdx.x; y < H; y += kThreads) {
            unsigned char* row = st + y * S;
            bool reach = false;
            for (int x = 0; x < W; ++x) { const unsigned char s = row[x]; if (s == 2) reach = true; else if (s == 1 && reach) { row[x] = 2; local = 1; } else if (s == 0) reach = false; }
            reach = false;
            for (int x = W - 1; x >= 0; --x) { const unsigned char s = row[x]; if (s == 2) reach = true; else if (s == 1 && reach) { row[x] = 2; local = 1; } else if (s == 0) reach = false; }
        }
        __syncthreads();
        for (int x = threadIdx.x; x < W; x += kThreads) {
            unsigned char* col = st + x;
            bool reach = false;
            for (int y = 0; y < H; ++y) { const unsigned char s = col[y * S]; if (s == 2) reach = true; else if (s == 1 && reach) { col[y * S] = 2; local = 1; } else if (s == 0) reach = false; }
            reach = false;
            for (int y = H - 1; y >= 0; --y) { const unsigned char s = col[y * S]; if (s == 2) reach = true; else if (s == 1 && reach) { col[y * S] = 2; local = 1; } else if (s == 0) reach = false; }
        }
        if (local) changed = 1;
        __syncthreads();
        if (!changed) break;
        __syncthreads();
    }
    for (int i = threadIdx.x; i < H * W; i += kThreads) {
        const int y = i / W, x = i - y * W;
        m[i] = st[y * S + x] == 2 ? 0.f : (255.f - a[i] * 255.f) / 255.f;
    }
}

// ---- 256 x 256 masks (the generator's UV maps): bit-parallel flood.
// The label image is kept as two 256 x 256 bit matrices (passable P, reached R), row-major AND column-major, 64 pixels per
// 64-bit word.  Carrying the fill along a whole line is then integer arithmetic: with X = R & P, the multi-word sum P + X
// ripples a carry from every reached pixel to the end of its run of passable pixels, so ((P + X) ^ P) & P | X is the line
// filled towards higher indices; the other direction is the same on bit-reversed words.  One thread owns one row (then
// one column); between the row pass and the column pass R is transposed with one ballot per pixel column.  A round costs
// ~10 us instead of ~100 us of byte-wise LDS sweeps; results are identical (the 4-connected component is unique).
constexpr int kN = 256, kNW = kN / 64;
typedef unsigned long long u64;

__device__ __forceinline__ void fill_line(const u64 (&P)[kNW], u64 (&R)[kNW]) {
    u64 X[kNW], F[kNW];
    unsigned carry = 0;
#pragma unroll
    for (int w = 0; w < kNW; ++w) {            // towards higher bit indices
        X[w] = R[w] & P[w];
        const u64 s1 = P[w] + X[w], s2 = s1 + carry;
        carry = (s1 < P[w]) | (s2 < s1);
        F[w] = ((s2 ^ P[w]) & P[w]) | X[w];
    }
    carry = 0;
#pragma unroll
    for (int w = kNW - 1; w >= 0; --w) {       // towards lower bit indices: the same on the reversed line
        const u64 p = __brevll(P[w]), x = __brevll(F[w]);
        const u64 s1 = p + x, s2 = s1 + carry;
        carry = (s1 < p) | (s2 < s1);
        R[w] = __brevll(((s2 ^ p) & p) | x);
    }
}

// dst[c][wave] = bit c of the 64 rows of this wave (src row words in registers): 256 x 256 bit transpose.
__device__ __forceinline__ void transpose_bits(const u64 (&r)[kNW], u64* dst, int wave, int lane) {
#pragma unroll
    for (int w = 0; w < kNW; ++w)
        for (int bit = 0; bit < 64; ++bit) {
            const u64 word = __ballot((r[w] >> bit) & 1ull);
            if (lane == 0) dst[(w * 64 + bit) * kNW + wave] = word;
        }
}

__global__ __launch_bounds__(kN) void fill_mouth256_kernel(const float* __restrict__ alpha, float* __restrict__ mouth) {
    __shared__ u64 Prow[kN * kNW], Pcol[kN * kNW], Rrow[kN * kNW], Rcol[kN * kNW];
    __shared__ int changed;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* a = alpha + (int64_t)blockIdx.x * kN * kN;
    float* m = mouth + (int64_t)blockIdx.x * kN * kN;
    const float seed = a[0] * 255.f;
    for (int wd = wave; wd < kN * kNW; wd += kN / 64) {          // one 64-pixel word per wave iteration (coalesced reads)
        const float v = a[wd * 64 + lane] * 255.f;
        const u64 word = __ballot(v >= seed && v <= seed + 254.f);
        if (lane == 0) Prow[wd] = word;
    }
    __syncthreads();
    u64 P[kNW], R[kNW], Pc[kNW];
#pragma unroll
    for (int w = 0; w < kNW; ++w) { P[w] = Prow[tid * kNW + w]; R[w] = 0; }
    if (tid == 0) R[0] = 1ull;                                    // the seed pixel (0,0) is passable by definition
    transpose_bits(P, Pcol, wave, lane);
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kNW; ++w) Pc[w] = Pcol[tid * kNW + w];
    for (int round = 0; round < 2 * kN; ++round) {
        if (tid == 0) changed = 0;
        u64 before[kNW];
#pragma unroll
        for (int w = 0; w < kNW; ++w) before[w] = R[w];
        fill_line(P, R);                                          // along the row this thread owns
        transpose_bits(R, Rcol, wave, lane);
        __syncthreads();
        u64 C[kNW];
#pragma unroll
        for (int w = 0; w < kNW; ++w) C[w] = Rcol[tid * kNW + w];
        fill_line(Pc, C);                                         // along the column this thread owns
        transpose_bits(C, Rrow, wave, lane);
        __syncthreads();
        bool diff = false;
#pragma unroll
        for (int w = 0; w < kNW; ++w) { R[w] = Rrow[tid * kNW + w]; diff |= R[w] != before[w]; }
        if (diff) changed = 1;
        __syncthreads();
        if (!changed) break;
        __syncthreads();
    }
#pragma unroll
    for (int w = 0; w < kNW; ++w) Rrow[tid * kNW + w] = R[w];
    __syncthreads();
    for (int wd = wave; wd < kN * kNW; wd += kN / 64) {
        const int i = wd * 64 + lane;
        const bool reached = (Rrow[wd] >> lane) & 1ull;
        m[i] = reached ? 0.f : (255.f - a[i] * 255.f) / 255.f;
    }
}

// ---- blur_mouth_edge=True (renderer.py:732-734): copyImg = cv2.blur(cv2.erode(copyImg, ones(3,3), iterations=3), (5,5)) on the filled
// float image (255 where the flood reached, alpha*255 elsewhere), then (255 - copyImg) / 255.  Three 3x3 erosions are one 7x7 minimum
// (border pixels see +inf: cv2's morphologyDefaultBorderValue); the normalised 5x5 box filter of a CV_32F image sums in double and
// multiplies by the double 1/25 before rounding to float (boxFilter: sumType CV_64F), border BORDER_REFLECT_101.
// A workgroup owns a 32 x 32 output tile: filled image with a 5-pixel halo -> LDS, 7x7 minimum with a 2-pixel halo -> LDS, box sum.
constexpr int kET = 32, kEH = kET + 10, kEM = kET + 4;

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

__global__ __launch_bounds__(256) void mouth_edge_kernel(const float* __restrict__ alpha, const float* __restrict__ mouth,
                                                         float* __restrict__ out, int H, int W) {
    __shared__ float filled[kEH * kEH];      // image coordinates (ty0 - 5 + r, tx0 - 5 + c); +inf outside the image
    __shared__ float eroded[kEM * kEM];      // image coordinates (ty0 - 2 + r, tx0 - 2 + c)
    const int64_t img = (int64_t)blockIdx.z * H * W;
    const int ty0 = blockIdx.y * kET, tx0 = blockIdx.x * kET;
    // The box filter reads the eroded image at reflected coordinates; with H, W >= 3 a reflected index stays within 2 pixels of the
    // border it left, i.e. inside the tile's own halo range only for interior tiles -- so eroded[] is evaluated AT the reflected
    // coordinate (its 7x7 window is fetched through `at`), never assumed to sit in the neighbouring slot.
    auto at = [&](int y, int x) -> float {
        if (y < 0 || y >= H || x < 0 || x >= W) return __builtin_inff();
        const int64_t i = img + (int64_t)y * W + x;
        return mouth[i] == 0.f ? 255.f : alpha[i] * 255.f;
    };
    for (int i = threadIdx.x; i < kEH * kEH; i += 256) filled[i] = at(ty0 - 5 + i / kEH, tx0 - 5 + i % kEH);
    __syncthreads();
    for (int i = threadIdx.x; i < kEM * kEM; i += 256) {
        const int r = i / kEM, c = i % kEM;
        const int y = ty0 - 2 + r, x = tx0 - 2 + c;
        const int yr = reflect101(y, H), xr = reflect101(x, W);
        float m = __builtin_inff();
        if (yr == y && xr == x) {
            for (int dy = 0; dy < 7; ++dy)
                for (int dx = 0; dx < 7; ++dx) m = fminf(m, filled[(r + dy) * kEH + c + dx]);
        } else {                                  // a reflected border sample: its own window, from memory
            for (int dy = -3; dy <= 3; ++dy)
                for (int dx = -3; dx <= 3; ++dx) m = fminf(m, at(yr + dy, xr + dx));
        }
        eroded[i] = m;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kET * kET; i += 256) {
        const int r = i / kET, c = i % kET;
        const int y = ty0 + r, x = tx0 + c;
        if (y >= H || x >= W) continue;
        double sum = 0.0;
        for (int dx = 0; dx < 5; ++dx) {          // row sums first, then the column of row sums (RowSum / ColumnSum order; exact in double)
            double col = 0.0;
            for (int dy = 0; dy < 5; ++dy) col += (double)eroded[(r + dy) * kEM + c + dx];
            sum += col;
        }
        const float blurred = (float)(sum * (1.0 / 25.0));
        out[img + (int64_t)y * W + x] = (255.f - blurred) / 255.f;
    }
}

}  // namespace

extern "C" int ia_mouth_edge_blur(const float* alpha, const float* mouth, float* out, int B, int H, int W, void* stream) {
    IA_REQUIRE(alpha && mouth && out, "null pointer argument");
    IA_REQUIRE(B > 0 && H >= 3 && W >= 3, "mask must be at least 3 x 3");
    hipLaunchKernelGGL(mouth_edge_kernel, dim3((W + kET - 1) / kET, (H + kET - 1) / kET, B), dim3(256), 0, (hipStream_t)stream,
                       alpha, mouth, out, H, W);
    return ia::check_launch("ia_mouth_edge_blur");
}

extern "C" int ia_fill_mouth(const float* alpha, float* mouth, int B, int H, int W, void* stream) {
    IA_REQUIRE(alpha && mouth, "null pointer argument");
    IA_REQUIRE(B > 0 && H > 0 && W > 0, "empty tensor");
    if (H == kN && W == kN) {
        hipLaunchKernelGGL(fill_mouth256_kernel, dim3(B), dim3(kN), 0, (hipStream_t)stream, alpha, mouth);
        return ia::check_launch("ia_fill_mouth");
    }
    const size_t lds = (size_t)H * (W + 4);
    if (lds > 150 * 1024) return ia::fail(IA_ERR_UNSUPPORTED, "mask %dx%d does not fit the LDS label image", H, W);
    if (const int rs = ia::reserve_lds((const void*)fill_mouth_kernel, (size_t)(lds), "fill_mouth")) return rs;
    hipLaunchKernelGGL(fill_mouth_kernel, dim3(B), dim3(kThreads), lds, (hipStream_t)stream, alpha, mouth, H, W);
    return ia::check_launch("ia_fill_mouth");
}

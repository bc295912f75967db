// Probe ("settle Winograd with a number"): what does the INPUT side of a Winograd F(2x2, 3x3) convolution cost
// on split-format activations?  A Winograd form of the fp16-pair convolutions does 16 products per 2x2 output tile and channel pair
// instead of 36 (2.25x fewer MFMAs), but its B operand is the TRANSFORMED patch V = B^T d B, and the fp16 hi / lo split has to be
// applied to V, i.e. inside the consumer: per 16-channel chunk a workgroup must
//   (1) read the 4x4 input tiles of its 2x2 output tiles from the DMA'd patch (hi / lo planes, 16-byte units of 8 channels),
//   (2) rebuild fp32 (hi + lo * 2^-11), (3) transform (32 adds per tile and channel), (4) split to hi / lo again, (5) write the
//   16 transformed positions back to LDS as MFMA operands.
// This kernel does exactly that and nothing else (no MFMAs, no weights), for one layer: C channels at H x W, a workgroup per
// 8 x 32 output pixels (64 tiles), 16 channels per chunk, the patch DMA'd into a two-stage LDS ring like conv_split_kernel's.
// Variants: MODE 0 = all five steps; 1 = no split (fp32 -> one cvt); 2 = no transform either (copy through registers);
// 3 = DMA only.  Prints microseconds per layer pass, to be set against the MFMA time the layer would save.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/probes/wino_transform.hip -o /tmp/wino && /tmp/wino
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char lds_char;

__device__ __forceinline__ void dma_piece(u32x4 rsrc, unsigned lds_addr, unsigned voffset, unsigned soffset) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory", "m0");
}

constexpr int TY = 4, TX = 16;                 // Winograd tiles per workgroup: 4 x 16 (8 x 32 output pixels)
constexpr int PH = 2 * TY + 2, PW = 2 * TX + 2; // input patch 10 x 34
constexpr int NPOS = PH * PW;                   // 340 positions of 16 bytes per (plane, channel group)
constexpr int CG = 2;                           // channel groups of 8 per chunk (16 channels, the K step of v_mfma_f32_32x32x16_f16)
constexpr int UNITS = 2 * CG * NPOS;            // 16-byte units per chunk: [plane][cg][pos]
constexpr int PIECES = (UNITS + 63) / 64;       // DMA pieces of 64 units
constexpr int OUT_UNITS = 2 * CG * 16 * TY * TX; // transformed operands per chunk: [plane][cg][position 16][tile 64]

__device__ __forceinline__ void split(float v, _Float16& hi, _Float16& lo) {
    v = fminf(fmaxf(v, -65504.f), 65504.f);
    hi = fabsf(v) < 6.103515625e-5f ? (_Float16)0.f : (_Float16)v;
    lo = (_Float16)((v - (float)hi) * 2048.f);
}

template <int MODE>
__global__ __launch_bounds__(256) void wino_input_kernel(const h16x8* __restrict__ xs, int C, int H, int W, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    h16x8* patch = reinterpret_cast<h16x8*>(lds);                      // [2 stages][UNITS]
    h16x8* vout = patch + 2 * PIECES * 64;                             // [OUT_UNITS]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = W / (2 * TX);
    const int by = blockIdx.x / tiles_x, bx = blockIdx.x - by * tiles_x;
    const int y0 = by * 2 * TY - 1, x0 = bx * 2 * TX - 1;               // patch origin (padding 1)
    const int C8 = C / 8;
    const long HW = (long)H * W;
    u32x4 rs;
    {
        const unsigned long long a = (unsigned long long)xs;
        rs[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
        rs[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
        rs[2] = (unsigned)(2l * C8 * HW * 16);
        rs[3] = 0x00020000u;
    }
    const unsigned lds_base = (unsigned)(unsigned long long)(lds_char*)lds;
    // this thread's DMA units (the same for every chunk): unit u = 64 * piece + lane -> (plane, cg, pos) -> global byte offset
    constexpr int PPW = (PIECES + 3) / 4;
    unsigned voff[PPW];
#pragma unroll
    for (int s = 0; s < PPW; ++s) {
        const int u = 64 * (wave + 4 * s) + lane;
        const int pos = u % NPOS, cg = (u / NPOS) % CG, plane = u / (NPOS * CG);
        const int y = y0 + pos / PW, x = x0 + pos % PW;
        const bool ok = u < UNITS && y >= 0 && y < H && x >= 0 && x < W;
        voff[s] = ok ? (unsigned)((((long)plane * C8 + cg) * HW + (long)y * W + x) * 16) : 0x7ffffff0u;
    }
    auto issue = [&](int chunk, int stage) {
#pragma unroll
        for (int s = 0; s < PPW; ++s)
            dma_piece(rs, lds_base + (unsigned)((stage * PIECES + wave + 4 * s) * 64 * 16), voff[s], (unsigned)((long)chunk * CG * HW * 16));
    };
    const int chunks = C8 / CG;
    // work item of this thread: tile t (0..63) x channel group g (0..1); 128 items, 256 threads: the two halves of the workgroup take
    // the two PLANE-pairs of the reconstruction?  No: both planes are needed per value.  Threads 128..255 take every other chunk.
    const int item = tid & 127, half = tid >> 7;
    const int t = item & 63, g = item >> 6;
    const int ty = t / TX, tx = t - ty * TX;
    float acc = 0.f;
    issue(0, 0);
    for (int ch = 0; ch < chunks; ++ch) {
        const int stage = ch & 1;
        if (ch + 1 < chunks) issue(ch + 1, stage ^ 1);
        if (ch + 1 < chunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (MODE < 3 && (ch & 1) == half) {
            const h16x8* ph = patch + stage * PIECES * 64 + (0 * CG + g) * NPOS;
            const h16x8* pl = patch + stage * PIECES * 64 + (1 * CG + g) * NPOS;
            float d[4][4][8];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pos = (2 * ty + i) * PW + 2 * tx + j;
                    const h16x8 vh = ph[pos], vl = pl[pos];
#pragma unroll
                    for (int c = 0; c < 8; ++c) d[i][j][c] = (float)vh[c] + (float)vl[c] * (1.f / 2048.f);
                }
            if (MODE < 2) {
                // V = B^T d B, B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]: rows, then columns
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float r[4][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        r[0][j] = d[0][j][c] - d[2][j][c];
                        r[1][j] = d[1][j][c] + d[2][j][c];
                        r[2][j] = d[2][j][c] - d[1][j][c];
                        r[3][j] = d[1][j][c] - d[3][j][c];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        d[i][0][c] = r[i][0] - r[i][2];
                        d[i][1][c] = r[i][1] + r[i][2];
                        d[i][2][c] = r[i][2] - r[i][1];
                        d[i][3][c] = r[i][1] - r[i][3];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    h16x8 oh, ol;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        if (MODE == 0) { _Float16 a, b; split(d[i][j][c], a, b); oh[c] = a; ol[c] = b; }
                        else { oh[c] = (_Float16)d[i][j][c]; ol[c] = (_Float16)0.f; }
                    }
                    vout[((0 * CG + g) * 16 + 4 * i + j) * 64 + t] = oh;      // [plane][cg][position][tile]: 64 tiles = one MFMA column block
                    if (MODE == 0) vout[((1 * CG + g) * 16 + 4 * i + j) * 64 + t] = ol;
                }
        }
        __syncthreads();
        acc += (float)vout[tid][0] + (float)patch[stage * PIECES * 64 + tid][1];        // keep both buffers live
    }
    if (acc == 123.456f) sink[blockIdx.x] = acc;
}

template <int MODE>
float run(const h16x8* xs, int C, int H, int W, float* sink, int reps) {
    const int grid = (H / (2 * TY)) * (W / (2 * TX));
    const size_t lds = (size_t)(2 * PIECES * 64 + OUT_UNITS) * 16;
    hipFuncSetAttribute((const void*)wino_input_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(wino_input_kernel<MODE>, dim3(grid), dim3(256), lds, 0, xs, C, H, W, sink);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(wino_input_kernel<MODE>, dim3(grid), dim3(256), lds, 0, xs, C, H, W, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    struct Layer { int C, H; const char* name; double conv_us; };
    // conv_us: the layer's stride-1 fp16-pair convolution alone on the same pool this round (tools/bench_conv_layers.py, BENCH_EPI=1)
    const Layer layers[] = {{128, 256, "128 -> 128 @256^2", 61.8}, {256, 256, "256 -> 256 @256^2", 206.8}, {128, 512, "128 -> 128 @512^2", 232.3},
                            {256, 128, "256 -> 256 @128^2", 71.2}, {512, 64, "512 -> 512 @64^2", 76.2}};
    for (const Layer& L : layers) {
        const size_t n_units = (size_t)2 * (L.C / 8) * L.H * L.H;
        std::vector<_Float16> h(n_units * 8);
        unsigned s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (_Float16)(((s >> 8) & 0xffff) / 65536.f * 4.f - 2.f); }
        h16x8* xs; float* sink;
        hipMalloc(&xs, n_units * 16); hipMalloc(&sink, 1 << 20);
        hipMemcpy(xs, h.data(), n_units * 16, hipMemcpyHostToDevice);
        const float t0 = run<0>(xs, L.C, L.H, L.H, sink, 20), t1 = run<1>(xs, L.C, L.H, L.H, sink, 20), t2 = run<2>(xs, L.C, L.H, L.H, sink, 20),
                    t3 = run<3>(xs, L.C, L.H, L.H, sink, 20);
        // per output-channel tile of 128 the transform is repeated: a layer with O = C output channels runs it C / 128 times
        const int o_tiles = L.C / 128 > 0 ? L.C / 128 : 1;
        printf("%-20s  transform+split %7.1f us | no split %7.1f | no transform %7.1f | DMA only %7.1f   x %d output-channel tiles = %7.1f us"
               "   (the layer's convolution today: %.1f us)\n", L.name, t0, t1, t2, t3, o_tiles, t0 * o_tiles, L.conv_us);
        hipFree(xs); hipFree(sink);
    }
    return 0;
}

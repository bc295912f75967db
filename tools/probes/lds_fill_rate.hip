// Probe: how fast can ONE CU fill LDS from L2-resident data -- LDS-DMA (buffer_load_dwordx4 ... lds) against plain 16-byte loads
// + ds_write_b128 -- as a function of the bytes per chunk, the ring depth and the number of busy CUs?  The convolution kernels of
// conv_split.hip stage a chunk (20 .. 66 KB) per barrier; this isolates the staging from the MFMAs.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/probes/lds_fill_rate.hip -o /tmp/lds_fill && /tmp/lds_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;

__device__ __forceinline__ void dma_piece(u32x4 rsrc, unsigned lds_addr, int voffset, int soffset) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory", "m0");
}
__device__ __forceinline__ void wait_vmcnt_n(int n) {
    switch (n) {
#define W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15) W(16) W(17) W(18) W(19) W(20) W(21) W(22) W(23) W(24)
        W(25) W(26) W(27) W(28) W(29) W(30) W(31) W(32) W(33) W(34) W(35) W(36) W(37) W(38) W(39) W(40) W(41) W(42) W(43) W(44) W(45) W(46) W(47) W(48)
#undef W
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// MODE 0: LDS-DMA ring of NS stages; MODE 1: plain loads into registers one chunk ahead + ds_write_b128.
// P = 1 KB pieces per wave per chunk; 8 waves; a chunk is 8 * P KB.  `shared_src`: every workgroup reads the same bytes (weights-like).
template <int MODE, int P>
__global__ __launch_bounds__(512) void fill_kernel(const char* src, unsigned src_bytes, int chunks, int reps, int ns, int shared_src, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned chunk_bytes = 8 * P * 1024;
    const unsigned wg_off = shared_src ? 0 : (blockIdx.x * 2654435761u) % (src_bytes / 2) / chunk_bytes * chunk_bytes;
    const unsigned long long a = (unsigned long long)(src + wg_off);
    u32x4 rs;
    rs[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    rs[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    rs[2] = src_bytes / 2;
    rs[3] = 0x00020000u;
    const unsigned lds_base = (unsigned)(unsigned long long)(lds_char*)lds;
    float acc = 0.f;
    for (int rep = 0; rep < reps; ++rep) {
        if (MODE == 0) {
            int issued = 0;
            for (int k = 0; k < ns - 1 && issued < chunks; ++k, ++issued)
#pragma unroll
                for (int j = 0; j < P; ++j) dma_piece(rs, lds_base + k * chunk_bytes + (wave * P + j) * 1024, (wave * P + j) * 1024 + lane * 16, issued * chunk_bytes);
            int cur = 0;
            for (int ch = 0; ch < chunks; ++ch) {
                wait_vmcnt_n((issued - ch - 1) * P);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (issued < chunks) {
                    const int st = cur == 0 ? ns - 1 : cur - 1;
#pragma unroll
                    for (int j = 0; j < P; ++j) dma_piece(rs, lds_base + st * chunk_bytes + (wave * P + j) * 1024, (wave * P + j) * 1024 + lane * 16, issued * chunk_bytes);
                    ++issued;
                }
                acc += *reinterpret_cast<const float*>(lds + cur * chunk_bytes + tid * 16);      // one read per chunk keeps the data live
                cur = cur + 1 == ns ? 0 : cur + 1;
            }
        } else {
            f32x4 r[P];
            const char* base = src + wg_off;
#pragma unroll
            for (int j = 0; j < P; ++j) r[j] = *reinterpret_cast<const f32x4*>(base + (wave * P + j) * 1024 + lane * 16);
            for (int ch = 0; ch < chunks; ++ch) {
                const int st = ch & 1;
#pragma unroll
                for (int j = 0; j < P; ++j) *reinterpret_cast<f32x4*>(lds + st * chunk_bytes + (wave * P + j) * 1024 + lane * 16) = r[j];
                if (ch + 1 < chunks)
#pragma unroll
                    for (int j = 0; j < P; ++j) r[j] = *reinterpret_cast<const f32x4*>(base + (size_t)(ch + 1) * chunk_bytes + (wave * P + j) * 1024 + lane * 16);
                __syncthreads();
                acc += *reinterpret_cast<const float*>(lds + st * chunk_bytes + tid * 16);
            }
        }
        __syncthreads();
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int MODE, int P>
void run(const char* src, unsigned src_bytes, int wgs, int ns, int shared_src, float* sink) {
    const int chunks = 64, reps = 20;
    const size_t lds = (size_t)(MODE == 0 ? ns : 2) * 8 * P * 1024;
    if (lds > 160 * 1024) return;
    auto k = fill_kernel<MODE, P>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<wgs, 512, lds>>>(src, src_bytes, chunks, 2, ns, shared_src, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<wgs, 512, lds>>>(src, src_bytes, chunks, reps, ns, shared_src, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_chunk_us = ms * 1e3 / (chunks * reps), gbs = 8.0 * P * 1024 / per_chunk_us / 1e3;
    printf("%s P=%d (%3d KB/chunk) wgs=%3d ns=%d %s: %6.3f us/chunk  %6.1f GB/s per CU  %6.2f TB/s chip\n", MODE == 0 ? "dma  " : "plain", P, 8 * P, wgs, ns,
           shared_src ? "shared  " : "distinct", per_chunk_us, gbs, gbs * wgs / 1e3);
}

int main() {
    const unsigned src_bytes = 256u << 20;
    char* src; float* sink;
    hipMalloc(&src, src_bytes); hipMalloc(&sink, 64);
    hipMemset(src, 1, src_bytes);
    for (int shared_src = 1; shared_src >= 0; --shared_src)
        for (int wgs : {64, 256}) {
            run<0, 2>(src, src_bytes, wgs, 2, shared_src, sink);
            run<0, 2>(src, src_bytes, wgs, 6, shared_src, sink);
            run<1, 2>(src, src_bytes, wgs, 2, shared_src, sink);
            run<0, 4>(src, src_bytes, wgs, 2, shared_src, sink);
            run<0, 4>(src, src_bytes, wgs, 4, shared_src, sink);
            run<1, 4>(src, src_bytes, wgs, 2, shared_src, sink);
            run<0, 8>(src, src_bytes, wgs, 2, shared_src, sink);
            run<1, 8>(src, src_bytes, wgs, 2, shared_src, sink);
        }
    return 0;
}

#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// Probe: buffer_load_dwordx4 ... lds with out-of-range lanes -> what lands in LDS?
__global__ void k(const float* src, int nbytes, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 2];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = -7.f;
    __syncthreads();
    auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
    int lane = threadIdx.x;
    int voff = (lane % 3 == 2) ? 0x7ffffff0 : lane * 16;     // every third lane is out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
    // second instruction with an immediate LDS offset via M0 base: second KB
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 256), 16, lane * 16 + 1024, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    float *src, *out; std::vector<float> h(1024), o(512);
    for (int i = 0; i < 1024; ++i) h[i] = i + 1;
    hipMalloc(&src, 4096); hipMalloc(&out, 2048);
    hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
    k<<<1, 64>>>(src, 1024 + 512 /* second load: lanes >= 32 out of range */, out);
    hipMemcpy(o.data(), out, 2048, hipMemcpyDeviceToHost);
    for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
    for (int l = 30; l < 34; ++l) printf("2nd lane %d: %g %g %g %g\n", l, o[256+l*4], o[256+l*4+1], o[256+l*4+2], o[256+l*4+3]);
    return 0;
}

// LDS-DMA (buffer_load ... lds) pieces shared by the convolution kernels whose operands are DMA'd straight into LDS
// (conv_split.hip, conv_up.hip).
#pragma once
#include "conv_common.h"

namespace {

typedef __attribute__((address_space(3))) char lds_char;

// One LDS-DMA piece: 64 lanes x 16 bytes from a buffer resource into LDS at lds_addr + 16 * lane (out-of-range lanes write zeros).
// Issued as inline assembly on purpose: for the builtin form the compiler's wait-count pass cannot tell which LDS stage a later
// ds_read touches and inserts `s_waitcnt vmcnt(0)` in front of the first operand read of every chunk, which drains the whole ring
// (seen in the ISA of the first ring build).  The kernel orders its reads behind the DMA itself: wait_vmcnt + s_barrier.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // (m0 is a reserved register: naming it as clobbered is the point)
__device__ __forceinline__ void dma_piece(u32x4 rsrc, unsigned lds_addr, int voffset, int soffset) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory", "m0");
}
#pragma clang diagnostic pop

__device__ __forceinline__ u32x4 buffer_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);      // stride 0: raw buffer, byte offsets
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}

// s_waitcnt vmcnt(n) for a wave-uniform n (the instruction takes an immediate).  Loads -- LDS-DMA included -- return in order, so
// "at most n outstanding" means everything issued before the youngest n has landed.
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (__builtin_amdgcn_readfirstlane(n)) {
#define IA_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
#define IA_W8(k) IA_W(k) IA_W(k + 1) IA_W(k + 2) IA_W(k + 3) IA_W(k + 4) IA_W(k + 5) IA_W(k + 6) IA_W(k + 7)
        IA_W8(1) IA_W8(9) IA_W8(17) IA_W8(25) IA_W8(33) IA_W8(41) IA_W8(49) IA_W(57) IA_W(58) IA_W(59) IA_W(60) IA_W(61) IA_W(62) IA_W(63)
#undef IA_W8
#undef IA_W
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

}  // namespace

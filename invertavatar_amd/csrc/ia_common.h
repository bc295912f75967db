// Shared host-side helpers for libia_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/ia_hip.h"

namespace ia {

// Range watch of the fp16 hi / lo split (the split clamps at +-65504 silently).  Every translation unit that splits owns
// one device word, set by split_f16 when a value is outside the fp16 range (or not finite) -- a compare per element in kernels that
// are memory-bound, an atomic only when it fires -- and registers a reader; ia_split_saturation_poll sums the words.
struct SatProbe {
    hipError_t (*read)(unsigned int* h_word, int reset, hipStream_t s);
    SatProbe* next;
};
void register_sat_probe(SatProbe* p);

// Thread-local last-error text; besides the range-watch words the only mutable state in the library.
char* error_buffer();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(IA_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return IA_OK;
}

// Opt a kernel in to `bytes` of dynamic LDS.  The attribute call is the capacity check: a device whose workgroups cannot hold the
// tile (a 64 KB-LDS part, where gfx950 has 160 KB) refuses it, and the entry point reports IA_ERR_UNSUPPORTED instead of a
// generic launch failure, so callers with an unfused composition can take it.
inline int reserve_lds(const void* kernel, size_t bytes, const char* what) {
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) return IA_OK;
    (void)hipGetLastError();
    return fail(IA_ERR_UNSUPPORTED, "%s: %zu bytes of LDS per workgroup are not available on this device (%s)", what, bytes,
                hipGetErrorString(e));
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int kWave = 64;        // CDNA wavefront
constexpr int kNumCU = 256;      // MI355X
constexpr int kNumXCD = 8;

// Streaming kernels: enough blocks to fill 256 CUs x 8 blocks, grid-stride beyond that.
inline int streaming_grid(int64_t work_items, int block) {
    int64_t g = ceil_div(work_items, block);
    int64_t cap = (int64_t)kNumCU * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// fp32 <-> storage conversions used by every kernel template.
template <class T> struct Num;
template <> struct Num<float> {
    using compute_t = float;
    __device__ static float load(const float* p) { return *p; }
    __device__ static void store(float* p, float v) { *p = v; }
};
template <> struct Num<__half> {
    using compute_t = float;
    __device__ static float load(const __half* p) { return __half2float(*p); }
    __device__ static void store(__half* p, float v) { *p = __float2half(v); }
};
template <> struct Num<double> {
    using compute_t = double;
    __device__ static double load(const double* p) { return *p; }
    __device__ static void store(double* p, double v) { *p = v; }
};

// fp32 value -> fp16 pair for the "fp32 products from fp16 pairs" convolutions (conv_mfma.hip HM = 2, conv_split.hip):
// hi = fp16(v), lo = fp16((v - hi) * 2^11).  v saturates at the fp16 range; a high part that would be a denormal (flushed by
// the MFMA) is dropped so that the value rides entirely in the scaled low part.  hi + lo * 2^-11 carries 22 mantissa bits.
constexpr float kSplitLoScale = 2048.f;
}  // namespace ia
namespace {
__device__ unsigned int ia_tu_saturated;         // this translation unit's range-watch word
hipError_t ia_tu_read_saturated(unsigned int* h_word, int reset, hipStream_t s) {
    hipError_t e = h_word ? hipMemcpyFromSymbolAsync(h_word, HIP_SYMBOL(ia_tu_saturated), sizeof(unsigned int), 0, hipMemcpyDeviceToHost, s)
                          : hipSuccess;          // (no destination: clear only, nothing for the host to wait for)
    static const unsigned int zero = 0;
    if (e == hipSuccess && reset) e = hipMemcpyToSymbolAsync(HIP_SYMBOL(ia_tu_saturated), &zero, sizeof(unsigned int), 0, hipMemcpyHostToDevice, s);
    return e;
}
ia::SatProbe ia_tu_probe{ia_tu_read_saturated, nullptr};
struct IaTuProbeRegistration { IaTuProbeRegistration() { ia::register_sat_probe(&ia_tu_probe); } } ia_tu_probe_registration;
}  // namespace
namespace ia {
// Range watch of one thread: `see` is a compare and an OR per value (no control flow: a branch with an atomic behind every split
// value broke the producers' loops into ~5 basic blocks per element and made the HBM-bound ones issue-bound, r05 ISA), `report`
// is the thread's one conditional atomic, called once when the thread has split its last value.
struct SatWatch {
    unsigned bad = 0u;
    __device__ __forceinline__ void see(float v) { bad |= (fabsf(v) <= 65504.f) ? 0u : 1u; }      // outside the fp16 range, or NaN
    __device__ __forceinline__ void report() const { if (bad) atomicOr(&ia_tu_saturated, 1u); }
};

// The same interface without a register: every out-of-range value goes to the device word at once (a branch and an atomic per value).
// For the one producer that has no register to spare (the fused-ToRGB convolution epilogue, 256 VGPRs).
struct SatWatchNow {
    __device__ __forceinline__ void see(float v) const { if (!(fabsf(v) <= 65504.f)) atomicOr(&ia_tu_saturated, 1u); }
    __device__ __forceinline__ void report() const {}
};

template <class Watch>
__device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo, Watch& watch) {
    watch.see(v);                                                     // clamped below, and reported by the caller's watch
    v = fminf(fmaxf(v, -65504.f), 65504.f);
    hi = fabsf(v) < 6.103515625e-5f ? (_Float16)0.f : (_Float16)v;
    lo = (_Float16)((v - (float)hi) * kSplitLoScale);
}

// Weight side of the pair form: hi = fp16(v), lo = fp16(v - hi), unscaled (the caller has scaled v so that both are normal numbers).
__device__ __forceinline__ void split_f16_unscaled_lo(float v, _Float16& hi, _Float16& lo) {
    v = fminf(fmaxf(v, -65504.f), 65504.f);
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// One-plane form of the same format (fp16 operands, the arithmetic of the reference's fp16 blocks): the saturated value rounded once.
template <class Watch>
__device__ __forceinline__ _Float16 round_f16(float v, Watch& watch) {
    watch.see(v);
    return (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);
}

}  // namespace ia

#define IA_REQUIRE(cond, ...) \
    do { if (!(cond)) return ia::fail(IA_ERR_INVALID_ARG, __VA_ARGS__); } while (0)

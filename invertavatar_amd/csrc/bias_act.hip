// ia_bias_act: y = clamp(act(x + b) * gain) and its first/second-order gradient forms.
//
// HBM-bound streaming op: 16-byte vector loads/stores per lane, grid-stride over at most
// 256 CUs x 8 workgroups.  Algorithmic traffic = 2 * sizeof(T) bytes per element.
// Semantics follow the reference's definition of the op (torch_utils/ops/bias_act.py:93-122
// for the forward, bias_act.cu:36-150 for the gradient forms); the layout (vector width,
// grid-stride, wave64 blocks) is this library's own.
#include "ia_common.h"

namespace {

template <class S> struct Params {
    const void* x; const void* b; const void* xref; const void* yref; const void* dy; void* y;
    int64_t numel; int size_b; int64_t step_b; int grad;
    S alpha, gain, clamp;
};

// One element.  `v` is x (+bias) for grad 0 and dy-side input for grad > 0; `xr` is xref (+bias).
template <int ACT, class S>
__device__ __forceinline__ S act_eval(int grad, S v, S xr, S& yref, S gain, S alpha) {
    const S one = 1, two = 2;
    const S range = 80, half_range = 40;
    const S selu_s = (S)1.0507009873554804934193349852946;
    const S selu_a = (S)1.6732632423543772848170429916717;
    const S yy = (gain != 0) ? yref / gain : 0;  // un-gained forward output
    S r = 0;
    if (ACT == IA_ACT_LINEAR) {
        if (grad < 2) r = v;
    } else if (ACT == IA_ACT_RELU) {
        if (grad == 0) r = v > 0 ? v : 0;
        else if (grad == 1) r = yy > 0 ? v : 0;
    } else if (ACT == IA_ACT_LRELU) {
        if (grad == 0) r = v > 0 ? v : v * alpha;
        else if (grad == 1) r = yy > 0 ? v : v * alpha;
    } else if (ACT == IA_ACT_TANH) {
        if (grad == 0) {
            S e = exp(v), ie = one / e;
            r = v < -range ? -one : (v > range ? one : (e - ie) / (e + ie));
        } else if (grad == 1) r = v * (one - yy * yy);
        else r = v * (one - yy * yy) * (-two * yy);
    } else if (ACT == IA_ACT_SIGMOID) {
        if (grad == 0) r = v < -range ? 0 : one / (exp(-v) + one);
        else if (grad == 1) r = v * yy * (one - yy);
        else r = v * yy * (one - yy) * (one - two * yy);
    } else if (ACT == IA_ACT_ELU) {
        if (grad == 0) r = v >= 0 ? v : exp(v) - one;
        else if (grad == 1) r = yy >= 0 ? v : v * (yy + one);
        else r = yy >= 0 ? 0 : v * (yy + one);
    } else if (ACT == IA_ACT_SELU) {
        if (grad == 0) r = v >= 0 ? selu_s * v : (selu_s * selu_a) * (exp(v) - one);
        else if (grad == 1) r = yy >= 0 ? v * selu_s : v * (yy + selu_s * selu_a);
        else r = yy >= 0 ? 0 : v * (yy + selu_s * selu_a);
    } else if (ACT == IA_ACT_SOFTPLUS) {
        if (grad == 0) r = v > range ? v : log(exp(v) + one);
        else if (grad == 1) r = v * (one - exp(-yy));
        else { S e = exp(-yy); r = v * e * (one - e); }
    } else if (ACT == IA_ACT_SWISH) {
        if (grad == 0) r = v < -range ? 0 : v / (exp(-v) + one);
        else {
            S e = exp(xr), d = e + one;
            if (grad == 1) r = xr > half_range ? v : v * e * (xr + d) / (d * d);
            else r = xr > half_range ? 0 : v * e * (xr * (two - d) + two * d) / (d * d * d);
            yref = xr < -range ? 0 : xr / (exp(-xr) + one) * gain;  // swish keeps x, not y
        }
    }
    return r;
}

template <class T, int ACT, int VEC>
__global__ __launch_bounds__(256) void bias_act_kernel(Params<typename ia::Num<T>::compute_t> p) {
    using S = typename ia::Num<T>::compute_t;
    const T* x = (const T*)p.x; const T* b = (const T*)p.b;
    const T* xref = (const T*)p.xref; const T* yref = (const T*)p.yref; const T* dy = (const T*)p.dy;
    T* y = (T*)p.y;
    const int64_t nvec = p.numel / VEC;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool bias_uniform = (p.step_b % VEC) == 0;
    for (int64_t vi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; vi < nvec; vi += stride) {
        const int64_t base = vi * VEC;
        T in[VEC], out[VEC], r0[VEC], r1[VEC], r2[VEC];
        // 16-byte (or VEC*sizeof(T)) vector load; the host guarantees alignment when VEC > 1.
        struct alignas(sizeof(T) * VEC) Pack { T v[VEC]; };
        *(Pack*)in = *(const Pack*)(x + base);
        if (xref) *(Pack*)r0 = *(const Pack*)(xref + base);
        if (yref) *(Pack*)r1 = *(const Pack*)(yref + base);
        if (dy) *(Pack*)r2 = *(const Pack*)(dy + base);
        S bias0 = 0;
        if (b && bias_uniform) bias0 = ia::Num<T>::load(b + (base / p.step_b) % p.size_b);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            S bv = bias0;
            if (b && !bias_uniform) bv = ia::Num<T>::load(b + ((base + j) / p.step_b) % p.size_b);
            S v = ia::Num<T>::load(in + j);
            S xr = xref ? ia::Num<T>::load(r0 + j) : (S)0;
            S yr = yref ? ia::Num<T>::load(r1 + j) : (S)0;
            S g = dy ? ia::Num<T>::load(r2 + j) : (S)1;
            if (p.grad == 0) v += bv; else xr += bv;
            S r = act_eval<ACT, S>(p.grad, v, xr, yr, p.gain, p.alpha);
            r *= p.gain * g;
            if (p.clamp >= 0) {
                if (p.grad == 0) r = (r > -p.clamp && r < p.clamp) ? r : (r >= 0 ? p.clamp : -p.clamp);
                else r = (yr > -p.clamp && yr < p.clamp) ? r : (S)0;
            }
            ia::Num<T>::store(out + j, r);
        }
        *(Pack*)(y + base) = *(Pack*)out;
    }
}

template <class T, int VEC>
int launch_act(const Params<typename ia::Num<T>::compute_t>& p, int act, hipStream_t s) {
    const int block = 256;
    const int grid = ia::streaming_grid(p.numel / VEC, block);
#define IA_CASE(A) case A: hipLaunchKernelGGL((bias_act_kernel<T, A, VEC>), dim3(grid), dim3(block), 0, s, p); break;
    switch (act) {
        IA_CASE(IA_ACT_LINEAR) IA_CASE(IA_ACT_RELU) IA_CASE(IA_ACT_LRELU) IA_CASE(IA_ACT_TANH) IA_CASE(IA_ACT_SIGMOID)
        IA_CASE(IA_ACT_ELU) IA_CASE(IA_ACT_SELU) IA_CASE(IA_ACT_SOFTPLUS) IA_CASE(IA_ACT_SWISH)
        default: return ia::fail(IA_ERR_INVALID_ARG, "no kernel found for the specified activation func (act=%d)", act);
    }
#undef IA_CASE
    return ia::check_launch("ia_bias_act");
}

template <class T>
int dispatch(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y,
             int64_t numel, int size_b, int64_t step_b, int grad, int act, float alpha, float gain, float clamp,
             hipStream_t s) {
    using S = typename ia::Num<T>::compute_t;
    Params<S> p{x, b, xref, yref, dy, y, numel, size_b, step_b, grad, (S)alpha, (S)gain, (S)clamp};
    constexpr int VEC = 16 / sizeof(T);
    auto aligned = [](const void* q) { return q == nullptr || ((uintptr_t)q & 15) == 0; };
    const bool vec_ok = (numel % VEC == 0) && aligned(x) && aligned(y) && aligned(xref) && aligned(yref) && aligned(dy);
    return vec_ok ? launch_act<T, VEC>(p, act, s) : launch_act<T, 1>(p, act, s);
}

}  // namespace

extern "C" int ia_bias_act(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y,
                           int dtype, int64_t numel, int size_b, int64_t step_b,
                           int grad, int act, float alpha, float gain, float clamp, void* stream) {
    IA_REQUIRE(x && y, "x and y must be device pointers");
    IA_REQUIRE(numel >= 0 && numel <= INT32_MAX, "x is too large");
    IA_REQUIRE(grad >= 0 && grad <= 2, "grad must be 0, 1 or 2");
    IA_REQUIRE(b == nullptr || (size_b > 0 && step_b > 0), "b has wrong number of elements");
    if (b == nullptr) { size_b = 1; step_b = 1; }
    if (numel == 0) return IA_OK;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case IA_F32: return dispatch<float>(x, b, xref, yref, dy, y, numel, size_b, step_b, grad, act, alpha, gain, clamp, s);
        case IA_F16: return dispatch<__half>(x, b, xref, yref, dy, y, numel, size_b, step_b, grad, act, alpha, gain, clamp, s);
        case IA_F64: return dispatch<double>(x, b, xref, yref, dy, y, numel, size_b, step_b, grad, act, alpha, gain, clamp, s);
        default: return ia::fail(IA_ERR_INVALID_ARG, "unsupported dtype %d", dtype);
    }
}

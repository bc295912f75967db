// Squeeze-and-excitation tail of an IR-SE50 residual unit in two launches:
//     out = v * sigmoid(W2 relu(W1 mean_hw(v))) + shortcut
// Replaces SEModule.forward + the residual add of bottleneck_IR_SE.forward (encoder_inversion/models/helpers.py:84-100, :121-124):
// adaptive average pool, two 1x1 convolutions, ReLU, sigmoid, the broadcast multiply and the add -- seven ATen / library launches
// of a few microseconds each, 24 units per trunk, four trunk passes per few-shot inversion.
//   ia_se_pool       : pooled[b][c] = mean over H x W of v (one workgroup per plane; v may be a strided view: the stride-2 result of a
//                      unit's second convolution is the stride-1 result sub-sampled)
//   ia_se_gate_apply : every workgroup re-derives the gate of its plane from `pooled` (C x C/16 + C/16 multiply-adds: cheaper than a
//                      third launch) and streams out = v * gate + shortcut; the shortcut is a strided view too (MaxPool2d(1, s) = x[::s, ::s])
#include "ia_common.h"

namespace {

struct View { const float* p; int64_t sb, sc, sy, sx; };      // element (b, c, y, x) at p[b*sb + c*sc + y*sy + x*sx]

__global__ __launch_bounds__(256) void se_pool_kernel(View v, float* __restrict__ pooled, int C, int H, int W) {
    __shared__ float part[4];
    const int plane = blockIdx.x, b = plane / C, c = plane - b * C;
    const float* base = v.p + b * v.sb + c * v.sc;
    float acc = 0.f;
    for (int i = threadIdx.x; i < H * W; i += 256) {
        const int y = i / W, x = i - y * W;
        acc += base[y * v.sy + x * v.sx];
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) pooled[plane] = (part[0] + part[1] + part[2] + part[3]) / (float)(H * W);      // fixed order
}

__global__ __launch_bounds__(256) void se_gate_apply_kernel(View v, View sc, const float* __restrict__ pooled, const float* __restrict__ w1,
                                                           const float* __restrict__ w2, float* __restrict__ out, int C, int R, int H, int W) {
    __shared__ float hidden[64];
    __shared__ float gate_s;
    const int plane = blockIdx.x, b = plane / C, c = plane - b * C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // hidden[j] = relu(sum_i w1[j][i] * pooled[b][i]): one wave per hidden unit, round-robin
    for (int j = wave; j < R; j += 4) {
        float a = 0.f;
        for (int i = lane; i < C; i += 64) a = fmaf(w1[(int64_t)j * C + i], pooled[b * C + i], a);
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
        if (lane == 0) hidden[j] = fmaxf(a, 0.f);
    }
    __syncthreads();
    if (tid == 0) {
        float a = 0.f;
        for (int j = 0; j < R; ++j) a = fmaf(w2[(int64_t)c * R + j], hidden[j], a);
        gate_s = 1.f / (1.f + expf(-a));
    }
    __syncthreads();
    const float gate = gate_s;
    const float* vb = v.p + b * v.sb + c * v.sc;
    const float* sb = sc.p + b * sc.sb + c * sc.sc;
    float* ob = out + (int64_t)plane * H * W;
    for (int i = tid + blockIdx.y * 256; i < H * W; i += 256 * gridDim.y) {
        const int y = i / W, x = i - y * W;
        ob[i] = fmaf(vb[y * v.sy + x * v.sx], gate, sb[y * sc.sy + x * sc.sx]);
    }
}

// The same with EIGHT channels per workgroup and, optionally, the result also in split format for the convolution that reads it next:
// ys = split(out * next_scale[b][c] + next_shift[b][c]) -- the eval-mode BatchNorm in front of the NEXT residual unit's first convolution
// (ia_act_split's format, two planes): a trunk is a chain of dependent launches, and the stand-alone split of every unit's input was one
// of seven per unit.  One thread = one pixel of the 8-channel group (strided views: scalar loads, coalesced along x).
typedef _Float16 h16x8_se __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void se_gate_apply8_kernel(View v, View sc, const float* __restrict__ pooled, const float* __restrict__ w1,
                                                            const float* __restrict__ w2, float* __restrict__ out, const float* __restrict__ next_scale,
                                                            const float* __restrict__ next_shift, h16x8_se* __restrict__ ys, int C, int R, int H, int W) {
    __shared__ float hidden[64];
    __shared__ float gate_s[8];
    const int C8 = C / 8, b = blockIdx.x / C8, c8 = blockIdx.x - b * C8, c0 = c8 * 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int j = wave; j < R; j += 4) {
        float a = 0.f;
        for (int i = lane; i < C; i += 64) a = fmaf(w1[(int64_t)j * C + i], pooled[b * C + i], a);
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
        if (lane == 0) hidden[j] = fmaxf(a, 0.f);
    }
    __syncthreads();
    if (tid < 8) {
        float a = 0.f;
        for (int j = 0; j < R; ++j) a = fmaf(w2[(int64_t)(c0 + tid) * R + j], hidden[j], a);
        gate_s[tid] = 1.f / (1.f + expf(-a));
    }
    __syncthreads();
    float gate[8], ns[8], nb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        gate[k] = gate_s[k];
        ns[k] = ys ? next_scale[b * C + c0 + k] : 1.f;
        nb[k] = ys ? next_shift[b * C + c0 + k] : 0.f;
    }
    const int64_t hw = (int64_t)H * W;
    ia::SatWatch watch;
    for (int64_t i = tid + (int64_t)blockIdx.y * 256; i < hw; i += 256 * (int64_t)gridDim.y) {
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        h16x8_se hi, lo;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float vv = v.p[b * v.sb + (c0 + k) * v.sc + y * v.sy + x * v.sx];
            const float ss = sc.p[b * sc.sb + (c0 + k) * sc.sc + y * sc.sy + x * sc.sx];
            const float o = fmaf(vv, gate[k], ss);
            out[((int64_t)b * C + c0 + k) * hw + i] = o;
            if (ys) { _Float16 h_, l_; ia::split_f16(fmaf(o, ns[k], nb[k]), h_, l_, watch); hi[k] = h_; lo[k] = l_; }
        }
        if (ys) {
            ys[((int64_t)(b * 2) * C8 + c8) * hw + i] = hi;
            ys[((int64_t)(b * 2 + 1) * C8 + c8) * hw + i] = lo;
        }
    }
    watch.report();
}

}  // namespace

extern "C" int ia_se_gate_split(const float* v, const int64_t* v_strides, const float* shortcut, const int64_t* shortcut_strides, const float* w1,
                                const float* w2, float* pooled_scratch, float* out, const float* next_scale, const float* next_shift, void* ys,
                                int B, int C, int R, int H, int W, void* stream) {
    IA_REQUIRE(v && v_strides && shortcut && shortcut_strides && w1 && w2 && pooled_scratch && out, "null pointer argument");
    IA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "empty tensor");
    IA_REQUIRE(C % 8 == 0, "the split format stores channels in groups of 8 (C = %d)", C);
    IA_REQUIRE(R >= 1 && R <= 64, "the squeeze width (channels / reduction) must be 1 .. 64");
    IA_REQUIRE((ys == nullptr) == (next_scale == nullptr) && (ys == nullptr) == (next_shift == nullptr), "ys, next_scale and next_shift come together");
    IA_REQUIRE((int64_t)B * C * H * W <= INT32_MAX, "tensor is too large");
    const View vv{v, v_strides[0], v_strides[1], v_strides[2], v_strides[3]};
    const View ss{shortcut, shortcut_strides[0], shortcut_strides[1], shortcut_strides[2], shortcut_strides[3]};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(se_pool_kernel, dim3((unsigned)(B * C)), dim3(256), 0, s, vv, pooled_scratch, C, H, W);
    int st = ia::check_launch("ia_se_gate_split(pool)");
    if (st != IA_OK) return st;
    const int64_t hw = (int64_t)H * W;
    int chunks = (int)((hw + 255) / 256);
    const int want = (2 * ia::kNumCU + B * (C / 8) - 1) / (B * (C / 8));      // ~2 workgroups per CU over the launch
    if (chunks > want) chunks = want < 1 ? 1 : want;
    hipLaunchKernelGGL(se_gate_apply8_kernel, dim3((unsigned)(B * (C / 8)), (unsigned)chunks), dim3(256), 0, s, vv, ss, pooled_scratch, w1, w2, out,
                       next_scale, next_shift, static_cast<h16x8_se*>(ys), C, R, H, W);
    return ia::check_launch("ia_se_gate_split(apply)");
}

extern "C" int ia_se_gate(const float* v, const int64_t* v_strides, const float* shortcut, const int64_t* shortcut_strides, const float* w1,
                          const float* w2, float* pooled_scratch, float* out, int B, int C, int R, int H, int W, void* stream) {
    IA_REQUIRE(v && v_strides && shortcut && shortcut_strides && w1 && w2 && pooled_scratch && out, "null pointer argument");
    IA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "empty tensor");
    IA_REQUIRE(R >= 1 && R <= 64, "the squeeze width (channels / reduction) must be 1 .. 64");
    IA_REQUIRE((int64_t)B * C <= INT32_MAX / 2, "too many planes");
    const View vv{v, v_strides[0], v_strides[1], v_strides[2], v_strides[3]};
    const View ss{shortcut, shortcut_strides[0], shortcut_strides[1], shortcut_strides[2], shortcut_strides[3]};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(se_pool_kernel, dim3((unsigned)(B * C)), dim3(256), 0, s, vv, pooled_scratch, C, H, W);
    int st = ia::check_launch("ia_se_gate(pool)");
    if (st != IA_OK) return st;
    const int chunks = H * W > 16384 ? 4 : 1;      // (few planes x many pixels: the 64-channel 128^2 units)
    hipLaunchKernelGGL(se_gate_apply_kernel, dim3((unsigned)(B * C), (unsigned)chunks), dim3(256), 0, s, vv, ss, pooled_scratch, w1, w2, out, C, R, H, W);
    return ia::check_launch("ia_se_gate(apply)");
}

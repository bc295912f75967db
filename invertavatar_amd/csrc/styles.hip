// ia_styles_demod: every affine style vector and every demodulation coefficient of one synthesis network in two
// launches (instead of one FullyConnectedLayer GEMV + one demod launch per layer).
//
//   styles_l[b, i]        = dot(ws[b, widx_l, :], A_l[i, :]) * wgain_l + bias_l[i] * bgain_l      (SynthesisLayer / ToRGBLayer
//       .affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1): training/networks_stylegan2.py:96-127,298,345)
//   demod[b, doff_l + o]  = rsqrt( sum_i styles[b, soff_l + i]^2 * wsq_l[o, i] + 1e-8 )            (:60-64)
//
// Layers are described by a caller-owned device table (int64 [L][8]: A, bias, wsq, I, O, widx, soff, doff; float [L][2]:
// wgain, bgain) plus two row->layer maps.  Both kernels are weight-streaming (one wave per output row, 16-byte loads):
// ~18 MB of affine weights and ~13 MB of wsq per backbone, i.e. a few microseconds at HBM rate.
#include "ia_common.h"

namespace {

struct Table {
    const int64_t* layers;    // [L][8]
    const float* gains;       // [L][2]
};

__global__ __launch_bounds__(256) void styles_kernel(const float* __restrict__ ws, Table t, const int* __restrict__ row_layer,
                                                     float* __restrict__ styles, int B, int num_ws, int w_dim, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int l = row_layer[row];
    const int64_t* d = t.layers + (int64_t)l * 8;
    const float* A = (const float*)d[0];
    const float* bias = (const float*)d[1];
    const int I = (int)d[3], widx = (int)d[5], soff = (int)d[6];
    const int i = row - soff;
    const float wgain = t.gains[2 * l], bgain = t.gains[2 * l + 1];
    const float* a = A + (int64_t)i * w_dim;
    for (int b = 0; b < B; ++b) {
        const float* w = ws + ((int64_t)b * num_ws + widx) * w_dim;
        float acc = 0.f;
        for (int k = lane * 4; k < w_dim; k += 256) {
            const float4 av = *(const float4*)(a + k), wv = *(const float4*)(w + k);
            acc = fmaf(av.x, wv.x, acc); acc = fmaf(av.y, wv.y, acc); acc = fmaf(av.z, wv.z, acc); acc = fmaf(av.w, wv.w, acc);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) styles[(int64_t)B * soff + (int64_t)b * I + i] = acc * wgain + (bias ? bias[i] * bgain : 0.f);   // layer-major: [l][B][I]
    }
}

__global__ __launch_bounds__(256) void demod_rows_kernel(const float* __restrict__ styles, Table t, const int* __restrict__ row_layer,
                                                         float* __restrict__ demod, int B, int srows, int drows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= drows) return;
    const int l = row_layer[row];
    const int64_t* d = t.layers + (int64_t)l * 8;
    const float* wsq = (const float*)d[2];
    const int I = (int)d[3], O = (int)d[4], soff = (int)d[6], doff = (int)d[7];
    const int o = row - doff;
    const float* q = wsq + (int64_t)o * I;
    for (int b = 0; b < B; ++b) {
        const float* s = styles + (int64_t)B * soff + (int64_t)b * I;
        float acc = 0.f;
        for (int k = lane; k < I; k += 64) { const float sv = s[k]; acc = fmaf(sv * sv, q[k], acc); }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) demod[(int64_t)B * doff + (int64_t)b * O + o] = 1.0f / sqrtf(acc + 1e-8f);
    }
}

}  // namespace

extern "C" int ia_styles_demod(const float* ws, int B, int num_ws, int w_dim, const int64_t* layer_table, const float* layer_gains,
                               const int* style_row_layer, int style_rows, const int* demod_row_layer, int demod_rows,
                               float* styles, float* demod, void* stream) {
    IA_REQUIRE(ws && layer_table && layer_gains && style_row_layer && styles, "null pointer argument");
    IA_REQUIRE(B > 0 && num_ws > 0 && w_dim > 0 && w_dim % 4 == 0 && style_rows > 0, "bad dimensions");
    IA_REQUIRE(demod_rows == 0 || (demod_row_layer && demod), "demodulation outputs requested without buffers");
    Table t{layer_table, layer_gains};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(styles_kernel, dim3((style_rows + 3) / 4), dim3(256), 0, s, ws, t, style_row_layer, styles, B, num_ws, w_dim, style_rows);
    int st = ia::check_launch("ia_styles_demod(styles)");
    if (st != IA_OK || demod_rows == 0) return st;
    hipLaunchKernelGGL(demod_rows_kernel, dim3((demod_rows + 3) / 4), dim3(256), 0, s, styles, t, demod_row_layer, demod, B, style_rows, demod_rows);
    return ia::check_launch("ia_styles_demod(demod)");
}

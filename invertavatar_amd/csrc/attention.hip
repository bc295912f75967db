// ia_attention: multi-head self-attention of the transformer-refined UNet decoders (eval_updated_os.py encoders) in one launch:
//     out = softmax(Q K^T * scale) V      per (batch, head)
// Replaces Attention.forward of encoder_inversion/models/mmseg/mix_transformer.py:83-116 between the q / kv projections and the
// output projection (sr_ratio = 1: `attn = (q @ k.transpose(-2, -1)) * self.scale; attn = attn.softmax(dim=-1); x = attn @ v`),
// i.e. two batched GEMMs, a softmax over an [N, N] matrix per head (268 MB at N = 4096) and two layout copies in ATen.
//
// fp32 operands, fp32 accumulation on v_mfma_f32_32x32x2_f32: the arithmetic of the reference (no reduced-precision products).
// A workgroup of 4 waves owns 32 query rows of one (batch, head) and walks the keys in tiles of 32 with an online softmax, so the
// [N, N] score matrix never exists.  Both products are formed TRANSPOSED, which makes every per-query quantity per-LANE:
//   S^T tile = K_tile Q^T   (A = K rows from LDS, B = Q rows from LDS; the head dimension is split over the 4 waves and the partial
//                            tiles are added through LDS in wave order): lane (q, half) holds 16 keys of its query -> row max / sum
//                            are in-lane reductions + one cross-half shuffle;
//   O^T tile += V_tile^T P^T (A = V straight from global memory, coalesced along the head dimension; B = P^T = the registers S^T
//                            just left: no transpose, no LDS round trip; each wave owns hd/4 output dimensions).
// Key order inside a tile follows the MFMA's register layout (register r of half h is key (r & 3) + 8 (r >> 2) + 4 h); the sum over
// keys is associative up to rounding, the order is fixed, results are run-to-run identical.
#include "ia_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kQT = 32, kKT = 32, kAttWaves = 4;

struct AttnParams {
    const float* q; const float* k; const float* v; float* out;
    int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;      // batch / row strides (floats); head h starts at column h * hd
    int N, M, hd, heads;
    float scale;
};

__device__ __forceinline__ int key_of(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int HD>
__global__ __launch_bounds__(kAttWaves * 64) void attention_kernel(AttnParams p) {
    constexpr int LD = HD + 1;                          // LDS row stride: rows of a tile land on different banks
    constexpr int DW = HD / kAttWaves;                  // head dimensions per wave (S^T: its slice of the contraction; O^T: its output rows)
    constexpr int NF = DW / 32;                         // 32-row fragments of O^T per wave
    extern __shared__ float lds[];
    float* Qs = lds;                                    // [32][LD]
    float* Ks = Qs + kQT * LD;                          // [32][LD]
    float* Sp = Ks + kKT * LD;                          // [4 waves][16][64]: partial S^T tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int q0 = qt * kQT;
    const float* qb = p.q + b * p.q_bs + (int64_t)q0 * p.q_rs + h * HD;
    const float* kb = p.k + b * p.k_bs + h * HD;
    const float* vb = p.v + b * p.v_bs + h * HD;

    // Q tile -> LDS (rows beyond N: zeros; their results are never stored)
    for (int i = tid; i < kQT * (HD / 4); i += kAttWaves * 64) {
        const int r = i / (HD / 4), c4 = i - r * (HD / 4);
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q0 + r < p.N) val = *reinterpret_cast<const float4*>(qb + (int64_t)r * p.q_rs + 4 * c4);
        float* d = Qs + r * LD + 4 * c4;
        d[0] = val.x; d[1] = val.y; d[2] = val.z; d[3] = val.w;
    }
    // first K tile into registers (8 float4 per thread), committed to LDS at the top of the loop
    constexpr int KV4 = kKT * (HD / 4) / (kAttWaves * 64);
    float4 kreg[KV4];
    auto load_k = [&](int key0) {
#pragma unroll
        for (int j = 0; j < KV4; ++j) {
            const int i = tid + j * kAttWaves * 64;
            const int r = i / (HD / 4), c4 = i - r * (HD / 4);
            kreg[j] = key0 + r < p.M ? *reinterpret_cast<const float4*>(kb + (int64_t)(key0 + r) * p.k_rs + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    load_k(0);

    f32x16 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;               // running max / sum of the lane's query (both halves keep the same values)

    for (int key0 = 0; key0 < p.M; key0 += kKT) {
        __syncthreads();                                // the previous tile's readers of Ks / Sp are done (first pass: Q stores)
#pragma unroll
        for (int j = 0; j < KV4; ++j) {
            const int i = tid + j * kAttWaves * 64;
            const int r = i / (HD / 4), c4 = i - r * (HD / 4);
            float* d = Ks + r * LD + 4 * c4;
            d[0] = kreg[j].x; d[1] = kreg[j].y; d[2] = kreg[j].z; d[3] = kreg[j].w;
        }
        __syncthreads();
        if (key0 + kKT < p.M) load_k(key0 + kKT);       // next tile's loads fly under this tile's arithmetic
        // ---- partial S^T over this wave's slice of the head dimension: D[key][query]
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
        const float* ka = Ks + l31 * LD + wave * DW + half;
        const float* qa = Qs + l31 * LD + wave * DW + half;
#pragma unroll 8
        for (int d = 0; d < DW; d += 2) st = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[d], qa[d], st, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) Sp[(wave * 16 + r) * 64 + lane] = st[r];
        __syncthreads();
        float s[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = Sp[(0 * 16 + r) * 64 + lane];
#pragma unroll
            for (int w = 1; w < kAttWaves; ++w) v += Sp[(w * 16 + r) * 64 + lane];
            s[r] = key0 + key_of(r, half) < p.M ? v * p.scale : -INFINITY;
        }
        // ---- online softmax of the lane's query over the tile's 32 keys (16 here, 16 in the other half-wave)
        float tmax = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);
        const float fac = expf(m_run - m_new);        // (m_run = -inf on the first tile: fac = 0, the accumulators are zero anyway)
        float tsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = expf(s[r] - m_new); tsum += s[r]; }
        tsum += __shfl_xor(tsum, 32);
        l_run = l_run * fac + tsum;
        m_run = m_new;
        // ---- O^T[dv][query] = O^T * fac + V_tile^T P^T: k-step r pairs key_of(r, 0) (lanes 0-31) with key_of(r, 1) (lanes 32-63)
#pragma unroll
        for (int f = 0; f < NF; ++f) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[f][r] *= fac;
            const float* va = vb + wave * DW + f * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + key_of(r, half);
                const float a = key < p.M ? va[(int64_t)key * p.v_rs] : 0.f;
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[r], acc[f], 0, 0, 0);
            }
        }
    }
    // ---- out[b][q][h * hd + dv] = O^T[dv][q] / l: register quad g of fragment f = 4 consecutive output dimensions
    if (q0 + l31 >= p.N) return;
    const float inv = 1.f / l_run;
    float* ob = p.out + b * p.o_bs + (int64_t)(q0 + l31) * p.o_rs + h * HD + wave * DW;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int dv = f * 32 + 8 * g + 4 * half;
            *reinterpret_cast<float4*>(ob + dv) = make_float4(acc[f][4 * g] * inv, acc[f][4 * g + 1] * inv, acc[f][4 * g + 2] * inv, acc[f][4 * g + 3] * inv);
        }
}

}  // namespace

// Measured r03 (tools/bench_attention.py, B = 1, 4 heads): 21 vs 66 us (ATen: two batched matmuls + softmax + permutes) at 64 tokens,
// 65 vs 139 us at 256; at 1024 / 4096 tokens the library GEMMs win (375 vs 79 us, 2964 vs 790 us: this kernel's per-tile barriers and
// fp32 MFMAs do not scale) -- so `supported` is the token range where the launch-bound ATen sequence loses.
extern "C" int ia_attention_supported(int head_dim, int N, int M) { return head_dim == 256 && N > 0 && M > 0 && (int64_t)N * M <= 131072; }

extern "C" int ia_attention(const float* q, const float* k, const float* v, float* out, int B, int heads, int N, int M, int head_dim,
                            int64_t q_batch_stride, int64_t q_row_stride, int64_t k_batch_stride, int64_t k_row_stride,
                            int64_t v_batch_stride, int64_t v_row_stride, int64_t out_batch_stride, int64_t out_row_stride,
                            float scale, void* stream) {
    IA_REQUIRE(q && k && v && out, "null pointer argument");
    IA_REQUIRE(B > 0 && heads > 0 && N > 0 && M > 0, "empty tensor");
    if (!ia_attention_supported(head_dim, N, M))
        return ia::fail(IA_ERR_UNSUPPORTED, "ia_attention covers head_dim 256 (the 1024-dim / 4-head blocks of transformer_block) up to N * M = 131072 "
                        "(got head_dim %d, N %d, M %d)", head_dim, N, M);
    IA_REQUIRE(q_row_stride % 4 == 0 && k_row_stride % 4 == 0 && out_row_stride % 4 == 0 && q_batch_stride % 4 == 0 && k_batch_stride % 4 == 0 &&
               out_batch_stride % 4 == 0, "row / batch strides must keep 16-byte alignment");
    IA_REQUIRE(heads <= 65535 && B <= 65535, "too many heads / batch elements for one launch");
    AttnParams p{q, k, v, out, q_batch_stride, q_row_stride, k_batch_stride, k_row_stride, v_batch_stride, v_row_stride, out_batch_stride,
                 out_row_stride, N, M, head_dim, heads, scale};
    constexpr int HD = 256;
    const size_t lds = sizeof(float) * (size_t)(kQT * (HD + 1) + kKT * (HD + 1) + kAttWaves * 16 * 64);
    auto kern = attention_kernel<HD>;
    if (const int rs = ia::reserve_lds((const void*)kern, lds, "attention")) return rs;
    const dim3 grid((unsigned)((N + kQT - 1) / kQT), (unsigned)heads, (unsigned)B);
    hipLaunchKernelGGL(kern, grid, dim3(kAttWaves * 64), lds, (hipStream_t)stream, p);
    return ia::check_launch("ia_attention");
}

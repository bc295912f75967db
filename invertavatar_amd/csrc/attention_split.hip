// ia_attention_sx: softmax(Q K^T * scale) V of the transformer blocks on the LARGE token grids (32^2 / 64^2 tokens, head_dim 256) in one
// launch, the score matrix never in memory (reference: encoder_inversion/models/mmseg/mix_transformer.py:103-115 -- q @ k^T * scale,
// softmax(dim=-1), attn @ v, transpose(1, 2).reshape(B, N, C): two batched library GEMMs, a scale, a softmax over [heads, N, M], permutes).
//
// Arithmetic: both products are fp32-equivalent GEMMs on the fp16 pipe like ia_matmul_sx (fp16 hi / lo pairs, three
// v_mfma_f32_32x32x16_f16 per k-step, hi x hi in one fp32 accumulator, the two cross terms -- both at 2^11 -- in a second one), the
// softmax is exact (maximum first, then exp and sum in the pass that accumulates the output).  Operands: ia_tokens_split(q), ia_tokens_split(k) and V split
// ALONG THE KEYS in accumulator-row order (ia_tokens_split_t with perm = 1).
//
// Layout trick that removes every transposition: a wave computes S^T = K Q^T for 32 keys x its 32 queries -- the MFMA's columns (lanes)
// are QUERIES, so all keys of a query sit in the registers of two lanes (l and l + 32): the row maximum / sum are register reductions
// plus one cross-half exchange, and the probabilities of a lane ARE a B fragment of the second product O^T = V^T P^T (k = keys, columns =
// queries) as they stand: registers 8 s .. 8 s + 7 of the C layout are keys {0..3, 8..11} + 4 half + 16 s of the block, which is exactly
// the key order ia_tokens_split_t(perm = 1) gives the octets of V^T.  O^T lives in 2 x 8 accumulator fragments (256 features x 32
// queries, both accumulators) and is never rescaled: a first pass over the keys finds every query's maximum (one more K Q^T).
// A workgroup owns 32 queries of one head; its four waves take a quarter of the keys each (one wave per SIMD: every SIMD of the machine
// works at 4 096 tokens, 128 workgroups at 1 024), share the queries' fragments in LDS (32 KB, loaded once), exchange their maxima after
// pass 1 and add their partial outputs through LDS in wave order at the end.  K and V stream from L2 through register rings.
#include "ia_common.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

constexpr int kOut = 0x7ffffff0;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t att_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ h16x8 att_load16(__amdgpu_buffer_rsrc_t r, int voffset, int soffset) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voffset, soffset, 0);
    return __builtin_bit_cast(h16x8, v);
}

struct AttParams {
    const h16x8* qs;     // [2][C8][N][8]
    const h16x8* ks;     // [2][C8][M][8]
    const h16x8* vp;     // [2][M/8][C][8], octets in accumulator-row key order
    float* out;          // [N][C]
    int N, M, C8;
    float scale;
};

struct NoWatchA { __device__ __forceinline__ void see(float) const {} };

template <int HD8>      // octets per head (32: head_dim 256)
__global__ __launch_bounds__(256) void attention_sx_kernel(AttParams p) {
    constexpr int STEPS = HD8 / 2, NF = HD8 / 4;          // k-steps of the first product; 32-feature fragments of a head
    constexpr int KRING = 4;
    extern __shared__ __attribute__((aligned(16))) h16x8 s_q[];      // [2 planes][STEPS][64 lanes] Q fragments (later: the partial outputs, 128 KB) | statistics
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, half = lane >> 5;
    const int h = blockIdx.y, q0 = blockIdx.x * 32;
    const int C = p.C8 * 8, M8 = p.M >> 3;
    const unsigned q_bytes = (unsigned)p.C8 * (unsigned)p.N * 16u, k_bytes = (unsigned)p.C8 * (unsigned)p.M * 16u, v_bytes = (unsigned)M8 * (unsigned)C * 16u;
    const char* qb = reinterpret_cast<const char*>(p.qs);
    const char* kb_ = reinterpret_cast<const char*>(p.ks);
    const char* vb = reinterpret_cast<const char*>(p.vp);
    const __amdgpu_buffer_rsrc_t rq0 = att_rsrc(qb, q_bytes), rq1 = att_rsrc(qb + q_bytes, q_bytes);
    const __amdgpu_buffer_rsrc_t rk0 = att_rsrc(kb_, k_bytes), rk1 = att_rsrc(kb_ + k_bytes, k_bytes);
    const __amdgpu_buffer_rsrc_t rv0 = att_rsrc(vb, v_bytes), rv1 = att_rsrc(vb + v_bytes, v_bytes);

    // ---- Q fragments of the workgroup's 32 queries -> LDS, [plane][step][lane]: wave w fetches steps w, w + 4, ...; every wave reads all
    h16x8* sq = s_q + lane;
    {
        const int q = q0 + l31;
        const int qo = q < p.N ? (half * p.N + q) * 16 : kOut;
#pragma unroll
        for (int i = 0; i < STEPS / 4; ++i) {
            const int st = wave + 4 * i;
            const int so = (h * HD8 + 2 * st) * p.N * 16;
            sq[(0 * STEPS + st) * 64] = att_load16(rq0, qo, so);
            sq[(1 * STEPS + st) * 64] = att_load16(rq1, qo, so);
        }
    }
    __syncthreads();

    const int nblk_all = (p.M + 31) >> 5, per = (nblk_all + 3) >> 2;
    const int blk0 = min(wave * per, nblk_all), nblk = min(blk0 + per, nblk_all) - blk0;      // this wave's key blocks [blk0, blk0 + nblk)
    const int total = nblk * STEPS;
    f32x16 oa[NF], ox[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) { oa[f][r] = 0.f; ox[f][r] = 0.f; }
    // K ring: flattened step g = blk * STEPS + st
    h16x8 ka[KRING][2];
    auto k_load = [&](int slot, int g, bool live, bool lo_too) {
        const int blk = g / STEPS, st = g - blk * STEPS;
        const int row = (blk0 + blk) * 32 + l31;
        const int off = (live && row < p.M) ? (half * p.M + row) * 16 : kOut;
        const int so = (h * HD8 + 2 * st) * p.M * 16;
        ka[slot][0] = att_load16(rk0, off, so);
        if (lo_too) ka[slot][1] = att_load16(rk1, off, so);
    };
    // V ring: ONE 16-key step of V^T fragments (NF x 2 planes); slot f is refilled with the next step's fragment right behind its MFMAs, so
    // a refill has a whole step (and, across blocks, the S phase) to land
    h16x8 va[NF][2];
    const int v_lane = (h * HD8 * 8 + l31) * 16;                     // feature (row of V^T) of this lane inside an octet row of C features
    const int nstep = nblk * 2;
    auto v_load = [&](int f, int t, bool live) {                     // step t = 2 blk + s2: octet 2 t + half
        const int oct = 4 * blk0 + 2 * t + half;
        const int off = (live && oct < M8) ? v_lane + (oct * C + f * 32) * 16 : kOut;
        va[f][0] = att_load16(rv0, off, 0);
        va[f][1] = att_load16(rv1, off, 0);
    };
    auto k_prologue = [&](bool lo_too) {
#pragma unroll
        for (int s = 0; s < KRING; ++s) {
            k_load(s, s < total ? s : 0, s < total, lo_too);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // S^T = K Q^T for the 32 keys of block blk, scaled, keys past M at -inf: this lane's 16 keys of query l31 (the other 16 are in lane ^ 32)
    // full = false (pass 1): the hi x hi product alone -- the reference value of a softmax need not be the exact maximum
    auto s_block = [&](int blk, float (&sv)[16], auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        f32x16 sa, sx;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = 0.f; sx[r] = 0.f; }
        h16x8 qn[2][2];                                              // Q fragments of the step after the one being multiplied
        qn[0][0] = sq[(0 * STEPS + 0) * 64];
        if constexpr (FULL) qn[0][1] = sq[(1 * STEPS + 0) * 64];
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int slot = st % KRING;
            if (st + 1 < STEPS) {
                qn[(st + 1) & 1][0] = sq[(0 * STEPS + st + 1) * 64];
                if constexpr (FULL) qn[(st + 1) & 1][1] = sq[(1 * STEPS + st + 1) * 64];
            }
            const h16x8 b_hi = qn[st & 1][0];
            sa = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka[slot][0], b_hi, sa, 0, 0, 0);
            if constexpr (FULL) {
                const h16x8 b_lo = qn[st & 1][1];
                sx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka[slot][0], b_lo, sx, 0, 0, 0);
                sx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka[slot][1], b_hi, sx, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int g = blk * STEPS + st + KRING;
            k_load(slot, g < total ? g : 0, g < total, FULL);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (blk0 + blk) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            sv[r] = key < p.M ? (sa[r] + sx[r] * (1.0f / 2048.0f)) * p.scale : -INFINITY;
        }
    };

    // ---- pass 1: the maximum score of every query.  (The one-pass form rescales O^T by exp(m_old - m_new) whenever a maximum moves; with
    // the accumulators in AGPRs that rescale -- conditional or not -- cost the register allocator 90 - 280 spilled registers.  The first
    // pass needs only a reference value near the maximum, so it runs the hi x hi product alone: 16 of a block's 160 MFMAs, and the accumulators are then never touched by the vector ALU.)
    k_prologue(false);
    float m_run = -INFINITY;
    for (int blk = 0; blk < nblk; ++blk) {
        float sv[16];
        s_block(blk, sv, std::false_type{});
#pragma unroll
        for (int r = 0; r < 16; ++r) m_run = fmaxf(m_run, sv[r]);
    }
    // the four waves hold the maxima of their quarters: exchange through LDS
    float* s_stat = reinterpret_cast<float*>(reinterpret_cast<char*>(s_q) + 131072);      // behind the partial-output region: [2][4 waves][64 lanes] maxima, sums
    s_stat[wave * 64 + lane] = fmaxf(m_run, __shfl_xor(m_run, 32));
    __syncthreads();
    const float m_q = fmaxf(fmaxf(s_stat[lane], s_stat[64 + lane]), fmaxf(s_stat[128 + lane], s_stat[192 + lane]));

    // ---- pass 2: p = exp(s - max), O^T += V^T P^T, l += sum p
    k_prologue(true);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        v_load(f, 0, true);
        __builtin_amdgcn_sched_barrier(0);
    }
    NoWatchA nw;
    float l_run = 0.f;
    for (int blk = 0; blk < nblk; ++blk) {
        float sv[16];
        s_block(blk, sv, std::true_type{});
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sv[r] = expf(sv[r] - m_q);
            l_run += sv[r];
        }
        // two 16-key steps; registers 8 s2 .. 8 s2 + 7 are this lane's B fragment
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            h16x8 ph, pl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                _Float16 hh, ll;
                ia::split_f16(sv[8 * s2 + j], hh, ll, nw);
                ph[j] = hh;
                pl[j] = ll;
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                oa[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va[f][0], ph, oa[f], 0, 0, 0);
                ox[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va[f][0], pl, ox[f], 0, 0, 0);
                ox[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va[f][1], ph, ox[f], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int tn = 2 * blk + s2 + 1;
                v_load(f, tn < nstep ? tn : 0, tn < nstep);          // the slot is free: the next step's fragment goes out now
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- the partial outputs of the four waves meet in LDS (over the Q fragments, which nobody reads any more) and are added in wave order;
    // wave w finishes fragments 2 w, 2 w + 1: O[q][h * hd + feature] = sum_w (oa + ox * 2^-11) / sum_w l
    s_stat[256 + wave * 64 + lane] = l_run + __shfl_xor(l_run, 32);
    __syncthreads();                                             // every wave is past its last Q read
    float4* part = reinterpret_cast<float4*>(s_q);                   // [4 waves][NF][4 quads][64 lanes] float4 = 128 KB, the Q fragments' 32 KB included
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            float4 v;
            v.x = oa[f][4 * rq] + ox[f][4 * rq] * (1.0f / 2048.0f);
            v.y = oa[f][4 * rq + 1] + ox[f][4 * rq + 1] * (1.0f / 2048.0f);
            v.z = oa[f][4 * rq + 2] + ox[f][4 * rq + 2] * (1.0f / 2048.0f);
            v.w = oa[f][4 * rq + 3] + ox[f][4 * rq + 3] * (1.0f / 2048.0f);
            part[((wave * NF + f) * 4 + rq) * 64 + lane] = v;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    const float l_tot = (s_stat[256 + lane] + s_stat[320 + lane]) + (s_stat[384 + lane] + s_stat[448 + lane]);
    const float inv = 1.f / l_tot;
    const int q = q0 + l31;
    if (q >= p.N) return;
    float* ob = p.out + (int64_t)q * C + h * HD8 * 8 + 4 * half;
#pragma unroll
    for (int i = 0; i < NF / 4; ++i) {
        const int f = wave * (NF / 4) + i;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            float4 v = part[((0 * NF + f) * 4 + rq) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 u = part[((w * NF + f) * 4 + rq) * 64 + lane];
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
            *reinterpret_cast<float4*>(ob + f * 32 + 8 * rq) = v;
        }
    }
}

}  // namespace

extern "C" int ia_attention_sx_supported(int head_dim, int N, int M) {
    return head_dim == 256 && N > 0 && M > 0 && M % 16 == 0;
}

extern "C" int ia_attention_sx(const void* q_split, const void* k_split, const void* v_split_t, float* out, int heads, int N, int M, int head_dim,
                               float scale, void* stream) {
    IA_REQUIRE(q_split && k_split && v_split_t && out, "q_split, k_split, v_split_t and out must be device pointers");
    IA_REQUIRE(heads > 0 && heads <= 65535 && N > 0 && M > 0, "empty attention");
    if (!ia_attention_sx_supported(head_dim, N, M))
        return ia::fail(IA_ERR_UNSUPPORTED, "ia_attention_sx covers head_dim 256 and M %% 16 == 0 (got head_dim %d, M %d)", head_dim, M);
    const int64_t C = (int64_t)heads * head_dim;
    IA_REQUIRE(C * N <= (int64_t)1 << 30 && C * M <= (int64_t)1 << 30, "token matrices too large for 32-bit plane offsets");
    AttParams p{static_cast<const h16x8*>(q_split), static_cast<const h16x8*>(k_split), static_cast<const h16x8*>(v_split_t), out, N, M, (int)(C / 8), scale};
    const size_t lds = (size_t)4 * 8 * 4 * 64 * 16 /* partial outputs, over the Q fragments */ + 2048 /* statistics */;
    auto k = attention_sx_kernel<32>;
    if (const int st = ia::reserve_lds((const void*)k, lds, "ia_attention_sx")) return st;
    hipLaunchKernelGGL(k, dim3((unsigned)((N + 31) / 32), (unsigned)heads), dim3(256), lds, (hipStream_t)stream, p);
    return ia::check_launch("ia_attention_sx");
}

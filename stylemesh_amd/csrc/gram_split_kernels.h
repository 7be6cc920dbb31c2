// K5a-s / K5c-s: the Gram contraction S = (mF)(mF)^T and its backward GEMM dF = m0 (D0 F) + m1 (D1 F) on the fp16
// matrix cores at fp32 accuracy - the same fp16x2 split as conv_split_kernel.h (x s = h + l, three partial products
// per fp32 product, fp32 accumulate; round 1's bf16x3 form left in round 6). Replace the same reference operators as gram_masked_kernel /
// gram_backward_kernel (bool-mask gather + torch.bmm and its backward, content_and_style_losses.py:74-80,136-143).
//
// Both kernels run the same pipeline per K stage: global loads into one of two register sets - re-issued straight
// after the set's previous contents were converted and stored, i.e. two stages ahead of their own conversion -,
// fp32 -> operand-part conversion while storing into one of two LDS buffers one stage ahead, fragment reads + MFMAs on
// the current buffer, one barrier per stage.
#pragma once
#include <type_traits>

#include "common.h"

// resident waves per SIMD the register budgets are set for (A/B: tools/build_variant.sh -f gram -DSM_GRAM_FWD_W1=4 ...)
#ifndef SM_GRAM_FWD_W1
#define SM_GRAM_FWD_W1 3   // grouped forward, 64-channel tiles: 4 waves per SIMD (128 VGPRs) spilled 13 registers (round 4's
#endif                     // verdict); 3 (168) fits without scratch - A/B in profiles/r05/gram_fwd_waves_ab.txt
#ifndef SM_GRAM_FWD_W2
#define SM_GRAM_FWD_W2 2
#endif
#ifndef SM_GRAM_BWD_W1
#define SM_GRAM_BWD_W1 2
#endif
#ifndef SM_GRAM_BWD_W2
#define SM_GRAM_BWD_W2 2
#endif

namespace sm {

constexpr int GRAM_MAX_GROUP = 24;   // problems per grouped launch (kernel-argument tables)

// fp16 x 2 operands scaled by a power of two from the tensor's recorded max |x| (conv_split_kernel.h), three partial
// products. Fragments travel as raw 16-byte units.
constexpr int GNP = 2;   // parts per operand
typedef _Float16 g_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float gram_pow2_scale(float amax, float& inv) {
    const unsigned bits = __builtin_bit_cast(unsigned, amax);
    const int ex = (int)((bits >> 23) & 0xff);
    if (ex < 16 || ex > 250) { inv = 1.f; return 1.f; }
    inv = __builtin_bit_cast(float, (unsigned)(ex - 14) << 23);
    return __builtin_bit_cast(float, (unsigned)(268 - ex) << 23);
}
// eight fp32 values (already scaled) -> h, l fp16x8 units
__device__ __forceinline__ void split2x8(const float (&x)[8], f32x4& vh, f32x4& vl) {
    g_f16x8 h, l;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float xs = __builtin_amdgcn_fmed3f(x[c], -65000.f, 65000.f);
        const _Float16 a = (_Float16)xs;
        h[c] = a;
        l[c] = (_Float16)(xs - (float)a);
    }
    vh = __builtin_bit_cast(f32x4, h);
    vl = __builtin_bit_cast(f32x4, l);
}
__device__ __forceinline__ void mfma_parts(f32x16& acc, const f32x4 (&fa)[GNP], const f32x4 (&fb)[GNP]) {
#define SM_H(x_) __builtin_bit_cast(g_f16x8, x_)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(SM_H(fa[1]), SM_H(fb[0]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(SM_H(fa[0]), SM_H(fb[1]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(SM_H(fa[0]), SM_H(fb[0]), acc, 0, 0, 0);
#undef SM_H
}
// eight fp32 values times eight per-position factors (the operand scale where the 0/1 mask is set, 0 elsewhere) -> h, l
// fp16x8 units. No clamp: every position with a non-zero factor is live data below the recorded bound, and a masked-out
// (possibly stale, always finite) value times 0 is 0.
__device__ __forceinline__ void split2x8_scaled(const float (&x)[8], const float (&sm)[8], f32x4& vh, f32x4& vl) {
    g_f16x8 h, l;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float xs = x[c] * sm[c];
        const _Float16 a = (_Float16)xs;
        h[c] = a;
        l[c] = (_Float16)(xs - (float)a);
    }
    vh = __builtin_bit_cast(f32x4, h);
    vl = __builtin_bit_cast(f32x4, l);
}
// eight fp32 values, scaled by `scale` first -> the two operand units
__device__ __forceinline__ void split_parts(const float (&x)[8], float scale, f32x4 (&v)[GNP]) {
    float y[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) y[c] = x[c] * scale;
    split2x8(y, v[0], v[1]);
}
// eight fp32 values x eight per-position factors (operand scale or 0) -> the two operand units
__device__ __forceinline__ void masked_parts(const float (&x)[8], const float (&sm)[8], f32x4 (&v)[GNP]) {
    split2x8_scaled(x, sm, v[0], v[1]);
}

// ---------------------------------------------------------------------------------------------------
// D [C][C] fp32 (symmetric) -> MFMA A-fragment image [C/16 chunks][2 parts][2 k-groups][C rows][8] fp16
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gram_d_pack_body(const float* __restrict__ D, f32x4* __restrict__ P, int C,
                                                 const float* __restrict__ amax_d, int block_x) {
    float scale = 1.f;
    if (GNP == 2) {   // both matrices share one bound (max |D0|, |D1| of the style-loss kernel)
        float inv;
        scale = gram_pow2_scale(amax_read(amax_d), inv);
    }
    const int u = block_x * 256 + threadIdx.x;   // (row, 8-column group)
    const int groups = C / 8;
    if (u >= C * groups) return;
    const int row = u / groups, g = u - row * groups;
    float x[8];
    const f32x4 a = *reinterpret_cast<const f32x4*>(D + (size_t)row * C + g * 8);
    const f32x4 b = *reinterpret_cast<const f32x4*>(D + (size_t)row * C + g * 8 + 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) { x[c] = a[c]; x[4 + c] = b[c]; }
    f32x4 v[GNP];
    split_parts(x, scale, v);
    const int chunk = g >> 1, kg = g & 1;
    f32x4* d = P + (size_t)(chunk * 2 * GNP + kg) * C + row;
#pragma unroll
    for (int part = 0; part < GNP; ++part) d[2 * part * (size_t)C] = v[part];
}
__global__ __launch_bounds__(256) void gram_d_pack_kernel(const float* __restrict__ D0, const float* __restrict__ D1,
                                                          f32x4* __restrict__ P0, f32x4* __restrict__ P1, int C,
                                                          const float* __restrict__ amax_d) {
    gram_d_pack_body(blockIdx.y ? D1 : D0, blockIdx.y ? P1 : P0, C, amax_d, blockIdx.x);
}
// GROUPED: the derivative matrices of many (level, layer) problems in one launch
struct GramPackProb {
    const float* D0;
    const float* D1;          // nullptr: one matrix
    f32x4* P0;
    f32x4* P1;
    const float* amax_d;
    int C, blocks;            // blocks per matrix
};
struct GramPackGroup {
    GramPackProb p[GRAM_MAX_GROUP];
    int first_block[GRAM_MAX_GROUP + 1];
    int n;
};
__global__ __launch_bounds__(256) void gram_d_pack_group_kernel(GramPackGroup G) {
    int g = 0;
    for (int i = 1; i < G.n; ++i)
        if ((int)blockIdx.x >= G.first_block[i]) g = i;
    const GramPackProb P = G.p[g];
    const int local = blockIdx.x - G.first_block[g];
    const int which = local / P.blocks;
    gram_d_pack_body(which ? P.D1 : P.D0, which ? P.P1 : P.P0, P.C, P.amax_d, local - which * P.blocks);
}

// ---------------------------------------------------------------------------------------------------
// K5c-s: dF[c][q] = m0[q] (D0 F)[c][q] + m1[q] (D1 F)[c][q]; GEMM M = C, N = positions, K = C per live mask.
// Block tile (64 MI) x 128, 4 waves 2 x 2. A stage = (16-channel chunk, mask k): the B operand is m_k F, so both
// masks accumulate into one set of accumulators; masks that are zero on the whole 128-position tile cost nothing
// (the passed / failed angle masks partition the valid pixels, most tiles see one of them).
// ---------------------------------------------------------------------------------------------------
// MI = 1: 64-row blocks, waves 2 x 2 with 32 x 64 tiles; MI = 2: 128-row blocks, waves 4 x 1 with 32 x 128 tiles (one
// 32-row weight fragment set per wave: the loop is sensitive to the number of vector-memory instructions per MFMA,
// see conv_split_kernel.h)
struct GramBwdProb {
    const float* feat;
    const float* mask0;
    const float* mask1;       // nullptr: one mask
    const f32x4* P0;          // operand images of D0 / D1 (gram_d_pack_kernel)
    const f32x4* P1;
    float* dfeat;
    const float* amax_feat;
    const float* amax_d;
    float* amax_out;          // optional: records max |dF| (the operand bound of the conv that consumes dfeat)
    int C, plane, q_begin, q_end, relu_gate, n_ptiles;
};
struct GramBwdGroup {
    GramBwdProb p[GRAM_MAX_GROUP];
    int first_block[GRAM_MAX_GROUP + 1];
    int n;
};

template <int MI>
__device__ __forceinline__ void gram_backward_body(const GramBwdProb& G, const int block_x, const int block_y) {
    const float* __restrict__ feat = G.feat;
    const float* __restrict__ mask0 = G.mask0;
    const float* __restrict__ mask1 = G.mask1;
    const f32x4* __restrict__ P0 = G.P0;
    const f32x4* __restrict__ P1 = G.P1;
    float* __restrict__ dfeat = G.dfeat;
    const float* __restrict__ amax_feat = G.amax_feat;
    const float* __restrict__ amax_d = G.amax_d;
    const int C = G.C, plane = G.plane, q_begin = G.q_begin, q_end = G.q_end;
    const bool RELU_GATE = G.relu_gate != 0;
    const float amax_seen = amax_peek(G.amax_out);
    constexpr int BN = 128;
    constexpr int KS = 2;                 // MFMA K-steps (16 channels each) per stage
    constexpr int SLICE = KS * 2 * GNP * BN;    // [kstep][part][kgroup][position] units of 8 channels
    float f_scale = 1.f, out_scale = 1.f;
    if (GNP == 2) {
        float inv_f, inv_d;
        f_scale = gram_pow2_scale(amax_read(amax_feat), inv_f);
        gram_pow2_scale(amax_read(amax_d), inv_d);
        out_scale = inv_f * inv_d;
    }
    __shared__ __attribute__((aligned(16))) f32x4 Bs[2][SLICE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    constexpr int NJ = 2 * MI;                       // 32-position MFMA tiles per wave
    const int wm = (MI == 2 ? wave : (wave >> 1)) * 32, wn = (MI == 2 ? 0 : (wave & 1)) * 64;
    const int m0 = block_y * (64 * MI);
    const int q0 = q_begin + block_x * BN;

    // staging units of this thread: k-group b_kg of every K-step, position b_px; its two mask values
    const int b_kg = tid >> 7, b_px = tid & 127;
    const bool b_in = q0 + b_px < q_end;
    const float mv0 = b_in ? mask0[q0 + b_px] : 0.f;
    const float mv1 = (b_in && mask1) ? mask1[q0 + b_px] : 0.f;
    const int live0 = __syncthreads_or(mv0 != 0.f), live1 = __syncthreads_or(mv1 != 0.f);
    const int nlive = (live0 != 0) + (live1 != 0);

    f32x16 acc[1][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

    if (nlive > 0) {
        // Round 6: a stage of the B operand = a 32-channel chunk of F, scaled and split ONCE whatever the number of live
        // masks; a mask is applied where a wave reads its fragments (a column of the MFMA = a position: the lane's own two
        // mask values select the fragment or zeros). Rounds 2-5 staged m_k F per (chunk, mask): on a tile both masks touch
        // - 30 % of relu3_1's, 50-65 % of relu4_1's / relu5_1's - every load, conversion, LDS store and barrier twice.
        // The operand values, and the order (chunk, mask, k-step) in which their products reach an accumulator, are the
        // same: the sums keep their bits (the conv epilogues' SM_EPI_GRAM form reproduces them).
        const int n_chunks = C / (16 * KS);      // 2, 4, 8 or 16
        const int kfix = live0 ? 0 : 1;
        const float* bsrc = feat + (size_t)b_kg * 8 * plane + q0 + b_px;
        const int a_off = lhi * C + m0 + wm + l31;
        // this lane's mask values at its NJ columns, as two bit sets
        unsigned keep0 = 0u, keep1 = 0u;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = q0 + wn + j * 32 + l31;
            const bool in = q < q_end;
            keep0 |= ((in && mask0[in ? q : q0] != 0.f) ? 1u : 0u) << j;
            keep1 |= ((in && mask1 && mask1[in ? q : q0] != 0.f) ? 1u : 0u) << j;
        }
        f32x4 ra[2][KS][1][GNP];
        float rb[KS][8];
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#define SM_LOAD_A(set_, chunk_, k_)                                                         \
    {                                                                                       \
        const f32x4* p_ = ((k_) ? P1 : P0) + (size_t)(chunk_) * KS * 2 * GNP * C + a_off;    \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks)                                   \
            _Pragma("unroll") for (int part = 0; part < GNP; ++part)                         \
                ra[set_][ks][0][part] = p_[(ks * 2 * GNP + part * 2) * C];                   \
    }
    // (beyond the last chunk the last one is re-read: unconditional loads keep the compiler's vmcnt bookkeeping exact,
    // see conv_split_kernel.h)
#define SM_LOAD_B(chunk_)                                                                   \
    {                                                                                       \
        const float* p_ = bsrc + (size_t)min((chunk_), n_chunks - 1) * 16 * KS * plane;     \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks)                                   \
            _Pragma("unroll") for (int c = 0; c < 8; ++c) rb[ks][c] = side_load(p_ + (size_t)(ks * 16 + c) * plane); \
    }
#define SM_STORE_B(buf_)                                                                    \
    {                                                                                       \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                 \
            f32x4 v_[GNP];                                                                   \
            float y_[8];                                                                    \
            _Pragma("unroll") for (int c = 0; c < 8; ++c) y_[c] = rb[ks][c] * f_scale;      \
            /* (a stale value of a dead position may leave fp16's range: clamped - the mask drops it anyway) */ \
            split2x8(y_, v_[0], v_[1]);                                                     \
            f32x4* d_ = &Bs[buf_][ks * 2 * GNP * BN + b_kg * BN + b_px];                     \
            _Pragma("unroll") for (int part = 0; part < GNP; ++part) d_[2 * part * BN] = v_[part]; \
        }                                                                                   \
    }
    // the MFMAs of (chunk in buffer buf_, mask keep_) with the A fragments of set_
#define SM_MFMAS(buf_, set_, keep_)                                                         \
    {                                                                                       \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                 \
            f32x4 fa[1][GNP], fb[NJ][GNP];                                                    \
            const f32x4* bf_ = &Bs[buf_][ks * 2 * GNP * BN + lhi * BN + wn + l31];           \
            _Pragma("unroll") for (int part = 0; part < GNP; ++part) {                       \
                fa[0][part] = ra[set_][ks][0][part];                                        \
                _Pragma("unroll") for (int j = 0; j < NJ; ++j)                              \
                    fb[j][part] = ((keep_) >> j) & 1u ? bf_[part * 2 * BN + j * 32] : zero4; \
            }                                                                               \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j) mfma_parts(acc[0][j], fa[0], fb[j]); \
        }                                                                                   \
    }
        // chunk c lives in buffer c & 1; while its stage(s) run, chunk c + 1 is converted into the other buffer and
        // chunk c + 2 loaded into the registers just stored
        auto pipeline = [&](auto two_c) {
            constexpr bool TWO = decltype(two_c)::value;
            const unsigned keepA = TWO ? keep0 : (kfix ? keep1 : keep0);
            const int kA = TWO ? 0 : kfix;
            SM_LOAD_B(0)
            SM_LOAD_A(0, 0, kA)
            if (TWO) {
                SM_LOAD_A(1, 0, 1)
            } else {
                SM_LOAD_A(1, min(1, n_chunks - 1), kA)
            }
            SM_STORE_B(0)
            SM_LOAD_B(1)
            __syncthreads();
#define SM_CHUNK(c_, buf_)                                                                  \
    {                                                                                       \
        SM_STORE_B(1 - (buf_))                                                              \
        SM_LOAD_B((c_) + 2)                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        if (TWO) {                                                                          \
            SM_MFMAS(buf_, 0, keep0)                                                        \
            __builtin_amdgcn_sched_barrier(0);                                              \
            SM_LOAD_A(0, min((c_) + 1, n_chunks - 1), 0)                                    \
            SM_MFMAS(buf_, 1, keep1)                                                        \
            __builtin_amdgcn_sched_barrier(0);                                              \
            SM_LOAD_A(1, min((c_) + 1, n_chunks - 1), 1)                                    \
        } else {                                                                            \
            SM_MFMAS(buf_, buf_, keepA)                                                     \
            __builtin_amdgcn_sched_barrier(0);                                              \
            SM_LOAD_A(buf_, min((c_) + 2, n_chunks - 1), kA)                                \
        }                                                                                   \
        __syncthreads();                                                                    \
    }
            for (int c = 0; c < n_chunks; c += 2) {      // (n_chunks is even)
                SM_CHUNK(c, 0)
                SM_CHUNK(c + 1, 1)
            }
#undef SM_CHUNK
        };
        if (nlive == 2) pipeline(std::true_type{}); else pipeline(std::false_type{});
#undef SM_LOAD_A
#undef SM_LOAD_B
#undef SM_STORE_B
#undef SM_MFMAS
    }
    // epilogue: 32x32 C/D layout, column (position) = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float vmax = 0.f;
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
        const int q = q0 + wn + nj * 32 + l31;
        if (q >= q_end) continue;
        const size_t o0 = (size_t)(m0 + wm + 4 * lhi) * plane + q;
        float gate[16];
        if (RELU_GATE) {
#pragma unroll
            for (int r = 0; r < 16; ++r) gate[r] = feat[o0 + (size_t)((r & 3) + 8 * (r >> 2)) * plane];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[0][nj][r];
            if (GNP == 2) v *= out_scale;
            if (RELU_GATE) v = gate[r] > 0.f ? v : 0.f;
            side_store(v, dfeat + o0 + (size_t)((r & 3) + 8 * (r >> 2)) * plane);
            vmax = fmaxf(vmax, fabsf(v));
        }
    }
    record_amax(G.amax_out, vmax, amax_seen);
}

template <int MI, bool RELU_GATE_>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MI == 1 ? SM_GRAM_BWD_W1 : SM_GRAM_BWD_W2, MI == 1 ? SM_GRAM_BWD_W1 : SM_GRAM_BWD_W2))) void gram_backward_split_kernel(
    const float* __restrict__ feat, const float* __restrict__ mask0, const float* __restrict__ mask1,
    const f32x4* __restrict__ P0, const f32x4* __restrict__ P1, float* __restrict__ dfeat, int C, int plane, int q_begin,
    int q_end, const float* __restrict__ amax_feat, const float* __restrict__ amax_d) {
    const GramBwdProb G{feat, mask0, mask1, P0, P1, dfeat, amax_feat, amax_d, nullptr, C, plane, q_begin, q_end, RELU_GATE_ ? 1 : 0, 0};
    gram_backward_body<MI>(G, blockIdx.x, blockIdx.y);
}

// GROUPED: one launch over the (level, layer) problems of one row-tile class; block -> (problem, position tile, row tile)
template <int MI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MI == 1 ? SM_GRAM_BWD_W1 : SM_GRAM_BWD_W2, MI == 1 ? SM_GRAM_BWD_W1 : SM_GRAM_BWD_W2))) void gram_backward_group_kernel(GramBwdGroup G) {
    int g = 0;
    for (int i = 1; i < G.n; ++i)
        if ((int)blockIdx.x >= G.first_block[i]) g = i;
    const GramBwdProb P = G.p[g];
    const int local = blockIdx.x - G.first_block[g];
    gram_backward_body<MI>(P, local % P.n_ptiles, local / P.n_ptiles);
}

// ---------------------------------------------------------------------------------------------------
// K5a-s: S_k[i][j] = sum_q m_k[q] F[i][q] F[j][q] = (m_k F)(m_k F)^T for 0/1 masks, GROUPED: one launch covers the
// (level, layer) problems of one tile class (TS = 64 MI channels per tile). A block = (problem, position range, tile
// pair tm <= tn, mask); its partial sums are added into the problem's PRE-ZEROED S_k with fp32 atomics (C^2 floats of
// traffic per block, no slab reduction; summation order - like the texture scatter's - is not fixed).
// K runs over positions in stages of SP = 16 KS positions (KS MFMA K-steps per barrier); a stage whose mask values are
// all zero is skipped before anything is loaded (the masks of a level are sparse and disjoint). The loop is bound by
// the latency of the feature-map loads: two stages are in flight per block (MI = 1: 2 x 64 channels x 64 positions =
// 32 KB, MI = 2 off-diagonal: 2 x 256 x 32 = 64 KB), which is what one CU needs outstanding to draw its share of the
// HBM bandwidth (the 16-position stages of the first version kept 8 KB in flight and ran at 1.9 TB/s).
// LDS operand image per buffer: [K-step][part][k-group][channel] 16-byte units (8 positions), strides padded so that
// the eight lanes of a ds_write_b128 group land on eight different bank groups.
// ---------------------------------------------------------------------------------------------------
struct GramProb {
    const float* feat;
    const float* mask0;
    const float* mask1;       // nullptr: one mask
    float* S0;
    float* S1;
    const float* amax_feat;
    int C, plane, q_begin, q_end, qb, n_ranges;
};
struct GramGroup {
    GramProb p[GRAM_MAX_GROUP];
    int first_block[GRAM_MAX_GROUP + 1];
    int n;
};
constexpr int gram_ks(int MI) { return MI == 1 ? 4 : 2; }
constexpr int gram_kg_stride(int MI) { return MI == 1 ? 64 + 1 : 128 + 2; }
constexpr int gram_ks_stride(int MI) {
    int v = 2 * GNP * gram_kg_stride(MI);
    while (v % 8 != (MI == 1 ? 2 : 4)) ++v;
    return v;
}
constexpr size_t gram_group_lds_bytes(int MI, bool diag_only = false) {
    return (size_t)(diag_only ? 2 : 4) * gram_ks(MI) * gram_ks_stride(MI) * 16;
}

// MI = 1 (126 VGPRs): four blocks per CU when every problem of the launch is a single diagonal tile (C = 64: the Bt
// half of the LDS image is not allocated then) - a block keeps ~one 16 KB stage in flight, a CU needs ~50 KB
template <int MI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MI == 1 ? SM_GRAM_FWD_W1 : 2, MI == 1 ? SM_GRAM_FWD_W1 : 2))) void gram_group_kernel(GramGroup G) {
    constexpr int TS = 64 * MI;                 // tile size (channels)
    constexpr int KS = gram_ks(MI);             // MFMA K-steps per stage
    constexpr int SP = 16 * KS;                 // positions per stage
    constexpr int NG = 2 * KS;                  // 8-position k-groups per stage
    constexpr int CP = 256 / NG;                // channels per staging pass
    constexpr int R = TS / CP;                  // staging passes
    static_assert(R == 2, "two staging units per thread and operand");
    constexpr int KG = gram_kg_stride(MI), PART = 2 * KG, KSS = gram_ks_stride(MI), BUF = KS * KSS;
    extern __shared__ __attribute__((aligned(16))) f32x4 gsm[];
    f32x4* As = gsm;                            // [2][BUF]
    f32x4* Bt = gsm + 2 * BUF;                  // [2][BUF]
    __shared__ int live_list[64];               // qb <= 64 SP positions
    __shared__ int wave_count[4];

    int g = 0;
    for (int i = 1; i < G.n; ++i)
        if ((int)blockIdx.x >= G.first_block[i]) g = i;
    const GramProb P = G.p[g];
    int local = blockIdx.x - G.first_block[g];
    const int T = P.C / TS, pairs = T * (T + 1) / 2;
    const int range = local % P.n_ranges;
    local /= P.n_ranges;
    int rem = local % pairs;
    const int which = local / pairs;
    int tm = 0;
    while (rem >= T - tm) { rem -= T - tm; ++tm; }
    const int tn = tm + rem;
    const bool diag = tm == tn;
    const float* mask = which ? P.mask1 : P.mask0;
    float* S = which ? P.S1 : P.S0;
    const int C = P.C, plane = P.plane;

    float f_scale = 1.f, out_scale = 1.f;
    if (GNP == 2) {
        float inv;
        f_scale = gram_pow2_scale(amax_read(P.amax_feat), inv);
        out_scale = inv * inv;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int wm = (wave >> 1) * (32 * MI), wn = (wave & 1) * (32 * MI);
    const int qs = P.q_begin + range * P.qb;
    const int qe = min(qs + P.qb, P.q_end);
    const int n_st = (qe - qs + SP - 1) / SP;

    // ---- compact list of the stages with any non-zero mask value (order preserved)
    {
        // four threads per stage (n_st <= 64), SP / 16 float4 each, all loads in flight together (unconditional,
        // addresses clamped into the range)
        const int st = tid >> 2;
        const int q = qs + min(st, max(n_st - 1, 0)) * SP + (tid & 3) * (SP / 4);
        f32x4 m[SP / 16];
#pragma unroll
        for (int e = 0; e < SP / 16; ++e) m[e] = *reinterpret_cast<const f32x4*>(mask + min(q + 4 * e, qe - 4));   // qe, q: multiples of 4
        bool lv = false;
#pragma unroll
        for (int e = 0; e < SP / 16; ++e)
            lv |= (q + 4 * e < qe) & ((m[e][0] != 0.f) | (m[e][1] != 0.f) | (m[e][2] != 0.f) | (m[e][3] != 0.f));
        const unsigned long long b4 = __ballot(lv && st < n_st);
        const bool leader = (lane & 3) == 0 && ((b4 >> lane) & 0xFull) != 0ull;   // first thread of a live stage
        const unsigned long long b = __ballot(leader);
        if (lane == 0) wave_count[wave] = __popcll(b);
        __syncthreads();
        int off = 0;
        for (int w = 0; w < wave; ++w) off += wave_count[w];
        if (leader) live_list[off + __popcll(b & ((1ull << lane) - 1ull))] = st;
        __syncthreads();
    }
    const int n_live = wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
    if (n_live == 0) return;   // nothing to add

    f32x16 acc[MI][MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging units of this thread: k-group u_kg (8 consecutive positions) of channels u_ch, u_ch + CP
    const int u_kg = tid % NG, u_ch = tid / NG;
    const float* a_src = P.feat + (size_t)(tm * TS + u_ch) * plane + u_kg * 8;
    const float* b_src = P.feat + (size_t)(tn * TS + u_ch) * plane + u_kg * 8;
    const int u_dst = (u_kg >> 1) * KSS + (u_kg & 1) * KG + u_ch;
#define SM_GRAM_MFMA(acc_, fa_, fb_) mfma_parts(acc_, fa_, fb_)
#define SM_LOAD(set_, i_)                                                                               \
    {                                                                                                   \
        const int q_ = qs + live_list[min((i_), n_live - 1)] * SP;                                      \
        const f32x4 m0_ = *reinterpret_cast<const f32x4*>(mask + q_ + u_kg * 8), m1_ = *reinterpret_cast<const f32x4*>(mask + q_ + u_kg * 8 + 4); \
        const int qe_ = (i_) < n_live ? qe : 0;   /* stages past the list (odd counts are padded) add zeros */ \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                 \
            rM[set_][c] = (q_ + u_kg * 8 + c < qe_ && m0_[c] != 0.f) ? f_scale : 0.f;                   \
            rM[set_][4 + c] = (q_ + u_kg * 8 + 4 + c < qe_ && m1_[c] != 0.f) ? f_scale : 0.f;           \
        }                                                                                               \
        _Pragma("unroll") for (int r = 0; r < R; ++r) {                                                 \
            const float* pa_ = a_src + (size_t)r * CP * plane + q_;                                     \
            const f32x4 a0_ = side_load(reinterpret_cast<const f32x4*>(pa_)), a1_ = side_load(reinterpret_cast<const f32x4*>(pa_ + 4)); \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) { rA[set_][r][c] = a0_[c]; rA[set_][r][4 + c] = a1_[c]; } \
            if constexpr (!DIAG) {                                                                                \
                const float* pb_ = b_src + (size_t)r * CP * plane + q_;                                 \
                const f32x4 b0_ = side_load(reinterpret_cast<const f32x4*>(pb_)), b1_ = side_load(reinterpret_cast<const f32x4*>(pb_ + 4)); \
                _Pragma("unroll") for (int c = 0; c < 4; ++c) { rB[set_][r][c] = b0_[c]; rB[set_][r][4 + c] = b1_[c]; } \
            }                                                                                           \
        }                                                                                               \
    }
#define SM_STORE(set_, buf_)                                                                            \
    {                                                                                                   \
        _Pragma("unroll") for (int r = 0; r < R; ++r) {                                                 \
            f32x4 v_[GNP];                                                                               \
            masked_parts(rA[set_][r], rM[set_], v_);                                        \
            f32x4* d_ = As + (buf_) * BUF + u_dst + r * CP;                                             \
            _Pragma("unroll") for (int part = 0; part < GNP; ++part) d_[part * PART] = v_[part];         \
            if constexpr (!DIAG) {                                                                                \
                masked_parts(rB[set_][r], rM[set_], v_);                                    \
                f32x4* e_ = Bt + (buf_) * BUF + u_dst + r * CP;                                         \
                _Pragma("unroll") for (int part = 0; part < GNP; ++part) e_[part * PART] = v_[part];     \
            }                                                                                           \
        }                                                                                               \
    }
#define SM_STAGE(i_, par_)                                                                              \
    {                                                                                                   \
        SM_STORE(1 - (par_), 1 - (par_))                                                                \
        SM_LOAD(1 - (par_), (i_) + 3)   /* straight back into the set just stored: two stages of lead */ \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        const f32x4* af_ = As + (par_) * BUF + lhi * KG + wm + l31;                                     \
        const f32x4* bf_ = (DIAG ? As : Bt) + (par_) * BUF + lhi * KG + wn + l31;                       \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                             \
            f32x4 fa[MI][GNP], fb[MI][GNP];                                                               \
            _Pragma("unroll") for (int part = 0; part < GNP; ++part)                                     \
                _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                        \
                    fa[i][part] = af_[ks * KSS + part * PART + i * 32];                                 \
                    fb[i][part] = bf_[ks * KSS + part * PART + i * 32];                                 \
                }                                                                                       \
            _Pragma("unroll") for (int i = 0; i < MI; ++i)                                              \
                _Pragma("unroll") for (int j = 0; j < MI; ++j) SM_GRAM_MFMA(acc[i][j], fa[i], fb[j]);   \
        }                                                                                               \
        __syncthreads();                                                                                \
    }
    // the pipeline, compiled once per tile kind: a load under a runtime `if (!diag)` would make the compiler's waitcnt
    // pass drain the whole load queue at every later wait (conv_split_kernel.h)
    auto pipeline = [&](auto diag_c) {
    constexpr bool DIAG = decltype(diag_c)::value;
    float rA[2][R][8], rB[2][R][8], rM[2][8];
    SM_LOAD(0, 0)
    SM_LOAD(1, 1)
    SM_STORE(0, 0)
    SM_LOAD(0, 2)
    __syncthreads();
    // stages in unconditional pairs (a load under a condition makes the compiler's waitcnt pass drain the whole queue
    // at every later wait, see conv_split_kernel.h): an odd count runs one padding stage of zeros
    for (int i = 0; i < n_live; i += 2) {
        SM_STAGE(i, 0)
        SM_STAGE(i + 1, 1)
    }
    };
    if (diag) pipeline(std::true_type{}); else pipeline(std::false_type{});
#undef SM_LOAD
#undef SM_STORE
#undef SM_STAGE
#undef SM_GRAM_MFMA
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < MI; ++nj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tm * TS + wm + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const int col = tn * TS + wn + nj * 32 + l31;
                atomicAdd(&S[(size_t)row * C + col], GNP == 2 ? acc[mi][nj][r] * out_scale : acc[mi][nj][r]);
            }
}

}  // namespace sm

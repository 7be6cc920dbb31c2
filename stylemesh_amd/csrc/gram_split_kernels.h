// K5a-s / K5c-s: the Gram contraction S = (mF)(mF)^T and its backward GEMM dF = m0 (D0 F) + m1 (D1 F) on the bf16
// matrix cores at fp32 accuracy - the same bf16x3 split as conv_split_kernel.h (x = h + m + l, six partial products
// per fp32 product, fp32 accumulate). Replace the same reference operators as gram_masked_kernel /
// gram_backward_kernel (bool-mask gather + torch.bmm and its backward, content_and_style_losses.py:74-80,136-143).
//
// Both kernels run the same pipeline per 16-deep K stage: global loads two stages ahead into one of two register
// sets, fp32 -> 3 x bf16x8 conversion while storing into one of two LDS buffers one stage ahead, fragment reads +
// 6 MFMAs per output tile on the current buffer, one barrier per stage.
#pragma once
#include "common.h"

namespace sm {

// NP = 3: bf16 x 3 operands, six partial products; NP = 2: fp16 x 2 operands scaled by a power of two from the
// tensor's recorded max |x| (conv_split_kernel.h), three partial products. Fragments travel as raw 16-byte units.
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
template <int NP>
__device__ __forceinline__ void mfma_parts(f32x16& acc, const f32x4 (&fa)[NP], const f32x4 (&fb)[NP]) {
    if constexpr (NP == 3) {
#define SM_B(x_) __builtin_bit_cast(bf16x8, x_)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(SM_B(fa[2]), SM_B(fb[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(SM_B(fa[0]), SM_B(fb[2]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(SM_B(fa[1]), SM_B(fb[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(SM_B(fa[1]), SM_B(fb[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(SM_B(fa[0]), SM_B(fb[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(SM_B(fa[0]), SM_B(fb[0]), acc, 0, 0, 0);
#undef SM_B
    } else {
#define SM_H(x_) __builtin_bit_cast(g_f16x8, x_)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(SM_H(fa[1]), SM_H(fb[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(SM_H(fa[0]), SM_H(fb[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(SM_H(fa[0]), SM_H(fb[0]), acc, 0, 0, 0);
#undef SM_H
    }
}
// eight fp32 values -> NP operand units (NP = 2: scaled by `scale` first)
template <int NP>
__device__ __forceinline__ void split_parts(const float (&x)[8], float scale, f32x4 (&v)[NP]) {
    if constexpr (NP == 3) {
        split3x8(x, v[0], v[1], v[2]);
    } else {
        float y[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) y[c] = x[c] * scale;
        split2x8(y, v[0], v[1]);
    }
}

// ---------------------------------------------------------------------------------------------------
// D [C][C] fp32 (symmetric) -> MFMA A-fragment image [C/16 chunks][3 parts][2 k-groups][C rows][8] bf16
// ---------------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(256) void gram_d_pack_kernel(const float* __restrict__ D0, const float* __restrict__ D1,
                                                          f32x4* __restrict__ P0, f32x4* __restrict__ P1, int C,
                                                          const float* __restrict__ amax_d) {
    const float* D = blockIdx.y ? D1 : D0;
    f32x4* P = blockIdx.y ? P1 : P0;
    float scale = 1.f;
    if (NP == 2) {   // both matrices share one bound (max |D0|, |D1| of the style-loss kernel)
        float inv;
        scale = gram_pow2_scale(amax_read(amax_d), inv);
    }
    const int u = blockIdx.x * 256 + threadIdx.x;   // (row, 8-column group)
    const int groups = C / 8;
    if (u >= C * groups) return;
    const int row = u / groups, g = u - row * groups;
    float x[8];
    const f32x4 a = *reinterpret_cast<const f32x4*>(D + (size_t)row * C + g * 8);
    const f32x4 b = *reinterpret_cast<const f32x4*>(D + (size_t)row * C + g * 8 + 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) { x[c] = a[c]; x[4 + c] = b[c]; }
    f32x4 v[NP];
    split_parts<NP>(x, scale, v);
    const int chunk = g >> 1, kg = g & 1;
    f32x4* d = P + (size_t)(chunk * 2 * NP + kg) * C + row;
#pragma unroll
    for (int part = 0; part < NP; ++part) d[2 * part * (size_t)C] = v[part];
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
template <int MI, bool RELU_GATE, int NP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gram_backward_split_kernel(
    const float* __restrict__ feat, const float* __restrict__ mask0, const float* __restrict__ mask1,
    const f32x4* __restrict__ P0, const f32x4* __restrict__ P1, float* __restrict__ dfeat, int C, int plane, int q_begin,
    int q_end, const float* __restrict__ amax_feat, const float* __restrict__ amax_d) {
    constexpr int BN = 128;
    constexpr int KS = 2;                 // MFMA K-steps (16 channels each) per stage
    constexpr int SLICE = KS * 2 * NP * BN;    // [kstep][part][kgroup][position] units of 8 channels
    float f_scale = 1.f, out_scale = 1.f;
    if (NP == 2) {
        float inv_f, inv_d;
        f_scale = gram_pow2_scale(amax_read(amax_feat), inv_f);
        gram_pow2_scale(amax_read(amax_d), inv_d);
        out_scale = inv_f * inv_d;
    }
    __shared__ __attribute__((aligned(16))) f32x4 Bs[2][SLICE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    constexpr int NJ = 2 * MI;                       // 32-position MFMA tiles per wave
    const int wm = (MI == 2 ? wave : (wave >> 1)) * 32, wn = (MI == 2 ? 0 : (wave & 1)) * 64;
    const int m0 = blockIdx.y * (64 * MI);
    const int q0 = q_begin + blockIdx.x * BN;

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
        const int n_chunks = C / (16 * KS);
        const int n_stages = n_chunks * nlive;   // even: n_chunks is 2, 4, 8 or 16
        const int kfix = live0 ? 0 : 1;
        const float* bsrc = feat + (size_t)b_kg * 8 * plane + q0 + b_px;
        const int a_off = lhi * C + m0 + wm + l31;
        f32x4 ra[2][KS][1][NP];
        float rb[2][KS][8];
        // stage s -> (chunk, mask); beyond the last stage the last one is re-read (unconditional loads keep the
        // compiler's vmcnt bookkeeping exact, see conv_split_kernel.h)
#define SM_STAGE_OF(s_, chunk_, k_)                                                         \
    const int sc_ = min((s_), n_stages - 1);                                                \
    const int k_ = nlive == 2 ? (sc_ & 1) : kfix;                                           \
    const int chunk_ = nlive == 2 ? (sc_ >> 1) : sc_;
#define SM_LOAD_A(set_, s_)                                                                 \
    {                                                                                       \
        SM_STAGE_OF(s_, chunk_, k_)                                                         \
        const f32x4* p_ = (k_ ? P1 : P0) + (size_t)chunk_ * KS * 2 * NP * C + a_off;        \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks)                                   \
            _Pragma("unroll") for (int part = 0; part < NP; ++part)                         \
                ra[set_][ks][0][part] = p_[(ks * 2 * NP + part * 2) * C];                   \
    }
#define SM_LOAD_B(set_, s_)                                                                 \
    {                                                                                       \
        SM_STAGE_OF(s_, chunk_, k_)                                                         \
        (void)k_;                                                                           \
        const float* p_ = bsrc + (size_t)chunk_ * 16 * KS * plane;                          \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks)                                   \
            _Pragma("unroll") for (int c = 0; c < 8; ++c) rb[set_][ks][c] = p_[(size_t)(ks * 16 + c) * plane]; \
    }
#define SM_STORE_B(set_, s_, buf_)                                                          \
    {                                                                                       \
        SM_STAGE_OF(s_, chunk_, k_)                                                         \
        (void)chunk_;                                                                       \
        const float mv_ = k_ ? mv1 : mv0;                                                   \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                 \
            float x_[8];                                                                    \
            _Pragma("unroll") for (int c = 0; c < 8; ++c) x_[c] = (mv_ != 0.f) ? rb[set_][ks][c] : 0.f; \
            f32x4 v_[NP];                                                                   \
            split_parts<NP>(x_, f_scale, v_);                                               \
            f32x4* d_ = &Bs[buf_][ks * 2 * NP * BN + b_kg * BN + b_px];                     \
            _Pragma("unroll") for (int part = 0; part < NP; ++part) d_[2 * part * BN] = v_[part]; \
        }                                                                                   \
    }
#define SM_STAGE(s_, par_)                                                                  \
    {                                                                                       \
        SM_STORE_B(1 - (par_), (s_) + 1, 1 - (par_))                                        \
        SM_LOAD_B(par_, (s_) + 2)                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                 \
            f32x4 fa[1][NP], fb[NJ][NP];                                                    \
            const f32x4* bf_ = &Bs[par_][ks * 2 * NP * BN + lhi * BN + wn + l31];           \
            _Pragma("unroll") for (int part = 0; part < NP; ++part) {                       \
                fa[0][part] = ra[par_][ks][0][part];                                        \
                _Pragma("unroll") for (int j = 0; j < NJ; ++j) fb[j][part] = bf_[part * 2 * BN + j * 32]; \
            }                                                                               \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j) mfma_parts<NP>(acc[0][j], fa[0], fb[j]); \
        }                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        SM_LOAD_A(par_, (s_) + 2)                                                           \
        __syncthreads();                                                                    \
    }
        SM_LOAD_B(0, 0)
        SM_LOAD_B(1, 1)
        SM_LOAD_A(0, 0)
        SM_LOAD_A(1, 1)
        SM_STORE_B(0, 0, 0)
        __syncthreads();
        for (int s = 0; s < n_stages; s += 2) {
            SM_STAGE(s, 0)
            SM_STAGE(s + 1, 1)
        }
#undef SM_STAGE_OF
#undef SM_LOAD_A
#undef SM_LOAD_B
#undef SM_STORE_B
#undef SM_STAGE
    }
    // epilogue: 32x32 C/D layout, column (position) = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
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
            if (NP == 2) v *= out_scale;
            if (RELU_GATE) v = (gate[r] > 0.f) ? v : 0.f;
            dfeat[o0 + (size_t)((r & 3) + 8 * (r >> 2)) * plane] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// K5a-s: S_k[i][j] = sum_q m_k[q] F[i][q] F[j][q] = (m_k F)(m_k F)^T for 0/1 masks. grid = (position ranges, tile
// pairs tm <= tn of (64 MI)^2, masks). K runs over positions in stages of 16; a stage whose 16 mask values are all
// zero is skipped before anything is loaded (the masks of a level are sparse and disjoint). Partial sums of the
// range go to slab blockIdx.x, exactly like gram_masked_kernel.
// ---------------------------------------------------------------------------------------------------
// ATOMIC: all position ranges accumulate into ONE pre-zeroed slab with fp32 atomics instead of writing a slab each
// (no reduction pass, C^2 floats of traffic per block instead of written + re-read; summation order - like the
// texture scatter's - is then not fixed).
template <int MI, bool ATOMIC, int NP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gram_split_kernel(
    const float* __restrict__ feat, const float* __restrict__ mask0, const float* __restrict__ mask1, float* S0, float* S1,
    int C, int plane, int q_begin, int q_end, int qb, const float* __restrict__ amax_feat) {
    constexpr int TS = 64 * MI;           // tile size (channels)
    constexpr int SLICE = 2 * NP * TS;    // [part][kgroup][channel] units of 8 positions
    float f_scale = 1.f, out_scale = 1.f;
    if (NP == 2) {
        float inv;
        f_scale = gram_pow2_scale(amax_read(amax_feat), inv);
        out_scale = inv * inv;
    }
    constexpr int MAX_STAGES = 256;       // qb <= 4096 positions
    __shared__ __attribute__((aligned(16))) f32x4 As[2][SLICE];
    __shared__ __attribute__((aligned(16))) f32x4 Bt[2][SLICE];
    __shared__ int live_list[MAX_STAGES];
    __shared__ int wave_count[4];
    const int T = C / TS;
    int tm = 0, rem = blockIdx.y;
    while (rem >= T - tm) { rem -= T - tm; ++tm; }
    const int tn = tm + rem;
    const bool diag = tm == tn;
    const float* mask = blockIdx.z ? mask1 : mask0;
    float* S = (blockIdx.z ? S1 : S0) + (ATOMIC ? (size_t)0 : (size_t)blockIdx.x * C * C);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int wm = (wave >> 1) * (32 * MI), wn = (wave & 1) * (32 * MI);
    const int qs = q_begin + blockIdx.x * qb;
    const int qe = min(qs + qb, q_end);
    const int n_st = (qe - qs + 15) / 16;

    // ---- compact list of the stages with any non-zero mask value (order preserved)
    {
        bool lv = false;
        if (tid < n_st) {
            const int q = qs + tid * 16;
#pragma unroll
            for (int e = 0; e < 16; e += 4) {
                if (q + e < qe) {   // qe, q are multiples of 4
                    const f32x4 m = *reinterpret_cast<const f32x4*>(mask + q + e);
                    lv |= (m[0] != 0.f) | (m[1] != 0.f) | (m[2] != 0.f) | (m[3] != 0.f);
                }
            }
        }
        const unsigned long long b = __ballot(lv);
        if (lane == 0) wave_count[wave] = __popcll(b);
        __syncthreads();
        int off = 0;
        for (int w = 0; w < wave; ++w) off += wave_count[w];
        if (lv) live_list[off + __popcll(b & ((1ull << lane) - 1ull))] = tid;
        __syncthreads();
    }
    const int n_live = wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];

    f32x16 acc[MI][MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (n_live > 0) {
        // staging unit of this thread: channel u_ch of the tile, k-group u_kg (8 consecutive positions)
        const int u_ch = (tid >> 1) % TS, u_kg = tid & 1;
        const bool u_on = tid < 2 * TS;   // 64-channel tiles use half the block for staging
        const float* a_src = feat + (size_t)(tm * TS + u_ch) * plane + u_kg * 8;
        const float* b_src = feat + (size_t)(tn * TS + u_ch) * plane + u_kg * 8;
        float rA[2][8], rB[2][8], rM[2][8];
#define SM_LOAD(set_, i_)                                                                               \
    {                                                                                                   \
        const int q_ = qs + live_list[min((i_), n_live - 1)] * 16;                                      \
        const f32x4 a0_ = *reinterpret_cast<const f32x4*>(a_src + q_), a1_ = *reinterpret_cast<const f32x4*>(a_src + q_ + 4); \
        const f32x4 m0_ = *reinterpret_cast<const f32x4*>(mask + q_ + u_kg * 8), m1_ = *reinterpret_cast<const f32x4*>(mask + q_ + u_kg * 8 + 4); \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                 \
            rA[set_][c] = a0_[c]; rA[set_][4 + c] = a1_[c];                                             \
            rM[set_][c] = (q_ + u_kg * 8 + c < qe) ? m0_[c] : 0.f;                                      \
            rM[set_][4 + c] = (q_ + u_kg * 8 + 4 + c < qe) ? m1_[c] : 0.f;                              \
        }                                                                                               \
        if (!diag) {                                                                                    \
            const f32x4 b0_ = *reinterpret_cast<const f32x4*>(b_src + q_), b1_ = *reinterpret_cast<const f32x4*>(b_src + q_ + 4); \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) { rB[set_][c] = b0_[c]; rB[set_][4 + c] = b1_[c]; } \
        }                                                                                               \
    }
#define SM_STORE(set_, buf_)                                                                            \
    if (u_on) {                                                                                         \
        float x_[8];                                                                                    \
        f32x4 v_[NP];                                                                                   \
        _Pragma("unroll") for (int c = 0; c < 8; ++c) x_[c] = (rM[set_][c] != 0.f) ? rA[set_][c] : 0.f; \
        split_parts<NP>(x_, f_scale, v_);                                                               \
        f32x4* d_ = &As[buf_][u_kg * TS + u_ch];                                                        \
        _Pragma("unroll") for (int part = 0; part < NP; ++part) d_[2 * part * TS] = v_[part];           \
        if (!diag) {                                                                                    \
            _Pragma("unroll") for (int c = 0; c < 8; ++c) x_[c] = (rM[set_][c] != 0.f) ? rB[set_][c] : 0.f; \
            split_parts<NP>(x_, f_scale, v_);                                                           \
            f32x4* e_ = &Bt[buf_][u_kg * TS + u_ch];                                                    \
            _Pragma("unroll") for (int part = 0; part < NP; ++part) e_[2 * part * TS] = v_[part];       \
        }                                                                                               \
    }
#define SM_STAGE(i_, par_)                                                                              \
    {                                                                                                   \
        SM_STORE(1 - (par_), 1 - (par_))                                                                \
        SM_LOAD(par_, (i_) + 2)                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        f32x4 fa[MI][NP], fb[MI][NP];                                                                   \
        const f32x4* af_ = &As[par_][lhi * TS + wm + l31];                                              \
        const f32x4* bf_ = (diag ? &As[par_][0] : &Bt[par_][0]) + lhi * TS + wn + l31;                  \
        _Pragma("unroll") for (int part = 0; part < NP; ++part)                                         \
            _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                            \
                fa[i][part] = af_[part * 2 * TS + i * 32];                                              \
                fb[i][part] = bf_[part * 2 * TS + i * 32];                                              \
            }                                                                                           \
        _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                  \
            _Pragma("unroll") for (int j = 0; j < MI; ++j) mfma_parts<NP>(acc[i][j], fa[i], fb[j]);     \
        __syncthreads();                                                                                \
    }
        SM_LOAD(0, 0)
        SM_LOAD(1, 1)
        SM_STORE(0, 0)
        __syncthreads();
        for (int i = 0; i < n_live; i += 2) {
            SM_STAGE(i, 0)
            if (i + 1 < n_live) SM_STAGE(i + 1, 1)
        }
#undef SM_LOAD
#undef SM_STORE
#undef SM_STAGE
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < MI; ++nj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tm * TS + wm + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const int col = tn * TS + wn + nj * 32 + l31;
                const float v = NP == 2 ? acc[mi][nj][r] * out_scale : acc[mi][nj][r];
                if (ATOMIC) {
                    if (n_live > 0) atomicAdd(&S[(size_t)row * C + col], v);
                } else {
                    S[(size_t)row * C + col] = v;
                }
            }
}

}  // namespace sm

// K5 / K6: masked Gram matrices on the fp32 matrix cores, the style-loss value and its derivative matrices,
// the Gram backward GEMM, and the masked content MSE.
//
// Reference operators replaced (lukasHoel/stylemesh, model/losses/content_and_style_losses.py):
// masked_features (:136-143, bool-mask gather -> here: multiply by the 0/1 mask, same sums), GramMatrix
// (:74-80, torch.bmm), nn.MSELoss (:265) and the loss loops of ContentAndStyleLoss.forward (:301-348),
// plus their autograd backward (bmm backward, index_put_, mse backward).
#include <algorithm>

#include "common.h"
#include <cstdlib>

#include "gram_split_kernels.h"

namespace sm {

// ---------------------------------------------------------------------------------------------------
// K5a: S_k[i][j] = sum_q m_k[q] F[i][q] F[j][q]   (split over q, upper-triangular 64x64 tiles, atomics)
// ---------------------------------------------------------------------------------------------------
constexpr int GRAM_QC = 32;     // positions per LDS chunk
constexpr int GRAM_LD = 33;     // padded row length: conflict-free column reads

// positions per block: every block writes its own partial tile (a slab costs C^2 floats of HBM traffic, written here
// and read back by the reduction), so as few position ranges as still fill the chip: ~512 blocks counting the tile
// pairs of the split kernel (128-channel tiles) and two masks
__host__ inline int gram_qb(int C, int n_pos) {
    const int T = (C % 128 == 0) ? C / 128 : C / 64, pairs = T * (T + 1) / 2;
    const int target = C <= 64 ? 2048 : (C <= 128 ? 768 : 512);   // thin layers: slabs are small, ranges are latency-bound
    int qb = (int)(((long long)n_pos * pairs * 2 + target - 1) / target);
    qb = (qb + GRAM_QC - 1) / GRAM_QC * GRAM_QC;
    return std::max(256, std::min(qb, 4096));
}

template <int NMASK>
__global__ __launch_bounds__(256) void gram_masked_kernel(const float* __restrict__ feat, const float* __restrict__ mask0,
                                                          const float* __restrict__ mask1, float* S0, float* S1, int C,
                                                          int plane, int q_begin, int q_end, int qb) {
    __shared__ float FsA[64 * GRAM_LD];
    __shared__ float FsB[64 * GRAM_LD];
    __shared__ float Ms[2][GRAM_QC];
    // tile pair (tm <= tn) from blockIdx.y
    const int T = C / 64;
    int tm = 0, rem = blockIdx.y;
    while (rem >= T - tm) { rem -= T - tm; ++tm; }
    const int tn = tm + rem;
    const bool diag = tm == tn;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x16 acc[NMASK];
#pragma unroll
    for (int k = 0; k < NMASK; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    const int qs = q_begin + blockIdx.x * qb;
    const int qe = min(qs + qb, q_end);
    for (int q0 = qs; q0 < qe; q0 += GRAM_QC) {
        // stage 64 rows x 32 positions of each operand: 512 float4 per operand, 2 per thread
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int i = tid + it * 256;
            const int row = i >> 3, c4 = (i & 7) * 4;
            const int q = q0 + c4;
            float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
            if (q < qe) {  // q_end and chunk starts are multiples of 4
                va = *reinterpret_cast<const float4*>(feat + (size_t)(tm * 64 + row) * plane + q);
                if (!diag) vb = *reinterpret_cast<const float4*>(feat + (size_t)(tn * 64 + row) * plane + q);
            }
            float* da = FsA + row * GRAM_LD + c4;
            da[0] = va.x; da[1] = va.y; da[2] = va.z; da[3] = va.w;
            if (!diag) {
                float* db = FsB + row * GRAM_LD + c4;
                db[0] = vb.x; db[1] = vb.y; db[2] = vb.z; db[3] = vb.w;
            }
        }
        if (tid < GRAM_QC) {
            const int q = q0 + tid;
            Ms[0][tid] = (q < qe) ? mask0[q] : 0.f;
            if (NMASK > 1) Ms[1][tid] = (q < qe) ? mask1[q] : 0.f;
        }
        __syncthreads();
        // masks are 0/1: a mask that is zero on the whole chunk contributes exact zeros - skip its MFMAs
        // (every wave reads the same 32 values, so the branch is block-uniform)
        bool live[NMASK];
#pragma unroll
        for (int mk = 0; mk < NMASK; ++mk) live[mk] = __ballot(Ms[mk][lane & 31] != 0.f) != 0ull;
        const float* Bsrc = diag ? FsA : FsB;
#pragma unroll
        for (int mk = 0; mk < NMASK; ++mk) {
            if (!live[mk]) continue;
#pragma unroll
            for (int kk = 0; kk < GRAM_QC / 2; ++kk) {
                const int k = kk * 2 + lhi;
                const float a = FsA[(wm + l31) * GRAM_LD + k];
                const float b = Bsrc[(wn + l31) * GRAM_LD + k];
                acc[mk] = __builtin_amdgcn_mfma_f32_32x32x2f32(a * Ms[mk][k], b, acc[mk], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // partial sums of this position range go to slab blockIdx.x (plain stores; summed by style_loss_kernel)
#pragma unroll
    for (int mk = 0; mk < NMASK; ++mk) {
        float* S = (mk == 0 ? S0 : S1) + (size_t)blockIdx.x * C * C;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tm * 64 + wm + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            const int col = tn * 64 + wn + l31;
            S[(size_t)row * C + col] = acc[mk][r];
        }
    }
}

// second level of the position split: sums groups of GRAM_GROUP raw slabs into the reduced slabs at the front
constexpr int GRAM_GROUP = 32;
__global__ __launch_bounds__(256) void gram_reduce_kernel(const float* __restrict__ raw, float* __restrict__ red, int cc,
                                                          int n_raw) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int s0 = blockIdx.y * GRAM_GROUP, s1 = min(s0 + GRAM_GROUP, n_raw);
    float v = 0.f;
    for (int s = s0; s < s1; ++s) v += raw[(size_t)s * cc + idx];
    red[(size_t)blockIdx.y * cc + idx] = v;
}

struct GramPlan {
    int qb, n_raw, n_red;   // positions per block, raw slabs, slabs the consumer sums (== n_raw when no 2nd level)
};
__host__ inline GramPlan gram_plan(int C, int n_pos) {
    GramPlan p;
    p.qb = gram_qb(C, n_pos);
    p.n_raw = (n_pos + p.qb - 1) / p.qb;
    p.n_red = p.n_raw > GRAM_GROUP ? (p.n_raw + GRAM_GROUP - 1) / GRAM_GROUP : p.n_raw;
    return p;
}

// upper-triangular tile storage: element (i,j) lives at [i][j] when tile(i) <= tile(j), else at [j][i]
__device__ __forceinline__ float sym_read(const float* S, int C, int i, int j, int n_slabs) {
    const size_t o = ((i >> 6) <= (j >> 6)) ? (size_t)i * C + j : (size_t)j * C + i;
    float v = 0.f;
    for (int s = 0; s < n_slabs; ++s) v += S[(size_t)s * C * C + o];
    return v;
}

// ---------------------------------------------------------------------------------------------------
// K5b: loss value + derivative matrices D_k for one (level, layer)
// ---------------------------------------------------------------------------------------------------
struct StyleTerms {
    const float* target[4];
    int mask[4];
    int n;
    int skip_if_empty[2];
};

__device__ __forceinline__ float block_sum256(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

#ifndef SM_STYLE_EPT
#define SM_STYLE_EPT 8
#endif
constexpr int STYLE_EPT = SM_STYLE_EPT;   // elements per thread
__device__ __forceinline__ void style_loss_body(const float* __restrict__ S0, const float* __restrict__ S1,
                                                const float* __restrict__ counts, const float* __restrict__ factor,
                                                const StyleTerms& terms, float weight, int C, float* __restrict__ D0,
                                                float* __restrict__ D1, float* loss_out, float* history, int hist_len,
                                                int hist_slot, int n_slabs, float* amax_d, int block_x) {
    __shared__ float red[4];
    float dmax = 0.f;   // max |D0|, |D1|: operand bound of the fp16x2 Gram backward
    const float dseen = amax_peek(amax_d);
    const float f = *factor;
    const float inv_c2 = 1.f / ((float)C * (float)C);
    float loss = 0.f;
    // STYLE_EPT elements per thread (one atomic on the single loss address per block). Two phases: every load of all
    // elements first (predicated, no stores in between - a store to D / history may alias a later element's loads, which
    // would put one memory round trip per element on the critical path: 44 us for the step's 20 problems), then the math
    // and the stores.
    float N[2], invN[2];
    bool empty[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float* S = k == 0 ? S0 : S1;
        N[k] = (S != nullptr) ? counts[k] : 0.f;
        empty[k] = !(N[k] > 0.f);
        invN[k] = empty[k] ? 0.f : 1.f / N[k];
    }
    const int cc = C * C;
    float Gs[STYLE_EPT][2], Y[STYLE_EPT][4], Hs[STYLE_EPT];
#pragma unroll
    for (int it = 0; it < STYLE_EPT; ++it) {
        const int idx = min((block_x * STYLE_EPT + it) * 256 + (int)threadIdx.x, cc - 1);
        const int i = idx / C, j = idx - i * C;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float* S = k == 0 ? S0 : S1;
            Gs[it][k] = (S != nullptr && !empty[k]) ? sym_read(S, C, i, j, n_slabs) : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) Y[it][t] = t < terms.n ? terms.target[t][idx] : 0.f;
        Hs[it] = 0.f;
        if (history)
            for (int h = 0; h < hist_len; ++h) Hs[it] += history[(size_t)h * cc + idx];
    }
#pragma unroll
    for (int it = 0; it < STYLE_EPT; ++it) {
        const int idx = (block_x * STYLE_EPT + it) * 256 + threadIdx.x;
        if (idx >= cc) continue;
        float G[2], d[2] = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 2; ++k) G[k] = empty[k] ? 0.f : Gs[it][k] / N[k];   // N == 0: masked_features -> zeros
        float navg = 1.f;
        float Gavg0 = G[0];
        if (history) {  // gram_mode 'average': mean over the current and up to 9 detached previous Grams (:319-323)
            navg = (float)(hist_len + 1);
            Gavg0 = (G[0] + Hs[it]) / navg;
            history[(size_t)hist_slot * cc + idx] = G[0];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t >= terms.n) continue;
            const int k = terms.mask[t];
            if (empty[k] && terms.skip_if_empty[k]) continue;
            const float g = (k == 0) ? Gavg0 : G[k];
            const float diff = g - Y[it][t];
            loss += diff * diff;
            // dL/dF = 2 dL/dS F (S symmetric), dL/dS = dL/dG / N, dL/dG = weight f (2/C^2) (G - Y) [/ navg]
            d[k] += diff * (4.f * weight * f * inv_c2 * invN[k] / ((k == 0) ? navg : 1.f));
        }
        D0[idx] = d[0];
        if (D1) D1[idx] = d[1];
        dmax = fmaxf(dmax, fmaxf(fabsf(d[0]), fabsf(d[1])));
    }
    record_amax(amax_d, dmax, dseen);
    const float tot = block_sum256(loss, red);
    if (threadIdx.x == 0 && tot != 0.f) atomicAdd(loss_out, tot * weight * f * inv_c2);
}
__global__ __launch_bounds__(256) void style_loss_kernel(const float* __restrict__ S0, const float* __restrict__ S1,
                                                         const float* __restrict__ counts,
                                                         const float* __restrict__ factor, StyleTerms terms,
                                                         float weight, int C, float* __restrict__ D0,
                                                         float* __restrict__ D1, float* loss_out, float* history,
                                                         int hist_len, int hist_slot, int n_slabs, float* amax_d) {
    style_loss_body(S0, S1, counts, factor, terms, weight, C, D0, D1, loss_out, history, hist_len, hist_slot, n_slabs,
                    amax_d, blockIdx.x);
}
// GROUPED: the terms of many (level, layer) problems in one launch, all adding into one loss value
struct StyleProb {
    const float* S0;
    const float* S1;
    const float* counts;
    const float* factor;
    StyleTerms terms;
    float weight;
    int C;
    float* D0;
    float* D1;
    float* history;
    int hist_len, hist_slot, n_slabs;
    float* amax_d;
};
constexpr int STYLE_MAX_GROUP = 20;
struct StyleGroup {
    StyleProb p[STYLE_MAX_GROUP];
    int first_block[STYLE_MAX_GROUP + 1];
    int n;
};
__global__ __launch_bounds__(256) void style_loss_group_kernel(StyleGroup G, float* loss_out) {
    int g = 0;
    for (int i = 1; i < G.n; ++i)
        if ((int)blockIdx.x >= G.first_block[i]) g = i;
    const StyleProb& P = G.p[g];
    style_loss_body(P.S0, P.S1, P.counts, P.factor, P.terms, P.weight, P.C, P.D0, P.D1, loss_out, P.history, P.hist_len,
                    P.hist_slot, P.n_slabs, P.amax_d, blockIdx.x - G.first_block[g]);
}

// ---------------------------------------------------------------------------------------------------
// K5c: dF[c][q] = m0[q] (D0 F)[c][q] + m1[q] (D1 F)[c][q]   -  GEMM M = C, N = positions, K = C
// Block tile 64 (channels) x 256 (positions), 4 waves side by side, K-chunk 16 channels.
// ---------------------------------------------------------------------------------------------------
template <int NMASK, bool RELU_GATE>
__global__ __launch_bounds__(256) void gram_backward_kernel(const float* __restrict__ feat,
                                                            const float* __restrict__ mask0,
                                                            const float* __restrict__ mask1, const float* __restrict__ D0,
                                                            const float* __restrict__ D1, float* __restrict__ dfeat, int C,
                                                            int plane, int q_begin, int q_end) {
    constexpr int KC = 16, BN = 256;
    __shared__ __attribute__((aligned(16))) float As[NMASK][KC * 64];
    __shared__ __attribute__((aligned(16))) float Bs[KC * BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int m0 = blockIdx.y * 64;
    const int q0 = q_begin + blockIdx.x * BN;
    const int wn = wave * 64;
    f32x16 acc[NMASK][2][2];
#pragma unroll
    for (int k = 0; k < NMASK; ++k)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[k][a][b][r] = 0.f;

    // masks are 0/1 and partition the valid pixels: a 32-position tile that is all-passed needs only D0, all-failed
    // only D1, outside the mask neither (its output is exactly 0). Wave-uniform flags per (mask, n-tile).
    bool live[NMASK][2];
    bool any_live = false;
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
        const int q = q0 + wn + nj * 32 + l31;
        live[0][nj] = __ballot(q < q_end && mask0[q] != 0.f) != 0ull;
        if (NMASK > 1) live[NMASK - 1][nj] = __ballot(q < q_end && mask1[q] != 0.f) != 0ull;
        any_live |= live[0][nj] | live[NMASK - 1][nj];
    }
    // block-uniform: is any position of the 256-wide tile inside any mask?
    __shared__ int block_live;
    if (tid == 0) block_live = 0;
    __syncthreads();
    if (any_live && lane == 0) atomicOr(&block_live, 1);
    __syncthreads();
    const bool skip_block = block_live == 0;

    for (int k0 = 0; k0 < (skip_block ? 0 : C); k0 += KC) {
        // D rows k0..k0+15, columns m0..m0+63 (D symmetric: D[k][m] == D[m][k]); 256 float4 per matrix
        {
            const int row = tid >> 4, c4 = (tid & 15) * 4;
            *reinterpret_cast<float4*>(&As[0][row * 64 + c4]) =
                *reinterpret_cast<const float4*>(D0 + (size_t)(k0 + row) * C + m0 + c4);
            if (NMASK > 1)
                *reinterpret_cast<float4*>(&As[NMASK - 1][row * 64 + c4]) =
                    *reinterpret_cast<const float4*>(D1 + (size_t)(k0 + row) * C + m0 + c4);
        }
        // F rows k0..k0+15, positions q0..q0+255: 1024 float4, 4 per thread
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = tid + it * 256;
            const int row = i >> 6, c4 = (i & 63) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q0 + c4 < q_end) v = *reinterpret_cast<const float4*>(feat + (size_t)(k0 + row) * plane + q0 + c4);
            *reinterpret_cast<float4*>(&Bs[row * BN + c4]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int kp = 0; kp < KC / 2; ++kp) {
            const int k = kp * 2 + lhi;
            const float b0 = Bs[k * BN + wn + l31], b1 = Bs[k * BN + wn + 32 + l31];
#pragma unroll
            for (int mk = 0; mk < NMASK; ++mk) {
                const float a0 = As[mk][k * 64 + l31], a1 = As[mk][k * 64 + 32 + l31];
                if (live[mk][0]) {
                    acc[mk][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[mk][0][0], 0, 0, 0);
                    acc[mk][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[mk][1][0], 0, 0, 0);
                }
                if (live[mk][1]) {
                    acc[mk][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[mk][0][1], 0, 0, 0);
                    acc[mk][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[mk][1][1], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
        const int q = q0 + wn + nj * 32 + l31;
        if (q >= q_end) continue;
        const float w0 = mask0[q];
        const float w1 = (NMASK > 1) ? mask1[q] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const size_t o = (size_t)c * plane + q;
                float v = w0 * acc[0][mi][nj][r];
                if (NMASK > 1) v += w1 * acc[NMASK - 1][mi][nj][r];
                if (RELU_GATE) v = (feat[o] > 0.f) ? v : 0.f;
                dfeat[o] = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------------
// K6: masked content MSE + gradient
// ---------------------------------------------------------------------------------------------------
constexpr int MSE_CG = 16;   // channels per block
__global__ __launch_bounds__(256) void mse_masked_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                         const float* __restrict__ mask, const float* __restrict__ count,
                                                         const float* __restrict__ factor, float weight,
                                                         float* __restrict__ dpred, float* loss_out, int C, int plane,
                                                         int q_begin, int q_end, int relu_gate) {
    __shared__ float red[4];
    // a block covers MSE_CG channels of its 1024 positions: one atomic on the (single) loss address per block, and
    // thousands of same-address atomics serialise at ~2 ns each
    const int q = q_begin + (blockIdx.x * 256 + threadIdx.x) * 4;
    const float N = *count;
    const float coef = (N > 0.f) ? weight * (*factor) / ((float)C * N) : 0.f;
    float part = 0.f;
    if (q < q_end) {
        const float4 m = *reinterpret_cast<const float4*>(mask + q);
        const int c_end = min(C, (int)(blockIdx.y + 1) * MSE_CG);
        for (int c = blockIdx.y * MSE_CG; c < c_end; ++c) {
        const size_t o = (size_t)c * plane + q;
        const float4 p = *reinterpret_cast<const float4*>(pred + o);
        const float4 t = *reinterpret_cast<const float4*>(target + o);
        float4 d = make_float4(m.x * (p.x - t.x), m.y * (p.y - t.y), m.z * (p.z - t.z), m.w * (p.w - t.w));
        part += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;  // m in {0,1}: m^2 = m
        const float c2 = 2.f * coef;
        float4 g = make_float4(c2 * d.x, c2 * d.y, c2 * d.z, c2 * d.w);
        if (relu_gate) {  // pred is the post-ReLU activation itself
            g.x = p.x > 0.f ? g.x : 0.f;
            g.y = p.y > 0.f ? g.y : 0.f;
            g.z = p.z > 0.f ? g.z : 0.f;
            g.w = p.w > 0.f ? g.w : 0.f;
        }
        *reinterpret_cast<float4*>(dpred + o) = g;
        }
    }
    const float tot = block_sum256(part, red);
    if (threadIdx.x == 0 && tot != 0.f) atomicAdd(loss_out, tot * coef);
}

}  // namespace sm

// positions per block of the grouped Gram forward: equal K ranges for every problem of a launch (equal block work);
// long enough to amortise the block's C^2-float atomic epilogue, short enough that the launch has >= ~4 blocks per CU
static int gram_group_qb(int MI, long long total_block_positions) {
    // total_block_positions = sum over problems of positions x tile pairs x masks
    const int sp = MI == 1 ? 64 : 32;
    static const int target = getenv("SM_GRAM_TARGET_BLOCKS") ? atoi(getenv("SM_GRAM_TARGET_BLOCKS")) : 1024;
    long long qb = total_block_positions / target;
    qb = (qb + sp - 1) / sp * sp;
    return (int)std::max<long long>(8 * sp, std::min<long long>(qb, MI == 1 ? 4096 : 2048));
}

template <int MI>
static int launch_gram_group(const sm_gram_problem* problems, const int* idx, int n, hipStream_t s) {
    if (n == 0) return 0;
    constexpr int TS = 64 * MI;
    long long tot = 0;
    for (int i = 0; i < n; ++i) {
        const sm_gram_problem& q = problems[idx[i]];
        const int T = q.C / TS;
        tot += (long long)q.H * sm::row_stride(q.W) * (T * (T + 1) / 2) * (q.mask1 ? 2 : 1);
    }
    const int qb = gram_group_qb(MI, tot);
    for (int i0 = 0; i0 < n; i0 += sm::GRAM_MAX_GROUP) {
        sm::GramGroup G{};
        G.n = std::min(n - i0, sm::GRAM_MAX_GROUP);
        G.first_block[0] = 0;
        for (int i = 0; i < G.n; ++i) {
            const sm_gram_problem& q = problems[idx[i0 + i]];
            const int Wp = sm::row_stride(q.W), T = q.C / TS;
            sm::GramProb& P = G.p[i];
            P = sm::GramProb{q.feat, q.mask0, q.mask1, q.S0, q.S1, q.amax_feat, q.C, sm::plane_size(q.H, q.W), Wp,
                             (q.H + 1) * Wp, qb, (q.H * Wp + qb - 1) / qb};
            G.first_block[i + 1] = G.first_block[i] + P.n_ranges * (T * (T + 1) / 2) * (q.mask1 ? 2 : 1);
        }
        bool diag_only = true;
        for (int i = 0; i < G.n; ++i) diag_only &= G.p[i].C == TS;
        const size_t lds = sm::gram_group_lds_bytes(MI, diag_only), lds_max = sm::gram_group_lds_bytes(MI);
        auto k = sm::gram_group_kernel<MI>;
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
            if (e != hipSuccess) return (int)e;
            attr_done = true;
        }
        hipLaunchKernelGGL(k, dim3(G.first_block[G.n]), dim3(256), lds, s, G);
        SM_LAUNCH_CHECK();
    }
    return 0;
}

// All problems in at most two launches (64-channel tiles for C % 128 != 0, 128-channel tiles otherwise). S0 / S1 must
// be zero on entry.
static int gram_masked_grouped_impl(const sm_gram_problem* problems, int n, hipStream_t s) {
    if (n < 1 || n > 256) return (int)hipErrorInvalidValue;
    int idx1[256], idx2[256], n1 = 0, n2 = 0;
    for (int i = 0; i < n; ++i) {
        const sm_gram_problem& q = problems[i];
        if (q.C % 64 != 0 || q.feat == nullptr || q.mask0 == nullptr || q.S0 == nullptr || (q.mask1 && !q.S1) ||
            q.amax_feat == nullptr)
            return (int)hipErrorInvalidValue;
        if (q.C % 128 == 0) idx2[n2++] = i; else idx1[n1++] = i;
    }
    const int rc = launch_gram_group<2>(problems, idx2, n2, s);
    if (rc) return rc;
    return launch_gram_group<1>(problems, idx1, n1, s);
}


static int fill_style_terms(sm::StyleTerms& t, const float* const* targets, const int* term_mask, int n_terms,
                            const int* skip_if_empty, bool have1) {
    if (n_terms < 1 || n_terms > 4) return (int)hipErrorInvalidValue;
    t.n = n_terms;
    for (int i = 0; i < n_terms; ++i) {
        t.target[i] = targets[i];
        t.mask[i] = term_mask[i];
        if (term_mask[i] == 1 && !have1) return (int)hipErrorInvalidValue;
    }
    t.skip_if_empty[0] = skip_if_empty ? skip_if_empty[0] : 0;
    t.skip_if_empty[1] = skip_if_empty ? skip_if_empty[1] : 0;
    return 0;
}

template <int MI>
static int launch_gram_bwd_group(const sm_gram_bwd_problem* problems, const int* idx, int n, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += sm::GRAM_MAX_GROUP) {
        sm::GramBwdGroup G{};
        G.n = std::min(n - i0, sm::GRAM_MAX_GROUP);
        G.first_block[0] = 0;
        for (int i = 0; i < G.n; ++i) {
            const sm_gram_bwd_problem& q = problems[idx[i0 + i]];
            const int Wp = sm::row_stride(q.W), np = (q.H * Wp + 127) / 128;
            const bool two = q.mask1 && q.D1;
            sm::f32x4* P0 = reinterpret_cast<sm::f32x4*>(q.ws);
            G.p[i] = sm::GramBwdProb{q.feat, q.mask0, two ? q.mask1 : nullptr, P0, P0 + (size_t)6 * q.C * q.C / 16, q.dfeat,
                                     q.amax_feat, q.amax_d, q.amax_out, q.C, sm::plane_size(q.H, q.W), Wp, (q.H + 1) * Wp,
                                     q.relu_gate, np};
            G.first_block[i + 1] = G.first_block[i] + np * (q.C / (64 * MI));
        }
        hipLaunchKernelGGL((sm::gram_backward_group_kernel<MI>), dim3(G.first_block[G.n]), dim3(256), 0, s, G);
        SM_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" {

int sm_gram_num_slabs(int C, int H, int W) { return sm::gram_plan(C, H * sm::row_stride(W)).n_red; }

int sm_gram_workspace_slabs(int C, int H, int W) {
    const sm::GramPlan p = sm::gram_plan(C, H * sm::row_stride(W));
    return p.n_red == p.n_raw ? p.n_raw : p.n_red + p.n_raw;
}

int sm_gram_masked(const float* feat, const float* mask0, const float* mask1, float* S0, float* S1, int C, int H, int W,
                   void* stream) {
    if (C % 64 != 0) return (int)hipErrorInvalidValue;
    const int Wp = sm::row_stride(W), plane = sm::plane_size(H, W);
    const int q_begin = Wp, q_end = (H + 1) * Wp;
    const int T = C / 64;
    const sm::GramPlan p = sm::gram_plan(C, q_end - q_begin);
    const bool two_level = p.n_red != p.n_raw;
    const size_t cc = (size_t)C * C;
    float* raw0 = two_level ? S0 + p.n_red * cc : S0;   // reduced slabs first, raw slabs behind them
    float* raw1 = (two_level && S1) ? S1 + p.n_red * cc : S1;
    dim3 grid(p.n_raw, T * (T + 1) / 2);
    hipStream_t s = (hipStream_t)stream;
    if (mask1)
        hipLaunchKernelGGL(sm::gram_masked_kernel<2>, grid, dim3(256), 0, s, feat, mask0, mask1, raw0, raw1, C, plane,
                           q_begin, q_end, p.qb);
    else
        hipLaunchKernelGGL(sm::gram_masked_kernel<1>, grid, dim3(256), 0, s, feat, mask0, mask1, raw0, raw1, C, plane,
                           q_begin, q_end, p.qb);
    SM_LAUNCH_CHECK();
    if (two_level) {
        dim3 rg((unsigned)(cc / 256), p.n_red);
        hipLaunchKernelGGL(sm::gram_reduce_kernel, rg, dim3(256), 0, s, raw0, S0, (int)cc, p.n_raw);
        if (mask1) hipLaunchKernelGGL(sm::gram_reduce_kernel, rg, dim3(256), 0, s, raw1, S1, (int)cc, p.n_raw);
        SM_LAUNCH_CHECK();
    }
    return 0;
}

int sm_gram_split_num_slabs(void) { return 1; }

int sm_gram_masked_split2_grouped(const sm_gram_problem* problems, int n_problems, void* stream) {
    return gram_masked_grouped_impl(problems, n_problems, (hipStream_t)stream);
}

static int gram_masked_split_impl(const float* feat, const float* mask0, const float* mask1, float* S0, float* S1, int C,
                                  int H, int W, bool zero_fill, const float* amax_feat, void* stream) {
    if (C % 64 != 0) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    if (zero_fill) {   // the position ranges add into the slab
        const size_t cc = (size_t)C * C;
        hipError_t e = hipMemsetAsync(S0, 0, cc * sizeof(float), s);
        if (e == hipSuccess && mask1) e = hipMemsetAsync(S1, 0, cc * sizeof(float), s);
        if (e != hipSuccess) return (int)e;
    }
    const sm_gram_problem q{feat, mask0, mask1, S0, S1, amax_feat, C, H, W};
    return gram_masked_grouped_impl(&q, 1, s);
}

int sm_gram_masked_split(const float* feat, const float* mask0, const float* mask1, float* S0, float* S1, int C, int H,
                         int W, const float* amax_feat, void* stream) {
    return gram_masked_split_impl(feat, mask0, mask1, S0, S1, C, H, W, true, amax_feat, stream);
}

int sm_gram_masked_split_acc(const float* feat, const float* mask0, const float* mask1, float* S0, float* S1, int C,
                             int H, int W, const float* amax_feat, void* stream) {
    return gram_masked_split_impl(feat, mask0, mask1, S0, S1, C, H, W, false, amax_feat, stream);
}

size_t sm_gram_backward_split_ws_bytes(int C) { return (size_t)2 * 6 * C * C; }

int sm_gram_backward_split(const float* feat, const float* mask0, const float* mask1, const float* D0, const float* D1,
                           float* dfeat, int C, int H, int W, int relu_gate, void* ws, const float* amax_feat,
                           const float* amax_d, void* stream) {
    if (C % 64 != 0 || ws == nullptr || amax_feat == nullptr || amax_d == nullptr) return (int)hipErrorInvalidValue;
    const int Wp = sm::row_stride(W), plane = sm::plane_size(H, W);
    const int q_begin = Wp, q_end = (H + 1) * Wp;
    hipStream_t s = (hipStream_t)stream;
    const bool two = mask1 && D1;
    sm::f32x4* P0 = reinterpret_cast<sm::f32x4*>(ws);
    sm::f32x4* P1 = P0 + (size_t)6 * C * C / 16;
    hipLaunchKernelGGL(sm::gram_d_pack_kernel, dim3((C * (C / 8) + 255) / 256, two ? 2 : 1), dim3(256), 0, s, D0, D1,
                       P0, P1, C, amax_d);
    SM_LAUNCH_CHECK();
    const float* m1 = two ? mask1 : nullptr;
#define SM_GBS(MI_, RG)                                                                                               \
    hipLaunchKernelGGL((sm::gram_backward_split_kernel<MI_, RG>), dim3((q_end - q_begin + 127) / 128, C / (64 * MI_)), \
                       dim3(256), 0, s, feat, mask0, m1, P0, P1, dfeat, C, plane, q_begin, q_end, amax_feat, amax_d)
    // 128-row tiles only when they still fill the chip: deep layers of small levels have a handful of position tiles,
    // and a block's K loop (C x live masks) is serial
    if (C % 128 == 0 && (long long)((q_end - q_begin + 127) / 128) * (C / 128) >= 256) {
        if (relu_gate) SM_GBS(2, true); else SM_GBS(2, false);
    } else {
        if (relu_gate) SM_GBS(1, true); else SM_GBS(1, false);
    }
#undef SM_GBS
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_style_loss(const float* S0, const float* S1, const float* counts, const float* factor,
                  const float* const* targets, const int* term_mask, int n_terms, const int* skip_if_empty, float weight,
                  int C, float* D0, float* D1, float* loss_out, float* history, int hist_len, int hist_slot,
                  int n_slabs, float* amax_d_out, void* stream) {
    if ((C * C) % 256 != 0) return (int)hipErrorInvalidValue;
    sm::StyleTerms t;
    const int rc = fill_style_terms(t, targets, term_mask, n_terms, skip_if_empty, S1 != nullptr && D1 != nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(sm::style_loss_kernel, dim3((C * C + 256 * sm::STYLE_EPT - 1) / (256 * sm::STYLE_EPT)), dim3(256), 0, (hipStream_t)stream, S0, S1, counts, factor,
                       t, weight, C, D0, D1, loss_out, history, hist_len, hist_slot, n_slabs, amax_d_out);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_gram_backward(const float* feat, const float* mask0, const float* mask1, const float* D0, const float* D1,
                     float* dfeat, int C, int H, int W, int relu_gate, void* stream) {
    if (C % 64 != 0) return (int)hipErrorInvalidValue;
    const int Wp = sm::row_stride(W), plane = sm::plane_size(H, W);
    const int q_begin = Wp, q_end = (H + 1) * Wp;
    dim3 grid((q_end - q_begin + 255) / 256, C / 64);
    hipStream_t s = (hipStream_t)stream;
#define SM_GB(NM, RG)                                                                                                  \
    hipLaunchKernelGGL((sm::gram_backward_kernel<NM, RG>), grid, dim3(256), 0, s, feat, mask0, mask1, D0, D1, dfeat, C, \
                       plane, q_begin, q_end)
    if (mask1 && D1) {
        if (relu_gate) SM_GB(2, true); else SM_GB(2, false);
    } else {
        if (relu_gate) SM_GB(1, true); else SM_GB(1, false);
    }
#undef SM_GB
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_mse_masked(const float* pred, const float* target, const float* mask, const float* count, const float* factor,
                  float weight, float* dpred, float* loss_out, int C, int H, int W, int relu_gate, void* stream) {
    const int Wp = sm::row_stride(W), plane = sm::plane_size(H, W);
    const int q_begin = Wp, q_end = (H + 1) * Wp;
    dim3 grid(((q_end - q_begin) / 4 + 255) / 256, (C + sm::MSE_CG - 1) / sm::MSE_CG);
    hipLaunchKernelGGL(sm::mse_masked_kernel, grid, dim3(256), 0, (hipStream_t)stream, pred, target, mask, count, factor,
                       weight, dpred, loss_out, C, plane, q_begin, q_end, relu_gate);
    SM_LAUNCH_CHECK();
    return 0;
}


int sm_style_loss_grouped(const sm_style_problem* problems, int n_problems, float* loss_out, void* stream) {
    if (n_problems < 1 || loss_out == nullptr) return (int)hipErrorInvalidValue;
    for (int i0 = 0; i0 < n_problems; i0 += sm::STYLE_MAX_GROUP) {
        sm::StyleGroup G{};
        G.n = std::min(n_problems - i0, sm::STYLE_MAX_GROUP);
        G.first_block[0] = 0;
        for (int i = 0; i < G.n; ++i) {
            const sm_style_problem& q = problems[i0 + i];
            if ((q.C * q.C) % 256 != 0 || q.S0 == nullptr || q.D0 == nullptr) return (int)hipErrorInvalidValue;
            sm::StyleProb& P = G.p[i];
            const int rc = fill_style_terms(P.terms, q.targets, q.term_mask, q.n_terms, q.skip_if_empty, q.S1 && q.D1);
            if (rc) return rc;
            P.S0 = q.S0; P.S1 = q.S1; P.counts = q.counts; P.factor = q.factor; P.weight = q.weight; P.C = q.C;
            P.D0 = q.D0; P.D1 = q.D1; P.history = q.history; P.hist_len = q.hist_len; P.hist_slot = q.hist_slot;
            P.n_slabs = q.n_slabs; P.amax_d = q.amax_d_out;
            G.first_block[i + 1] = G.first_block[i] + (q.C * q.C + 256 * sm::STYLE_EPT - 1) / (256 * sm::STYLE_EPT);
        }
        hipLaunchKernelGGL(sm::style_loss_group_kernel, dim3(G.first_block[G.n]), dim3(256), 0, (hipStream_t)stream, G, loss_out);
        SM_LAUNCH_CHECK();
    }
    return 0;
}

int sm_gram_backward_split2_grouped(const sm_gram_bwd_problem* problems, int n_problems, void* stream) {
    if (n_problems < 1 || n_problems > 256) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    int idx1[256], idx2[256], n1 = 0, n2 = 0;
    for (int i = 0; i < n_problems; ++i) {
        const sm_gram_bwd_problem& q = problems[i];
        if (q.C % 64 != 0 || q.ws == nullptr || q.amax_feat == nullptr || q.amax_d == nullptr || q.D0 == nullptr)
            return (int)hipErrorInvalidValue;
        if (q.dfeat == nullptr) continue;   // pack only: the caller's conv epilogue consumes the operand images (SM_EPI_GRAM)
        if (q.C % 128 == 0) idx2[n2++] = i; else idx1[n1++] = i;
    }
    // operand images of all derivative matrices in one launch
    for (int i0 = 0; i0 < n_problems; i0 += sm::GRAM_MAX_GROUP) {
        sm::GramPackGroup G{};
        G.n = std::min(n_problems - i0, sm::GRAM_MAX_GROUP);
        G.first_block[0] = 0;
        for (int i = 0; i < G.n; ++i) {
            const sm_gram_bwd_problem& q = problems[i0 + i];
            const bool two = q.mask1 && q.D1;
            sm::f32x4* P0 = reinterpret_cast<sm::f32x4*>(q.ws);
            const int blocks = (q.C * (q.C / 8) + 255) / 256;
            G.p[i] = sm::GramPackProb{q.D0, two ? q.D1 : nullptr, P0, P0 + (size_t)6 * q.C * q.C / 16, q.amax_d, q.C, blocks};
            G.first_block[i + 1] = G.first_block[i] + blocks * (two ? 2 : 1);
        }
        hipLaunchKernelGGL(sm::gram_d_pack_group_kernel, dim3(G.first_block[G.n]), dim3(256), 0, s, G);
        SM_LAUNCH_CHECK();
    }
    const int rc = launch_gram_bwd_group<2>(problems, idx2, n2, s);
    if (rc) return rc;
    return launch_gram_bwd_group<1>(problems, idx1, n1, s);
}

int sm_sizeof_problem(int which) {
    switch (which) {
        case 0: return (int)sizeof(sm_conv_problem);
        case 1: return (int)sizeof(sm_plane_problem);
        case 2: return (int)sizeof(sm_gram_problem);
        case 3: return (int)sizeof(sm_style_problem);
        case 4: return (int)sizeof(sm_gram_bwd_problem);
        case 5: return (int)sizeof(sm_cover_problem);
        case 6: return (int)sizeof(sm_view_masks_desc);
        case 7: return (int)sizeof(sm_view_layer_mask);
        case 8: return (int)sizeof(sm_view_resize);
        case 9: return (int)sizeof(sm_view_lists_desc);
        case 10: return (int)sizeof(sm_view_list);
        case 11: return (int)sizeof(sm_call);
        default: return -1;
    }
}

}  // extern "C"

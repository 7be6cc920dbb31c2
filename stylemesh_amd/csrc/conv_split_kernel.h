// K3s/K4s: the 3x3 convolutions on the fp16 matrix cores at fp32 accuracy ("fp16x2 split, 3 products").
//
// Every fp32 operand x is scaled by a power of two s (from the bound its producer recorded) and written as h + l, two
// fp16 numbers: h = fp16(x s), l = fp16(x s - h) - 2 x 11 = 22 significand bits. A product x y is evaluated as the three
// partial products hh' + hl' + lh' (each exact in fp32; the dropped ll' is < 2^-22 of the product), summed in the MFMA's
// fp32 accumulators; the epilogue multiplies by the exact inverse scales. v_mfma_f32_32x32x16_f16 retires 16x the MACs
// per cycle of v_mfma_f32_32x32x2_f32, so three of them per fp32 MAC leave a 5.3x higher ceiling (2.5 PFLOP/s / 3 =
// 833 TFLOP/s fp32-equivalent, against 157 TFLOP/s). Measured against an fp64 convolution: tests/test_kernels_gpu.py.
// (Round 1's bf16x3 form - three parts, six products - shared this source as NP = 3 until round 6; profiles/r02 keeps
// its measurements.)
//
// Same GEMM view, tile scheduling, grouped launch, active-tile list, tail split-K and epilogues as conv.hip.
// Differences: weights arrive pre-split from the host ([9][Cin/16][2 parts][2 k-groups][Cout][8 ci] fp16, the exact
// image of an LDS weight stage); activations stay fp32 in HBM and are split when they are staged into LDS, as
// [ky slice][2 parts][2 k-groups][BN+2 positions][8 ci] fp16 - a tap shift is again a pure offset (16 bytes per
// position). One K-stage = one tap of a 16-channel chunk = ONE fp16 MFMA K-step (12 MFMAs per wave):
//   * weights: the global stage image is the MFMA A-fragment layout, so each wave loads its fragments straight into
//     a register ring three stages ahead - no LDS copy, no per-stage barrier;
//   * activations: the three ky slices of a chunk live in a ring of FOUR LDS slots; the next chunk's slice k is
//     loaded at tap 3k and written at the end of tap 3k+2 into the slot that the current chunk stopped reading
//     three taps earlier (one barrier per three stages); the next stage's fragments are read under the current
//     stage's MFMAs.
// Two blocks per CU (register-limited): the two waves that share a SIMD's matrix pipe belong to different blocks and
// cover each other's store / barrier phases.
#pragma once
#include "conv_common.h"

// tuning knobs (A/B builds: build.sh -DSM_SPLIT2_AD=1 ..., compared with tools/ab_libs.sh). What the loop is sensitive to
// is the NUMBER of vector-memory instructions (round-2 ablation: no weight loads +15 %, no activation loads +8 %, no
// conversion arithmetic / no LDS fragment reads / no epilogue stores +-1 %): hence the 32 x 128 wave tiles (SM_SPLIT_WGM =
// 4 in conv.hip) and the buffer loads. Any s_setprio, pinned fragment reads or a third wave per SIMD lose 8-14 %
// (profiles/r02 - r05 keep the records; the switches left with round 6).
#ifndef SM_SPLIT_PREFETCH_B
#define SM_SPLIT_PREFETCH_B 1  // read the next stage's activation fragments under this stage's MFMAs
#endif
#ifndef SM_SPLIT_WAVES64
#define SM_SPLIT_WAVES64 4     // resident waves per SIMD of the 64 x 128 (resident-input) variant: 118 VGPRs with 32-channel phases
#endif

namespace sm {

// ---- helpers of the SM_EPI_GRAM epilogue: the arithmetic of gram_backward_body (gram_split_kernels.h), restated here so
// that the fused form reproduces its bits (same operand scales, same fp16 pairs, same product order)
typedef _Float16 cg_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float conv_gram_pow2_scale(float amax, float& inv) {
    const unsigned bits = __builtin_bit_cast(unsigned, amax);
    const int ex = (int)((bits >> 23) & 0xff);
    if (ex < 16 || ex > 250) { inv = 1.f; return 1.f; }
    inv = __builtin_bit_cast(float, (unsigned)(ex - 14) << 23);
    return __builtin_bit_cast(float, (unsigned)(268 - ex) << 23);
}
__device__ __forceinline__ void conv_gram_split(const float (&x)[8], float sm, f32x4& vh, f32x4& vl) {
    cg_f16x8 h, l;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float xs = x[c] * sm;
        const _Float16 a = (_Float16)xs;
        h[c] = a;
        l[c] = (_Float16)(xs - (float)a);
    }
    vh = __builtin_bit_cast(f32x4, h);
    vl = __builtin_bit_cast(f32x4, l);
}
__device__ __forceinline__ void conv_gram_mfma(f32x16& acc, const f32x4 (&fa)[2], const f32x4 (&fb)[2]) {
#define SM_H(x_) __builtin_bit_cast(cg_f16x8, x_)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(SM_H(fa[1]), SM_H(fb[0]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(SM_H(fa[0]), SM_H(fb[1]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(SM_H(fa[0]), SM_H(fb[0]), acc, 0, 0, 0);
#undef SM_H
}

#ifndef SM_SPLIT2_PAIR_ROWS
#define SM_SPLIT2_PAIR_ROWS 1  // forward convs with the pooling epilogue: the lower segment of a pair re-uses the upper one's rows
#endif
constexpr int SM_SPLIT_NP = 2;      // parts per operand
constexpr int SM_SPLIT_SLOTS = 4;   // ky-slice ring of a block
constexpr size_t conv_split_lds_bytes(int BM, int BN) {
    return (size_t)(SM_SPLIT_SLOTS * 2 * SM_SPLIT_NP * (BN / 32 * 34)) * 16;   // 34 staged positions per 32-position segment
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// Live operands are scaled below 2^15. Positions of tiles the active-tile lists skip hold STALE values of earlier
// views, which the halo of a live tile reads (their products only reach outputs nobody uses): clamped, so that a stale
// value above the current maximum cannot become an fp16 infinity (and inf - inf a NaN) in a plane that masked sums
// later multiply by zero.
#define SM_F16_CLAMP 65000.f

// power-of-two scale that maps a tensor whose largest magnitude is amax into [2^14, 2^15) (1 for amax == 0 / denormal
// / non-finite), and its inverse
__device__ __forceinline__ float pow2_scale_for(float amax, float& inv) {
    const unsigned bits = __builtin_bit_cast(unsigned, amax);
    const int ex = (int)((bits >> 23) & 0xff);          // amax in [2^(ex-127), 2^(ex-126))
    if (ex < 16 || ex > 250) { inv = 1.f; return 1.f; }
    inv = __builtin_bit_cast(float, (unsigned)(ex - 14) << 23);      // 2^(ex - 141)
    return __builtin_bit_cast(float, (unsigned)(268 - ex) << 23);    // 2^(141 - ex)
}

#ifndef SM_SPLIT2_AD
#define SM_SPLIT2_AD 3
#endif
#ifndef SM_SPLIT2_WAVES
#define SM_SPLIT2_WAVES 2      // resident waves per SIMD of the 128-row fp16x2 variant
#endif
// resident waves per SIMD the register budget of a variant is set for
constexpr int conv_split_waves(int BM, int BN) {
    return BM == 256 ? 1 : (BM == 64 && BN == 128) ? SM_SPLIT_WAVES64 : SM_SPLIT2_WAVES;
}
// RES (round 5): RESIDENT INPUT, for the 64-output-channel launches (conv1_2 forward / data gradient, conv2_1's data
// gradient: K = 576 / 1152). The 64 x 256 tile stages every chunk's three ky slices for each of its eight free segments -
// 3 x 34 / 32 = 3.2 staged positions per output position and 16 channels - and spends twice the staging per MFMA of the
// 128-row tile. Here a tile is a QUAD: four vertically adjacent 32-position segments (list entries q, q + Wp, q + 2 Wp,
// q + 3 Wp; sm_cover_segments quad modes), and the block stages the (4 + 2) rows x 34 positions it needs of 64 input
// channels ONCE, already scaled and split: 6 x 34 x 64 channels x 2 parts x 2 B = 51 KB (in phases, see below), 1.59 staged
// positions per output position. The 36 stages of a 64-channel phase then are MFMAs, weight-fragment loads and LDS
// fragment reads only - a tap shift (ky, kx) is the offset ky * 34 + kx - with no barrier and no conversion inside the
// loop; Cin = 128 takes two phases. Same chunk / tap / product order as the ring kernel: the sums have its bits.
constexpr int SM_RES_ROWS = 6, SM_RES_RP = SM_RES_ROWS * 34;                    // staged rows / positions of a quad
// Round 6: a PHASE stages 32 channels (26 KB, one staging task per thread: 118 VGPRs) instead of 64 (51 KB, 149 VGPRs) - FOUR
// blocks per CU instead of three; the launches -4.5 % (profiles/r06/resident_phase32.txt), c3 +0.5 %. (What these launches wait
// for is neither the matrix pipe nor one latency chain: a persistent, cross-tile pipelined form - list entry, loads, weight
// ring and claim of the next quad under the current quad's loop - was built, is bit-identical and changes nothing; with every
// MFMA removed the launches get SLOWER, weight-fragment loads are worth 11-19 %, staging 16-28 %, the epilogue 12-51 %:
// profiles/r06/respipe_ablation.txt, LABNOTES 10.6.)
#ifndef SM_RES_PHASE
#define SM_RES_PHASE 32   // channels staged per phase (32 or 64)
#endif
constexpr int SM_RES_CC = SM_RES_PHASE / 16;                                    // 16-channel chunks of a phase
constexpr size_t conv_resident_lds_bytes() { return (size_t)(SM_RES_CC * 2 * 2 * SM_RES_RP) * 16; }   // [chunk][part][k-group][RP] units
// ---- the stores of a whole tile: scale, bias / ReLU, addend, the Gram term (accg: already summed), gate, border.
// gate_bits (Gram epilogue): one byte per (8-channel group, position of the block), bit c = channel 8 g + c of the gate
// operand > 0. Returns the lane's max |output|.
template <int BM, int BN, int WGM, int WGN, int FLAGS, bool RES>
__device__ __forceinline__ float conv_split_store_tile(const ConvProblem& P, const int (&qs)[BN / 32], const bool (&live)[BN / 32],
                                                       f32x16 (&acc)[BM / WGM / 32][BN / WGN / 32], const int m0,
                                                       const float out_scale, const f32x4 (&bias4)[BM / WGM / 32][4],
                                                       const f32x16 (&accg)[(FLAGS & SM_EPI_GRAM) ? BN / WGN / 32 : 1],
                                                       const float g_oscale, const unsigned char* gate_bits) {
    constexpr int MI = BM / WGM / 32, NJ = BN / WGN / 32, SEG = BN / 32;
    constexpr bool GRAM = (FLAGS & SM_EPI_GRAM) != 0;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int lhi = lane >> 5;
    const int wm = (wave / WGN) * (32 * MI);
    const int wn = (wave % WGN) * (32 * NJ);
    const int q_end = (P.H + 1) * P.Wp;
    float vmax = 0.f;
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
        int q_seg = qs[0];
        bool alive = live[0];
#pragma unroll
        for (int k = 1; k < SEG; ++k)
            if (wn / 32 + nj == k) { q_seg = qs[k]; alive = live[k]; }
        const int q = q_seg + l31;
        if (!alive || q >= q_end) continue;
        // (quads: the runs of a row group are disjoint within their rows, but a run that passes the end of its row would
        // continue on the first columns of the next one, which another run of the quad's next segment covers - a position
        // stored twice, added twice under SM_EPI_ADD: a lane stays in its segment's row)
        if (RES && q / P.Wp != q_seg / P.Wp) continue;
        const bool inside = interior(q, P.H, P.W, P.Wp);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const size_t o0 = (size_t)(m0 + wm + mi * 32 + 4 * lhi) * P.plane + q;
            // independent loads of all 16 rows first, then the 16 stores (no load -> store -> load chains)
            float prev[16], gate[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const size_t o = o0 + (size_t)((r & 3) + 8 * (r >> 2)) * P.plane;
                if (FLAGS & SM_EPI_ADD) prev[r] = P.out[o];
                if ((FLAGS & SM_EPI_RELU_MASK) && !GRAM) gate[r] = P.gate[o];
            }
            if constexpr (GRAM) {   // (MI == 1) the gate bits of this lane's 16 rows: byte g = channels wm + 8 g + 0..7
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned b8 = gate_bits[(wm / 8 + g) * BN + wn + nj * 32 + l31];
#pragma unroll
                    for (int k = 0; k < 4; ++k) gate[4 * g + k] = ((b8 >> (4 * lhi + k)) & 1u) ? 1.f : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const size_t o = o0 + (size_t)((r & 3) + 8 * (r >> 2)) * P.plane;
                float v = acc[mi][nj][r];
                v *= out_scale;
                if (FLAGS & SM_EPI_BIAS_RELU) v = fmaxf(v + bias4[mi][r >> 2][r & 3], 0.f);
                if (FLAGS & SM_EPI_ADD) v += prev[r];
                if constexpr (GRAM) v += accg[nj][r] * g_oscale;
                if (FLAGS & SM_EPI_RELU_MASK) v = (gate[r] > 0.f) ? v : 0.f;
                v = inside ? v : 0.f;
                P.out[o] = v;
                vmax = fmaxf(vmax, fabsf(v));
            }
        }
    }
    return vmax;
}

// ---- the epilogue of a WHOLE tile (same 32x32 C/D layout as conv3x3_mfma_kernel: column = lane & 31, row = (r & 3) +
// 8 * (r >> 2) + 4 * (lane >> 5); column tile j of the wave is in acc[.][j]): scale, bias / ReLU / gate / addend, the pooling
// or the Gram epilogue, stores. `bias`: the launch's bias vector; `smem4`: LDS the Gram epilogue may stage its
// operand in, GPH channels at a time ((2 * GPH / 8 * BN) * 16 + (BM / 8) * BN bytes, free of readers; the sums do not depend
// on GPH). Returns the lane's max |output|.
template <int BM, int BN, int WGM, int WGN, int FLAGS, bool RES, int GPH = 64>
__device__ __forceinline__ float conv_split_epilogue(const ConvArgs& a, const ConvProblem& P, const int (&qs_)[BN / 32],
                                                     const bool (&live_)[BN / 32],
                                                     f32x16 (&acc)[BM / WGM / 32][BN / WGN / 32], const int m0,
                                                     const float out_scale, const float* bias, f32x4* smem4) {
    constexpr int MI = BM / WGM / 32, NJ = BN / WGN / 32, SEG = BN / 32;
    constexpr int NP = SM_SPLIT_NP, SEGP = 34, BNP = SEG * SEGP, SLICE = 2 * NP * BNP;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int lhi = lane >> 5;
    const int wm = (wave / WGN) * (32 * MI);
    const int wn = (wave % WGN) * (32 * NJ);
    // (copies: every entry is read here, unconditionally - selecting among the caller's array elements by a computed segment
    // number becomes an indexed load from a scratch copy of the array otherwise)
    int qs[SEG];
    bool live[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) {
        qs[i] = qs_[i];
        live[i] = live_[i];
    }
    const int q_end = (P.H + 1) * P.Wp;
    // the 32 bias values of this lane's rows, as 8 float4 (rows (r&3) + 8(r>>2) + 4 lhi: groups of 4). (Hoisting these
    // loads above the main loop - so that they do not queue behind the last, unused prefetches - bought nothing.)
    f32x4 bias4[MI][4];
    if (FLAGS & SM_EPI_BIAS_RELU) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bias4[mi][g] = *reinterpret_cast<const f32x4*>(bias + m0 + wm + mi * 32 + 4 * lhi + 8 * g);
    }
    float vmax = 0.f;   // max |output| of this lane: the operand scale of the conv that consumes this tensor (NP = 2)
    if constexpr ((FLAGS & SM_EPI_POOL) != 0) {
        // Forward conv BELOW a 2x2 max-pool: the tile's segments come in vertical pairs (entries 2k, 2k + 1 of the list:
        // the same 32 columns of image rows 2Y and 2Y + 1, first column even), so every pooling window lies inside one
        // wave - rows in two accumulator tiles of the same lane, columns in neighbouring lanes. The epilogue stores the
        // POOLED map and the pool's argmax codes (the formats of maxpool_fwd_codes_kernel); the full-resolution output,
        // which only the pool would read, is never written: no pool pass, 1.75 plane sizes of HBM traffic less.
        static_assert(FLAGS == (SM_EPI_BIAS_RELU | SM_EPI_POOL) && NJ % 2 == 0, "forward epilogue, segment pairs per wave");
        const int Ho = P.H >> 1, Wo = P.W >> 1, Wpo = row_stride(Wo), plane_o = plane_size(Ho, Wo);
#pragma unroll
        for (int pj = 0; pj < NJ; pj += 2) {
            int q_seg = qs[0];
            bool alive = live[0];
#pragma unroll
            for (int k = 1; k < SEG; ++k)
                if (wn / 32 + pj == k) { q_seg = qs[k]; alive = live[k]; }
            if (!alive) continue;                                  // (wave-uniform: a padding pair)
            const int q = q_seg + l31;                             // this lane's position in the upper row
            const int yy = q / P.Wp - 1, xx = q - (yy + 1) * P.Wp - 1;
            // even lanes own a window; a segment that runs past the end of its row holds nothing there
            const bool ok = ((l31 | yy | xx) & 1) == 0 && (unsigned)yy < (unsigned)(2 * Ho) && (unsigned)xx < (unsigned)(2 * Wo);
            const int qo = ((yy >> 1) + 1) * Wpo + (xx >> 1) + 1;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                unsigned codes[4] = {0u, 0u, 0u, 0u};              // one dword per 8-channel group: this lane's 4 nibbles
                const size_t o0 = (size_t)(m0 + wm + mi * 32 + 4 * lhi) * plane_o + qo;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float t = acc[mi][pj][r], b = acc[mi][pj + 1][r];
                    t *= out_scale;
                    b *= out_scale;
                    const float bv = bias4[mi][r >> 2][r & 3];
                    t = fmaxf(t + bv, 0.f);
                    b = fmaxf(b + bv, 0.f);
                    // the window partner (lane ^ 1) through a DPP quad permute [1, 0, 3, 2] (round 5; an LDS permute before: -1 % on the launch)
                    const float tp = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, true));
                    const float bp = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, b), 0xB1, 0xF, 0xF, true));
                    // first maximum in row-major order (strict '>' scan, as max_pool2d_with_indices); 4: maximum <= 0
                    float m = t;
                    unsigned c = 0u;
                    if (tp > m) { m = tp; c = 1u; }
                    if (b > m) { m = b; c = 2u; }
                    if (bp > m) { m = bp; c = 3u; }
                    if (!(m > 0.f)) c = 4u;
                    vmax = ok ? fmaxf(vmax, m) : vmax;         // (bound of the POOLED map: what the next conv reads)
                    if (ok) P.pool_out[o0 + (size_t)((r & 3) + 8 * (r >> 2)) * plane_o] = m;
                    codes[r >> 2] |= c << (4 * ((r & 3) + 4 * lhi));
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned full = codes[g] | (unsigned)__shfl_xor((int)codes[g], 32, 64);   // channels 8g + 0..3 | 4..7
                    if (ok && lhi == 0) P.pool_code[(size_t)((m0 + wm + mi * 32) / 8 + g) * plane_o + qo] = full;
                }
            }
        }
        return vmax;
    }
    // SM_EPI_GRAM: the output layer is a style layer of C = BM channels (the block's row tile holds all of them) and this
    // epilogue adds its masked Gram backward, sum_k m_k(q) (D_k F)(q), F = the gate operand - C / 16 sixteen-channel steps
    // per mask on the matrix cores for each of the wave's 32-position column tiles, in gram_backward_body's operand format and product order (the sum has its bits);
    // the gradient plane is then written once instead of written (Gram backward), read (here) and written again.
    constexpr bool GRAM = (FLAGS & SM_EPI_GRAM) != 0;
    float g_fscale = 1.f, g_oscale = 1.f;
    if constexpr (GRAM) {
        static_assert(MI == 1 && (BM == 64 || BM == 128) && (FLAGS & SM_EPI_RELU_MASK) && !(FLAGS & SM_EPI_ADD),
                      "the Gram term replaces the addend of a data gradient whose row tile holds all C = BM channels");
        float inv_f, inv_d;
        g_fscale = conv_gram_pow2_scale(amax_read(P.gram_amax_feat), inv_f);
        conv_gram_pow2_scale(amax_read(P.gram_amax_d), inv_d);
        g_oscale = inv_f * inv_d;
    }
    // Round 4: the operand F is staged ONCE per block through the (now idle) slice ring - 64 channels at a time, already
    // scaled and split, in the main loop's unit format [part][k-group][position] - instead of being loaded and converted
    // by every wave that shares the positions (2 of 4 waves on the 64 x 256 tile, all 4 on the 128 x 128 one) in a
    // load -> convert -> MFMA chain per 32 channels; the derivative matrices' fragments are fetched once per k-step for the
    // wave's NJ column tiles, whose NJ accumulators take the MFMAs interleaved. Per column tile the products arrive in the
    // order they always did (chunk, mask, k-step): the sums keep their bits.
    f32x16 accg[GRAM ? NJ : 1];
    if constexpr (GRAM) {
        static_assert(GPH == 64 || GPH == 32, "staging phases of 64 channels (the whole ring) or 32 (half a resident window)");
        constexpr int KG = GPH / 8;             // k-groups of a phase: the staged image is [part][k-group][position]
        constexpr int PH = BM / GPH;            // phases
        constexpr int CH = GPH / 32;            // 32-channel chunks of a phase
        constexpr int GU = KG * BN / 256;       // staging units (k-group, position) per thread and phase
        static_assert(GPH != 64 || (2 * 8 * BN) * 16 + (BM / 8) * BN <= (RES ? conv_resident_lds_bytes() : (size_t)SM_SPLIT_SLOTS * SLICE * 16),
                      "a phase of the Gram operand + the gate bits of all channels fit the slice ring");
        f32x4* Gs = smem4;
        // F is also the ReLU gate of this launch's output: the staging threads - which hold the raw values - leave one bit
        // per (channel, position), x > 0, behind the operand (byte [channel / 8][position]); the store loop below reads its
        // gate from there instead of loading the layer a second time (conv1_2's data gradient moved 1.27 GB, a third of it
        // this second read)
        unsigned char* Gb = reinterpret_cast<unsigned char*>(smem4 + 2 * KG * BN);
        float mk[NJ][2];
        bool anyk[NJ][2], alive_j[NJ];
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
            int q_seg = qs[0];
            alive_j[nj] = live[0];
#pragma unroll
            for (int k = 1; k < SEG; ++k)
                if (wn / 32 + nj == k) { q_seg = qs[k]; alive_j[nj] = live[k]; }
            const int q = q_seg + l31;
            const bool valid = alive_j[nj] && q < q_end;
            const int qc = valid ? q : q_seg;                          // (lanes past the plane's end load a valid address)
            mk[nj][0] = valid ? P.gram_mask0[qc] : 0.f;
            mk[nj][1] = (valid && P.gram_mask1) ? P.gram_mask1[qc] : 0.f;
            anyk[nj][0] = __ballot(mk[nj][0] != 0.f) != 0ull;
            anyk[nj][1] = __ballot(mk[nj][1] != 0.f) != 0ull;
#pragma unroll
            for (int r = 0; r < 16; ++r) accg[nj][r] = 0.f;
        }
        // staging unit u of this thread: position tid % BN of the block, k-group tid / BN + u * (256 / BN)
        const int g_pos = tid & (BN - 1);
        int g_q = qs[0];
#pragma unroll
        for (int k = 1; k < SEG; ++k)
            if ((g_pos >> 5) == k) g_q = qs[k];
        g_q += g_pos & 31;
        if (g_q >= q_end) g_q -= g_pos & 31;                           // (as above: a valid address, masked to zero later)
        const f32x4* gp = P.gram_p + lhi * BM + wm + l31;            // operand unit (k-step t, part p): gp[(t * 4 + p * 2) * BM]
        const f32x4* gp1 = gp + (size_t)6 * BM * BM / 16;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ph = 0; ph < PH; ++ph) {
            __syncthreads();                                           // the ring's (the previous phase's) last readers are through
            {
                float rb[GU][8];
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    const int kg = tid / BN + u * (256 / BN);
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        rb[u][c] = P.gate[(size_t)(ph * GPH + kg * 8 + c) * P.plane + g_q];
                }
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    const int kg = tid / BN + u * (256 / BN);
                    f32x4 vh, vl;
                    conv_gram_split(rb[u], g_fscale, vh, vl);
                    Gs[kg * BN + g_pos] = vh;
                    Gs[(KG + kg) * BN + g_pos] = vl;
                    unsigned bits = 0u;
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        bits |= (rb[u][c] > 0.f ? 1u : 0u) << c;
                    Gb[(ph * KG + kg) * BN + g_pos] = (unsigned char)bits;
                }
            }
            __syncthreads();
#pragma unroll
            for (int chunk = 0; chunk < CH; ++chunk)                   // 32 channels
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int t = (ph * CH + chunk) * 2 + ks;      // k-step of 16 channels
                        const f32x4* gk = k == 0 ? gp : gp1;
                        f32x4 fa[2];
                        fa[0] = gk[(t * 4 + 0) * BM];
                        fa[1] = gk[(t * 4 + 2) * BM];
#pragma unroll
                        for (int nj = 0; nj < NJ; ++nj) {
                            if (!alive_j[nj] || !anyk[nj][k]) continue;   // (wave-uniform)
                            const f32x4* gf = Gs + ((chunk * 2 + ks) * 2 + lhi) * BN + wn + nj * 32 + l31;
                            const bool keep = mk[nj][k] != 0.f;
                            f32x4 fb[2];
                            fb[0] = keep ? gf[0] : zero4;
                            fb[1] = keep ? gf[KG * BN] : zero4;
                            conv_gram_mfma(accg[nj], fa, fb);
                        }
                    }
        }
    }
    return fmaxf(vmax, conv_split_store_tile<BM, BN, WGM, WGN, FLAGS, RES>(P, qs, live, acc, m0, out_scale, bias4, accg, g_oscale,
                                                                            reinterpret_cast<const unsigned char*>(smem4 + 2 * (GPH / 8) * BN)));
}

template <int BM, int BN, int WGM, int WGN, int FLAGS, bool UNPOOL = false, bool RES = false>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(conv_split_waves(BM, BN), conv_split_waves(BM, BN))))
void conv3x3_split_kernel(ConvArgs a) {
    constexpr int NP = SM_SPLIT_NP;
    static_assert(!RES || (BM == 64 && BN == 128 && WGM == 2 && WGN == 2),
                  "resident input: 64 x 128 tiles (a quad of segments), waves 2 x 2");
    constexpr int MI = BM / WGM / 32;     // 32-row MFMA tiles per wave: 2 (128-row blocks) or 1 (64-row blocks)
    constexpr int NJ = BN / WGN / 32;     // 32-position MFMA tiles per wave: 2 (waves 2 x 2) or 4 (waves 4 x 1)
    static_assert((MI == 1 || MI == 2) && BM == WGM * MI * 32 && (NJ == 2 || NJ == 4) && BN == WGN * NJ * 32 &&
                      WGM * WGN == 4,
                  "wave tile is (32 MI) x (32 NJ)");
    static_assert(BN == 128 || BN == 256, "activation staging: BN / 128 (k-group, position) units per thread + a 2 x 2 x 8 halo");
    constexpr int NU = BN / 128;          // staging units per thread and slice
    constexpr int KC = 16;
    // A block's BN positions are BN / 32 SEGMENTS of 32 consecutive positions - with an active-tile list ANY live segments
    // of one problem (ConvArgs::tile_list holds segments, round 3: the dead work inside 128-position tiles was 8 % of all
    // MFMAs), else consecutive ones. Every segment is staged with its own halo: 34 positions per segment and slice.
    constexpr int SEG = BN / 32;
    constexpr int SEGP = 34;
    constexpr int BNP = SEG * SEGP;       // staged positions per slice
    constexpr int SLICE = 2 * NP * BNP;   // 16-byte units of one ky slice of a chunk: [part][kgroup][position]
    extern __shared__ __attribute__((aligned(16))) f32x4 smem4[];
    f32x4* Bs = smem4;   // [4 slots][SLICE]

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int lhi = lane >> 5;
    const int wm = (wave / WGN) * (32 * MI);
    const int wn = (wave % WGN) * (32 * NJ);

    int tile, split;
    conv_unit(a, blockIdx.x, tile, split);   // XCD-aware order of whole tiles and of the tail's (tile, K-split) units
    const int m_tile = tile / a.n_tiles;
    const int n_glob = tile - m_tile * a.n_tiles;
    ConvProblem P = a.p[0];
    int qs[SEG];        // first position of each segment (index into the padded plane); block-uniform
    bool live[SEG];     // false: a padding entry of the list (nothing is stored for it)
    if (a.tile_list) {
        const int* e = a.tile_list + (size_t)n_glob * SEG;   // SEG entries (problem << 24) | first position, 0xFFFFFF = padding
        const int gsel = e[0] >> 24;
#pragma unroll
        for (int g = 1; g < SM_MAX_GROUP; ++g)
            if (g == gsel) P = a.p[g];
        const int s0 = e[0] & 0xFFFFFF;
#pragma unroll
        for (int i = 0; i < SEG; ++i) {
            const int sg = e[i] & 0xFFFFFF;
            live[i] = sg != 0xFFFFFF;
            qs[i] = live[i] ? sg : s0;                      // (a padding entry stages segment 0's data again)
        }
    } else {
        int n_tile = n_glob;
#pragma unroll
        for (int g = 1; g < SM_MAX_GROUP; ++g)
            if (g < a.n_problems && n_glob >= a.tile_begin[g]) {
                P = a.p[g];
                n_tile = n_glob - a.tile_begin[g];
            }
#pragma unroll
        for (int i = 0; i < SEG; ++i) {
            live[i] = true;
            qs[i] = P.Wp + n_tile * BN + 32 * i;
        }
    }
    const int n_chunks = a.Cin_pad / KC;
    const int ch_begin = split < 0 ? 0 : split * a.chunks_per_split;
    const int ch_end = split < 0 ? n_chunks : min(n_chunks, ch_begin + a.chunks_per_split);
    const int m0 = m_tile * BM;

    const float amax_seen = split < 0 ? amax_peek(a.amax_out) : 0.f;   // whole tiles record their output's bound
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- staging plan
    // weights: the global image of a stage, [part][kgroup][Cout] units of 8 fp16, IS the MFMA A-fragment layout
    // (row = lane & 31, k-group = lane >> 5), so every wave loads its own 3 MI fragments (MI row tiles x 3 parts) of a
    // stage straight into registers, three stages ahead: no LDS copy of the weights and no per-stage barrier.
    // (buffer loads: scalar base in an SGPR resource + one 32-bit lane offset + a scalar stage offset - no 64-bit
    // address arithmetic per load)
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.wt), 0, 9 * (a.Cin_pad / KC) * 2 * NP * a.Cout * 16, 0x00020000);
    const int a_voff = (lhi * a.Cout + m0 + wm + l31) * 16;   // bytes
    const int a_part = 2 * a.Cout * 16;            // bytes between the parts of a stage
    const int a_stage_bytes = 2 * NP * a.Cout * 16;   // bytes per (tap, chunk) stage
    // operand scale from the producer's recorded max |x| (one vector load of the bound's slots per wave)
    float inv_in;
    const float in_scale = pow2_scale_for(a.amax_in ? amax_read(a.amax_in) : 1.f, inv_in);
    const float out_scale = inv_in * a.w_scale_inv;
    // activations, per ky slice: thread -> unit (kgroup = tid / 128, position px = tid % 128), the 8 channels of the
    // k-group at stride `plane`; the 2 remaining halo positions x 2 k-groups x 8 channels = 32 single elements are
    // fetched one per lane (every half-wave does the same 32: identical values to identical addresses)
    const int b_kg = tid >> 7, b_px = tid & 127;
    // resource base one row + one float before the plane (inside the guard), so that every offset is >= 0
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(P.in) - P.Wp - 1, 0, 0x7ffffff0, 0x00020000);
    // unit u of this thread: position b_px + 128 u of the block = position (b_px & 31) of segment (b_px >> 5) + 4 u;
    // LDS position p of a segment holds input position qs - 1 + p (p = 0: left halo)
    int b_src[NU], b_dst[NU], b_q[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        int q_seg = qs[4 * u];
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if ((b_px >> 5) == k) q_seg = qs[4 * u + k];
        b_q[u] = q_seg + (b_px & 31);                              // position index + 1 of the staged element, row ky = 1
        b_src[u] = (b_kg * 8 * P.plane + b_q[u]) * 4;              // bytes, relative to the shifted base, row ky = 0
        b_dst[u] = b_kg * BNP + ((b_px >> 5) + 4 * u) * SEGP + (b_px & 31);   // + part * 2 * BNP (+ slot * SLICE)
    }
    // the two remaining positions of every segment (p = 32, 33): wave w fetches those of segment w (SEG = 4) or of
    // segments 2 w + (lane >> 5) (SEG = 8) - 2 positions x 2 k-groups x 8 channels = 32 single elements per segment
    const int h_kg = l31 >> 4, h_which = (l31 >> 3) & 1, h_c = l31 & 7;
    int h_seg = SEG == 4 ? wave : 2 * wave + lhi;
    int h_qs = qs[0];
#pragma unroll
    for (int k = 1; k < SEG; ++k)
        if (h_seg == k) h_qs = qs[k];
    const int h_q = h_qs + 32 + h_which;                          // position index + 1 of the halo element, row ky = 1
    const int h_src = ((h_kg * 8 + h_c) * P.plane + h_q) * 4;     // bytes, same base
    // UNPOOL: the operand is the max-pool backward of the pooled gradient `in`, taken on the fly. Geometry of the pooled
    // planes, and row / column (in the un-pooled image) of the CENTRE-row position of every staging unit of this thread
    // (slice ky reads row + ky - 1); -1 marks the padding column left of the image.
    const int up_Ho = P.H >> 1, up_Wo = P.W >> 1, up_Wp = row_stride(up_Wo), up_plane = plane_size(up_Ho, up_Wo);
    __amdgpu_buffer_rsrc_t gp_rsrc, code_rsrc;
    int up_y[NU], up_x[NU], up_hy = 0, up_hx = 0;
    if constexpr (UNPOOL) {
        gp_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.in), 0, 0x7ffffff0, 0x00020000);
        code_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(P.code), 0, 0x7ffffff0, 0x00020000);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int qc = b_q[u] - 1;
            up_y[u] = qc / P.Wp - 1;
            up_x[u] = qc - (up_y[u] + 1) * P.Wp - 1;
        }
        const int qh = h_q - 1;
        up_hy = qh / P.Wp - 1;
        up_hx = qh - (up_hy + 1) * P.Wp - 1;
    }
    // pooled offset (elements) and window parity of un-pooled (y, x); parity -1: outside every pooling window
#define SM_UP_MAP(y_, x_, off_, par_)                                                                    \
    {                                                                                                    \
        const bool ok_ = (unsigned)(y_) < (unsigned)(2 * up_Ho) && (unsigned)(x_) < (unsigned)(2 * up_Wo); \
        off_ = ok_ ? (((y_) >> 1) + 1) * up_Wp + ((x_) >> 1) + 1 : 0;                                    \
        par_ = ok_ ? ((((y_) & 1) << 1) | ((x_) & 1)) : -1;                                              \
    }
    const int h_dst = (h_kg * BNP + h_seg * SEGP + 32 + h_which) * 8 + h_c;   // in fp16 elements (+ part * 2 * BNP * 8)
    // PAIRS (round 4): a forward conv with the pooling epilogue takes its segments in vertical pairs - entries 2k, 2k + 1 =
    // the same 32 columns of image rows 2Y and 2Y + 1 - so slice ky of the LOWER segment holds the input row that slice
    // ky + 1 of the UPPER one holds. The lower segment's slices 0 and 1 are therefore not staged at all: its fragments for
    // taps ky = 0, 1 are read from the upper segment's slices 1, 2 (complete whenever the lower one's would be: the three
    // slices of a chunk are published before its first tap, and a ring slot is only re-used after the taps that read it,
    // see SM_NEXT_SLOT). Four staged rows per pair and chunk instead of six - the 64-row tile spends twice the staging
    // per MFMA of the 128-row tile, and staging is what bounds it. Bit-identical: the same input
    // values go through the same conversion. Slices 0 / 1 are staged with a mapping of their own over the UPPER segments:
    // one unit per thread for BN = 256 (instead of two); for BN = 128 half the threads convert and store.
    constexpr bool PAIRS = !UNPOOL && (FLAGS & SM_EPI_POOL) != 0 && SM_SPLIT2_PAIR_ROWS;
    const int a_kg = BN == 256 ? (tid >> 7) : ((tid >> 6) & 1);
    const int a_px = BN == 256 ? (tid & 127) : (tid & 63);
    const bool a_active = BN == 256 || tid < 128;
    int a_src = 0, a_dst = 0;
    if constexpr (PAIRS) {
        int q_seg = qs[0];
#pragma unroll
        for (int k = 1; k < SEG / 2; ++k)
            if ((a_px >> 5) == k) q_seg = qs[2 * k];
        a_src = (a_kg * 8 * P.plane + q_seg + (a_px & 31)) * 4;
        a_dst = a_kg * BNP + (2 * (a_px >> 5)) * SEGP + (a_px & 31);
    }
    // weight prefetch distance in stages = ring size; slot of a stage = tap % AD. A stage of the fp16x2 variant has half
    // the MFMA time to hide the same fetch latency behind: its ring is deeper
#ifndef SM_RES_AD
#define SM_RES_AD SM_SPLIT2_AD
#endif
    constexpr int AD = RES ? SM_RES_AD : SM_SPLIT2_AD;
    static_assert(9 % AD == 0, "ring slot of a stage is the same in every chunk");
    f32x4 ra[AD][MI][NP];
    // in-flight activation loads: one register set - a slice is loaded two stages before it is converted and stored
    // (register sets 1 and 2 only carry the prologue's three slices)
    constexpr int NSET = 3;
    float rbs[NSET][NU][8], rhs[NSET];
    unsigned rcs[UNPOOL ? NSET : 1][NU], rhc[UNPOOL ? NSET : 1];   // UNPOOL: argmax codes of the loaded gradients (one
    int rps[UNPOOL ? NSET : 1][NU], rhp[UNPOOL ? NSET : 1];        // nibble per channel) and each unit's window parity

#define SM_LOAD_A(tap_, chunk_)                                                                          \
    {                                                                                                    \
        const int so_ = ((tap_) * n_chunks + (chunk_)) * a_stage_bytes;                                  \
        _Pragma("unroll") for (int s = 0; s < NP; ++s)                                                   \
            _Pragma("unroll") for (int i = 0; i < MI; ++i)                                               \
                ra[(tap_) % AD][i][s] = __builtin_bit_cast(                                              \
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, a_voff, so_ + s * a_part + i * 512, 0)); \
    }
#define SM_LOAD_B(set_, ky_, chunk_)                                                                     \
    if constexpr (UNPOOL) {                                                                              \
        /* (everything that varies inside a wave - k-group, halo channel - goes into the VECTOR offset: a scalar  \
           offset the compiler cannot prove wave-uniform turns every load into a waterfall loop) */     \
        const int sc_ = (chunk_) * KC * up_plane * 4;   /* bytes, wave-uniform */                        \
        _Pragma("unroll") for (int u = 0; u < NU; ++u) {                                                 \
            int off_;                                                                                    \
            SM_UP_MAP(up_y[u] + (ky_) - 1, up_x[u], off_, rps[set_][u])                                  \
            _Pragma("unroll") for (int c = 0; c < 8; ++c)                                                \
                rbs[set_][u][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gp_rsrc, (off_ + b_kg * 8 * up_plane) * 4, sc_ + c * up_plane * 4, 0)); \
            /* the eight channels' codes: one dword of the [Cin / 8][plane] code image */               \
            rcs[set_][u] = __builtin_amdgcn_raw_buffer_load_b32(code_rsrc, (off_ + b_kg * up_plane) * 4, (chunk_) * 2 * up_plane * 4, 0); \
        }                                                                                                \
        int off_;                                                                                        \
        SM_UP_MAP(up_hy + (ky_) - 1, up_hx, off_, rhp[set_])                                             \
        rhs[set_] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gp_rsrc, (off_ + (h_kg * 8 + h_c) * up_plane) * 4, sc_, 0)); \
        rhc[set_] = __builtin_amdgcn_raw_buffer_load_b32(code_rsrc, (off_ + h_kg * up_plane) * 4, (chunk_) * 2 * up_plane * 4, 0); \
    } else {                                                                                             \
        const int so_ = ((chunk_) * KC * P.plane + (ky_) * P.Wp) * 4;                                    \
        if (PAIRS && (ky_) < 2) {   /* the upper segments only (unconditional loads: see the note at the loop) */ \
            _Pragma("unroll") for (int c = 0; c < 8; ++c)                                                \
                rbs[set_][0][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, a_src, so_ + c * P.plane * 4, 0)); \
        } else {                                                                                         \
            _Pragma("unroll") for (int u = 0; u < NU; ++u)                                               \
                _Pragma("unroll") for (int c = 0; c < 8; ++c)                                            \
                    rbs[set_][u][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, b_src[u], so_ + c * P.plane * 4, 0)); \
        }                                                                                                \
        rhs[set_] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, h_src, so_, 0)); \
    }
#define SM_STORE_B(set_, slot_, ky_)                                                                     \
    {                                                                                                    \
        f32x4* d_ = Bs + (slot_) * SLICE;                                                                \
        if constexpr (UNPOOL) {   /* gradient only at the window element that held the maximum */       \
            _Pragma("unroll") for (int u = 0; u < NU; ++u)                                               \
                _Pragma("unroll") for (int c = 0; c < 8; ++c)                                            \
                    rbs[set_][u][c] = ((int)((rcs[set_][u] >> (4 * c)) & 15u) == rps[set_][u]) ? rbs[set_][u][c] : 0.f; \
            rhs[set_] = ((int)((rhc[set_] >> (4 * h_c)) & 15u) == rhp[set_]) ? rhs[set_] : 0.f;          \
        }                                                                                                \
        if (PAIRS && (ky_) < 2) {                                                                        \
            f16x8 vh, vl;                                                                                \
            _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                              \
                const float xs_ = __builtin_amdgcn_fmed3f(rbs[set_][0][c] * in_scale, -SM_F16_CLAMP, SM_F16_CLAMP); \
                const _Float16 h_ = (_Float16)xs_;                                                       \
                vh[c] = h_; vl[c] = (_Float16)(xs_ - (float)h_);                                         \
            }                                                                                            \
            if (a_active) {                                                                              \
                d_[a_dst] = __builtin_bit_cast(f32x4, vh);                                               \
                d_[a_dst + 2 * BNP] = __builtin_bit_cast(f32x4, vl);                                     \
            }                                                                                            \
        } else {                                                                                         \
        _Pragma("unroll") for (int u = 0; u < NU; ++u) {                                                 \
            f16x8 vh, vl;                                                                                \
            _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                              \
                const float xs_ = __builtin_amdgcn_fmed3f(rbs[set_][u][c] * in_scale, -SM_F16_CLAMP, SM_F16_CLAMP); \
                const _Float16 h_ = (_Float16)xs_;                                                       \
                vh[c] = h_; vl[c] = (_Float16)(xs_ - (float)h_);                                         \
            }                                                                                            \
            d_[b_dst[u]] = __builtin_bit_cast(f32x4, vh);                                                \
            d_[b_dst[u] + 2 * BNP] = __builtin_bit_cast(f32x4, vl);                                      \
        }                                                                                                \
        }                                                                                                \
        const float xs_ = __builtin_amdgcn_fmed3f(rhs[set_] * in_scale, -SM_F16_CLAMP, SM_F16_CLAMP);    \
        const _Float16 h_ = (_Float16)xs_;                                                               \
        _Float16* e_ = reinterpret_cast<_Float16*>(d_);                                                  \
        e_[h_dst] = h_;                                                                                  \
        e_[h_dst + 2 * BNP * 8] = (_Float16)(xs_ - (float)h_);                                           \
    }
#define SM_READ_B(dst_, slot_, kx_)                                                                      \
    {                                                                                                    \
        const f32x4* bf_ = b_frag + (slot_) * SLICE + (kx_);                                             \
        _Pragma("unroll") for (int s = 0; s < NP; ++s)                                                   \
            _Pragma("unroll") for (int i = 0; i < NJ; ++i)                                               \
                dst_[i][s] = bf_[s * 2 * BNP + i * SEGP];                                                \
    }
    // PAIRS: n-tile i of a wave is segment wn / 32 + i, and wn / 32 is even - odd i = the LOWER segment of a pair, whose
    // taps ky = 0, 1 read the upper segment's (i - 1) slice ky + 1 (slot_up_)
#define SM_READ_B_KY(dst_, slot_, slot_up_, ky_, kx_)                                                    \
    if (PAIRS && (ky_) < 2) {                                                                            \
        const f32x4* bf_ = b_frag + (slot_) * SLICE + (kx_);                                             \
        const f32x4* bu_ = b_frag + (slot_up_) * SLICE + (kx_);                                          \
        _Pragma("unroll") for (int s = 0; s < NP; ++s)                                                   \
            _Pragma("unroll") for (int i = 0; i < NJ; ++i)                                               \
                dst_[i][s] = (i & 1) ? bu_[s * 2 * BNP + (i - 1) * SEGP] : bf_[s * 2 * BNP + i * SEGP];  \
    } else {                                                                                             \
        SM_READ_B(dst_, slot_, kx_)                                                                      \
    }

    if constexpr (RES) {
        constexpr int RP = SM_RES_RP;
        constexpr int RGRP = SM_RES_PHASE / 8;           // (chunk, k-group) groups of eight channels in a phase
        // Staging tasks: (group of 8 channels, window row r, block of four consecutive positions) - 8 x 6 x 9 = 432 per
        // phase, two per thread. A task loads its four positions of each channel with ONE 16-byte load (un-pooling input:
        // its two pooled elements with one 8-byte load + the codes of both) and builds the four positions' 8-channel units
        // in registers. (The single-position units of the first version issued 56 dword loads per thread: 3.0 us of a
        // block's 7 us staging time passed before the last of them was issued, 2.4 us now - profiles/r05/resident_kernel.txt; -3 % on
        // the launches.)
        // Window position (r, p) holds input position qs[0] - 1 + p + (r - 1) Wp.
        constexpr int RCB = 9, RT = RGRP * SM_RES_ROWS * RCB, RU = (RT + 255) / 256;
        f32x4* Rs = smem4;
        int r_src[RU], r_dst[RU], r_code[UNPOOL ? RU : 1], r_ypar[UNPOOL ? RU : 1];
        bool r_on[RU], r_ok[UNPOOL ? RU : 1][2];
        int r_p0[RU];                                    // window position of the task's element 0 (un-pooling: may be -1)
#pragma unroll
        for (int k = 0; k < RU; ++k) {
            const int t = tid + 256 * k;
            r_on[k] = t < RT;
            const int tt = r_on[k] ? t : t % RT;         // (idle tasks repeat a task's loads and store nothing)
            const int grp = tt / (SM_RES_ROWS * RCB), rem = tt - grp * (SM_RES_ROWS * RCB);
            const int r = rem / RCB, cb = rem - r * RCB;
            if constexpr (UNPOOL) {
                // quads of an un-pooling launch start on even columns (sm_cover_segments, flat quads) and rows 4 Y: the
                // task = the two pooled elements under image columns x .. x + 3, x = x0 - 2 + 4 cb (even), of image row
                // y = y0 - 1 + r; no gradient outside the pooled windows (and, row-locally, behind the row's end: only
                // outputs nobody stores read those)
                const int y0 = qs[0] / P.Wp - 1, x0 = qs[0] - (y0 + 1) * P.Wp - 1;   // pixel of the quad's first position
                const int y = y0 - 1 + r, x = x0 - 2 + 4 * cb;
                const bool yok = (unsigned)y < (unsigned)(2 * up_Ho);
                r_ok[k][0] = yok && (unsigned)x < (unsigned)(2 * up_Wo);
                r_ok[k][1] = yok && (unsigned)(x + 2) < (unsigned)(2 * up_Wo);
                // (rows / columns outside the image clamp into the pooled plane - its own padding row / columns: the 8-byte
                // load of elements xp, xp + 1 stays inside the plane of every channel, and of the code image, which has no
                // guard floats behind it)
                const int yp = min(max((y >> 1) + 1, 0), up_Ho + 1), xp = min(max((x >> 1) + 1, 0), up_Wp - 2);
                const int off_ = yp * up_Wp + xp;
                r_src[k] = (off_ + grp * 8 * up_plane) * 4;
                r_code[k] = (off_ + grp * up_plane) * 4;
                r_ypar[k] = (y & 1) << 1;
                r_p0[k] = 4 * cb - 1;
            } else {
                // (window rows below the plane's bottom padding row - the quads of a level's last rows - read that row
                // again: only outputs behind the last image row use them, and no load passes the plane's end by more than
                // the 36 floats of its own width, whatever the level's size)
                const int r_max = P.H + 1 - (qs[0] / P.Wp - 1);
                r_p0[k] = 4 * cb;
                r_src[k] = (grp * 8 * P.plane + qs[0] + 4 * cb + min(r, r_max) * P.Wp) * 4;  // bytes from the shifted base (row ky = 0)
            }
            r_dst[k] = ((grp >> 1) * 4 + (grp & 1)) * RP + r * SEGP + r_p0[k];   // + j (position in the task) + part * 2 * RP
        }
#pragma unroll
        for (int t = 0; t < AD; ++t) SM_LOAD_A(t, 0);
        // n-tile i of the wave = segment wn / 32 + i = window row wn / 32 + i + ky of tap row ky
        const f32x4* b_frag = Rs + lhi * RP + (wn / 32) * SEGP + l31;
        const int n_phases = a.Cin_pad / SM_RES_PHASE;
        for (int ph = 0; ph < n_phases; ++ph) {
            if (ph > 0) __syncthreads();                 // the previous phase's last fragment reads
            {
                if constexpr (UNPOOL) {
                    typedef float f32x2_ __attribute__((ext_vector_type(2)));
                    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
                    f32x2_ rb[RU][8];
                    u32x2_ rc[RU];
                    const int sc_ = ph * SM_RES_PHASE * up_plane * 4;
#pragma unroll
                    for (int k = 0; k < RU; ++k) {
#pragma unroll
                        for (int c = 0; c < 8; ++c)
                            rb[k][c] = __builtin_bit_cast(f32x2_, __builtin_amdgcn_raw_buffer_load_b64(gp_rsrc, r_src[k], sc_ + c * up_plane * 4, 0));
                        rc[k] = __builtin_bit_cast(u32x2_, __builtin_amdgcn_raw_buffer_load_b64(code_rsrc, r_code[k], ph * (SM_RES_PHASE / 8) * up_plane * 4, 0));
                    }
#pragma unroll
                    for (int k = 0; k < RU; ++k)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {            // position r_p0 + j: pooled element j >> 1, window parity (y, j & 1)
                            const int e = j >> 1;
                            const unsigned cw = e ? rc[k][1] : rc[k][0];
                            const int par = r_ypar[k] | (j & 1);
                            f16x8 vh, vl;
#pragma unroll
                            for (int c = 0; c < 8; ++c) {
                                float v = e ? rb[k][c][1] : rb[k][c][0];
                                v = (r_ok[k][e] && (int)((cw >> (4 * c)) & 15u) == par) ? v : 0.f;
                                const float xs_ = __builtin_amdgcn_fmed3f(v * in_scale, -SM_F16_CLAMP, SM_F16_CLAMP);
                                const _Float16 h_ = (_Float16)xs_;
                                vh[c] = h_;
                                vl[c] = (_Float16)(xs_ - (float)h_);
                            }
                            if (r_on[k] && (unsigned)(r_p0[k] + j) < (unsigned)SEGP) {
                                Rs[r_dst[k] + j] = __builtin_bit_cast(f32x4, vh);
                                Rs[r_dst[k] + j + 2 * RP] = __builtin_bit_cast(f32x4, vl);
                            }
                        }
                } else {
                    f32x4 rb[RU][8];
                    const int so_ = ph * SM_RES_PHASE * P.plane * 4;
#pragma unroll
                    for (int k = 0; k < RU; ++k)
#pragma unroll
                        for (int c = 0; c < 8; ++c)
                            rb[k][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, r_src[k], so_ + c * P.plane * 4, 0));
#pragma unroll
                    for (int k = 0; k < RU; ++k)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            f16x8 vh, vl;
#pragma unroll
                            for (int c = 0; c < 8; ++c) {
                                const float xs_ = __builtin_amdgcn_fmed3f(rb[k][c][j] * in_scale, -SM_F16_CLAMP, SM_F16_CLAMP);
                                const _Float16 h_ = (_Float16)xs_;
                                vh[c] = h_;
                                vl[c] = (_Float16)(xs_ - (float)h_);
                            }
                            if (r_on[k] && r_p0[k] + j < SEGP) {
                                Rs[r_dst[k] + j] = __builtin_bit_cast(f32x4, vh);
                                Rs[r_dst[k] + j + 2 * RP] = __builtin_bit_cast(f32x4, vl);
                            }
                        }
                }
            }
            __syncthreads();
            f32x4 fb[NJ][NP], fb_next[NJ][NP];
#pragma unroll
            for (int s = 0; s < NP; ++s)
#pragma unroll
                for (int i = 0; i < NJ; ++i) fb[i][s] = b_frag[s * 2 * RP + i * SEGP];
            for (int cc = 0; cc < SM_RES_CC; ++cc) {
                const int ch = ph * SM_RES_CC + cc;
                const int ch_next = ch + 1 < n_chunks ? ch + 1 : ch;   // (loads stay unconditional: see the ring loop)
                const f32x4* bc = b_frag + cc * 4 * RP;
                const f32x4* bn = b_frag + (cc < SM_RES_CC - 1 ? cc + 1 : cc) * 4 * RP;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    // the next stage's fragments are read under this stage's MFMAs
                    const f32x4* bf_ = tap < 8 ? bc + ((tap + 1) / 3) * SEGP + (tap + 1) % 3 : bn;
#pragma unroll
                    for (int s = 0; s < NP; ++s)
#pragma unroll
                        for (int i = 0; i < NJ; ++i) fb_next[i][s] = bf_[s * 2 * RP + i * SEGP];
                    f32x4 fa[NP];
#pragma unroll
                    for (int s = 0; s < NP; ++s) fa[s] = ra[tap % AD][0][s];
#define SM_RES_PRODUCT(pa_, pb_)                                                                         \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                       \
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[pa_]),           \
                                                           __builtin_bit_cast(f16x8, fb[j][pb_]), acc[0][j], 0, 0, 0);
                    SM_RES_PRODUCT(1, 0)
                    SM_RES_PRODUCT(0, 1)
                    SM_RES_PRODUCT(0, 0)
#undef SM_RES_PRODUCT
                    __builtin_amdgcn_sched_barrier(0);
                    if (tap + AD < 9) {
                        SM_LOAD_A(tap + AD, ch);
                    } else {
                        SM_LOAD_A(tap + AD - 9, ch_next);
                    }
#pragma unroll
                    for (int s = 0; s < NP; ++s)
#pragma unroll
                        for (int i = 0; i < NJ; ++i) fb[i][s] = fb_next[i][s];
                }
            }
        }
    } else {
    // prologue: the first AD weight stages into the register ring, chunk ch_begin's three slices into slots 0..2
#pragma unroll
    for (int t = 0; t < AD; ++t) SM_LOAD_A(t, ch_begin);
    {   // all three slices' loads in flight together (one memory round trip instead of three)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) SM_LOAD_B(ky, ky, ch_begin);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) SM_STORE_B(ky, ky, ky);
    }
    {
        const int ch1 = ch_begin + 1 < ch_end ? ch_begin + 1 : ch_begin;
        SM_LOAD_B(0, 0, ch1);      // stored at the end of tap 1 of the first chunk
    }
    __syncthreads();
    int base = 0;   // ring slot of the current chunk's ky = 0 slice
    // slot of slice ky of the current / of the next chunk
#define SM_CUR_SLOT(ky_) ((base + (ky_)) & 3)
#define SM_NEXT_SLOT(ky_) ((base + 3 + (ky_)) & 3)
    const f32x4* b_frag = Bs + lhi * BNP + (wn / 32) * SEGP + l31;   // n-tile i of the wave = segment wn / 32 + i
    f32x4 fb[NJ][NP], fb_next[NJ][NP];   // operand fragments as raw 16-byte units (8 fp16)
#if SM_SPLIT_PREFETCH_B
    SM_READ_B_KY(fb, 0, 1, 0, 0)
#endif
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        // Every load below is issued UNCONDITIONALLY (the last chunk re-reads its own data instead of the next
        // chunk's): a load under `if (more)` makes the compiler's waitcnt pass assume the no-load path at the join,
        // and every later wait for an OLDER load then drains the whole queue (vmcnt(0) instead of vmcnt(N)).
        const int ch_next = ch + 1 < ch_end ? ch + 1 : ch;
        const int ch_next2 = ch + 2 < ch_end ? ch + 2 : ch_next;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            // the next stage's activation fragments are read under this stage's MFMAs (its slice is complete: slices
            // are written a full barrier before their first use)
#if SM_SPLIT_PREFETCH_B
            if (tap < 8) {
                SM_READ_B_KY(fb_next, SM_CUR_SLOT((tap + 1) / 3), SM_CUR_SLOT((tap + 1) / 3 + 1), (tap + 1) / 3, (tap + 1) % 3)
            } else {
                SM_READ_B_KY(fb_next, SM_NEXT_SLOT(0), SM_NEXT_SLOT(1), 0, 0)
            }
#else
            SM_READ_B_KY(fb, SM_CUR_SLOT(ky), SM_CUR_SLOT(ky + 1), ky, kx)
#endif
            f32x4 fa[MI][NP];
#pragma unroll
            for (int s = 0; s < NP; ++s)
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i][s] = ra[tap % AD][i][s];
            // the partial products per output tile, smallest first; consecutive MFMAs target different accumulators
#define SM_PRODUCT(pa_, pb_)                                                                             \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                       \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                   \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i][pa_]),    \
                                                               __builtin_bit_cast(f16x8, fb[j][pb_]), acc[i][j], 0, 0, 0);
            SM_PRODUCT(1, 0)
            SM_PRODUCT(0, 1)
            SM_PRODUCT(0, 0)
#undef SM_PRODUCT
            // the ring slot just consumed is refilled with the weights of stage + AD (pinned below the MFMAs: hoisting
            // the loads would need a fourth set of fragment registers)
            __builtin_amdgcn_sched_barrier(0);
            if (tap + AD < 9) {
                SM_LOAD_A(tap + AD, ch);
            } else {
                SM_LOAD_A(tap + AD - 9, ch_next);
            }
            // next chunk's slice ky -> slot (base + 3 + ky) & 3: for ky = 0 the spare slot (the previous chunk's
            // ky = 2), for ky = 1, 2 the slot of this chunk's slice ky - 1, whose last readers passed the barrier of
            // tap 3 ky - 1. The slice is loaded at the end of tap 3 ky - 1 (for ky = 0: tap 8 of the previous chunk),
            // converted and written at the end of tap 3 ky + 1 - a stage WITHOUT a barrier, so that the conversion does
            // not sit on a barrier's critical path - and published by the barrier at the end of tap 3 ky + 2.
            if (kx == 1) { SM_STORE_B(0, SM_NEXT_SLOT(ky), ky); }
            if (kx == 2) {
                if (ky < 2) {
                    SM_LOAD_B(0, ky + 1, ch_next);
                } else {
                    SM_LOAD_B(0, 0, ch_next2);
                }
                __syncthreads();
            }
#if SM_SPLIT_PREFETCH_B
#pragma unroll
            for (int s = 0; s < NP; ++s)
#pragma unroll
                for (int i = 0; i < NJ; ++i) fb[i][s] = fb_next[i][s];
#endif
        }
        base = (base + 3) & 3;
    }
    }   // (ring kernel)
#undef SM_CUR_SLOT
#undef SM_NEXT_SLOT
#undef SM_UP_MAP
#undef SM_LOAD_A
#undef SM_LOAD_B
#undef SM_STORE_B
#undef SM_READ_B
#undef SM_READ_B_KY

    // ---- epilogue (same 32x32 C/D layout as conv3x3_mfma_kernel: column = lane & 31,
    //      row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)); column tile j of the wave is in acc[.][j]
    if (split >= 0) {
        float* wt = a.ws + ((size_t)(tile - a.n_whole) * a.splits + split) * (BM * BN);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    wt[(wm + mi * 32 + 4 * lhi + (r & 3) + 8 * (r >> 2)) * BN + wn + j * 32 + l31] =
                        acc[mi][j][r] * out_scale;   // power of two: exact
        return;
    }
    const float vmax = conv_split_epilogue<BM, BN, WGM, WGN, FLAGS, RES, (RES && SM_RES_PHASE == 32) ? 32 : 64>(a, P, qs, live, acc, m0, out_scale, a.bias, smem4);
    record_amax(a.amax_out, vmax, amax_seen);
}

}  // namespace sm

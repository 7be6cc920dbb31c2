// K3s/K4s: the 3x3 convolutions on the bf16 matrix cores at fp32 accuracy ("bf16x3 split, 6 products").
//
// Every fp32 operand x is written as h + m + l, three bf16 numbers (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m):
// 3 x 8 = 24 significand bits, i.e. the split is exact up to the last rounding of l), and a product x*y is evaluated
// as the six partial products of weight >= 2^-16:  hh' + (hm' + mh') + (mm' + hl' + lh').  The dropped terms
// (ml', lm', ll') are below 2^-23 |xy|, each partial product of two bf16 numbers is exact in fp32, and all sums are
// taken in the MFMA's fp32 accumulators. The result is as close to the exact dot product as the fp32 MFMA chain of
// conv.hip (tests/test_kernels_gpu.py measures both against an fp64 convolution) - but
// v_mfma_f32_32x32x16_bf16 retires 16x the MACs per cycle of v_mfma_f32_32x32x2_f32, so six of them per fp32 MAC
// still leave a 2.67x higher ceiling (2.5 PFLOP/s / 6 = 417 TFLOP/s fp32-equivalent, against 157 TFLOP/s).
//
// Same GEMM view, tile scheduling, grouped launch, active-tile list, tail split-K and epilogues as conv.hip.
// Differences: weights arrive pre-split from the host ([9][Cin/16][3 parts][2 k-groups][Cout][8 ci] bf16, the exact
// image of an LDS weight stage); activations stay fp32 in HBM and are split when they are staged into LDS, as
// [ky slice][3 parts][2 k-groups][BN+2 positions][8 ci] bf16 - a tap shift is again a pure offset (16 bytes per
// position). One K-stage = one tap of a 16-channel chunk = ONE bf16 MFMA K-step (24 MFMAs per wave):
//   * weights: the global stage image is the MFMA A-fragment layout, so each wave loads its fragments straight into
//     a register ring three stages ahead - no LDS copy, no per-stage barrier;
//   * activations: the three ky slices of a chunk live in a ring of FOUR LDS slots; the next chunk's slice k is
//     loaded at tap 3k and written at the end of tap 3k+2 into the slot that the current chunk stopped reading
//     three taps earlier (one barrier per three stages); the next stage's fragments are read under the current
//     stage's MFMAs.
// LDS: 4 x 12.2 KB = 48.8 KB; two blocks per CU (register-limited): the two waves that share a SIMD's matrix pipe
// belong to different blocks and cover each other's store / barrier phases.
#pragma once
#include "conv_common.h"

// tuning knobs (A/B builds: build.sh -DSM_SPLIT_AD=1 ..., compared with tools/ab_libs.sh). Measured on c3 / the layer
// micro-benchmark, relative to the defaults (187-190 TFLOP/s): AD=1 -22 %; no fragment prefetch -5 %; 3 waves per
// SIMD (AD=1, no prefetch, 168 VGPRs) -14 %; PIN_READS=1 (fragment reads pinned a full stage ahead: the densest
// MFMA stream, 245 VGPRs) -8 %; any s_setprio (MFMA cluster or the load/convert tail) -9 %: each of them fences the
// compiler's own interleaving of the tail instructions with the MFMAs. 3 waves per SIMD with AD=3 and no fragment
// prefetch needs 168 VGPRs: the 128-row variant spills ~25 dwords and loses 14 %. What the loop is sensitive to is the
// NUMBER of vector-memory instructions (ablation: no weight loads +15 %, no activation loads +8 %, no conversion
// arithmetic / no LDS fragment reads / no epilogue stores +-1 %): hence the 32 x 128 wave tiles (SM_SPLIT_WGM = 4 in
// conv.hip) and the buffer loads.
#ifndef SM_SPLIT_AD
#define SM_SPLIT_AD 3          // weight prefetch distance in stages (must divide 9)
#endif
#ifndef SM_SPLIT_PREFETCH_B
#define SM_SPLIT_PREFETCH_B 1  // read the next stage's activation fragments under this stage's MFMAs
#endif
#ifndef SM_SPLIT_PIN_READS
#define SM_SPLIT_PIN_READS 0
#endif
#ifndef SM_SPLIT_TAIL_PRIO
#define SM_SPLIT_TAIL_PRIO 0
#endif
#ifndef SM_SPLIT_MFMA_PRIO
#define SM_SPLIT_MFMA_PRIO 0
#endif
#ifndef SM_SPLIT_WAVES64
#define SM_SPLIT_WAVES64 3
#endif
#ifndef SM_SPLIT_WAVES
#define SM_SPLIT_WAVES 2       // resident waves per SIMD the register budget is set for
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
__device__ __forceinline__ unsigned Gs_gate_byte(const f32x4* smem, int index, int unit_offset) {
    return reinterpret_cast<const unsigned char*>(smem + unit_offset)[index];
}
__device__ __forceinline__ void conv_gram_mfma(f32x16& acc, const f32x4 (&fa)[2], const f32x4 (&fb)[2]) {
#define SM_H(x_) __builtin_bit_cast(cg_f16x8, x_)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(SM_H(fa[1]), SM_H(fb[0]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(SM_H(fa[0]), SM_H(fb[1]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(SM_H(fa[0]), SM_H(fb[0]), acc, 0, 0, 0);
#undef SM_H
}

// NP = number of parts an fp32 operand is split into:
//   NP = 3  bf16 x 3 (above): 6 partial products per fp32 product, no scaling needed (bf16 has the fp32 exponent range);
//   NP = 2  fp16 x 2: x s = h + l with h = fp16(x s), l = fp16(x s - h): 2 x 11 = 22 significand bits, products
//           hh' + hl' + lh' (each exact in fp32; the dropped ll' is < 2^-22 of the product) - THREE MFMAs
//           (v_mfma_f32_32x32x16_f16, same rate as bf16) per fp32 product instead of six. fp16 has 5 exponent bits, so
//           every operand tensor is scaled by a power of two s that puts its largest magnitude into [2^14, 2^15): the
//           producer of an activation / gradient tensor records max |x| (ConvArgs::amax_out, one atomic max per wave),
//           the consumer derives s from it (amax_in) and the epilogue multiplies the accumulators by 1 / (s s_w) (exact).
//           Elements more than 2^18 below the tensor's maximum lose low bits of l (absolute error <= 2^-40 max|x|).
//           Measured against an fp64 convolution: tests/test_kernels_gpu.py, tools/bench_conv_split.py.
#ifndef SM_ABL_NOCVT
#define SM_ABL_NOCVT 0          // 1 (ablation build, timing only): activations staged without the fp32 -> fp16-pair conversion
#endif
#ifndef SM_ABL_NOSEL
#define SM_ABL_NOSEL 0          // 1 (ablation build, timing only): the un-pooling input without its argmax selection
#endif
#ifndef SM_SPLIT2_PAIR_ROWS
#define SM_SPLIT2_PAIR_ROWS 1  // forward convs with the pooling epilogue: the lower segment of a pair re-uses the upper one's rows
#endif
#ifndef SM_SPLIT2_ILV
#define SM_SPLIT2_ILV 0        // experiment: 1 = no scheduling fence below a stage's MFMAs; 2 = tail instructions dealt between the MFMAs
#endif
#ifndef SM_SPLIT2_RING6
#define SM_SPLIT2_RING6 0      // fp16x2: six LDS slots (two whole chunks), ONE barrier per chunk instead of three
#endif
// (128-row tiles only: six slots of the 64 x 256 tile would be 99 KB per block - one block per CU)
constexpr int conv_split_slots(int NP, int BM = 128) { return (NP == 2 && BM == 128 && SM_SPLIT2_RING6) ? 6 : 4; }
constexpr size_t conv_split_lds_bytes(int BM, int BN, int NP = 3) {
    return (size_t)(conv_split_slots(NP, BM) * 2 * NP * (BN / 32 * 34)) * 16;   // 34 staged positions per 32-position segment
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

// ---- pair images (round 5). A feature map may be stored as PACKED fp16 PAIRS instead of fp32: word = h | l << 16 with
// h = fp16(x s), l = fp16(x s - h) - exactly the two operand parts the fp16x2 kernels build from an fp32 value while
// they stage it, so a consumer only un-packs (two byte permutes per channel pair instead of mul / clamp / cvt / cvt /
// sub / cvt per element: that conversion was 13.5 % of a four-level step, profiles/r05/staging_ablation_c2_c3.txt).
// Same 4 bytes per element, same [C][plane] addressing, zero word = zero. The power-of-two scale s must be known BEFORE
// the producer runs: it is derived from the bound the PREVIOUS step recorded for the tensor, with head-room
// (sm_pair_scales); sm_pair_check compares the bound this step recorded with it and invalidates the step on overflow
// (values beyond the fp16 range saturate at +-65000 s^-1; the engine skips the update and repeats the step).
__device__ __forceinline__ unsigned pair_encode(float v, float s) {
    const float xs = __builtin_amdgcn_fmed3f(v * s, -SM_F16_CLAMP, SM_F16_CLAMP);
    const _Float16 h = (_Float16)xs;
    const _Float16 l = (_Float16)(xs - (float)h);
    return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}
// eight pair words (channels c .. c + 7 of one position) -> the h unit and the l unit (8 fp16 each)
__device__ __forceinline__ void pair_units(const float (&w)[8], f32x4& vh, f32x4& vl) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned a = __builtin_bit_cast(unsigned, w[2 * k + 1]), b = __builtin_bit_cast(unsigned, w[2 * k]);
        vh[k] = __builtin_bit_cast(float, __builtin_amdgcn_perm(a, b, 0x05040100u));
        vl[k] = __builtin_bit_cast(float, __builtin_amdgcn_perm(a, b, 0x07060302u));
    }
}

// STAMP: debug build that records s_memtime stamps of every wave at the stage boundaries into the tail of ws.
// the 64-row variant needs 129 VGPRs: three of its waves fit a SIMD (SM_SPLIT_WAVES64)
#ifndef SM_SPLIT_BSETS
#define SM_SPLIT_BSETS 1
#endif
#ifndef SM_SPLIT2_AD
#define SM_SPLIT2_AD 3
#endif
#ifndef SM_SPLIT2_WAVES
#define SM_SPLIT2_WAVES 2      // resident waves per SIMD of the 128-row fp16x2 variant
#endif
// resident waves per SIMD the register budget of a variant is set for
constexpr int conv_split_waves(int BM, int BN, int NP) {
    return BM == 256 ? 1 : (BM == 64 && BN == 128) ? SM_SPLIT_WAVES64 : NP == 2 ? SM_SPLIT2_WAVES : SM_SPLIT_WAVES;
}
// KG (round 5) = wave GROUPS per block. KG = 2: a block is 512 threads = two groups of four waves; both compute the SAME
// output tile, group g over half g of the block's K-chunks, each with a slice ring and weight registers of its own (the
// groups only meet at the stages' barriers), and exchange accumulator halves through LDS before the epilogue, which each
// group runs for half of the wave's column tiles. For launches of <= one block per CU (every layer of a one-level view:
// 12 - 172 tiles): a lone 4-wave block is latency-bound - one wave per SIMD cannot cover its own load / convert / store
// stage tail with MFMAs (a stage takes ~800 cycles against ~475 per block when two blocks share a CU) - and getting the
// second wave per SIMD from MORE global K-splits doubles the partial slabs instead.
// PIN (round 5): the input planes hold packed fp16 pairs (above): staging un-packs instead of converting.
// RES (round 5): RESIDENT INPUT, for the 64-output-channel launches (conv1_2 forward / data gradient, conv2_1's data
// gradient: K = 576 / 1152). The 64 x 256 tile stages every chunk's three ky slices for each of its eight free segments -
// 3 x 34 / 32 = 3.2 staged positions per output position and 16 channels - and spends twice the staging per MFMA of the
// 128-row tile. Here a tile is a QUAD: four vertically adjacent 32-position segments (list entries q, q + Wp, q + 2 Wp,
// q + 3 Wp; sm_cover_segments quad modes), and the block stages the (4 + 2) rows x 34 positions it needs of 64 input
// channels ONCE, already scaled and split: 6 x 34 x 64 channels x 2 parts x 2 B = 51 KB (three blocks per CU), 1.59 staged
// positions per output position. The 36 stages of a 64-channel phase then are MFMAs, weight-fragment loads and LDS
// fragment reads only - a tap shift (ky, kx) is the offset ky * 34 + kx - with no barrier and no conversion inside the
// loop; Cin = 128 takes two phases. Same chunk / tap / product order as the ring kernel: the sums have its bits.
constexpr int SM_RES_ROWS = 6, SM_RES_RP = SM_RES_ROWS * 34;                    // staged rows / positions of a quad
constexpr size_t conv_resident_lds_bytes() { return (size_t)(4 * 2 * 2 * SM_RES_RP) * 16; }   // [chunk][part][k-group][RP] units
template <int BM, int BN, int WGM, int WGN, int FLAGS, bool STAMP = false, int NP = 3, bool UNPOOL = false, int KG = 1,
          bool PIN = false, bool RES = false>
__global__ __launch_bounds__(256 * KG)
__attribute__((amdgpu_waves_per_eu(conv_split_waves(BM, BN, NP), conv_split_waves(BM, BN, NP))))
void conv3x3_split_kernel(ConvArgs a) {
    static_assert(NP == 2 || NP == 3, "bf16 x 3 or fp16 x 2");
    static_assert(!RES || (NP == 2 && BM == 64 && BN == 128 && WGM == 2 && WGN == 2 && KG == 1 && !PIN && !STAMP),
                  "resident input: the fp16x2 kernel on 64 x 128 tiles (a quad of segments), waves 2 x 2");
    static_assert(!PIN || NP == 2, "pair images are the fp16x2 kernel's operand format");
    static_assert(KG == 1 || (KG == 2 && NP == 2 && conv_split_waves(BM, BN, NP) == 2 && !(FLAGS & SM_EPI_GRAM)),
                  "two wave groups: the fp16x2 kernel at two waves per SIMD, without the Gram epilogue");
    static_assert(KG == 1 || !SM_SPLIT2_RING6, "two wave groups: the four-slot ring (a barrier at the end of every chunk)");
    static_assert(!UNPOOL || NP == 2, "the unpool input exists for the fp16x2 kernel");
    // activation register sets: 1 = a slice is loaded two stages before it is converted; 3 = one set per ky slice, loaded a
    // whole chunk (nine stages) ahead. (the un-pooling input and the pair rows keep one set)
    constexpr int BSETS = (UNPOOL || ((FLAGS & SM_EPI_POOL) != 0 && NP == 2 && SM_SPLIT2_PAIR_ROWS)) ? 1 : SM_SPLIT_BSETS;
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
    const int grp = KG == 1 ? 0 : (int)(threadIdx.x >> 8);        // wave group (wave-uniform)
    f32x4* Bs = smem4 + grp * (conv_split_slots(NP, BM) * SLICE);   // [4 slots][SLICE] of this group

    const int tid = KG == 1 ? (int)threadIdx.x : (int)(threadIdx.x & 255);   // thread / wave index INSIDE the group
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int lhi = lane >> 5;
    const int wm = (wave / WGN) * (32 * MI);
    const int wn = (wave % WGN) * (32 * NJ);
#define SM_TS(slot_)                                                                                     \
    if (STAMP && lane == 0) {                                                                            \
        reinterpret_cast<long long*>(a.ws + 15 * 1024 * 1024)[((size_t)(blockIdx.x * KG + grp) * 4 + wave) * 64 + (slot_)] = \
            __builtin_readcyclecounter();                                                                \
    }

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
    int ch_begin = split < 0 ? 0 : split * a.chunks_per_split;
    int ch_end = split < 0 ? n_chunks : min(n_chunks, ch_begin + a.chunks_per_split);
    if constexpr (KG == 2) {   // (the host keeps every unit's chunk count even: both groups run the same number of stages)
        const int half = (ch_end - ch_begin) >> 1;
        ch_begin += grp * half;
        ch_end = ch_begin + half;
    }
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
    // weights: the global image of a stage, [part][kgroup][Cout] units of 8 bf16, IS the MFMA A-fragment layout
    // (row = lane & 31, k-group = lane >> 5), so every wave loads its own 3 MI fragments (MI row tiles x 3 parts) of a
    // stage straight into registers, three stages ahead: no LDS copy of the weights and no per-stage barrier.
    // (buffer loads: scalar base in an SGPR resource + one 32-bit lane offset + a scalar stage offset - no 64-bit
    // address arithmetic per load)
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.wt), 0, 9 * (a.Cin_pad / KC) * 2 * NP * a.Cout * 16, 0x00020000);
    const int a_voff = (lhi * a.Cout + m0 + wm + l31) * 16;   // bytes
    const int a_part = 2 * a.Cout * 16;            // bytes between the parts of a stage
    const int a_stage_bytes = 2 * NP * a.Cout * 16;   // bytes per (tap, chunk) stage
    // NP = 2: operand scale from the producer's recorded max |x| (one vector load of the bound's slots per wave)
    float in_scale = 1.f, out_scale = 1.f;
    if constexpr (PIN) {
        out_scale = a.pair_in[1] * a.w_scale_inv;          // {scale, 1 / scale} of the stored pairs
    } else if (NP == 2) {
        float inv;
        in_scale = pow2_scale_for(a.amax_in ? amax_read(a.amax_in) : 1.f, inv);
        out_scale = inv * a.w_scale_inv;
    }
    // outputs as pairs (fp16x2 kernel; wave-uniform), the ReLU gate planes as pairs (x > 0 <=> word != 0)
    const bool pout = NP == 2 && a.pair_out != nullptr;
    const float po_scale = pout ? a.pair_out[0] : 1.f;
    const bool gate_pair = NP == 2 && a.pair_gate != nullptr;
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
    const int h_dst = (h_kg * BNP + h_seg * SEGP + 32 + h_which) * 8 + h_c;   // in bf16 elements (+ part * 2 * BNP * 8)
    // PAIRS (round 4): a forward conv with the pooling epilogue takes its segments in vertical pairs - entries 2k, 2k + 1 =
    // the same 32 columns of image rows 2Y and 2Y + 1 - so slice ky of the LOWER segment holds the input row that slice
    // ky + 1 of the UPPER one holds. The lower segment's slices 0 and 1 are therefore not staged at all: its fragments for
    // taps ky = 0, 1 are read from the upper segment's slices 1, 2 (complete whenever the lower one's would be: the three
    // slices of a chunk are published before its first tap, and a ring slot is only re-used after the taps that read it,
    // see SM_NEXT_SLOT). Four staged rows per pair and chunk instead of six - the 64-row tile spends twice the staging
    // per MFMA of the 128-row tile, and staging is what bounds it. Bit-identical: the same input
    // values go through the same conversion. Slices 0 / 1 are staged with a mapping of their own over the UPPER segments:
    // one unit per thread for BN = 256 (instead of two); for BN = 128 half the threads convert and store.
    constexpr bool PAIRS = NP == 2 && !UNPOOL && (FLAGS & SM_EPI_POOL) != 0 && BSETS == 1 && !SM_SPLIT2_RING6 &&
                           SM_SPLIT2_PAIR_ROWS;
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
#ifndef SM_RES_ABL
#define SM_RES_ABL 0           // ablation builds (timing only): 1 = no MFMA loop, 2 = no staging, 3 = no weight loads in the loop
#endif
    constexpr int AD = RES ? SM_RES_AD : NP == 2 ? SM_SPLIT2_AD : SM_SPLIT_AD;
    static_assert(9 % AD == 0, "ring slot of a stage is the same in every chunk");
    f32x4 ra[AD][MI][NP];
    // in-flight activation loads: SM_SPLIT_BSETS = 1: one register set, a slice is loaded two stages before it is
    // converted and stored; 3: one set per ky slice, re-loaded right after its store - a slice's loads then have a
    // whole chunk (nine stages) to arrive
    // (register sets 1 and 2 of the one-set variant only carry the prologue's three slices)
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
    if constexpr (NP == 3) {                                                                             \
        f32x4* d_ = Bs + (slot_) * SLICE;                                                                \
        _Pragma("unroll") for (int u = 0; u < NU; ++u) {                                                 \
            bf16x8 vh, vm, vl;                                                                           \
            _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                              \
                __bf16 h, m, l;                                                                          \
                split3(rbs[set_][u][c], h, m, l);                                                        \
                vh[c] = h; vm[c] = m; vl[c] = l;                                                         \
            }                                                                                            \
            d_[b_dst[u]] = __builtin_bit_cast(f32x4, vh);                                                \
            d_[b_dst[u] + 2 * BNP] = __builtin_bit_cast(f32x4, vm);                                      \
            d_[b_dst[u] + 4 * BNP] = __builtin_bit_cast(f32x4, vl);                                      \
        }                                                                                                \
        __bf16 h, m, l;                                                                                  \
        split3(rhs[set_], h, m, l);                                                                      \
        __bf16* e_ = reinterpret_cast<__bf16*>(d_);                                                      \
        e_[h_dst] = h;                                                                                   \
        e_[h_dst + 2 * BNP * 8] = m;                                                                     \
        e_[h_dst + 4 * BNP * 8] = l;                                                                     \
    } else {                                                                                             \
        f32x4* d_ = Bs + (slot_) * SLICE;                                                                \
        if constexpr (UNPOOL && !SM_ABL_NOSEL) {   /* gradient only at the window element that held the maximum */ \
            _Pragma("unroll") for (int u = 0; u < NU; ++u)                                               \
                _Pragma("unroll") for (int c = 0; c < 8; ++c)                                            \
                    rbs[set_][u][c] = ((int)((rcs[set_][u] >> (4 * c)) & 15u) == rps[set_][u]) ? rbs[set_][u][c] : 0.f; \
            rhs[set_] = ((int)((rhc[set_] >> (4 * h_c)) & 15u) == rhp[set_]) ? rhs[set_] : 0.f;          \
        }                                                                                                \
        if constexpr (PIN) {   /* stored pairs: un-pack (a stale word of an unlisted tile is still a finite pair) */ \
            if (PAIRS && (ky_) < 2) {                                                                    \
                f32x4 vh, vl;                                                                            \
                pair_units(rbs[set_][0], vh, vl);                                                        \
                if (a_active) { d_[a_dst] = vh; d_[a_dst + 2 * BNP] = vl; }                              \
            } else {                                                                                     \
                _Pragma("unroll") for (int u = 0; u < NU; ++u) {                                         \
                    f32x4 vh, vl;                                                                        \
                    pair_units(rbs[set_][u], vh, vl);                                                    \
                    d_[b_dst[u]] = vh;                                                                   \
                    d_[b_dst[u] + 2 * BNP] = vl;                                                         \
                }                                                                                        \
            }                                                                                            \
        } else                                                                                           \
        if constexpr (SM_ABL_NOCVT) {   /* (ablation, timing only: raw bits instead of the scaled fp16 pairs) */ \
            _Pragma("unroll") for (int u = 0; u < ((PAIRS && (ky_) < 2) ? 1 : NU); ++u) {               \
                f32x4 vh = {rbs[set_][u][0], rbs[set_][u][1], rbs[set_][u][2], rbs[set_][u][3]};         \
                f32x4 vl = {rbs[set_][u][4], rbs[set_][u][5], rbs[set_][u][6], rbs[set_][u][7]};         \
                if (PAIRS && (ky_) < 2) { if (a_active) { d_[a_dst] = vh; d_[a_dst + 2 * BNP] = vl; } }  \
                else { d_[b_dst[u]] = vh; d_[b_dst[u] + 2 * BNP] = vl; }                                 \
            }                                                                                            \
        } else                                                                                           \
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
        if constexpr (PIN) {                                                                             \
            const unsigned w_ = __builtin_bit_cast(unsigned, rhs[set_]);                                 \
            unsigned short* e_ = reinterpret_cast<unsigned short*>(d_);                                  \
            e_[h_dst] = (unsigned short)(w_ & 0xffffu);                                                  \
            e_[h_dst + 2 * BNP * 8] = (unsigned short)(w_ >> 16);                                        \
        } else {                                                                                         \
        const float xs_ = __builtin_amdgcn_fmed3f(rhs[set_] * in_scale, -SM_F16_CLAMP, SM_F16_CLAMP);    \
        const _Float16 h_ = (_Float16)xs_;                                                               \
        _Float16* e_ = reinterpret_cast<_Float16*>(d_);                                                  \
        e_[h_dst] = h_;                                                                                  \
        e_[h_dst + 2 * BNP * 8] = (_Float16)(xs_ - (float)h_);                                           \
        }                                                                                                \
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

    SM_TS(0)
#ifdef SM_RES_TRACE   // (debug build: per block {start, staged, loop done, end, HW_ID, XCC_ID} into the tail of ws; tools/res_trace.py)
#define SM_RT(slot_)                                                                                     \
    if (RES && tid == 0) reinterpret_cast<long long*>(a.ws + 15 * 1024 * 1024)[(size_t)blockIdx.x * 16 + (slot_)] = \
        (long long)__builtin_amdgcn_s_memrealtime();
    if (RES && tid == 0) {
        reinterpret_cast<long long*>(a.ws + 15 * 1024 * 1024)[(size_t)blockIdx.x * 16 + 4] = __builtin_amdgcn_s_getreg(63492);
        reinterpret_cast<long long*>(a.ws + 15 * 1024 * 1024)[(size_t)blockIdx.x * 16 + 5] = __builtin_amdgcn_s_getreg(63508);
    }
#else
#define SM_RT(slot_)
#endif
    SM_RT(0)
    if constexpr (RES) {
        constexpr int RP = SM_RES_RP;
        constexpr int RGRP = 8;                          // (chunk, k-group) groups of eight channels in a 64-channel phase
        // Staging tasks: (group of 8 channels, window row r, block of four consecutive positions) - 8 x 6 x 9 = 432 per
        // phase, two per thread. A task loads its four positions of each channel with ONE 16-byte load (un-pooling input:
        // its two pooled elements with one 8-byte load + the codes of both) and builds the four positions' 8-channel units
        // in registers. (The single-position units of the first version issued 56 dword loads per thread: 3.0 us of a
        // block's 7 us staging time passed before the last of them was issued, 2.4 us now - tools/res_trace.py; -3 % on
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
            const int tt = r_on[k] ? t : tid;            // (idle tasks load a valid address and store nothing)
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
        const int n_phases = a.Cin_pad / 64;
        for (int ph = 0; ph < n_phases; ++ph) {
            if (ph > 0) __syncthreads();                 // the previous phase's last fragment reads
            if (SM_RES_ABL != 2) {
                if constexpr (UNPOOL) {
                    typedef float f32x2_ __attribute__((ext_vector_type(2)));
                    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
                    f32x2_ rb[RU][8];
                    u32x2_ rc[RU];
                    const int sc_ = ph * 64 * up_plane * 4;
#pragma unroll
                    for (int k = 0; k < RU; ++k) {
#pragma unroll
                        for (int c = 0; c < 8; ++c)
                            rb[k][c] = __builtin_bit_cast(f32x2_, __builtin_amdgcn_raw_buffer_load_b64(gp_rsrc, r_src[k], sc_ + c * up_plane * 4, 0));
                        rc[k] = __builtin_bit_cast(u32x2_, __builtin_amdgcn_raw_buffer_load_b64(code_rsrc, r_code[k], ph * 8 * up_plane * 4, 0));
                    }
                    SM_RT(6)
#ifdef SM_RES_TRACE
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    SM_RT(7)
#endif
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
                    const int so_ = ph * 64 * P.plane * 4;
#pragma unroll
                    for (int k = 0; k < RU; ++k)
#pragma unroll
                        for (int c = 0; c < 8; ++c)
                            rb[k][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, r_src[k], so_ + c * P.plane * 4, 0));
                    SM_RT(6)
#ifdef SM_RES_TRACE
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    SM_RT(7)
#endif
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
#ifdef SM_RES_TRACE   // every wave's "converted and stored" time (slots 10 .. 13) and "loads arrived" (wave 3: slot 14)
            if (lane == 0) reinterpret_cast<long long*>(a.ws + 15 * 1024 * 1024)[(size_t)blockIdx.x * 16 + 10 + wave] =
                (long long)__builtin_amdgcn_s_memrealtime();
#endif
            __syncthreads();
            SM_RT(1)
            f32x4 fb[NJ][NP], fb_next[NJ][NP];
#pragma unroll
            for (int s = 0; s < NP; ++s)
#pragma unroll
                for (int i = 0; i < NJ; ++i) fb[i][s] = b_frag[s * 2 * RP + i * SEGP];
            for (int cc = 0; cc < (SM_RES_ABL == 1 ? 0 : 4); ++cc) {
                const int ch = ph * 4 + cc;
                const int ch_next = ch + 1 < n_chunks ? ch + 1 : ch;   // (loads stay unconditional: see the ring loop)
                const f32x4* bc = b_frag + cc * 4 * RP;
                const f32x4* bn = b_frag + (cc < 3 ? cc + 1 : cc) * 4 * RP;
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
                    if (SM_RES_ABL != 3) {
                    if (tap + AD < 9) {
                        SM_LOAD_A(tap + AD, ch);
                    } else {
                        SM_LOAD_A(tap + AD - 9, ch_next);
                    }
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
        if constexpr (BSETS == 1) {
            SM_LOAD_B(0, 0, ch1);      // stored at the end of tap 1 of the first chunk
        } else {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) SM_LOAD_B(ky, ky, ch1);   // stored at the end of taps 1, 4, 7 of the first chunk
        }
    }
    __syncthreads();
    SM_TS(1)
    int base = 0;   // ring slot of the current chunk's ky = 0 slice
    constexpr bool RING6 = conv_split_slots(NP, BM) == 6;
    // slot of slice ky of the current / of the next chunk
#define SM_CUR_SLOT(ky_) (RING6 ? base + (ky_) : (base + (ky_)) & 3)
#define SM_NEXT_SLOT(ky_) (RING6 ? (3 - base) + (ky_) : (base + 3 + (ky_)) & 3)
    const f32x4* b_frag = Bs + lhi * BNP + (wn / 32) * SEGP + l31;   // n-tile i of the wave = segment wn / 32 + i
    f32x4 fb[NJ][NP], fb_next[NJ][NP];   // operand fragments as raw 16-byte units (8 bf16 / fp16)
#if SM_SPLIT_PREFETCH_B
    SM_READ_B_KY(fb, 0, 1, 0, 0)
#endif
    // SM_SPLIT2_ILV = 2 (experiment): a stage's instruction stream is dictated - the next stage's fragment reads first, then
    // its MFMAs with the tail's instructions dealt between them (a lone wave per SIMD cannot hide a tail that FOLLOWS
    // its MFMAs): per MFMA `valu_` VALU instructions (the convert-and-store stage) or one vector-memory load
#if SM_SPLIT2_ILV == 2
#define SM_ILV_PIPE(valu_)                                                                               \
    if constexpr (NP == 2) {                                                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, NJ * NP, 0);                                         \
        _Pragma("unroll") for (int g_ = 0; g_ < MI * NJ * 3; ++g_) {                                     \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                           \
            if ((valu_) > 0) __builtin_amdgcn_sched_group_barrier(0x002, (valu_), 0);                    \
            else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                      \
        }                                                                                                \
    }
#else
#define SM_ILV_PIPE(valu_)
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
#if SM_SPLIT2_ILV == 2
            __builtin_amdgcn_sched_barrier(0);   // a scheduling region = one stage
#endif
#if defined(SM_ABL_NOREAD)
            if (ch == ch_begin && tap == 0) { SM_READ_B(fb_next, 0, 1) }
#elif SM_SPLIT_PREFETCH_B
            if (tap < 8) {
                SM_READ_B_KY(fb_next, SM_CUR_SLOT((tap + 1) / 3), SM_CUR_SLOT((tap + 1) / 3 + 1), (tap + 1) / 3, (tap + 1) % 3)
            } else {
                SM_READ_B_KY(fb_next, SM_NEXT_SLOT(0), SM_NEXT_SLOT(1), 0, 0)
            }
#else
            SM_READ_B_KY(fb, SM_CUR_SLOT(ky), SM_CUR_SLOT(ky + 1), ky, kx)
#endif
#if SM_SPLIT_PIN_READS
            // keep the fragment reads HERE, a full stage ahead of their use: left alone, the scheduler sinks them to the
            // end of the stage (shorter live ranges) and the next stage's first MFMAs wait out the LDS round trip
            __builtin_amdgcn_sched_barrier(0);
#endif
            f32x4 fa[MI][NP];
#pragma unroll
            for (int s = 0; s < NP; ++s)
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i][s] = ra[tap % AD][i][s];
            // the partial products per output tile, smallest first; consecutive MFMAs target different accumulators
#define SM_PRODUCT(pa_, pb_)                                                                             \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                       \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                 \
            if constexpr (NP == 3)                                                                       \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i][pa_]), \
                                                                    __builtin_bit_cast(bf16x8, fb[j][pb_]), acc[i][j], 0, 0, 0); \
            else                                                                                         \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i][pa_]), \
                                                                   __builtin_bit_cast(f16x8, fb[j][pb_]), acc[i][j], 0, 0, 0); \
        }
#if SM_SPLIT_MFMA_PRIO
            __builtin_amdgcn_s_setprio(SM_SPLIT_MFMA_PRIO);
#endif
#ifdef SM_ABL_NOMFMA
            _Pragma("unroll") for (int s_ = 0; s_ < NP; ++s_) {
                _Pragma("unroll") for (int i = 0; i < MI; ++i) asm volatile("" :: "v"(fa[i][s_]));
                _Pragma("unroll") for (int j = 0; j < NJ; ++j) asm volatile("" :: "v"(fb[j][s_]));
            }
            if constexpr (false)
#endif
            if constexpr (NP == 3) {
                SM_PRODUCT(2, 0)
                SM_PRODUCT(0, 2)
                SM_PRODUCT(1, 1)
                SM_PRODUCT(1, 0)
                SM_PRODUCT(0, 1)
                SM_PRODUCT(0, 0)
            } else {
                SM_PRODUCT(1, 0)
                SM_PRODUCT(0, 1)
                SM_PRODUCT(0, 0)
            }
#undef SM_PRODUCT
            // the ring slot just consumed is refilled with the weights of stage + AD (pinned below the MFMAs: hoisting
            // the loads would need a fourth set of fragment registers)
#if SM_SPLIT2_ILV == 0
            __builtin_amdgcn_sched_barrier(0);
#endif
            if (STAMP && ch - ch_begin == 1) SM_TS(32 + tap)      // this stage's MFMAs are issued
#if SM_SPLIT_TAIL_PRIO
            __builtin_amdgcn_s_setprio(SM_SPLIT_TAIL_PRIO);   // the load / convert / store tail outranks the partner's MFMAs
#elif SM_SPLIT_MFMA_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
#ifndef SM_ABL_NOA   // (ablation builds, timing only: -DSM_ABL_NOA / _NOB / _NOREAD / _NOMFMA drop one ingredient of the loop)
            if (tap + AD < 9) {
                SM_LOAD_A(tap + AD, ch);
            } else {
                SM_LOAD_A(tap + AD - 9, ch_next);
            }
#endif
            // next chunk's slice ky -> slot (base + 3 + ky) & 3: for ky = 0 the spare slot (the previous chunk's
            // ky = 2), for ky = 1, 2 the slot of this chunk's slice ky - 1, whose last readers passed the barrier of
            // tap 3 ky - 1. The slice is loaded at the end of tap 3 ky - 1 (for ky = 0: tap 8 of the previous chunk),
            // converted and written at the end of tap 3 ky + 1 - a stage WITHOUT a barrier, so that the conversion does
            // not sit on a barrier's critical path - and published by the barrier at the end of tap 3 ky + 2.
            if constexpr (BSETS == 1) {
#ifndef SM_ABL_NOB
                if (kx == 1) { SM_STORE_B(0, SM_NEXT_SLOT(ky), ky); SM_ILV_PIPE(6 * NU) }
                if (kx == 0) { SM_ILV_PIPE(0) }
#endif
                // six slots: the next chunk is written into the other half of the ring, which nobody reads after the
                // barrier at the end of tap 7 of the previous chunk (tap 8 already prefetches from the new half): that
                // one barrier per chunk both publishes the three new slices and releases the old half
                if (RING6 && tap == 7) __syncthreads();
                if (kx == 2) {
#ifndef SM_ABL_NOB
                    if (ky < 2) {
                        SM_LOAD_B(0, ky + 1, ch_next);
                    } else {
                        SM_LOAD_B(0, 0, ch_next2);
                    }
#endif
                    SM_ILV_PIPE(0)
                    if (STAMP && ch - ch_begin == 1) SM_TS(41 + tap)   // at the barrier
                    if (!RING6) __syncthreads();
                }
            } else {
                if (kx == 1) {   // slice ky of the next chunk out of its register set, the chunk after that into it
                    SM_STORE_B(ky, SM_NEXT_SLOT(ky), ky);
                    SM_LOAD_B(ky, ky, ch_next2);
                }
                if (RING6 ? tap == 7 : kx == 2) __syncthreads();
            }
#if SM_SPLIT_TAIL_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
#if SM_SPLIT_PREFETCH_B
#pragma unroll
            for (int s = 0; s < NP; ++s)
#pragma unroll
                for (int i = 0; i < NJ; ++i) fb[i][s] = fb_next[i][s];
#endif
            if (STAMP && ch - ch_begin < 2) SM_TS(2 + (ch - ch_begin) * 12 + tap)
        }
        base = RING6 ? 3 - base : (base + 3) & 3;
    }
    }   // (ring kernel)
    SM_TS(30)
    SM_RT(2)
#undef SM_CUR_SLOT
#undef SM_NEXT_SLOT
#undef SM_UP_MAP
#undef SM_LOAD_A
#undef SM_LOAD_B
#undef SM_STORE_B
#undef SM_READ_B
#undef SM_READ_B_KY
#undef SM_ILV_PIPE

    // ---- KG = 2: the two groups' partial sums meet. Wave w of group g keeps column tiles [g NJ/2, (g + 1) NJ/2) of its
    // 32 MI x 32 NJ tile and hands the other half to wave w of the other group through the (now idle) slice rings; both
    // form group 0's sum + group 1's sum - one fixed order, deterministic - and each runs the epilogue for the tiles it
    // kept, in acc[.][0 .. NJ/2). (The loop's last barrier is behind every fragment read of the rings.)
    constexpr int NJE = NJ / KG;          // column tiles per wave in the epilogue
    const int nj0 = grp * NJE;            // first of them
    if constexpr (KG == 2) {
        static_assert(NJ % 2 == 0 && NJE % 2 == 0, "whole segment pairs per group");
        static_assert((size_t)8 * MI * NJE * 16 * 64 * 4 <= 2 * conv_split_lds_bytes(BM, BN, NP), "exchange fits the two rings");
        float* X = reinterpret_cast<float*>(smem4);
        float* mine = X + (size_t)((grp * 4 + wave) * (MI * NJE * 16)) * 64 + lane;
        const float* theirs = X + (size_t)(((1 - grp) * 4 + wave) * (MI * NJE * 16)) * 64 + lane;
        if (grp == 0) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < NJE; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mine[((mi * NJE + j) * 16 + r) * 64] = acc[mi][NJE + j][r];
        } else {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < NJE; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mine[((mi * NJE + j) * 16 + r) * 64] = acc[mi][j][r];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < NJE; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][j][r] = acc[mi][j][r] + theirs[((mi * NJE + j) * 16 + r) * 64];
        } else {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < NJE; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][j][r] = theirs[((mi * NJE + j) * 16 + r) * 64] + acc[mi][NJE + j][r];
        }
        SM_TS(50)
    }

    // ---- epilogue (same 32x32 C/D layout as conv3x3_mfma_kernel: column = lane & 31,
    //      row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)); column tile nj0 + j of the wave is in acc[.][j]
    if (split >= 0) {
        float* wt = a.ws + ((size_t)(tile - a.n_whole) * a.splits + split) * (BM * BN);
#pragma unroll
        for (int j = 0; j < NJE; ++j)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    wt[(wm + mi * 32 + 4 * lhi + (r & 3) + 8 * (r >> 2)) * BN + wn + (nj0 + j) * 32 + l31] =
                        NP == 2 ? acc[mi][j][r] * out_scale : acc[mi][j][r];   // power of two: exact
        SM_TS(51)
        return;
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
                bias4[mi][g] = *reinterpret_cast<const f32x4*>(a.bias + m0 + wm + mi * 32 + 4 * lhi + 8 * g);
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
        for (int pj = 0; pj < NJE; pj += 2) {
            int q_seg = qs[0];
            bool alive = live[0];
#pragma unroll
            for (int k = 1; k < SEG; ++k)
                if (wn / 32 + nj0 + pj == k) { q_seg = qs[k]; alive = live[k]; }
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
                    if (NP == 2) { t *= out_scale; b *= out_scale; }
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
                    if (ok) P.pool_out[o0 + (size_t)((r & 3) + 8 * (r >> 2)) * plane_o] =
                        pout ? __builtin_bit_cast(float, pair_encode(m, po_scale)) : m;
                    codes[r >> 2] |= c << (4 * ((r & 3) + 4 * lhi));
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned full = codes[g] | (unsigned)__shfl_xor((int)codes[g], 32, 64);   // channels 8g + 0..3 | 4..7
                    if (ok && lhi == 0) P.pool_code[(size_t)((m0 + wm + mi * 32) / 8 + g) * plane_o + qo] = full;
                }
            }
        }
        record_amax(a.amax_out, vmax, amax_seen);
        SM_RT(3)
        return;
    }
    // SM_EPI_GRAM: the output layer is a style layer of C = BM channels (the block's row tile holds all of them) and this
    // epilogue adds its masked Gram backward, sum_k m_k(q) (D_k F)(q), F = the gate operand - C / 16 sixteen-channel steps
    // per mask on the matrix cores for each of the wave's 32-position column tiles, in gram_backward_body's operand format and product order (the sum has its bits);
    // the gradient plane is then written once instead of written (Gram backward), read (here) and written again.
#ifdef SM_ABL_NOGRAM   // (ablation build, timing only)
    constexpr bool GRAM = false;
#else
    constexpr bool GRAM = (FLAGS & SM_EPI_GRAM) != 0;
#endif
    float g_fscale = 1.f, g_oscale = 1.f;
    if constexpr (GRAM) {
        static_assert(NP == 2 && MI == 1 && (BM == 64 || BM == 128) && (FLAGS & SM_EPI_RELU_MASK) && !(FLAGS & SM_EPI_ADD),
                      "the Gram term replaces the addend of a data gradient whose row tile holds all C = BM channels");
        float inv_f, inv_d;
        g_fscale = conv_gram_pow2_scale(amax_read(P.gram_amax_feat), inv_f);
        if (gate_pair) inv_f = a.pair_gate[1];   // F = the gate planes, stored as pairs under their own scale
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
        constexpr int PH = BM / 64;             // phases of 64 channels (the ring holds 2 parts x 8 k-groups x BN units)
        constexpr int GU = 8 * BN / 256;        // staging units (k-group, position) per thread and phase
        static_assert((2 * 8 * BN) * 16 + (BM / 8) * BN <= (RES ? conv_resident_lds_bytes() : (size_t)conv_split_slots(NP, BM) * SLICE * 16),
                      "a phase of the Gram operand + the gate bits of all channels fit the slice ring");
        f32x4* Gs = smem4;
        // F is also the ReLU gate of this launch's output: the staging threads - which hold the raw values - leave one bit
        // per (channel, position), x > 0, behind the operand (byte [channel / 8][position]); the store loop below reads its
        // gate from there instead of loading the layer a second time (conv1_2's data gradient moved 1.27 GB, a third of it
        // this second read)
        unsigned char* Gb = reinterpret_cast<unsigned char*>(smem4 + 2 * 8 * BN);
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
                        rb[u][c] = P.gate[(size_t)(ph * 64 + kg * 8 + c) * P.plane + g_q];
                }
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    const int kg = tid / BN + u * (256 / BN);
                    f32x4 vh, vl;
                    if (gate_pair) pair_units(rb[u], vh, vl);
                    else conv_gram_split(rb[u], g_fscale, vh, vl);
                    Gs[kg * BN + g_pos] = vh;
                    Gs[(8 + kg) * BN + g_pos] = vl;
                    unsigned bits = 0u;
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        bits |= ((gate_pair ? __builtin_bit_cast(unsigned, rb[u][c]) != 0u : rb[u][c] > 0.f) ? 1u : 0u) << c;
                    Gb[(ph * 8 + kg) * BN + g_pos] = (unsigned char)bits;
                }
            }
            __syncthreads();
            SM_RT(8)
#pragma unroll
            for (int chunk = 0; chunk < 2; ++chunk)                    // 32 channels
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int t = (ph * 2 + chunk) * 2 + ks;       // k-step of 16 channels
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
                            fb[1] = keep ? gf[8 * BN] : zero4;
                            conv_gram_mfma(accg[nj], fa, fb);
                        }
                    }
        }
        SM_RT(9)
    }
#pragma unroll
    for (int nj = 0; nj < NJE; ++nj) {   // (KG = 2: column tile nj0 + nj of the wave, held in acc[.][nj])
        int q_seg = qs[0];
        bool alive = live[0];
#pragma unroll
        for (int k = 1; k < SEG; ++k)
            if (wn / 32 + nj0 + nj == k) { q_seg = qs[k]; alive = live[k]; }
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
                if (FLAGS & SM_EPI_ADD) prev[r] = P.addend ? P.addend[o] : P.out[o];
                if ((FLAGS & SM_EPI_RELU_MASK) && !GRAM) {
                    const float gv = P.gate[o];
                    gate[r] = gate_pair ? (__builtin_bit_cast(unsigned, gv) != 0u ? 1.f : 0.f) : gv;
                }
            }
            if constexpr (GRAM) {   // (MI == 1) the gate bits of this lane's 16 rows: byte g = channels wm + 8 g + 0..7
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned b8 = Gs_gate_byte(smem4, (wm / 8 + g) * BN + wn + nj * 32 + l31, 2 * 8 * BN);
#pragma unroll
                    for (int k = 0; k < 4; ++k) gate[4 * g + k] = ((b8 >> (4 * lhi + k)) & 1u) ? 1.f : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const size_t o = o0 + (size_t)((r & 3) + 8 * (r >> 2)) * P.plane;
                float v = acc[mi][nj][r];
                if (NP == 2) v *= out_scale;
                if (FLAGS & SM_EPI_BIAS_RELU) v = fmaxf(v + bias4[mi][r >> 2][r & 3], 0.f);
                if (FLAGS & SM_EPI_ADD) v += prev[r];
                if constexpr (GRAM) v += accg[nj][r] * g_oscale;
                if (FLAGS & SM_EPI_RELU_MASK) v = (gate[r] > 0.f) ? v : 0.f;
                v = inside ? v : 0.f;
                P.out[o] = pout ? __builtin_bit_cast(float, pair_encode(v, po_scale)) : v;
                vmax = fmaxf(vmax, fabsf(v));
            }
        }
    }
    record_amax(a.amax_out, vmax, amax_seen);
    SM_TS(31)
    SM_RT(3)
#undef SM_TS
#undef SM_RT
}

}  // namespace sm

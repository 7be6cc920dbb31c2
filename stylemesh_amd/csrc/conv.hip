// K3/K4: VGG 3x3 convolutions (forward and data gradient) as implicit-im2col GEMMs on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32), plus 2x2 max-pool forward / backward and the 64->3 first-layer data gradient.
//
// Reference operators replaced (lukasHoel/stylemesh): nn.Conv2d + F.relu + nn.MaxPool2d of
// model/losses/content_and_style_losses.py:11-32,49-69 and their autograd backward.
//
// GEMM view of one conv:  out[co][q] = sum_{tap,ci} Wt[tap][ci][co] * in[ci][q + d(tap)],
//   M = C_out, N = linear positions q of the padded plane (rows 1..H), K = 9 * C_in,
//   d(tap) = (ky-1)*Wp + (kx-1). Because the plane carries its own zero border, the im2col shift is a pure
//   offset in q: no per-tap bounds checks, and a tile of BN consecutive q needs just three contiguous input
//   segments per channel (one per ky) of BN+8 floats, staged in LDS once per K-chunk.
// Block = 256 threads = 4 waves; every wave owns a 64x64 output tile = 2x2 MFMA 32x32 tiles (64 accumulator
// VGPRs). A K-chunk = KC input channels x 9 taps. LDS: weights [9][KC][BM] + inputs [3][KC][BN+8] fp32
// (49.9 KB for 128x128x8 -> 3 blocks per CU, whose MFMA phases cover each other's staging phases).
#include <algorithm>
#include <cstdlib>

#include "conv_common.h"
#include "conv_tail.h"
#include "conv_split_kernel.h"

#ifndef SM_SPLIT2_BN256
#define SM_SPLIT2_BN256 1   // fp16x2, Cout % 128 != 0: 64 x 256 tiles instead of 64 x 128
#endif
#ifndef SM_SPLIT2_BM256
#define SM_SPLIT2_BM256 0
#endif
#ifndef SM_SPLIT_WGM
#define SM_SPLIT_WGM 4   // waves along the channel dimension of the 128 x 128 split tile: 4 (32x128 wave tiles: half the
                         // weight-fragment loads per MFMA, +2.5 %) or 2 (64x64)
#endif

namespace sm {

template <int BM, int BN, int KC, int WGM, int WGN, int FLAGS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void conv3x3_mfma_kernel(ConvArgs a) {
    static_assert(BM / WGM == 64 && BN / WGN == 64 && WGM * WGN == 4, "wave tile is 64x64");
    constexpr int BNP = BN + 8;  // 4 floats of halo on each side keeps every segment 16-byte aligned
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [9][KC][BM]
    float* Bs = smem + 9 * KC * BM;   // [3][KC][BNP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int lhi = lane >> 5;
    const int wm = (wave / WGN) * 64;
    const int wn = (wave % WGN) * 64;

    int tile, split;
    conv_unit(a, blockIdx.x, tile, split);   // XCD-aware order of whole tiles and of the tail's (tile, K-split) units
    const int m_tile = tile / a.n_tiles;
    const int n_glob = tile - m_tile * a.n_tiles;
    // which problem of the group does this position tile belong to (block-uniform scalar selects)
    ConvProblem P = a.p[0];
    int n_tile = n_glob;
    if (a.tile_list) {
        const int e = a.tile_list[n_glob];
        const int gsel = e >> 24;
        n_tile = e & 0xFFFFFF;
#pragma unroll
        for (int g = 1; g < SM_MAX_GROUP; ++g)
            if (g == gsel) P = a.p[g];
    } else {
#pragma unroll
        for (int g = 1; g < SM_MAX_GROUP; ++g)
            if (g < a.n_problems && n_glob >= a.tile_begin[g]) {
                P = a.p[g];
                n_tile = n_glob - a.tile_begin[g];
            }
    }
    const int c_begin = split < 0 ? 0 : split * a.chunks_per_split * KC;
    const int c_end = split < 0 ? a.Cin_pad : min(a.Cin_pad, c_begin + a.chunks_per_split * KC);
    const int m0 = m_tile * BM;
    const int q0 = P.Wp + n_tile * BN;  // first computed position = start of row 1

    const float amax_seen = split < 0 ? amax_peek(a.amax_out) : 0.f;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- staging plan: every thread owns NA float4 of the weight slab and NB float4 of the input segments per
    //      K-chunk. Source offsets are chunk-invariant up to a stride (Cout resp. plane floats per channel).
    constexpr int A_F4_PER_ROW = BM / 4;
    constexpr int A_F4 = 9 * KC * A_F4_PER_ROW;
    constexpr int B_F4_PER_ROW = BNP / 4;
    constexpr int B_F4 = 3 * KC * B_F4_PER_ROW;
    constexpr int NA = (A_F4 + 255) / 256;
    constexpr int NB = (B_F4 + 255) / 256;
    // weights: a pass of 256 threads covers RPP = 1024/BM rows (tap, ci) of the slab; RPP is a multiple of KC, so a
    // thread keeps its ci and advances by RPP/KC taps per pass: one base offset + a uniform stride.
    constexpr int RPP = 256 / A_F4_PER_ROW;
    static_assert(RPP % KC == 0, "a staging pass covers whole taps");
    const int a_r0 = tid / A_F4_PER_ROW, a_c4 = tid % A_F4_PER_ROW;
    const unsigned a_src0 = (unsigned)(((a_r0 / KC) * a.Cin_pad + (a_r0 % KC)) * a.Cout + m0 + a_c4 * 4);
    const unsigned a_step = (unsigned)((RPP / KC) * a.Cin_pad * a.Cout);
    const int a_dst0 = a_r0 * BM + a_c4 * 4;
    // the last pass may be partial: threads beyond the slab re-load pass 0 (valid memory) and skip the LDS store
    const bool a_tail_ok = A_F4 % 256 == 0 || tid < A_F4 - (NA - 1) * 256;
    const int a_last = a_tail_ok ? NA - 1 : 0;
    // inputs: rows (ky, ci) of BNP floats; generic split, 32-bit element offsets
    int b_src[NB];   // signed: the first tile's halo starts 4 floats BEFORE the plane
    int b_dst[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int i = min(tid + j * 256, B_F4 - 1);
        const int row = i / B_F4_PER_ROW, c4 = i - row * B_F4_PER_ROW;
        const int ky = row / KC, ci = row - ky * KC;
        b_src[j] = ci * P.plane + q0 + (ky - 1) * P.Wp - 4 + c4 * 4;
        b_dst[j] = row * BNP + c4 * 4;
    }
    f32x4 ra[NA], rb[NB];
    // (macros, not lambdas: capturing the register arrays by reference sends them to scratch memory)
#define SM_LOAD_CHUNK(c0_)                                                                          \
    {                                                                                               \
        const float* wsrc = a.wt + (size_t)(c0_) * a.Cout + a_src0;                                 \
        _Pragma("unroll") for (int j = 0; j < NA; ++j)                                              \
            ra[j] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)((j == NA - 1) ? a_last : j) * a_step); \
        const float* isrc = P.in + (size_t)(c0_) * P.plane;                                         \
        _Pragma("unroll") for (int j = 0; j < NB; ++j)                                              \
            rb[j] = *reinterpret_cast<const f32x4*>(isrc + b_src[j]);                               \
    }
#define SM_STORE_CHUNK()                                                                            \
    {                                                                                               \
        _Pragma("unroll") for (int j = 0; j < NA; ++j)                                              \
            if (j < NA - 1 || a_tail_ok)                                                            \
                *reinterpret_cast<f32x4*>(As + a_dst0 + j * RPP * BM) = ra[j];                      \
        _Pragma("unroll") for (int j = 0; j < NB; ++j)                                              \
            if (B_F4 % 256 == 0 || tid + j * 256 < B_F4)                                            \
                *reinterpret_cast<f32x4*>(Bs + b_dst[j]) = rb[j];                                   \
    }

    // software pipeline: the global loads of chunk c+1 are in flight while the MFMAs of chunk c run
    SM_LOAD_CHUNK(c_begin);
    SM_STORE_CHUNK();
    __syncthreads();
    for (int c0 = c_begin; c0 < c_end; c0 += KC) {
        const bool more = c0 + KC < c_end;
        if (more) SM_LOAD_CHUNK(c0 + KC);
        // 9 taps x KC/2 channel pairs = NSTEP k-steps of 4 MFMAs; the LDS fragments of step s+1 are requested
        // before the MFMAs of step s are issued, so the matrix pipe never waits on an LDS round trip.
        {
            constexpr int NSTEP = 9 * (KC / 2);
            const float* abase = As + lhi * BM + wm + l31;
            const float* bbase = Bs + lhi * BNP + 3 + wn + l31;
            float a0 = abase[0], a1 = abase[32], b0 = bbase[0], b1 = bbase[32];
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                float na0, na1, nb0, nb1;
                if (st + 1 < NSTEP) {
                    const int tap = (st + 1) / (KC / 2), kp = (st + 1) % (KC / 2);
                    const int ky = tap / 3, kx = tap % 3;
                    const float* ap = abase + (tap * KC + kp * 2) * BM;
                    const float* bp = bbase + (ky * KC + kp * 2) * BNP + kx;
                    na0 = ap[0]; na1 = ap[32]; nb0 = bp[0]; nb1 = bp[32];
                }
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                if (st + 1 < NSTEP) { a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; }
                // pin the emitted order: one MFMA, the next step's two LDS reads, the other three MFMAs - the reads
                // complete under ~190 cycles of matrix-pipe time
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            }
        }
        __syncthreads();
        if (more) {
            SM_STORE_CHUNK();
            __syncthreads();
        }
    }
#undef SM_LOAD_CHUNK
#undef SM_STORE_CHUNK

    // ---- epilogue. C/D layout of the 32x32 MFMA: column (pixel) = lane & 31,
    //      row (channel) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
    if (split >= 0) {   // tail unit: raw partial tile -> workspace, [BM][BN] row-major
        float* wt = a.ws + ((size_t)(tile - a.n_whole) * a.splits + split) * (BM * BN);
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    wt[(wm + mi * 32 + 4 * lhi + (r & 3) + 8 * (r >> 2)) * BN + wn + nj * 32 + l31] = acc[mi][nj][r];
        return;
    }
    const int q_end = (P.H + 1) * P.Wp;
    float vmax = 0.f;
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
        const int q = q0 + wn + nj * 32 + l31;
        if (q >= q_end) continue;
        const bool inside = interior(q, P.H, P.W, P.Wp);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int co_base = m0 + wm + mi * 32 + 4 * lhi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_base + (r & 3) + 8 * (r >> 2);
                const size_t o = (size_t)co * P.plane + q;
                float v = acc[mi][nj][r];
                if (FLAGS & SM_EPI_BIAS_RELU) v = fmaxf(v + a.bias[co], 0.f);
                if (FLAGS & SM_EPI_ADD) v += P.out[o];
                if (FLAGS & SM_EPI_RELU_MASK) v = (P.gate[o] > 0.f) ? v : 0.f;
                v = inside ? v : 0.f;
                P.out[o] = v;
                vmax = fmaxf(vmax, fabsf(v));
            }
        }
    }
    record_amax(a.amax_out, vmax, amax_seen);
}

// max |x| over the interior rows of C planes (producers without an amax epilogue: the deepest loss layer's gradient)
__global__ __launch_bounds__(256) void fmap_amax_kernel(const float* __restrict__ in, int plane, int q_begin, int q_end,
                                                        float* amax_out) {
    const float* p = in + (size_t)blockIdx.y * plane;
    const float seen = amax_peek(amax_out);
    float m = 0.f;
    for (int q = q_begin + (blockIdx.x * 256 + threadIdx.x) * 4; q < q_end; q += gridDim.x * 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + q);     // q_begin, q_end: multiples of 4
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    record_amax(amax_out, m, seen);
}

// SPLIT = false: exact fp32 MFMA kernel; true: fp16x2-split kernel of conv_split_kernel.h (KC must be 16).
template <int BM, int BN, int KC, int WGM, int WGN, int FLAGS, bool SPLIT = false, bool UNPOOL = false, bool RES = false>
static int launch_conv(const ConvArgs& a0, int n_list, size_t ws_floats, hipStream_t s) {
    ConvArgs a = a0;
    a.m_tiles = a.Cout / BM;
    a.tile_begin[0] = 0;
    for (int g = 0; g < a.n_problems; ++g)
        a.tile_begin[g + 1] = a.tile_begin[g] + (a.p[g].H * a.p[g].Wp + BN - 1) / BN;
    // the split kernels take SEGMENT lists (BN / 32 entries of 32 positions per tile), the fp32 kernel BN-position tiles
    a.list_segments = SPLIT ? 1 : 0;
    if (SPLIT && a.tile_list && n_list % (BN / 32) != 0) return (int)hipErrorInvalidValue;
    a.n_tiles = a.tile_list ? (SPLIT ? n_list / (BN / 32) : n_list) : a.tile_begin[a.n_problems];
    if (a.n_tiles == 0) return 0;
    constexpr size_t lds = RES ? conv_resident_lds_bytes()
                         : SPLIT ? conv_split_lds_bytes(BM, BN)
                                 : (size_t)(9 * KC * BM + 3 * KC * (BN + 8)) * sizeof(float);
    if (RES && (a.tile_list == nullptr || a.Cin_pad % 64 != 0)) return (int)hipErrorInvalidValue;   // quads of a list; 64-channel phases
    const int tiles = a.m_tiles * a.n_tiles, chunks = a.Cin_pad / KC;
    // Full rounds of one tile per resident block slot run whole; the tail of `rem` tiles is split along K so that
    // it becomes about one more (short) round of rem * splits small units. Pick the split count that minimises the
    // tail's duration ceil(rem * S / slots) / S, each split keeping >= 2 K-chunks. (The split kernel keeps two
    // blocks resident per CU, but they share the matrix pipes: measured, rounds of 2 x CUs tiles are slower.)
#ifndef SM_SPLIT2_SLOTS
#define SM_SPLIT2_SLOTS 1
#endif
#ifndef SM_SMALL_SLOTS
#define SM_SMALL_SLOTS 3
#endif
#ifndef SM_CONV_SPLIT_PENALTY_DEFAULT
#define SM_CONV_SPLIT_PENALTY_DEFAULT 3.f   // measured (profiles/r04/split_penalty_ab.txt): c3 +0.4 %, c2 +0.6 % at 2-4, c2 -5 % at 8
#endif
    // (the fp16x2 variant is not matrix-pipe-bound with one block per CU: its rounds take SM_SPLIT2_SLOTS blocks per CU)
    // (64 x 128 tiles - the small-grid choice of round 5, dispatch_conv_split2 - are sized for three blocks per CU)
    constexpr int SLOTS = SM_NUM_CU * (SPLIT ? ((BM == 64 && BN == 128) ? SM_SMALL_SLOTS : SM_SPLIT2_SLOTS) : 1);
    a.n_whole = tiles / SLOTS * SLOTS;
    a.splits = 1;
    a.chunks_per_split = chunks;
    int rem = tiles - a.n_whole;
    if (a.ws != nullptr && rem > 0 && chunks >= 4) {
        const int max_s = (int)std::min<size_t>({(size_t)chunks / 2, (size_t)16, ws_floats / ((size_t)rem * BM * BN)});
        // cost of the tail in units of one whole tile; every unit pays ~1 chunk of fixed prologue / epilogue time
        float best = (chunks + 1.f) / chunks;   // S = 1: one more full round
        // a split tail costs a second launch (the reduce / epilogue pass: 9-13 us + the slabs' round trip) whatever it
        // saves: SM_CONV_SPLIT_PENALTY / chunks tile times (a tile takes ~3.6 us per K-chunk: 3.3 ~ 12 us; 0 = rounds 1-3)
        static const float penalty = getenv("SM_CONV_SPLIT_PENALTY") ? (float)atof(getenv("SM_CONV_SPLIT_PENALTY")) : SM_CONV_SPLIT_PENALTY_DEFAULT;
        const float second_pass = penalty / chunks;
        for (int S = 2; S <= max_s; ++S) {
            const int cps = (chunks + S - 1) / S, S_eff = (chunks + cps - 1) / cps;
            const float cost = (float)((rem * S_eff + SLOTS - 1) / SLOTS) * (cps + 1.f) / chunks + second_pass;
            if (cost < best * 0.97f) { best = cost; a.splits = S_eff; a.chunks_per_split = cps; }
        }
    }
    if (RES) a.splits = 1;                       // resident input: whole tiles only (K <= 1152: nothing to split)
    if (a.splits == 1) { a.n_whole = tiles; rem = 0; }
    if constexpr (SPLIT) {
        static_assert(KC == 16, "one fp16 MFMA K-step per tap");
        // (> 64 KB of dynamic LDS needs the opt-in)
        auto k = conv3x3_split_kernel<BM, BN, WGM, WGN, FLAGS, UNPOOL, RES>;
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            attr_done = true;
        }
        hipLaunchKernelGGL(k, dim3(a.n_whole + rem * a.splits), dim3(256), lds, s, a);
    } else {
        hipLaunchKernelGGL((conv3x3_mfma_kernel<BM, BN, KC, WGM, WGN, FLAGS>), dim3(a.n_whole + rem * a.splits),
                           dim3(256), lds, s, a);
    }
    SM_LAUNCH_CHECK();
    if (rem > 0) {
        if constexpr ((FLAGS & SM_EPI_POOL) != 0)
            hipLaunchKernelGGL((conv_tail_pool_kernel<BM, BN>), dim3(rem, BM / 8, BN / 128), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((conv_tail_epilogue_kernel<BM, BN, FLAGS>), dim3(rem, BM * BN / 1024), dim3(256), 0, s, a);
        SM_LAUNCH_CHECK();
    }
    return 0;
}

template <int FLAGS, bool UNPOOL = false>
static int dispatch_conv_split2(const ConvArgs& a, int n_list, size_t ws_floats, hipStream_t s) {
#if SM_SPLIT2_BN256
    // 64 output channels: 64 x 256 tiles (the same 12 MFMAs per stage and wave as the 128-row tile), waves 2 x 2 with
    // 32 x 128 wave tiles: half the weight-fragment loads of the 1 x 4 layout (64 x 64 wave tiles, all four waves
    // fetching the same 64 rows), +3-5 % on the 64-channel layers
#ifndef SM_SPLIT2_W64GM
#define SM_SPLIT2_W64GM 2
#endif
    if (a.Cout % 128 != 0)
        return launch_conv<64, 256, 16, SM_SPLIT2_W64GM, 4 / SM_SPLIT2_W64GM, FLAGS, true, UNPOOL>(a, n_list, ws_floats, s);
#else
    if (a.Cout % 128 != 0) return launch_conv<64, 128, 16, 2, 2, FLAGS, true, UNPOOL>(a, n_list, ws_floats, s);
#endif
    return launch_conv<128, 128, 16, SM_SPLIT_WGM, 4 / SM_SPLIT_WGM, FLAGS, true, UNPOOL>(a, n_list, ws_floats, s);
}

// ---------------------------------------------------------------------------------------------------
// First layer forward (3 -> C_out channels, bias + ReLU): 27 multiply-adds per output are far too thin for the matrix
// cores - the MFMA tile kernel spends its time in the epilogue (64 x 256 tiles: 165 us for 7.6 GFLOP, 2 TB/s of stores).
// A streaming VALU kernel instead: a thread holds the 3 x 3 x 6 input window of FOUR consecutive positions in registers
// and walks the output channels, weights through the scalar cache (wave-uniform), one 16-byte store per channel
// (1 KB contiguous per wave): bound by the HBM write of the output planes. A block = 1024 positions (the tile unit of
// this layer's active-tile list, sm_conv_tile_positions(4, .)).
// ---------------------------------------------------------------------------------------------------
// (wt / bias arrive as separate __restrict__ parameters: read through the pointers inside ConvArgs the compiler cannot
// rule out that the output stores alias them, and re-loads the weights with VECTOR loads after the first store)
__global__ __launch_bounds__(256) void conv3x3_c3_fwd_kernel(ConvArgs a, const float* __restrict__ wt,
                                                             const float* __restrict__ bias) {
    const int n_glob = blockIdx.x;
    ConvProblem P = a.p[0];
    int n_tile = n_glob;
    if (a.tile_list) {
        const int e = a.tile_list[n_glob];
        const int gsel = e >> 24;
        n_tile = e & 0xFFFFFF;
#pragma unroll
        for (int g = 1; g < SM_MAX_GROUP; ++g)
            if (g == gsel) P = a.p[g];
    } else {
#pragma unroll
        for (int g = 1; g < SM_MAX_GROUP; ++g)
            if (g < a.n_problems && n_glob >= a.tile_begin[g]) {
                P = a.p[g];
                n_tile = n_glob - a.tile_begin[g];
            }
    }
    const float seen = amax_peek(a.amax_out);
    const int q = P.Wp + (n_tile * 256 + threadIdx.x) * 4;   // Wp % 4 == 0: 16-byte aligned
    float vmax = 0.f;
    if (q < (P.H + 1) * P.Wp) {
        float x[3][3][6];
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* r = P.in + (size_t)ci * P.plane + q + (ky - 1) * P.Wp;
                const f32x4 m = *reinterpret_cast<const f32x4*>(r);
                x[ci][ky][0] = r[-1]; x[ci][ky][1] = m[0]; x[ci][ky][2] = m[1]; x[ci][ky][3] = m[2]; x[ci][ky][4] = m[3];
                x[ci][ky][5] = r[4];
            }
        bool in4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) in4[j] = interior(q + j, P.H, P.W, P.Wp);
        const int Cout = a.Cout;
        // small grids (a single 256 x 341 level is 86 blocks on 256 CUs): blockIdx.y walks groups of output channels, every
        // group re-reading the 3-plane input window (a few KB, L2)
        const int co_per = Cout / gridDim.y, co_begin = blockIdx.y * co_per;
        for (int co = co_begin; co < co_begin + co_per; co += 4) {   // four output channels per pass: 27 weights each as scalar float4s
            // accumulators as pairs of output channels: the multiply-adds issue as v_pk_fma_f32 (two per instruction)
            f32x2 acc[2][4];
#pragma unroll
            for (int cp = 0; cp < 2; ++cp)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[cp][j] = f32x2{bias[co + 2 * cp], bias[co + 2 * cp + 1]};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(wt + ((ky * 3 + kx) * 4 + ci) * Cout + co);
                        const f32x2 w01 = {w[0], w[1]}, w23 = {w[2], w[3]};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float xv = x[ci][ky][j + kx];
                            const f32x2 x2 = {xv, xv};
                            acc[0][j] = __builtin_elementwise_fma(w01, x2, acc[0][j]);
                            acc[1][j] = __builtin_elementwise_fma(w23, x2, acc[1][j]);
                        }
                    }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = in4[j] ? fmaxf(acc[c >> 1][j][c & 1], 0.f) : 0.f;
                    vmax = fmaxf(vmax, v[j]);
                }
                *reinterpret_cast<f32x4*>(P.out + (size_t)(co + c) * P.plane + q) = v;
            }
        }
    }
    record_amax(a.amax_out, vmax, seen);
}

static int launch_conv_c3_fwd(const ConvArgs& a0, int n_list, hipStream_t s) {
    ConvArgs a = a0;
    a.tile_begin[0] = 0;
    for (int g = 0; g < a.n_problems; ++g)
        a.tile_begin[g + 1] = a.tile_begin[g] + (a.p[g].H * a.p[g].Wp + 1023) / 1024;
    const int n = a.tile_list ? n_list : a.tile_begin[a.n_problems];
    if (n == 0) return 0;
    int gy = 1;   // >= ~4 blocks per CU in total; groups of at least 4 channels that divide Cout
    while (n * gy < 4 * SM_NUM_CU && a.Cout % (gy * 2 * 4) == 0) gy *= 2;
    hipLaunchKernelGGL(conv3x3_c3_fwd_kernel, dim3(n, gy), dim3(256), 0, s, a, a.wt, a.bias);
    SM_LAUNCH_CHECK();
    return 0;
}

template <int FLAGS>
static int dispatch_conv(const ConvArgs& a, int n_list, size_t ws_floats, hipStream_t s) {
    if (a.Cin_pad == 4 && FLAGS == SM_EPI_BIAS_RELU && a.Cout % 4 == 0) return launch_conv_c3_fwd(a, n_list, s);
    if (a.Cin_pad == 4) {   // (other epilogues: the MFMA tile kernel; its tiles are not the 1024-position list units)
        if (a.tile_list != nullptr) return (int)hipErrorInvalidValue;
        return launch_conv<64, 256, 4, 1, 4, FLAGS>(a, n_list, ws_floats, s);
    }
    if (a.Cout % 128 != 0) return launch_conv<64, 256, 8, 1, 4, FLAGS>(a, n_list, ws_floats, s);
    return launch_conv<128, 128, 8, 2, 2, FLAGS>(a, n_list, ws_floats, s);
}

#ifndef SM_DGRAD_C3_XCD
#define SM_DGRAD_C3_XCD 1
#endif
// ---------------------------------------------------------------------------------------------------
// First-layer data gradient (64 -> 3 channels): far too thin for the matrix cores (M = 3), so a VALU kernel:
// one thread per position q, 3 accumulators, weights read through the scalar cache (wave-uniform).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dgrad_c3_body(const float* __restrict__ dz, const float* __restrict__ wd, float* out,
                                              int Cin, int H, int W, int Wp, int plane, int block_x) {
    // four consecutive positions per thread: per channel and row one float4 + its two neighbours feed 4 x 3 taps
    // (1.5 loads per output instead of 9), weights through the scalar cache (wave-uniform)
    const int q = Wp + (block_x * 256 + threadIdx.x) * 4;   // Wp % 4 == 0: 16-byte aligned
    if (q >= (H + 1) * Wp) return;
    float acc[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[c][j] = 0.f;
    // the next channel's three rows are requested before this channel's 108 multiply-adds (the compiler does not
    // pipeline the loop by itself: every iteration would wait out a full memory round trip)
    f32x4 nm[3];
    float nl[3], nr[3];
#define SM_DG_LOAD(ci_)                                                                 \
    {                                                                                   \
        const float* p_ = dz + (size_t)min((ci_), Cin - 1) * plane + q;                 \
        _Pragma("unroll") for (int ky = 0; ky < 3; ++ky) {                              \
            const float* r_ = p_ + (ky - 1) * Wp;                                       \
            nm[ky] = *reinterpret_cast<const f32x4*>(r_);                               \
            nl[ky] = r_[-1];                                                            \
            nr[ky] = r_[4];                                                             \
        }                                                                               \
    }
    SM_DG_LOAD(0)
    for (int ci = 0; ci < Cin; ++ci) {
        f32x4 cm[3];
        float cl[3], cr[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) { cm[ky] = nm[ky]; cl[ky] = nl[ky]; cr[ky] = nr[ky]; }
        SM_DG_LOAD(ci + 1)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const f32x4 m = cm[ky];
            const float x[6] = {cl[ky], m[0], m[1], m[2], m[3], cr[ky]};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float* w = wd + ((ky * 3 + kx) * Cin + ci) * 4;
                const float w0 = w[0], w1 = w[1], w2 = w[2];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0][j] = fmaf(w0, x[j + kx], acc[0][j]);
                    acc[1][j] = fmaf(w1, x[j + kx], acc[1][j]);
                    acc[2][j] = fmaf(w2, x[j + kx], acc[2][j]);
                }
            }
        }
    }
#undef SM_DG_LOAD
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = interior(q + j, H, W, Wp) ? acc[c][j] : 0.f;
        *reinterpret_cast<f32x4*>(out + (size_t)c * plane + q) = v;
    }
}

// ---------------------------------------------------------------------------------------------------
// 2x2 max-pool, floor output size.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void maxpool_fwd_body(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                 int Wp, int plane, int Ho, int Wo, int Wpo, int plane_o, int block_x) {
    const int c = blockIdx.y;
    const int q = Wpo + block_x * 256 + threadIdx.x;  // output position, rows 1..Ho
    if (q >= (Ho + 1) * Wpo) return;
    const int r = q / Wpo, x = q - r * Wpo;
    float v = 0.f;
    if (x >= 1 && x <= Wo) {
        const float* p = in + (size_t)c * plane + (2 * (r - 1) + 1) * Wp + 2 * (x - 1) + 1;
        v = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[Wp], p[Wp + 1]));
    }
    out[(size_t)c * plane_o + q] = v;
}
// The same with argmax codes: a thread pools EIGHT channels of its position (blockIdx.y = 8-channel group) and writes,
// beside the pooled values, one dword code[g][q] whose nibble c says which window element of channel 8 g + c holds the
// FIRST maximum in row-major order (0..3 = dy * 2 + dx; the rule of maxpool_bwd_relu_body / ATen), or 4 when the
// maximum is <= 0 (every ReLU gate of the window is closed) or the position is padding - what the data-gradient
// convolution below the pool needs to take the pool's backward on the fly (ConvProblem::code).
__device__ __forceinline__ void maxpool_fwd_codes_body(const float* __restrict__ in, float* __restrict__ out,
                                                       uint32_t* __restrict__ code, int H, int W, int Wp, int plane,
                                                       int Ho, int Wo, int Wpo, int plane_o, int block_x) {
    const int g = blockIdx.y;
    const int q = Wpo + block_x * 256 + threadIdx.x;  // output position, rows 1..Ho
    if (q >= (Ho + 1) * Wpo) return;
    const int r = q / Wpo, x = q - r * Wpo;
    const bool inside = x >= 1 && x <= Wo;
    const float* p = in + (size_t)g * 8 * plane + (2 * (r - 1) + 1) * Wp + 2 * (x - 1) + 1;
    float v00[8], v01[8], v10[8], v11[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {   // all loads of the eight windows in flight together
        const float* pc = inside ? p + (size_t)c * plane : in;
        v00[c] = pc[0]; v01[c] = pc[1]; v10[c] = pc[Wp]; v11[c] = pc[Wp + 1];
    }
    uint32_t codes = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float v = inside ? fmaxf(fmaxf(v00[c], v01[c]), fmaxf(v10[c], v11[c])) : 0.f;
        const uint32_t k = (!inside || !(v > 0.f)) ? 4u : (v00[c] == v ? 0u : (v01[c] == v ? 1u : (v10[c] == v ? 2u : 3u)));
        codes |= k << (4 * c);
        out[(size_t)(g * 8 + c) * plane_o + q] = v;
    }
    code[(size_t)g * plane_o + q] = codes;
}

__device__ __forceinline__ void maxpool_bwd_relu_body(const float* __restrict__ act, const float* __restrict__ pooled,
                                                      const float* __restrict__ dpooled, float* __restrict__ dact,
                                                      int H, int W, int Wp, int plane, int Ho, int Wo, int Wpo,
                                                      int plane_o, int block_x) {
    const int c = blockIdx.y;
    // one thread per position of the pooled plane, blocks of 256 positions from row 1 on - the forward kernel's blocks,
    // so that one tile list serves both
    const int qp = Wpo + block_x * 256 + threadIdx.x;
    if (qp >= (Ho + 1) * Wpo) return;
    const int yo = qp / Wpo - 1, xo = qp - (yo + 1) * Wpo - 1;
    if (xo < 0 || xo >= Wo) return;
    const size_t qo = (size_t)c * plane_o + (yo + 1) * Wpo + xo + 1;
    const float pm = pooled[qo];
    const float g = (pm > 0.f) ? dpooled[qo] : 0.f;  // max <= 0 -> every ReLU gate in the window is closed
    const size_t base = (size_t)c * plane + (2 * yo + 1) * Wp + 2 * xo + 1;
    const float v00 = act[base], v01 = act[base + 1], v10 = act[base + Wp];   // the 4th wins by elimination
    // The first maximum in row-major order gets the gradient (ATen max_pool2d_with_indices: strict '>' scan).
    // NOTE on exact ties: a constant image region (background pixels, the all-zero initial texture) makes every
    // window an exact 4-way tie in exact arithmetic; which copy is largest in fp32 then depends on last-bit
    // summation-order differences between tiles (here: whole vs K-split tiles; in the reference: on its conv
    // backend). A tolerance-based rule was tried and rejected: it fixes the degenerate case but flips genuine
    // near-ties away from the reference's choice (tests/test_engine_gpu.py would fail).
    const bool s00 = v00 == pm;
    const bool s01 = !s00 && v01 == pm;
    const bool s10 = !s00 && !s01 && v10 == pm;
    const bool s11 = !s00 && !s01 && !s10;
    dact[base] = s00 ? g : 0.f;
    dact[base + 1] = s01 ? g : 0.f;
    dact[base + Wp] = s10 ? g : 0.f;
    dact[base + Wp + 1] = s11 ? g : 0.f;
}

// Per-level launches of these memory-bound kernels are latency-bound on the small UV levels: the grouped forms take
// the levels of a view in ONE launch (blockIdx.x runs over the concatenated block ranges of the problems).
struct PlaneProblem {
    const float* a;
    const float* b;
    const float* c;
    float* out;
    int H, W;
};
struct PlaneGroup {
    PlaneProblem p[SM_MAX_GROUP];
    int block_begin[SM_MAX_GROUP + 1];
    int n;
    // optional list of the ACTIVE blocks, entry = (problem << 24) | block-in-problem (as the convs' tile lists: blocks
    // whose positions cannot influence the loss are simply absent); NULL = every block
    const int* tile_list;
};
__device__ __forceinline__ int locate_problem(const PlaneGroup& g, int bx, int& local) {
    if (g.tile_list) {
        const int e = g.tile_list[bx];
        local = e & 0xFFFFFF;
        return e >> 24;
    }
    int k = 0;
#pragma unroll
    for (int i = 1; i < SM_MAX_GROUP; ++i)
        if (i < g.n && bx >= g.block_begin[i]) k = i;
    local = bx - g.block_begin[k];
    return k;
}

__device__ __forceinline__ PlaneProblem pick_problem(const PlaneGroup& g, int k) {
    PlaneProblem P = g.p[0];
#pragma unroll
    for (int i = 1; i < SM_MAX_GROUP; ++i)
        if (i == k) P = g.p[i];
    return P;
}

__global__ __launch_bounds__(256) void conv3x3_dgrad_c3_kernel(PlaneGroup g, const float* __restrict__ wd, int Cin) {
    int bx;
    // a block is ~one image row of the 64 gradient planes and reads the rows above and below it as well: blocks are
    // dealt to the XCDs in contiguous runs (SM_DGRAD_C3_XCD), so that a row is fetched into one L2 instead of three
#if SM_DGRAD_C3_XCD
    const int b = xcd_linear(blockIdx.x, gridDim.x);
#else
    const int b = blockIdx.x;
#endif
    const PlaneProblem P = pick_problem(g, locate_problem(g, b, bx));
    dgrad_c3_body(P.a, wd, P.out, Cin, P.H, P.W, row_stride(P.W), plane_size(P.H, P.W), bx);
}

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(PlaneGroup g) {
    int bx;
    const PlaneProblem P = pick_problem(g, locate_problem(g, blockIdx.x, bx));
    const int Ho = P.H / 2, Wo = P.W / 2;
    maxpool_fwd_body(P.a, P.out, P.H, P.W, row_stride(P.W), plane_size(P.H, P.W), Ho, Wo, row_stride(Wo),
                     plane_size(Ho, Wo), bx);
}

__global__ __launch_bounds__(256) void maxpool_fwd_codes_kernel(PlaneGroup g) {
    int bx;
    const PlaneProblem P = pick_problem(g, locate_problem(g, blockIdx.x, bx));
    const int Ho = P.H / 2, Wo = P.W / 2;
    // (the code image travels in the problem's otherwise unused `c` slot)
    maxpool_fwd_codes_body(P.a, P.out, reinterpret_cast<uint32_t*>(const_cast<float*>(P.c)), P.H, P.W, row_stride(P.W),
                           plane_size(P.H, P.W), Ho, Wo, row_stride(Wo), plane_size(Ho, Wo), bx);
}

__global__ __launch_bounds__(256) void maxpool_bwd_relu_kernel(PlaneGroup g) {
    int bx;
    const PlaneProblem P = pick_problem(g, locate_problem(g, blockIdx.x, bx));
    const int Ho = P.H / 2, Wo = P.W / 2;
    maxpool_bwd_relu_body(P.a, P.b, P.c, P.out, P.H, P.W, row_stride(P.W), plane_size(P.H, P.W), Ho, Wo, row_stride(Wo),
                          plane_size(Ho, Wo), bx);
}

}  // namespace sm

extern "C" {

int sm_fmap_row_stride(int W) { return sm::row_stride(W); }
int sm_fmap_plane(int H, int W) { return sm::plane_size(H, W); }
int sm_abi_version(void) { return 11; }

}  // extern "C"
// SM_LIST_QUADS: the list holds vertical quads of segments and the launch has 64 output channels - the resident-input
// kernel (conv_split_kernel.h, RES). fp32 planes only.
template <int FLAGS, bool UNPOOL>
static int launch_conv_resident(sm::ConvArgs& a, int n_list, hipStream_t s) {
    a.ws = nullptr;
    return sm::launch_conv<64, 128, 16, 2, 2, FLAGS, true, UNPOOL, true>(a, n_list, 0, s);
}
static int conv_dispatch_resident(sm::ConvArgs& a, int n_list, int flags, bool unpool, hipStream_t s) {
    if (a.Cout != 64 || a.Cin_pad % 64 != 0 || a.tile_list == nullptr) return (int)hipErrorInvalidValue;
    if (unpool) {
        switch (flags) {
            case SM_EPI_RELU_MASK | SM_EPI_GRAM: return launch_conv_resident<SM_EPI_RELU_MASK | SM_EPI_GRAM, true>(a, n_list, s);
            case SM_EPI_RELU_MASK: return launch_conv_resident<SM_EPI_RELU_MASK, true>(a, n_list, s);
            case SM_EPI_RELU_MASK | SM_EPI_ADD: return launch_conv_resident<SM_EPI_RELU_MASK | SM_EPI_ADD, true>(a, n_list, s);
            default: return (int)hipErrorInvalidValue;
        }
    }
    switch (flags) {
        case SM_EPI_BIAS_RELU | SM_EPI_POOL: return launch_conv_resident<SM_EPI_BIAS_RELU | SM_EPI_POOL, false>(a, n_list, s);
        case SM_EPI_BIAS_RELU: return launch_conv_resident<SM_EPI_BIAS_RELU, false>(a, n_list, s);
        case 0: return launch_conv_resident<0, false>(a, n_list, s);
        case SM_EPI_RELU_MASK: return launch_conv_resident<SM_EPI_RELU_MASK, false>(a, n_list, s);
        case SM_EPI_RELU_MASK | SM_EPI_ADD: return launch_conv_resident<SM_EPI_RELU_MASK | SM_EPI_ADD, false>(a, n_list, s);
        default: return (int)hipErrorInvalidValue;
    }
}
static int conv_dispatch_flags_split2_t(sm::ConvArgs& a, int n_list, int flags, size_t ws_floats, bool unpool, hipStream_t s) {
    if (unpool) {   // the data gradients below a max-pool: gated by the pool input's own producer conv
        if (flags == (SM_EPI_RELU_MASK | SM_EPI_GRAM)) {   // + the Gram backward of the 64-channel output layer; whole tiles only
            a.ws = nullptr;
            if (a.Cout == 64)
                return sm::launch_conv<64, 256, 16, SM_SPLIT2_W64GM, 4 / SM_SPLIT2_W64GM, SM_EPI_RELU_MASK | SM_EPI_GRAM, true, true>(a, n_list, 0, s);
            if (a.Cout == 128)   // (four waves of 32 rows: the 128-row tile holds all channels of its positions)
                return sm::launch_conv<128, 128, 16, 4, 1, SM_EPI_RELU_MASK | SM_EPI_GRAM, true, true>(a, n_list, 0, s);
            return (int)hipErrorInvalidValue;
        }
        switch (flags) {
            case SM_EPI_RELU_MASK: return sm::dispatch_conv_split2<SM_EPI_RELU_MASK, true>(a, n_list, ws_floats, s);
            case SM_EPI_RELU_MASK | SM_EPI_ADD: return sm::dispatch_conv_split2<SM_EPI_RELU_MASK | SM_EPI_ADD, true>(a, n_list, ws_floats, s);
            default: return (int)hipErrorInvalidValue;
        }
    }
    switch (flags) {
        case SM_EPI_BIAS_RELU: return sm::dispatch_conv_split2<SM_EPI_BIAS_RELU, false>(a, n_list, ws_floats, s);
        case SM_EPI_BIAS_RELU | SM_EPI_POOL: return sm::dispatch_conv_split2<SM_EPI_BIAS_RELU | SM_EPI_POOL, false>(a, n_list, ws_floats, s);
        case 0: return sm::dispatch_conv_split2<0, false>(a, n_list, ws_floats, s);
        case SM_EPI_RELU_MASK: return sm::dispatch_conv_split2<SM_EPI_RELU_MASK, false>(a, n_list, ws_floats, s);
        case SM_EPI_RELU_MASK | SM_EPI_ADD: return sm::dispatch_conv_split2<SM_EPI_RELU_MASK | SM_EPI_ADD, false>(a, n_list, ws_floats, s);
        case SM_EPI_ADD: return sm::dispatch_conv_split2<SM_EPI_ADD>(a, n_list, ws_floats, s);
        default: return (int)hipErrorInvalidValue;
    }
}
extern "C" {
static int conv_dispatch_flags_split2(sm::ConvArgs& a, int n_list, int flags, size_t ws_floats, bool unpool, hipStream_t s) {
    if (flags & SM_LIST_QUADS) return conv_dispatch_resident(a, n_list, flags & ~SM_LIST_QUADS, unpool, s);
    return conv_dispatch_flags_split2_t(a, n_list, flags, ws_floats, unpool, s);
}

static int conv_dispatch_flags(sm::ConvArgs& a, int n_list, int flags, size_t ws_floats, hipStream_t s) {
    switch (flags) {
        case SM_EPI_BIAS_RELU: return sm::dispatch_conv<SM_EPI_BIAS_RELU>(a, n_list, ws_floats, s);
        case 0: return sm::dispatch_conv<0>(a, n_list, ws_floats, s);
        case SM_EPI_RELU_MASK: return sm::dispatch_conv<SM_EPI_RELU_MASK>(a, n_list, ws_floats, s);
        case SM_EPI_RELU_MASK | SM_EPI_ADD: return sm::dispatch_conv<SM_EPI_RELU_MASK | SM_EPI_ADD>(a, n_list, ws_floats, s);
        case SM_EPI_ADD: return sm::dispatch_conv<SM_EPI_ADD>(a, n_list, ws_floats, s);
        default: return (int)hipErrorInvalidValue;
    }
}

int sm_amax_floats(void) { return SM_AMAX_SLOTS * SM_AMAX_STRIDE; }

int sm_fmap_amax(const float* planes, int C, int H, int W, float* amax_out, void* stream) {
    if (C < 1 || amax_out == nullptr) return (int)hipErrorInvalidValue;
    const int Wp = sm::row_stride(W), plane = sm::plane_size(H, W), n = H * Wp;
    hipLaunchKernelGGL(sm::fmap_amax_kernel, dim3(std::min(8, (n / 4 + 255) / 256), C), dim3(256), 0, (hipStream_t)stream,
                       planes, plane, Wp, (H + 1) * Wp, amax_out);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_conv_tile_positions(int Cin_pad, int Cout) { return Cin_pad == 4 ? 1024 : (Cout % 128 != 0 ? 256 : 128); }
int sm_conv_split2_tile_positions(int Cout) { return (SM_SPLIT2_BN256 && Cout % 128 != 0) ? 256 : 128; }

int sm_conv3x3_grouped(const sm_conv_problem* problems, int n_problems, const float* wt, const float* bias,
                       int Cin_pad, int Cout, int flags, const int32_t* tile_list, int n_list, float* ws,
                       size_t ws_floats, float* amax_out, void* stream) {
    if (n_problems < 1 || n_problems > sm::SM_MAX_GROUP) return (int)hipErrorInvalidValue;
    if (Cout % 64 != 0 || Cin_pad % 4 != 0 || (Cin_pad > 4 && Cin_pad % 8 != 0)) return (int)hipErrorInvalidValue;
    sm::ConvArgs a{};
    for (int g = 0; g < n_problems; ++g)
    {
        if (problems[g].unpool_code != nullptr) return (int)hipErrorInvalidValue;   // fp16x2 kernel only
        a.p[g] = sm::ConvProblem{problems[g].in, problems[g].out, problems[g].gate, nullptr, problems[g].H, problems[g].W,
                                 sm::row_stride(problems[g].W), sm::plane_size(problems[g].H, problems[g].W)};
    }
    a.n_problems = n_problems;
    a.wt = wt;
    a.bias = bias;
    a.Cin_pad = Cin_pad;
    a.Cout = Cout;
    a.ws = ws;
    a.splits = 1;
    a.tile_list = tile_list;
    a.amax_out = amax_out;
    return conv_dispatch_flags(a, n_list, flags, ws_floats, (hipStream_t)stream);
}

int sm_conv3x3_grouped_split2(const sm_conv_problem* problems, int n_problems, const uint16_t* wt2, float w_scale_inv,
                              const float* bias, int Cin, int Cout, int flags, const int32_t* tile_list, int n_list,
                              float* ws, size_t ws_floats, const float* amax_in, float* amax_out, void* stream) {
    if (n_problems < 1 || n_problems > sm::SM_MAX_GROUP) return (int)hipErrorInvalidValue;
    if (Cout % 64 != 0 || Cin % 16 != 0 || amax_in == nullptr || !(w_scale_inv > 0.f)) return (int)hipErrorInvalidValue;
    sm::ConvArgs a{};
    int unpool = 0;   // problems whose input is a pooled gradient + argmax codes: all of a launch or none
    for (int g = 0; g < n_problems; ++g)
    {
        unpool += problems[g].unpool_code != nullptr;
        a.p[g] = sm::ConvProblem{problems[g].in, problems[g].out, problems[g].gate, problems[g].unpool_code,
                                 problems[g].H, problems[g].W, sm::row_stride(problems[g].W),
                                 sm::plane_size(problems[g].H, problems[g].W), problems[g].pool_out, problems[g].pool_code,
                                 reinterpret_cast<const sm::f32x4*>(problems[g].gram_ws), problems[g].gram_mask0,
                                 problems[g].gram_mask1, problems[g].gram_amax_feat, problems[g].gram_amax_d};
        if ((flags & SM_EPI_GRAM) != 0 && (problems[g].gram_ws == nullptr || problems[g].gram_mask0 == nullptr ||
                                          problems[g].gram_amax_feat == nullptr || problems[g].gram_amax_d == nullptr ||
                                          problems[g].gate == nullptr || problems[g].unpool_code == nullptr))
            return (int)hipErrorInvalidValue;
        if ((flags & SM_EPI_POOL) != 0 && (problems[g].pool_out == nullptr || problems[g].pool_code == nullptr ||
                                          problems[g].H < 2 || problems[g].W < 2))
            return (int)hipErrorInvalidValue;
    }
    // the pooling epilogue needs the segment PAIRS of a list (sm_cover_segments, pair_w) and 8-channel code groups
    if ((flags & SM_EPI_POOL) != 0 && (tile_list == nullptr || bias == nullptr || unpool != 0)) return (int)hipErrorInvalidValue;
    a.n_problems = n_problems;
    a.wt = reinterpret_cast<const float*>(wt2);
    a.bias = bias;
    a.Cin_pad = Cin;
    a.Cout = Cout;
    a.ws = ws;
    a.splits = 1;
    a.tile_list = tile_list;
    a.amax_in = amax_in;
    a.amax_out = amax_out;
    a.w_scale_inv = w_scale_inv;
    if (unpool != 0 && unpool != n_problems) return (int)hipErrorInvalidValue;
    return conv_dispatch_flags_split2(a, n_list, flags, ws_floats, unpool != 0, (hipStream_t)stream);
}

int sm_conv3x3(const float* in, const float* wt, const float* bias, float* out, const float* gate, int Cin_pad,
               int Cout, int H, int W, int flags, float* ws, size_t ws_floats, void* stream) {
    sm_conv_problem p{in, out, gate, H, W};
    return sm_conv3x3_grouped(&p, 1, wt, bias, Cin_pad, Cout, flags, nullptr, 0, ws, ws_floats, nullptr, stream);
}

static int make_plane_group(sm::PlaneGroup& g, const sm_plane_problem* p, int n, int kind, const int32_t* tile_list = nullptr) {
    if (n < 1 || n > sm::SM_MAX_GROUP) return (int)hipErrorInvalidValue;
    g.n = n;
    g.tile_list = tile_list;
    g.block_begin[0] = 0;
    for (int i = 0; i < n; ++i) {
        g.p[i] = sm::PlaneProblem{p[i].a, p[i].b, p[i].c, p[i].out, p[i].H, p[i].W};
        const int H = p[i].H, W = p[i].W, Ho = H / 2, Wo = W / 2;
        if (kind != 0 && (Ho < 1 || Wo < 1)) return (int)hipErrorInvalidValue;
        int work;   // threads of the problem
        if (kind == 0) work = H * sm::row_stride(W) / 4;            // dgrad_c3: four positions per thread
        else if (kind == 1) work = Ho * sm::row_stride(Wo);          // pool forward: output positions
        else work = Ho * sm::row_stride(Wo);                         // pool backward: the same blocks
        g.block_begin[i + 1] = g.block_begin[i] + (work + 255) / 256;
    }
    return 0;
}

int sm_conv3x3_dgrad_c3_grouped(const sm_plane_problem* problems, int n, const float* wd, int Cin, void* stream) {
    return sm_conv3x3_dgrad_c3_tiles(problems, n, wd, Cin, nullptr, 0, stream);
}
int sm_maxpool2x2_fwd_grouped(const sm_plane_problem* problems, int n, int C, void* stream) {
    return sm_maxpool2x2_fwd_tiles(problems, n, C, nullptr, 0, stream);
}
int sm_maxpool2x2_bwd_relu_grouped(const sm_plane_problem* problems, int n, int C, void* stream) {
    return sm_maxpool2x2_bwd_relu_tiles(problems, n, C, nullptr, 0, stream);
}

int sm_plane_tile_positions(int kind) { return kind == 0 ? 1024 : 256; }

int sm_conv3x3_dgrad_c3_tiles(const sm_plane_problem* problems, int n, const float* wd, int Cin, const int32_t* tile_list,
                              int n_list, void* stream) {
    sm::PlaneGroup g;
    if (int e = make_plane_group(g, problems, n, 0, tile_list)) return e;
    const int blocks = tile_list ? n_list : g.block_begin[n];
    if (blocks == 0) return 0;
    // (a one-position-per-thread form with four times the blocks for small grids measured SLOWER on the single-level
    // workload: 45 against 39 us - nine 4-byte loads per channel instead of 1.5 sixteen-byte ones)
    hipLaunchKernelGGL(sm::conv3x3_dgrad_c3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, wd, Cin);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_maxpool2x2_fwd_tiles(const sm_plane_problem* problems, int n, int C, const int32_t* tile_list, int n_list,
                            void* stream) {
    return sm_maxpool2x2_fwd_codes_tiles(problems, nullptr, n, C, tile_list, n_list, stream);
}

int sm_maxpool2x2_fwd_codes_tiles(const sm_plane_problem* problems, uint32_t* const* codes, int n, int C,
                                  const int32_t* tile_list, int n_list, void* stream) {
    sm::PlaneGroup g;
    if (int e = make_plane_group(g, problems, n, 1, tile_list)) return e;
    if (codes != nullptr) {
        if (C % 8 != 0) return (int)hipErrorInvalidValue;
        for (int i = 0; i < n; ++i) {
            if (codes[i] == nullptr) return (int)hipErrorInvalidValue;
            g.p[i].c = reinterpret_cast<const float*>(codes[i]);
        }
        const int blocks = tile_list ? n_list : g.block_begin[n];
        if (blocks == 0) return 0;
        hipLaunchKernelGGL(sm::maxpool_fwd_codes_kernel, dim3(blocks, C / 8), dim3(256), 0, (hipStream_t)stream, g);
        SM_LAUNCH_CHECK();
        return 0;
    }
    const int blocks = tile_list ? n_list : g.block_begin[n];
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(sm::maxpool_fwd_kernel, dim3(blocks, C), dim3(256), 0, (hipStream_t)stream, g);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_maxpool2x2_bwd_relu_tiles(const sm_plane_problem* problems, int n, int C, const int32_t* tile_list, int n_list,
                                 void* stream) {
    sm::PlaneGroup g;
    if (int e = make_plane_group(g, problems, n, 2, tile_list)) return e;
    const int blocks = tile_list ? n_list : g.block_begin[n];
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(sm::maxpool_bwd_relu_kernel, dim3(blocks, C), dim3(256), 0, (hipStream_t)stream, g);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_conv3x3_dgrad_c3(const float* dz, const float* wd, float* out, int Cin, int H, int W, void* stream) {
    sm_plane_problem p{dz, nullptr, nullptr, out, H, W};
    return sm_conv3x3_dgrad_c3_grouped(&p, 1, wd, Cin, stream);
}

int sm_maxpool2x2_fwd(const float* in, float* out, int C, int H, int W, void* stream) {
    sm_plane_problem p{in, nullptr, nullptr, out, H, W};
    return sm_maxpool2x2_fwd_grouped(&p, 1, C, stream);
}

int sm_maxpool2x2_bwd_relu(const float* act, const float* pooled, const float* dpooled, float* dact, int C, int H,
                           int W, void* stream) {
    sm_plane_problem p{act, pooled, dpooled, dact, H, W};
    return sm_maxpool2x2_bwd_relu_grouped(&p, 1, C, stream);
}

}  // extern "C"

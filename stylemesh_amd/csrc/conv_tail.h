// The K-split TAIL of a conv launch: tiles of the last, partially filled round are split along K into `splits` units that
// store raw partial tiles ("slabs") to ConvArgs::ws; the tile's output = epilogue(sum of its slabs in split order).
//
// The reduction is a SECOND PASS (conv_tail_epilogue_kernel / conv_tail_pool_kernel): a launch of its own behind the conv
// launch, 16 blocks per tile, 7.5 us per layer launch included. Round 6 built the alternative the r5 verdict asked for - the
// K-split unit that ARRIVES LAST at its tile (an atomic counter per tile) reduces the tile before it exits, no second launch -
// in three versions (agent-scope fences; device-scope sc1 slab stores / loads; all loads of a split in flight), bit-identical
// to this pass, and measured it in situ: 75 / 53 / 45 us per one-level conv layer against 35 (profiles/r06/tail_fused_ab.txt;
// git 8b07e84 has the code). One block owning a tile's whole latency chain behind its main loop - write-through stores,
// counter, loads that by construction miss this XCD's L2, gate loads, stores - is slower than a kernel boundary plus a 16x
// wider pass: the XCDs' L2s are only coherent through write-back / invalidate or bypass. The second pass stays.
#pragma once
#include "conv_common.h"

namespace sm {

__device__ __forceinline__ f32x4 slab_ld4(const float* ws, size_t idx) { return *reinterpret_cast<const f32x4*>(ws + idx); }
__device__ __forceinline__ f32x2 slab_ld2(const float* ws, size_t idx) { return *reinterpret_cast<const f32x2*>(ws + idx); }

// slice y (of BM * BN / 1024) of tail tile `tail`: one float4 of the BM x BN slab per thread - element e = y * 256 + tid of
// the slab's float4s (row e / (BN / 4), columns 4 (e % (BN / 4)) ...: float index 4 e). `v`: the sum of the tile's slabs there.
template <int BM, int BN, int FLAGS>
__device__ __forceinline__ float conv_tail_epilogue_apply(const ConvArgs& a, int tail, int y, int tid, f32x4 v) {
    const int tile = a.n_whole + tail;
    const int m_tile = tile / a.n_tiles, n_glob = tile - m_tile * a.n_tiles;
    const int e_ = y * 256 + tid;
    const int row = e_ / (BN / 4), c4 = (e_ - row * (BN / 4)) * 4;   // this thread's float4 of the BM x BN slab
    ConvProblem P = a.p[0];
    int n_tile = n_glob;
    int q = 0;
    bool alive = true;
    if (a.tile_list && a.list_segments) {
        // the split kernels' lists: BN / 32 entries per tile, (problem << 24) | first position of a segment, 0xFFFFFF = padding
        const int* e = a.tile_list + (size_t)n_glob * (BN / 32);
        const int gsel = e[0] >> 24;
#pragma unroll
        for (int g = 1; g < SM_MAX_GROUP; ++g)
            if (g == gsel) P = a.p[g];
        const int sg = e[c4 >> 5] & 0xFFFFFF;
        alive = sg != 0xFFFFFF;
        q = sg + (c4 & 31);
    } else {
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
        q = P.Wp + n_tile * BN + c4;
    }
    const int m0 = m_tile * BM, q_end = (P.H + 1) * P.Wp;
    float m = 0.f;            // max |output| of this thread (all lanes stay active for the wave reduction of the caller)
    if (alive && q < q_end) { // q_end and q are multiples of 4
        const int co = m0 + row;
        const size_t o = (size_t)co * P.plane + q;
        f32x4 prev;
        bool open[4] = {true, true, true, true};
        if (FLAGS & SM_EPI_ADD) prev = *reinterpret_cast<const f32x4*>(P.out + o);
        if (FLAGS & SM_EPI_RELU_MASK) {
            const f32x4 gw = *reinterpret_cast<const f32x4*>(P.gate + o);
#pragma unroll
            for (int j = 0; j < 4; ++j) open[j] = gw[j] > 0.f;
        }
        const float b = (FLAGS & SM_EPI_BIAS_RELU) ? a.bias[co] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x = v[j];
            if (FLAGS & SM_EPI_BIAS_RELU) x = fmaxf(x + b, 0.f);
            if (FLAGS & SM_EPI_ADD) x += prev[j];
            if (FLAGS & SM_EPI_RELU_MASK) x = open[j] ? x : 0.f;
            v[j] = interior(q + j, P.H, P.W, P.Wp) ? x : 0.f;
            m = fmaxf(m, fabsf(v[j]));
        }
        *reinterpret_cast<f32x4*>(P.out + o) = v;
    }
    return m;
}

// Tail second pass: one block per (tail tile, slice); out = epilogue(sum over its K-splits), 4 positions per thread.
// (many small blocks: the pass is latency-bound)
template <int BM, int BN, int FLAGS>
__global__ __launch_bounds__(256) void conv_tail_epilogue_kernel(ConvArgs a) {
    const float amax_seen = amax_peek(a.amax_out);   // beside the slab loads, not behind the stores
    const size_t wt = (size_t)blockIdx.x * a.splits * (BM * BN) + (size_t)(blockIdx.y * 256 + threadIdx.x) * 4;
    f32x4 v = slab_ld4(a.ws, wt);
    for (int s = 1; s < a.splits; ++s) v += slab_ld4(a.ws, wt + (size_t)s * (BM * BN));
    const float m = conv_tail_epilogue_apply<BM, BN, FLAGS>(a, blockIdx.x, blockIdx.y, threadIdx.x, v);
    record_amax(a.amax_out, m, amax_seen);
}

// Tail of a forward conv with SM_EPI_POOL (split kernels; the tile's segments are vertical pairs, see
// conv_split_kernel.h): (tail tile, 8-channel group y, two segment pairs z); thread = (channel of the group, one of
// the 32 pooling windows): sums the K-splits of its four elements, bias + ReLU, 2x2 maximum + argmax code; the eight
// channels' nibbles meet through three lane exchanges. Writes the pooled map and the code image only.
// the problem a (segment-list) tile belongs to: entry 0 of the tile carries it
template <int BN>
__device__ __forceinline__ ConvProblem conv_tail_problem(const ConvArgs& a, int tail) {
    const int tile = a.n_whole + tail;
    const int n_glob = tile - (tile / a.n_tiles) * a.n_tiles;
    const int gsel = a.tile_list[(size_t)n_glob * (BN / 32)] >> 24;
    ConvProblem P = a.p[0];
#pragma unroll
    for (int g = 1; g < SM_MAX_GROUP; ++g)
        if (g == gsel) P = a.p[g];
    return P;
}
// index into ws of the upper-row pair of window (y, z, tid) in split 0's slab (the lower row: + 32)
template <int BM, int BN>
__device__ __forceinline__ size_t conv_tail_pool_index(const ConvArgs& a, int tail, int y, int z, int tid) {
    const int c = tid & 7, w = tid >> 3;
    return (size_t)tail * a.splits * (BM * BN) + (size_t)(y * 8 + c) * BN + (z * 2 + (w >> 4)) * 64 + 2 * (w & 15);
}
// t / b: the sums of the tile's slabs at the window's upper / lower row pair
template <int BM, int BN>
__device__ __forceinline__ float conv_tail_pool_apply(const ConvArgs& a, const ConvProblem& P, int tail, int y, int z, int tid,
                                                      f32x2 t, f32x2 b) {
    const int tile = a.n_whole + tail;
    const int m_tile = tile / a.n_tiles, n_glob = tile - m_tile * a.n_tiles;
    const int c = tid & 7, w = tid >> 3;                         // channel of the group, window 0..31 of the block
    const int pair = z * 2 + (w >> 4), X = w & 15;               // segment pair of the tile, window of the pair
    const int* e = a.tile_list + (size_t)n_glob * (BN / 32);
    const int sg = e[2 * pair] & 0xFFFFFF;
    float m = 0.f;
    unsigned code = 4u;
    bool ok = false;
    int qo = 0;
    const int Ho = P.H >> 1, Wo = P.W >> 1, Wpo = row_stride(Wo), plane_o = plane_size(Ho, Wo);
    const int row = y * 8 + c;
    if (sg != 0xFFFFFF) {
        const int q = sg + 2 * X;
        const int yy = q / P.Wp - 1, xx = q - (yy + 1) * P.Wp - 1;
        ok = ((yy | xx) & 1) == 0 && (unsigned)yy < (unsigned)(2 * Ho) && (unsigned)xx < (unsigned)(2 * Wo);
        qo = ((yy >> 1) + 1) * Wpo + (xx >> 1) + 1;
        const float bv = a.bias[m_tile * BM + row];
        const float v00 = fmaxf(t[0] + bv, 0.f), v01 = fmaxf(t[1] + bv, 0.f), v10 = fmaxf(b[0] + bv, 0.f), v11 = fmaxf(b[1] + bv, 0.f);
        m = v00;
        code = 0u;
        if (v01 > m) { m = v01; code = 1u; }
        if (v10 > m) { m = v10; code = 2u; }
        if (v11 > m) { m = v11; code = 3u; }
        if (!(m > 0.f)) code = 4u;
        if (ok) P.pool_out[(size_t)(m_tile * BM + row) * plane_o + qo] = m;
    }
    unsigned word = code << (4 * c);
    word |= (unsigned)__shfl_xor((int)word, 1, 64);
    word |= (unsigned)__shfl_xor((int)word, 2, 64);
    word |= (unsigned)__shfl_xor((int)word, 4, 64);
    if (ok && c == 0) P.pool_code[(size_t)((m_tile * BM) / 8 + y) * plane_o + qo] = word;
    return ok ? m : 0.f;
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_tail_pool_kernel(ConvArgs a) {
    const float amax_seen = amax_peek(a.amax_out);
    const ConvProblem P = conv_tail_problem<BN>(a, blockIdx.x);
    const size_t wt = conv_tail_pool_index<BM, BN>(a, blockIdx.x, blockIdx.y, blockIdx.z, threadIdx.x);
    f32x2 t = slab_ld2(a.ws, wt), b = slab_ld2(a.ws, wt + 32);
    for (int s = 1; s < a.splits; ++s) {
        t += slab_ld2(a.ws, wt + (size_t)s * (BM * BN));
        b += slab_ld2(a.ws, wt + (size_t)s * (BM * BN) + 32);
    }
    const float m = conv_tail_pool_apply<BM, BN>(a, P, blockIdx.x, blockIdx.y, blockIdx.z, threadIdx.x, t, b);
    record_amax(a.amax_out, m, amax_seen);
}

}  // namespace sm

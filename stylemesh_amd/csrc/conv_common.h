// Launch arguments shared by the conv kernels (conv.hip: exact fp32 MFMA; conv_split_kernel.h: fp16x2-split MFMA).
#pragma once
#include "common.h"

namespace sm {

constexpr int SM_MAX_GROUP = 8;

// One feature-map problem of a grouped launch (same weights, different planes: the UV levels of a view).
struct ConvProblem {
    const float* in;
    float* out;
    const float* gate;
    // fp16x2 kernel, optional ("unpool" input): `in` is then the gradient of the 2x2 max-pooled map [Cin][plane(H/2,
    // W/2)] and `code` the pool's argmax codes [Cin / 8][plane(H/2, W/2)], one dword per position and 8-channel group,
    // nibble c = code of channel 8 g + c (0..3 = window element dy * 2 + dx that holds the first maximum, 4 = maximum
    // <= 0: no gradient through the ReLU); the operand the kernel stages is the gradient w.r.t. the pool's PRE-ReLU
    // input, computed on the fly: in[q/2] where code[q/2] == parity(q), else 0.
    const uint32_t* code;
    int H, W, Wp, plane;
    // fp16x2 kernel, flag SM_EPI_POOL (forward conv below a max-pool): the launch writes the 2x2-pooled map
    // [Cout][plane(H/2, W/2)] and the pool's argmax codes [Cout / 8][plane(H/2, W/2)] (same formats) INSTEAD of `out`.
    float* pool_out;
    uint32_t* pool_code;
    // fp16x2 kernel, flag SM_EPI_GRAM (data gradient whose output layer is a 64-channel STYLE layer): the launch adds
    // the masked Gram backward of that layer, sum_k m_k(q) (D_k F)(q), in its epilogue instead of reading it from `out`.
    // gram_p = the operand images of D0 / D1 (gram_d_pack_group_kernel; P1 = P0 + 6 C^2 / 16 units), gram_mask0 / 1 the
    // layer's mask planes (mask1 optional), gram_amax_feat / _d the bounds of F (= `gate`) and of D. F is `gate`.
    const f32x4* gram_p;
    const float* gram_mask0;
    const float* gram_mask1;
    const float* gram_amax_feat;
    const float* gram_amax_d;
};

struct ConvArgs {
    ConvProblem p[SM_MAX_GROUP];
    int tile_begin[SM_MAX_GROUP + 1];   // prefix sums of the problems' position-tile counts
    int n_problems;
    // optional compact list of ACTIVE position tiles, entry = (problem << 24) | tile-in-problem; NULL = all tiles.
    // Tiles that cannot influence the loss (outside the receptive-field-dilated level mask) are simply absent.
    // (the split kernels: BN / 32 entries per tile, (problem << 24) | 32-position SEGMENT, 0xFFFFFF = padding - any live
    // segments of ONE problem form a tile; list_segments = 1)
    const int* tile_list;
    int list_segments;
    const float* wt;
    const float* bias;
    int Cin_pad, Cout, n_tiles, m_tiles;   // n_tiles = position tiles of ALL problems
    // Work decomposition. The first n_whole tiles (a multiple of the CU count) are computed whole; the remaining
    // "tail" tiles - whose last, partially filled round would otherwise leave most CUs idle - are split along K
    // into `splits` units each, so the tail is made of many small units that spread over all CUs. Split units
    // store raw partial tiles to ws[(tail_tile * splits + split)][BM][BN]; conv_tail.h reduces them.
    float* ws;
    int n_whole;
    int splits;
    int chunks_per_split;
    // Operand scaling of the fp16x2 split kernel (conv_split_kernel.h). amax_in (device, optional): max |x| of
    // the input tensor(s), recorded by their producer; w_scale_inv: 1 / (power-of-two scale the host applied to the
    // weights). amax_out (device, optional, any kernel): this launch's max |output| is atomically max-ed into it (as
    // the bit pattern of a non-negative float) for the conv that consumes the output.
    const float* amax_in;
    float* amax_out;
    float w_scale_inv;
};

constexpr int SM_NUM_CU = 256;

// Block id -> (tile, split) of a conv launch. Whole tiles: XCD-aware order (neighbours in an XCD's share use the same
// weight slab and adjacent positions). Tail units (tile, K-split) are ordered (row of M-tiles, split, position tile)
// and dealt to the XCDs in contiguous runs of that order: the units that read the SAME slice of the weight image - an
// M-tile's rows of one K-range - run on one XCD (two at a run boundary), so a slice crosses the fabric into one L2
// instead of into all eight. (Round 2 dealt the splits of a tile round-robin: every XCD pulled the whole 9-19 MB image
// of a deep layer through its own L2 - 4.4 x the algorithmic HBM traffic on the single-level c2 workload, whose deep
// layers are all tail.) The tail is the range [n_whole, m_tiles * n_tiles) of the M-major tile order: a partial first
// row of M-tiles, then whole rows. split = -1 for whole tiles.
__device__ __forceinline__ void conv_unit(const ConvArgs& a, int bid, int& tile, int& split) {
    if (bid < a.n_whole) {
        tile = xcd_linear(bid, a.n_whole);
        split = -1;
        return;
    }
    const int S = a.splits;
    const int rem = a.m_tiles * a.n_tiles - a.n_whole;
    const int v = xcd_linear(bid - a.n_whole, rem * S);    // n_whole is a multiple of 8: XCD = (bid - n_whole) % 8
    const int m0 = a.n_whole / a.n_tiles, n0 = a.n_whole - m0 * a.n_tiles;
    const int c0 = min(rem, a.n_tiles - n0);               // tiles of the first (possibly partial) row
    if (v < c0 * S) {
        split = v / c0;
        tile = a.n_whole + (v - split * c0);
    } else {
        const int w = v - c0 * S;
        const int row = w / (S * a.n_tiles), inner = w - row * (S * a.n_tiles);
        split = inner / a.n_tiles;
        tile = (m0 + 1 + row) * a.n_tiles + (inner - split * a.n_tiles);
    }
}

}  // namespace sm

// Per-view constants: level masks, depth-interpolation weights, angle weights, layer-resolution masks, level
// factors, resized content targets; plus the dense <-> padded-planar image conversions.
//
// Reference code replaced (lukasHoel/stylemesh): erode / mask_depth / mask_interpolation_weight and the two
// gradient hooks' weights (model/model.py:195-254), the mask / factor part of calculate_pyramid
// (model/losses/content_and_style_losses.py:146-217) and the F.interpolate calls therein. The reference
// recomputes all of this every step; it only depends on the view, so it runs once per view here
// (RepeatingSampler feeds the same view 20-100 consecutive steps, data/abstract_dataset.py:498-512).
#include "common.h"

namespace sm {

// ATen upsample_nearest (legacy 'nearest'): src = min(floor(dst * (in / out)), in - 1), scale in fp32
__device__ __forceinline__ int nearest_src(int dst, int in_size, int out_size) {
    const float scale = (float)in_size / (float)out_size;
    return min((int)floorf((float)dst * scale), in_size - 1);
}

// ATen upsample_bilinear2d, align_corners=False: src = max(scale * (dst + 0.5) - 0.5, 0)
struct Lin {
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ Lin bilinear_src(int dst, int in_size, int out_size) {
    const float scale = (float)in_size / (float)out_size;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    Lin r;
    r.i0 = min((int)src, in_size - 1);
    r.i1 = r.i0 + ((r.i0 < in_size - 1) ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

template <typename F>
__device__ __forceinline__ float bilinear_at(F at, int y, int x, int h, int w, int H, int W) {
    const Lin ly = bilinear_src(y, h, H), lx = bilinear_src(x, w, W);
    return ly.l0 * (lx.l0 * at(ly.i0, lx.i0) + lx.l1 * at(ly.i0, lx.i1)) +
           ly.l1 * (lx.l0 * at(ly.i1, lx.i0) + lx.l1 * at(ly.i1, lx.i1));
}

// ---- level masks at the view resolution (model/model.py:204-239) -------------------------------------
__global__ __launch_bounds__(256) void level_masks_kernel(const int64_t* __restrict__ rounded,
                                                          const int64_t* __restrict__ other,
                                                          const float* __restrict__ interp_w,
                                                          const uint8_t* __restrict__ mask, int h, int w, int n_levels,
                                                          float* __restrict__ E, float* __restrict__ Wt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lvl = blockIdx.y;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    // erode(m) = m * [all 9 zero-padded neighbours are 1]
    bool all_any = true, all_r = true, all_o = true;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            bool r = false, o = false;
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
                const int j = yy * w + xx;
                const bool m = mask[j] != 0;
                r = m && rounded[j] == lvl;
                o = m && other[j] == lvl;
            }
            all_r &= r;
            all_o &= o;
            all_any &= (r || o);
        }
    const float wgt = interp_w[i];
    E[(size_t)lvl * h * w + i] = all_any ? 1.f : 0.f;
    Wt[(size_t)lvl * h * w + i] = (all_r ? wgt : 0.f) + (all_o ? (1.f - wgt) : 0.f);
}

// ---- per-level maps at the level resolution -----------------------------------------------------------
__global__ __launch_bounds__(256) void level_maps_kernel(const float* __restrict__ E, const float* __restrict__ Wt,
                                                         const float* __restrict__ angle_guidance,
                                                         const float* __restrict__ angle_deg, float thr, int h, int w,
                                                         int H, int W, float* __restrict__ M,
                                                         float* __restrict__ pixel_weight, uint8_t* __restrict__ passed,
                                                         float* m_sum) {
    __shared__ float red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float mval = 0.f;
    if (i < H * W) {
        const int y = i / W, x = i - y * W;
        const int sy = nearest_src(y, h, H), sx = nearest_src(x, w, W);
        mval = (E[sy * w + sx] > 0.f) ? 1.f : 0.f;
        M[i] = mval;
        if (pixel_weight) {
            float pw = 1.f;
            if (angle_guidance) pw = bilinear_at([&](int a, int b) { return angle_guidance[a * w + b]; }, y, x, h, w, H, W);
            if (Wt) pw *= Wt[sy * w + sx];
            pixel_weight[i] = pw;
        }
        if (passed) {
            bool p = true;
            if (angle_deg) p = bilinear_at([&](int a, int b) { return angle_deg[a * w + b]; }, y, x, h, w, H, W) < thr;
            passed[i] = p ? 1 : 0;
        }
    }
    float v = mval;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float s = red[0] + red[1] + red[2] + red[3];
        if (s != 0.f) atomicAdd(m_sum, s);
    }
}

// ---- layer-resolution masks (content_and_style_losses.py:172-174) --------------------------------------
__global__ __launch_bounds__(256) void layer_masks_kernel(const float* __restrict__ M, const uint8_t* __restrict__ passed,
                                                          int H, int W, int hl, int wl, int Wp, float* __restrict__ m_all,
                                                          float* __restrict__ m_pass, float* __restrict__ m_fail,
                                                          float* counts) {
    __shared__ float red[3][4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float a = 0.f, p = 0.f, f = 0.f;
    if (i < hl * wl) {
        const int y = i / wl, x = i - y * wl;
        const int j = nearest_src(y, H, hl) * W + nearest_src(x, W, wl);
        a = M[j];
        const bool ps = passed ? passed[j] != 0 : true;
        p = ps ? a : 0.f;
        f = ps ? 0.f : a;
        const int q = (y + 1) * Wp + x + 1;
        m_all[q] = a;
        if (m_pass) m_pass[q] = p;
        if (m_fail) m_fail[q] = f;
    }
    float v[3] = {a, p, f};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const float s = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        if (s != 0.f) atomicAdd(counts + threadIdx.x, s);
    }
}

struct FactorArgs {
    const float* counts[8];
    float* factors[8];
    float sizes[8];
    int n;
};
__global__ void level_factors_kernel(FactorArgs a) {
    if (threadIdx.x != 0) return;
    float means[8], s = 0.f;
    for (int i = 0; i < a.n; ++i) {
        means[i] = *a.counts[i] / a.sizes[i];  // torch.mean(mask_i)
        s += means[i];
    }
    for (int i = 0; i < a.n; ++i) *a.factors[i] = means[i] / s;
}

// ---- resizes ---------------------------------------------------------------------------------------------
// padded planar (h,w) -> padded planar (H,W), bilinear align_corners=False
__global__ __launch_bounds__(256) void fmap_resize_bilinear_kernel(const float* __restrict__ in, int h, int w, int wp,
                                                                   int plane_in, float* __restrict__ out, int H, int W,
                                                                   int Wp, int plane_out) {
    const int c = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i - y * W;
    const float* src = in + (size_t)c * plane_in;
    out[(size_t)c * plane_out + (y + 1) * Wp + x + 1] =
        bilinear_at([&](int a, int b) { return src[(a + 1) * wp + b + 1]; }, y, x, h, w, H, W);
}

// dense [C][h][w] -> padded planar (H,W); identity copy when sizes match
__global__ __launch_bounds__(256) void image_to_fmap_kernel(const float* __restrict__ in, int h, int w,
                                                            float* __restrict__ out, int H, int W, int Wp, int plane) {
    const int c = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i - y * W;
    const float* src = in + (size_t)c * h * w;
    float v;
    if (h == H && w == W) v = src[i];
    else v = bilinear_at([&](int a, int b) { return src[a * w + b]; }, y, x, h, w, H, W);
    out[(size_t)c * plane + (y + 1) * Wp + x + 1] = v;
}

__global__ __launch_bounds__(256) void fmap_to_image_kernel(const float* __restrict__ in, int H, int W, int Wp, int plane,
                                                            float* __restrict__ out) {
    const int c = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i - y * W;
    out[(size_t)c * H * W + i] = in[(size_t)c * plane + (y + 1) * Wp + x + 1];
}

// ---- dead-tile analysis (runtime/sparsity.py): which positions of a layer can reach the loss --------------
// need_src[y][x] = (through a conv: max over the 3x3 neighbourhood of need_out; through a 2x2 pool: need_out[y/2][x/2])
//                  OR (the layer is a loss layer: nearest-down-sampled level mask M)
__global__ __launch_bounds__(256) void need_step_kernel(const float* __restrict__ need_out, int ho, int wo, int mode,
                                                        const float* __restrict__ M, int H, int W,
                                                        float* __restrict__ need_src, int hs, int ws) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= hs * ws) return;
    const int y = i / ws, x = i - y * ws;
    float v = 0.f;
    if (mode == 1) {          // conv 3x3, same size
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = y + dy, xx = x + dx;
                if (yy >= 0 && yy < ho && xx >= 0 && xx < wo) v = fmaxf(v, need_out[yy * wo + xx]);
            }
    } else if (mode == 2) {   // 2x2 max-pool, floor size: the pooled pixel needs its whole window
        const int yy = y >> 1, xx = x >> 1;
        if (yy < ho && xx < wo) v = need_out[yy * wo + xx];
    }
    if (M) v = fmaxf(v, M[nearest_src(y, H, hs) * W + nearest_src(x, W, ws)]);
    need_src[i] = v > 0.f ? 1.f : 0.f;
}

// flags[t] = 1 if position tile t (bn consecutive positions q from row 1 of the padded plane) holds a needed pixel
__global__ __launch_bounds__(256) void tile_flags_kernel(const float* __restrict__ need, int h, int w, int Wp, int bn,
                                                         int n_tiles, uint8_t* __restrict__ flags) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= n_tiles) return;
    bool any = false;
    for (int j = lane; j < bn; j += 64) {
        const int q = t * bn + j;           // relative to row 1
        const int r = q / Wp, x = q - r * Wp;
        if (r < h && x >= 1 && x <= w) any |= need[r * w + x - 1] > 0.f;
    }
    if (__ballot(any) != 0ull && lane == 0) flags[t] = 1;
    else if (lane == 0) flags[t] = 0;
}

// ---------------------------------------------------------------------------------------------------
// Active lists of the split conv kernels: cover the needed positions of a plane with 32-position segments that may
// START at any multiple of 4 (not on a 32-position grid) and never overlap: greedy left to right over the flattened
// padded plane - the next segment starts at the first needed position not yet covered, rounded down to a multiple of 4
// (a data-gradient epilogue that adds into its output must never see a position twice, so segments are disjoint).
// Against segments on the aligned grid this drops another 3.6 % of the listed positions (5 bench views, FLOP-weighted:
// 0.659 -> 0.635 of dense; the exact need is 0.589).
//
// Round 4: the SAME greedy cover, computed in parallel. Round 3 packed a whole need map into LDS and walked it on ONE
// lane of one block per map: a chain of positions / 32 dependent LDS reads - 2.0 ms on average, 4.2 ms for a view's
// largest maps, on every view change, and a hard 160 KB limit on the plane size. The greedy has a tiny state: at a CHUNK
// boundary (2048 positions) all that matters is how far the last segment of the chunks before sticks out into this one,
// e = 0, 4, ..., 28 positions. So: (1) pack the need maps to bits in global memory (wave ballots, coalesced reads);
// (2) every chunk is walked for each of its 8 possible entry states, in parallel: (exit state, segment count) per entry;
// (3) one thread per map composes the chunk tables left to right - 400 table look-ups for the largest map instead of
// 25 000 dependent LDS reads - and leaves every chunk its true entry state and output offset; (4) every chunk is walked
// once more from its true entry state and writes its segments. Four short launches for all ~56 maps of a view, no limit
// on the plane size, the lists bit-identical to the one-lane walk (tests/test_round4_gpu.py restates it on the host).
// PAIR mode (a conv with the pooling epilogue) is the same pipeline with chunk = one row of the pooled need map: rows are
// independent there (entry state always 0), the scan is a plain prefix sum of the rows' counts.
// ---------------------------------------------------------------------------------------------------
#define SM_COVER_MAX 64
#define SM_COVER_CHUNK_WORDS 64                       // 2048 positions per chunk
struct CoverProblem {
    const float* need;     // [h][w] 0 / 1
    int32_t* starts;       // out: (tag << 24) | first position q of each segment (index into the padded plane)
    int32_t* count;        // out: number of segments
    int h, w, tag, cap;
    int pair_w;
    int quad;              // 1: QUAD mode (rows of the map are walked in groups: two pooled rows / four rows, header)
    int word_base;         // first word of this map's bits in the workspace
    int chunk_base;        // first chunk of this map in the workspace's table / state arrays
};                         // (56 bytes: 64 of them + three pointers stay below the 4 KB of kernel arguments)
struct CoverGroup {
    CoverProblem p[SM_COVER_MAX];
    uint32_t* bits;
    uint4* table;          // per chunk: 8 x uint16 = (segments << 3) | exit state / 4, one per entry state
    uint32_t* state;       // per chunk: (offset of its first segment << 3) | entry state / 4
};

__host__ __device__ inline int cover_row_words(int w) { return (w + 31) / 32 + 1; }   // pair mode: + a zero word behind a row
// words of a map's bit image (flat: the positions of rows 1 .. h of the padded plane + two zero tail words; pair: h rows
// of cover_row_words + one word the walk may read behind the last row) and its chunks (pair: one per pooled row)
// QUAD mode (round 5, the resident-input conv kernel): a chunk = a GROUP of rows of the need map - two pooled rows over a
// pooled need map (pair_w > 0), four rows otherwise - whose bits are OR-ed into one row image; the walk is row-local like
// the pair mode's, and every run becomes FOUR entries: the same 32 columns of image rows 4 Y .. 4 Y + 3.
__host__ __device__ inline int cover_group_rows(int pair_w, int quad) { return quad ? (pair_w > 0 ? 2 : 4) : 1; }
__host__ __device__ inline int cover_rows(int h, int pair_w, int quad) {
    const int G = cover_group_rows(pair_w, quad);
    return (h + G - 1) / G;
}
__host__ __device__ inline int cover_words(int h, int w, int pair_w, int quad = 0) {
    return (pair_w > 0 || quad) ? cover_rows(h, pair_w, quad) * cover_row_words(w) + 1 : (h * row_stride(w) + 31) / 32 + 2;
}
__host__ __device__ inline int cover_chunks(int h, int w, int pair_w, int quad = 0) {
    return (pair_w > 0 || quad) ? cover_rows(h, pair_w, quad)
                                : (h * row_stride(w) + SM_COVER_CHUNK_WORDS * 32 - 1) / (SM_COVER_CHUNK_WORDS * 32);
}

// (1) bits. One wave = 64 consecutive positions per ballot = two words; a block of 4 waves makes 64 words.
__global__ __launch_bounds__(256) void cover_pack_kernel(CoverGroup g) {
    const CoverProblem& P = g.p[blockIdx.y];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int first_word = blockIdx.x * 64 + wave * 16;
    const int n_words = cover_words(P.h, P.w, P.pair_w, P.quad);
    if (first_word >= n_words) return;
    const int Wp = row_stride(P.w), n_pos = P.h * Wp, rw = cover_row_words(P.w);
    const int G = cover_group_rows(P.pair_w, P.quad);
    for (int k = 0; k < 8; ++k) {
        const int wd = first_word + 2 * k;
        if (wd >= n_words) break;
        bool bit = false;
        if (P.pair_w > 0 || P.quad) {   // word wd = row (group) wd / rw, columns (wd % rw) * 32 ...; a row is a whole number of words
            const int wl = wd + (lane >> 5);
            const int Y = wl / rw, X = (wl - Y * rw) * 32 + (lane & 31);
            for (int r = 0; r < G; ++r)
                bit = bit || (Y * G + r < P.h && X < P.w && P.need[(size_t)(Y * G + r) * P.w + X] > 0.f);
        } else {
            const int i = wd * 32 + lane;
            if (i < n_pos) {
                const int r = i / Wp, x = i - r * Wp;
                bit = x >= 1 && x <= P.w && P.need[(size_t)r * P.w + x - 1] > 0.f;
            }
        }
        const unsigned long long m = __ballot(bit);
        if (lane == 0) {
            g.bits[P.word_base + wd] = (uint32_t)m;
            if (wd + 1 < n_words) g.bits[P.word_base + wd + 1] = (uint32_t)(m >> 32);
        }
    }
}

// The walk of one chunk from entry state e (positions relative to the chunk's first position): calls emit(st) for every
// segment whose first needed position lies in the chunk, returns the exit state. `words` holds the chunk's bits and at
// least one readable word behind them. FLAT mode: segment start = first needed position rounded down to 4, next cursor =
// start + 32. PAIR mode (n_bits = the row's width, align1): start = the first needed window itself, next cursor = + 16.
// (ROWQ: the flat QUAD mode - row-local like PAIR, start = the first needed position rounded down to even, next cursor = + 32)
template <bool PAIR, bool ROWQ = false, typename Emit>
__device__ __forceinline__ int cover_walk(const uint32_t* __restrict__ words, int n_bits, int e, Emit emit) {
    int cursor = e, end = e;     // end: one past the last position covered so far (skipping zeros covers nothing)
    while (cursor < n_bits) {
        const int wd = cursor >> 5, sh = cursor & 31;
        uint64_t win = ((uint64_t)words[wd] | ((uint64_t)words[wd + 1] << 32)) >> sh;   // >= 33 valid bits from cursor on
        const int valid = n_bits - cursor;                                              // bits of THIS chunk in the window
        if (valid < 64) win &= (1ull << valid) - 1ull;
        if (win == 0) { cursor += 64 - sh; continue; }
        const int p = cursor + __builtin_ctzll(win);
        const int st = PAIR ? p : ROWQ ? (p & ~1) : (p & ~3);   // (quads start on even columns: the un-pooling input's pairs)
        emit(st);
        cursor = end = st + (PAIR ? 16 : 32);
    }
    return end > n_bits ? end - n_bits : 0;
}

// (2) chunk tables. thread = (chunk, entry state); pair mode uses entry state 0 only.
__global__ __launch_bounds__(256) void cover_table_kernel(CoverGroup g) {
    const CoverProblem& P = g.p[blockIdx.y];
    const int chunk = blockIdx.x * 32 + (threadIdx.x >> 3), e8 = threadIdx.x & 7;
    if (chunk >= cover_chunks(P.h, P.w, P.pair_w, P.quad)) return;
    int n = 0, ex = 0;
    if (P.quad && P.pair_w == 0) {        // flat QUAD mode: a row group, runs of 32 positions
        if (e8 == 0) {
            const int rw = cover_row_words(P.w);
            cover_walk<false, true>(g.bits + P.word_base + chunk * rw, P.w, 0, [&](int) { ++n; });
        }
    } else if (P.pair_w < 0) {            // TILE mode: live tiles of -pair_w positions in this chunk (entry state unused)
        if (e8 == 0) {
            const int tw = -P.pair_w / 32;                                    // words per tile (bn >= 64) ...
            const uint32_t* wds = g.bits + P.word_base + chunk * SM_COVER_CHUNK_WORDS;
            const int n_words = cover_words(P.h, P.w, P.pair_w) - chunk * SM_COVER_CHUNK_WORDS;
            for (int t = 0; t * tw < SM_COVER_CHUNK_WORDS && t * tw < n_words; ++t) {
                uint32_t any = 0;
                for (int k = 0; k < tw && t * tw + k < n_words; ++k) any |= wds[t * tw + k];
                n += any != 0;
            }
        }
    } else if (P.pair_w > 0) {
        if (e8 == 0) {
            const int rw = cover_row_words(P.w);
            cover_walk<true>(g.bits + P.word_base + chunk * rw, P.w, 0, [&](int) { ++n; });
        }
    } else {
        const int n_pos = P.h * row_stride(P.w);
        const int bits_here = min(SM_COVER_CHUNK_WORDS * 32, n_pos - chunk * SM_COVER_CHUNK_WORDS * 32);
        ex = cover_walk<false>(g.bits + P.word_base + chunk * SM_COVER_CHUNK_WORDS, bits_here, e8 * 4, [&](int) { ++n; });
    }
    reinterpret_cast<uint16_t*>(g.table + P.chunk_base + chunk)[e8] = (uint16_t)((n << 3) | (ex >> 2));
}

// (3) compose the tables left to right: one thread per map, the table rows staged through LDS.
__global__ __launch_bounds__(64) void cover_scan_kernel(CoverGroup g) {
    const CoverProblem& P = g.p[blockIdx.x];
    const uint4* __restrict__ tab = g.table + P.chunk_base;
    uint32_t* __restrict__ state = g.state + P.chunk_base;
    constexpr int BATCH = 512;                       // table rows staged in LDS per round (a dependent global load per
    __shared__ uint4 rows[BATCH];                    // chunk would cost ~0.5 us each)
    __shared__ uint32_t out[BATCH];
    int e8 = 0, base = 0;                            // (meaningful on thread 0 only)
    const int n_chunks = cover_chunks(P.h, P.w, P.pair_w, P.quad);
    for (int c0 = 0; c0 < n_chunks; c0 += BATCH) {
        const int nb = min(BATCH, n_chunks - c0);
        for (int i = threadIdx.x; i < nb; i += 64) rows[i] = tab[c0 + i];
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 0; i < nb; ++i) {
                const uint4 row = rows[i];
                out[i] = ((uint32_t)base << 3) | (uint32_t)e8;
                const uint32_t d = e8 < 4 ? (e8 < 2 ? row.x : row.y) : (e8 < 6 ? row.z : row.w);
                const uint32_t t = (e8 & 1) ? (d >> 16) : (d & 0xffffu);
                base += (int)(t >> 3);
                e8 = (int)(t & 7u);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nb; i += 64) state[c0 + i] = out[i];
        __syncthreads();
    }
    if (threadIdx.x == 0) *P.count = P.quad ? 4 * base : P.pair_w > 0 ? 2 * base : base;
}

// (4) every chunk once more, from its true entry state, writing its segments.
__global__ __launch_bounds__(256) void cover_emit_kernel(CoverGroup g) {
    const CoverProblem& P = g.p[blockIdx.y];
    const int chunk = blockIdx.x * 256 + threadIdx.x;
    if (chunk >= cover_chunks(P.h, P.w, P.pair_w, P.quad)) return;
    const uint32_t s = g.state[P.chunk_base + chunk];
    int n = (int)(s >> 3);
    if (P.quad) {
        // chunk = image rows 4 chunk .. 4 chunk + 3 (rows behind the last image row: padding entries, which the conv's
        // epilogue skips); H = the height of the plane the segments index
        const bool pooled = P.pair_w > 0;
        const int rw = cover_row_words(P.w), Wp = row_stride(pooled ? P.pair_w : P.w), H = pooled ? 2 * P.h : P.h;
        auto emit = [&](int X0) {
            const int q = (4 * chunk + 1) * Wp + (pooled ? 2 * X0 : X0) + 1;
            if (4 * n + 3 < P.cap)
                for (int i = 0; i < 4; ++i)
                    P.starts[4 * n + i] = (P.tag << 24) | (4 * chunk + i < H ? q + i * Wp : 0xFFFFFF);
            ++n;
        };
        if (pooled) cover_walk<true>(g.bits + P.word_base + chunk * rw, P.w, 0, emit);
        else cover_walk<false, true>(g.bits + P.word_base + chunk * rw, P.w, 0, emit);
    } else if (P.pair_w < 0) {
        const int tw = -P.pair_w / 32, per_chunk = SM_COVER_CHUNK_WORDS / tw;
        const uint32_t* wds = g.bits + P.word_base + chunk * SM_COVER_CHUNK_WORDS;
        const int n_words = cover_words(P.h, P.w, P.pair_w) - chunk * SM_COVER_CHUNK_WORDS;
        for (int t = 0; t < per_chunk && t * tw < n_words; ++t) {
            uint32_t any = 0;
            for (int k = 0; k < tw && t * tw + k < n_words; ++k) any |= wds[t * tw + k];
            if (any) {
                if (n < P.cap) P.starts[n] = (P.tag << 24) | (chunk * per_chunk + t);
                ++n;
            }
        }
    } else if (P.pair_w > 0) {
        const int rw = cover_row_words(P.w), Wp = row_stride(P.pair_w);
        cover_walk<true>(g.bits + P.word_base + chunk * rw, P.w, 0, [&](int X0) {
            const int q = (2 * chunk + 1) * Wp + 2 * X0 + 1;     // chunk = pooled row Y: image rows 2Y and 2Y + 1
            if (2 * n + 1 < P.cap) {
                P.starts[2 * n] = (P.tag << 24) | q;
                P.starts[2 * n + 1] = (P.tag << 24) | (q + Wp);
            }
            ++n;
        });
    } else {
        const int Wp = row_stride(P.w), n_pos = P.h * Wp, first = chunk * SM_COVER_CHUNK_WORDS * 32;
        const int bits_here = min(SM_COVER_CHUNK_WORDS * 32, n_pos - first);
        cover_walk<false>(g.bits + P.word_base + chunk * SM_COVER_CHUNK_WORDS, bits_here, (int)(s & 7u) * 4, [&](int st) {
            if (n < P.cap) P.starts[n] = (P.tag << 24) | (Wp + first + st);
            ++n;
        });
    }
}

// ---------------------------------------------------------------------------------------------------
// Grouped forms of the per-view kernels above: one launch per KIND over all levels / (level, layer) pairs of a view
// (sm_view_masks, sm_view_lists). Same arithmetic, thread for thread; the problem tables travel as kernel arguments.
// ---------------------------------------------------------------------------------------------------
struct LevelMapProblem {
    const float* E;            // [h][w] (this level's slice)
    const float* Wt;           // or NULL
    float* M;
    float* pixel_weight;
    uint8_t* passed;
    float* m_sum;
    int H, W;
};
struct LevelMapGroup {
    LevelMapProblem p[SM_VIEW_MAX_LEVELS];
    const float* angle_guidance;
    const float* angle_deg;
    float thr;
    int h, w;
};
__global__ __launch_bounds__(256) void level_maps_group_kernel(LevelMapGroup g) {
    const LevelMapProblem& P = g.p[blockIdx.y];
    if (blockIdx.x * 256 >= P.H * P.W) return;
    __shared__ float red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int h = g.h, w = g.w, H = P.H, W = P.W;
    float mval = 0.f;
    if (i < H * W) {
        const int y = i / W, x = i - y * W;
        const int sy = nearest_src(y, h, H), sx = nearest_src(x, w, W);
        mval = (P.E[sy * w + sx] > 0.f) ? 1.f : 0.f;
        P.M[i] = mval;
        if (P.pixel_weight) {
            float pw = 1.f;
            if (g.angle_guidance) pw = bilinear_at([&](int a, int b) { return g.angle_guidance[a * w + b]; }, y, x, h, w, H, W);
            if (P.Wt) pw *= P.Wt[sy * w + sx];
            P.pixel_weight[i] = pw;
        }
        if (P.passed) {
            bool ps = true;
            if (g.angle_deg) ps = bilinear_at([&](int a, int b) { return g.angle_deg[a * w + b]; }, y, x, h, w, H, W) < g.thr;
            P.passed[i] = ps ? 1 : 0;
        }
    }
    float v = mval;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float s = red[0] + red[1] + red[2] + red[3];
        if (s != 0.f) atomicAdd(P.m_sum, s);
    }
}

// mask u8 -> float (the E plane of a view without depth scaling: model/model.py:240-254 uses the plain mask)
__global__ __launch_bounds__(256) void mask_to_float_kernel(const uint8_t* __restrict__ m, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = m[i] ? 1.f : 0.f;
}

struct LayerMaskProblem {
    const float* M;
    const uint8_t* passed;
    float* planes;             // 3 padded planes
    float* counts;
    int H, W, hl, wl;
};
struct LayerMaskGroup {
    LayerMaskProblem p[64];
};
__global__ __launch_bounds__(256) void layer_masks_group_kernel(LayerMaskGroup g) {
    const LayerMaskProblem& P = g.p[blockIdx.y];
    const int hl = P.hl, wl = P.wl;
    if (blockIdx.x * 256 >= hl * wl) return;
    __shared__ float red[3][4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int Wp = row_stride(wl), plane = plane_size(hl, wl);
    float a = 0.f, p = 0.f, f = 0.f;
    if (i < hl * wl) {
        const int y = i / wl, x = i - y * wl;
        const int j = nearest_src(y, P.H, hl) * P.W + nearest_src(x, P.W, wl);
        a = P.M[j];
        const bool ps = P.passed ? P.passed[j] != 0 : true;
        p = ps ? a : 0.f;
        f = ps ? 0.f : a;
        const int q = (y + 1) * Wp + x + 1;
        P.planes[q] = a;
        P.planes[plane + q] = p;
        P.planes[2 * plane + q] = f;
    }
    float v[3] = {a, p, f};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const float s = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        if (s != 0.f) atomicAdd(P.counts + threadIdx.x, s);
    }
}

struct FactorProblem {
    const float* counts;       // N_all of this (level, loss layer)
    float* factor;
    float size;
    int loss_layer;
};
struct FactorGroup {
    FactorProblem p[64];
    int n;
};
// thread k = loss layer k: the means of its levels, in table order (the order sm_level_factors sums them in)
__global__ void level_factors_group_kernel(FactorGroup a) {
    const int layer = threadIdx.x;
    float s = 0.f;
    bool any = false;
    for (int i = 0; i < a.n; ++i)
        if (a.p[i].loss_layer == layer) { s += *a.p[i].counts / a.p[i].size; any = true; }
    if (!any) return;
    for (int i = 0; i < a.n; ++i)
        if (a.p[i].loss_layer == layer) *a.p[i].factor = (*a.p[i].counts / a.p[i].size) / s;
}

struct ResizeProblem {
    const float* src;
    float* dst;
    int C, h, w, H, W;
};
struct ResizeGroup {
    ResizeProblem p[16];
};
__global__ __launch_bounds__(256) void fmap_resize_group_kernel(ResizeGroup g) {
    const ResizeProblem& P = g.p[blockIdx.z];
    const int c = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (c >= P.C || i >= P.H * P.W) return;
    const int y = i / P.W, x = i - y * P.W;
    const int wp = row_stride(P.w), Wp = row_stride(P.W);
    const float* src = P.src + (size_t)c * plane_size(P.h, P.w);
    P.dst[(size_t)c * plane_size(P.H, P.W) + (y + 1) * Wp + x + 1] =
        bilinear_at([&](int a, int b) { return src[(a + 1) * wp + b + 1]; }, y, x, P.h, P.w, P.H, P.W);
}

struct NeedProblem {
    const float* need_out;
    const float* M;            // or NULL
    float* need_src;
    int ho, wo, H, W, hs, ws;
};
struct NeedGroup {
    NeedProblem p[SM_VIEW_MAX_LEVELS];
    int mode;
};
__global__ __launch_bounds__(256) void need_step_group_kernel(NeedGroup g) {
    const NeedProblem& P = g.p[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.hs * P.ws) return;
    const int y = i / P.ws, x = i - y * P.ws;
    float v = 0.f;
    if (g.mode == 1) {
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = y + dy, xx = x + dx;
                if (yy >= 0 && yy < P.ho && xx >= 0 && xx < P.wo) v = fmaxf(v, P.need_out[yy * P.wo + xx]);
            }
    } else if (g.mode == 2) {
        const int yy = y >> 1, xx = x >> 1;
        if (yy < P.ho && xx < P.wo) v = P.need_out[yy * P.wo + xx];
    }
    if (P.M) v = fmaxf(v, P.M[nearest_src(y, P.H, P.hs) * P.W + nearest_src(x, P.W, P.ws)]);
    P.need_src[i] = v > 0.f ? 1.f : 0.f;
}

// The active lists of a view: list s = the runs of its levels back to back, each padded to a multiple of `group` entries
// with (level << 24) | 0xFFFFFF. Block s: the offsets from the covers' counts (one thread), then the copy.
struct ConcatList {
    const int32_t* staging;    // n_levels x staging_cap
    int32_t* out;
    int cap, group, staging_cap, pad;
};
struct ConcatGroup {
    ConcatList l[SM_VIEW_MAX_LISTS];
    const int32_t* counts;     // [list][level]
    int32_t* summary;
    int n_levels;
};
__global__ __launch_bounds__(256) void list_concat_kernel(ConcatGroup g) {
    const ConcatList& L = g.l[blockIdx.x];
    __shared__ int off[SM_VIEW_MAX_LEVELS + 1], live[SM_VIEW_MAX_LEVELS];
    if (threadIdx.x == 0) {
        int pos = 0;
        for (int k = 0; k < g.n_levels; ++k) {
            int n = g.counts[blockIdx.x * g.n_levels + k];
            n = n > L.staging_cap ? L.staging_cap : n;          // (an overflowing cover is reported through the count)
            live[k] = n;
            off[k] = pos;
            pos += n + (L.group - n % L.group) % L.group;
        }
        off[g.n_levels] = pos;
        int32_t* sum = g.summary + 9 * blockIdx.x;
        sum[0] = pos;
        for (int k = 0; k < SM_VIEW_MAX_LEVELS; ++k) sum[1 + k] = k < g.n_levels ? g.counts[blockIdx.x * g.n_levels + k] : 0;
    }
    __syncthreads();
    for (int k = 0; k < g.n_levels; ++k) {
        const int n = live[k], end = off[k + 1] - off[k];
        for (int i = threadIdx.x; i < end; i += 256)
            if (off[k] + i < L.cap) L.out[off[k] + i] = i < n ? L.staging[(size_t)k * L.staging_cap + i] : ((k << 24) | 0xFFFFFF);
    }
}

}  // namespace sm

extern "C" {

static void cover_extent(const sm_cover_problem& p, int* n_words, int* n_chunks) {
    *n_words = sm::cover_words(p.h, p.w, p.pair_w, p.quad);
    *n_chunks = sm::cover_chunks(p.h, p.w, p.pair_w, p.quad);
}

size_t sm_cover_segments_ws_bytes(const sm_cover_problem* problems, int n) {
    size_t words = 0, chunks = 0;
    for (int i = 0; i < n; ++i) {
        int w, c;
        cover_extent(problems[i], &w, &c);
        words += (size_t)((w + 3) & ~3);
        chunks += (size_t)c;
    }
    return words * 4 + chunks * 16 + chunks * 4 + 64;
}

int sm_cover_segments(const sm_cover_problem* problems, int n, void* ws, size_t ws_bytes, void* stream) {
    if (n < 1 || n > SM_COVER_MAX || ws == nullptr || ((uintptr_t)ws & 15)) return (int)hipErrorInvalidValue;
    if (ws_bytes < sm_cover_segments_ws_bytes(problems, n)) return (int)hipErrorInvalidValue;
    if (problems == nullptr) return (int)hipErrorInvalidValue;
    sm::CoverGroup g;
    int words = 0, chunks = 0, max_words = 0, max_chunks = 0;
    for (int i = 0; i < n; ++i) {
        const sm_cover_problem& p = problems[i];
        if (p.h < 1 || p.w < 1 || p.tag < 0 || p.tag > 127) return (int)hipErrorInvalidValue;
        if (p.pair_w > 0 && p.pair_w / 2 != p.w) return (int)hipErrorInvalidValue;
        if (p.pair_w < 0 && ((-p.pair_w) % 64 != 0 || (SM_COVER_CHUNK_WORDS * 32) % (-p.pair_w) != 0))
            return (int)hipErrorInvalidValue;                      // tiles of 64 .. 2048 positions that tile a chunk
        const long plane_rows = (p.pair_w > 0 ? 2l * p.h : (long)p.h) + 2;
        if (plane_rows * sm::row_stride(p.pair_w > 0 ? p.pair_w : p.w) >= 0xFFFFFFl)
            return (int)hipErrorInvalidValue;                      // a list entry holds the position in 24 bits
        int nw, nc;
        cover_extent(p, &nw, &nc);
        if (p.pair_w > 0 && (p.w + 15) / 16 + 1 > 8191) return (int)hipErrorInvalidValue;   // 13-bit counts per chunk
        if (p.quad != 0 && (p.quad != 1 || p.pair_w < 0 || (p.w + 31) / 32 + 1 > 8191)) return (int)hipErrorInvalidValue;
        g.p[i] = sm::CoverProblem{p.need, p.starts, p.count, p.h, p.w, p.tag, p.cap, p.pair_w, p.quad, words, chunks};
        words += (nw + 3) & ~3;
        chunks += nc;
        max_words = nw > max_words ? nw : max_words;
        max_chunks = nc > max_chunks ? nc : max_chunks;
    }
    g.bits = static_cast<uint32_t*>(ws);
    g.table = reinterpret_cast<uint4*>(g.bits + words);            // (words is a multiple of 4: 16-byte aligned)
    g.state = reinterpret_cast<uint32_t*>(g.table + chunks);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sm::cover_pack_kernel, dim3((max_words + 63) / 64, n), dim3(256), 0, st, g);
    hipLaunchKernelGGL(sm::cover_table_kernel, dim3((max_chunks + 31) / 32, n), dim3(256), 0, st, g);
    hipLaunchKernelGGL(sm::cover_scan_kernel, dim3(n), dim3(64), 0, st, g);
    hipLaunchKernelGGL(sm::cover_emit_kernel, dim3((max_chunks + 255) / 256, n), dim3(256), 0, st, g);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_need_step(const float* need_out, int ho, int wo, int mode, const float* M, int H, int W, float* need_src, int hs,
                 int ws, void* stream) {
    hipLaunchKernelGGL(sm::need_step_kernel, dim3((hs * ws + 255) / 256), dim3(256), 0, (hipStream_t)stream, need_out, ho,
                       wo, mode, M, H, W, need_src, hs, ws);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_tile_flags(const float* need, int h, int w, int bn, uint8_t* flags, void* stream) {
    const int Wp = sm::row_stride(w);
    const int n_tiles = (h * Wp + bn - 1) / bn;
    hipLaunchKernelGGL(sm::tile_flags_kernel, dim3((n_tiles + 3) / 4), dim3(256), 0, (hipStream_t)stream, need, h, w, Wp,
                       bn, n_tiles, flags);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_level_masks(const int64_t* rounded, const int64_t* other, const float* interp_w, const uint8_t* mask, int h,
                   int w, int n_levels, float* E, float* Wt, void* stream) {
    hipLaunchKernelGGL(sm::level_masks_kernel, dim3((h * w + 255) / 256, n_levels), dim3(256), 0, (hipStream_t)stream,
                       rounded, other, interp_w, mask, h, w, n_levels, E, Wt);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_level_maps(const float* E, const float* Wt, const float* angle_guidance, const float* angle_deg,
                  float angle_threshold, int h, int w, int H, int W, float* M, float* pixel_weight, uint8_t* passed,
                  float* m_sum, void* stream) {
    hipLaunchKernelGGL(sm::level_maps_kernel, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, E, Wt,
                       angle_guidance, angle_deg, angle_threshold, h, w, H, W, M, pixel_weight, passed, m_sum);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_layer_masks(const float* M, const uint8_t* passed, int H, int W, int hl, int wl, float* m_all, float* m_pass,
                   float* m_fail, float* counts, void* stream) {
    hipLaunchKernelGGL(sm::layer_masks_kernel, dim3((hl * wl + 255) / 256), dim3(256), 0, (hipStream_t)stream, M, passed,
                       H, W, hl, wl, sm::row_stride(wl), m_all, m_pass, m_fail, counts);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_level_factors(const float* const* counts_all, const float* sizes, int n, float* const* factors, void* stream) {
    if (n < 1 || n > 8) return (int)hipErrorInvalidValue;
    sm::FactorArgs a;
    a.n = n;
    for (int i = 0; i < n; ++i) {
        a.counts[i] = counts_all[i];
        a.factors[i] = factors[i];
        a.sizes[i] = sizes[i];
    }
    hipLaunchKernelGGL(sm::level_factors_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_fmap_resize_bilinear(const float* in, int C, int h, int w, float* out, int H, int W, void* stream) {
    hipLaunchKernelGGL(sm::fmap_resize_bilinear_kernel, dim3((H * W + 255) / 256, C), dim3(256), 0, (hipStream_t)stream,
                       in, h, w, sm::row_stride(w), sm::plane_size(h, w), out, H, W, sm::row_stride(W),
                       sm::plane_size(H, W));
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_image_to_fmap(const float* in, int C, int h, int w, float* out, int H, int W, void* stream) {
    hipLaunchKernelGGL(sm::image_to_fmap_kernel, dim3((H * W + 255) / 256, C), dim3(256), 0, (hipStream_t)stream, in, h,
                       w, out, H, W, sm::row_stride(W), sm::plane_size(H, W));
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_fmap_to_image(const float* in, int C, int H, int W, float* out, void* stream) {
    hipLaunchKernelGGL(sm::fmap_to_image_kernel, dim3((H * W + 255) / 256, C), dim3(256), 0, (hipStream_t)stream, in, H,
                       W, sm::row_stride(W), sm::plane_size(H, W), out);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_view_masks(const sm_view_masks_desc* d, void* stream) {
    if (d == nullptr || d->n_levels < 1 || d->n_levels > SM_VIEW_MAX_LEVELS || d->n_masks < 0 || d->n_masks > 64 ||
        d->n_resizes < 0 || d->n_resizes > 16 || d->h < 1 || d->w < 1 || d->E == nullptr)
        return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const bool depth = d->rounded != nullptr;
    if (depth && (d->other == nullptr || d->interp_w == nullptr || d->Wt == nullptr)) return (int)hipErrorInvalidValue;
    const size_t hw = (size_t)d->h * d->w;
    if (depth)
        hipLaunchKernelGGL(sm::level_masks_kernel, dim3((unsigned)((hw + 255) / 256), d->n_levels), dim3(256), 0, st, d->rounded,
                           d->other, d->interp_w, d->mask, d->h, d->w, d->n_levels, d->E, d->Wt);
    else
        hipLaunchKernelGGL(sm::mask_to_float_kernel, dim3((unsigned)((hw + 255) / 256)), dim3(256), 0, st, d->mask, d->E, (int)hw);
    // zero the accumulators (the levels' mask sums; the masks' counts - with the factor behind them, which the factor
    // kernel rewrites): adjacent ranges are merged, a view's tables are one or two memsets
    char* z_lo = nullptr;
    size_t z_len = 0;
    auto zero = [&](void* ptr, size_t bytes) -> hipError_t {
        char* c = static_cast<char*>(ptr);
        if (z_lo != nullptr && c == z_lo + z_len) { z_len += bytes; return hipSuccess; }
        hipError_t e = z_lo ? hipMemsetAsync(z_lo, 0, z_len, st) : hipSuccess;
        z_lo = c;
        z_len = bytes;
        return e;
    };
    sm::LevelMapGroup lm;
    int n_lm = 0, max_px = 0;
    for (int i = 0; i < d->n_levels; ++i) {
        const sm_view_level& L = d->levels[i];
        if (!L.has_maps) continue;
        if (L.M == nullptr || L.m_sum == nullptr || L.H < 1 || L.W < 1) return (int)hipErrorInvalidValue;
        if (hipError_t e = zero(L.m_sum, sizeof(float)); e != hipSuccess) return (int)e;
        lm.p[n_lm++] = sm::LevelMapProblem{d->E + (depth ? (size_t)i * hw : 0), depth ? d->Wt + (size_t)i * hw : nullptr, L.M,
                                           L.pixel_weight, L.passed, L.m_sum, L.H, L.W};
        max_px = L.H * L.W > max_px ? L.H * L.W : max_px;
    }
    if (n_lm == 0) return (int)hipErrorInvalidValue;
    for (int k = 0; k < d->n_masks; ++k) {       // (before the level-maps launch: one flush for both groups when adjacent)
        const sm_view_layer_mask& m = d->masks[k];
        const bool with_factor = m.factor == m.counts + 3;
        if (hipError_t e = zero(m.counts, (with_factor ? 4 : 3) * sizeof(float)); e != hipSuccess) return (int)e;
    }
    if (z_lo != nullptr)
        if (hipError_t e = hipMemsetAsync(z_lo, 0, z_len, st); e != hipSuccess) return (int)e;
    lm.angle_guidance = d->angle_guidance;
    lm.angle_deg = d->angle_degrees;
    lm.thr = d->angle_threshold;
    lm.h = d->h;
    lm.w = d->w;
    hipLaunchKernelGGL(sm::level_maps_group_kernel, dim3((max_px + 255) / 256, n_lm), dim3(256), 0, st, lm);
    if (d->n_masks > 0) {
        sm::LayerMaskGroup mg;
        sm::FactorGroup fg;
        fg.n = d->n_masks;
        int max_l = 0;
        for (int k = 0; k < d->n_masks; ++k) {
            const sm_view_layer_mask& m = d->masks[k];
            if (m.level < 0 || m.level >= d->n_levels || !d->levels[m.level].has_maps || m.loss_layer < 0 || m.loss_layer >= 64)
                return (int)hipErrorInvalidValue;
            const sm_view_level& L = d->levels[m.level];
            mg.p[k] = sm::LayerMaskProblem{L.M, L.passed, m.mask_planes, m.counts, L.H, L.W, m.hl, m.wl};
            fg.p[k] = sm::FactorProblem{m.counts, m.factor, (float)m.hl * (float)m.wl, m.loss_layer};
            max_l = m.hl * m.wl > max_l ? m.hl * m.wl : max_l;
        }
        hipLaunchKernelGGL(sm::layer_masks_group_kernel, dim3((max_l + 255) / 256, d->n_masks), dim3(256), 0, st, mg);
        hipLaunchKernelGGL(sm::level_factors_group_kernel, dim3(1), dim3(64), 0, st, fg);
    }
    if (d->n_resizes > 0) {
        sm::ResizeGroup rg;
        int max_px2 = 0, max_c = 0;
        for (int k = 0; k < d->n_resizes; ++k) {
            const sm_view_resize& r = d->resizes[k];
            rg.p[k] = sm::ResizeProblem{r.src, r.dst, r.C, r.h, r.w, r.H, r.W};
            max_px2 = r.H * r.W > max_px2 ? r.H * r.W : max_px2;
            max_c = r.C > max_c ? r.C : max_c;
        }
        hipLaunchKernelGGL(sm::fmap_resize_group_kernel, dim3((max_px2 + 255) / 256, max_c, d->n_resizes), dim3(256), 0, st, rg);
    }
    SM_LAUNCH_CHECK();
    return 0;
}

// the cover problems of a view's lists (one per list and level), in list order
static int view_cover_problems(const sm_view_lists_desc* d, sm_cover_problem* out, int32_t* counts_dev) {
    int n = 0;
    for (int s = 0; s < d->n_lists; ++s) {
        const sm_view_list& L = d->lists[s];
        if (L.layer < 0 || L.layer >= d->n_layers || L.group < 1 || L.out == nullptr) return -1;
        for (int g = 0; g < d->n_levels; ++g, ++n) {
            int pair_w = 0;
            if (L.mode == 1) {
                if (L.pair_layer < 0 || L.pair_layer >= d->n_layers) return -1;
                pair_w = d->lw[g][L.pair_layer];
            } else if (L.mode == 2) {
                pair_w = -L.bn;
            } else if (L.mode == 3) {          // quads over the pooled layer's need map (a conv with SM_EPI_POOL)
                if (L.pair_layer < 0 || L.pair_layer >= d->n_layers) return -1;
                pair_w = d->lw[g][L.pair_layer];
            } else if (L.mode != 0 && L.mode != 4) {
                return -1;
            }
            if (out) out[n] = sm_cover_problem{d->need[g][L.layer], L.staging + (size_t)g * L.staging_cap,
                                               counts_dev ? counts_dev + n : nullptr,
                                               d->lh[g][L.layer], d->lw[g][L.layer], g, L.staging_cap, pair_w,
                                               (L.mode == 3 || L.mode == 4) ? 1 : 0};
        }
    }
    return n;
}

static size_t view_cover_ws_bytes(const sm_view_lists_desc* d, const sm_cover_problem* probs, int n) {
    size_t most = 0;
    for (int i = 0; i < n; i += SM_COVER_MAX) {
        const size_t b = sm_cover_segments_ws_bytes(probs + i, n - i < SM_COVER_MAX ? n - i : SM_COVER_MAX);
        most = b > most ? b : most;
    }
    return most;
}

size_t sm_view_lists_ws_bytes(const sm_view_lists_desc* d) {
    if (d == nullptr || d->n_lists < 0 || d->n_lists > SM_VIEW_MAX_LISTS || d->n_levels < 1 || d->n_levels > SM_VIEW_MAX_LEVELS)
        return 0;
    static thread_local sm_cover_problem probs[SM_VIEW_MAX_LISTS * SM_VIEW_MAX_LEVELS];
    const int n = view_cover_problems(d, probs, nullptr);
    if (n < 0) return 0;
    // [cover counts: one int per problem, 256-byte aligned][cover scratch of the largest batch]
    return (((size_t)n * 4 + 255) & ~(size_t)255) + view_cover_ws_bytes(d, probs, n) + 256;
}

int sm_view_lists(const sm_view_lists_desc* d, void* stream) {
    if (d == nullptr || d->n_levels < 1 || d->n_levels > SM_VIEW_MAX_LEVELS || d->n_layers < 2 ||
        d->n_layers > SM_VIEW_MAX_LAYERS || d->n_lists < 0 || d->n_lists > SM_VIEW_MAX_LISTS || d->ws == nullptr ||
        ((uintptr_t)d->ws & 15) || d->summary == nullptr)
        return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    // ---- need maps: the deepest layer from its mask, then one launch per node, backwards, over all levels
    const int last = d->n_layers - 1;
    for (int j = last; j >= 0; --j) {
        // j == last: need[last] = mask only (mode 0). Otherwise node j (0-based) produces layer j + 1 from node_src[j]:
        // need[src] = step(need[j + 1]); handled when we reach layer j + 1's producer below.
        sm::NeedGroup ng;
        int max_px = 0;
        int src_layer, mode;
        if (j == last) {
            src_layer = last;
            mode = 0;
        } else {
            src_layer = d->node_src[j];
            mode = d->node_is_pool[j] ? 2 : 1;
            if (src_layer < 0 || src_layer > j) return (int)hipErrorInvalidValue;
        }
        for (int g = 0; g < d->n_levels; ++g) {
            const int out_layer = j + 1;
            ng.p[g] = sm::NeedProblem{mode == 0 ? nullptr : d->need[g][out_layer], d->injected[src_layer] ? d->M[g] : nullptr,
                                      d->need[g][src_layer], mode == 0 ? 0 : d->lh[g][out_layer],
                                      mode == 0 ? 0 : d->lw[g][out_layer], d->H[g], d->W[g], d->lh[g][src_layer],
                                      d->lw[g][src_layer]};
            if (ng.p[g].need_src == nullptr || (mode != 0 && ng.p[g].need_out == nullptr)) return (int)hipErrorInvalidValue;
            const int px = d->lh[g][src_layer] * d->lw[g][src_layer];
            max_px = px > max_px ? px : max_px;
        }
        ng.mode = mode;
        hipLaunchKernelGGL(sm::need_step_group_kernel, dim3((max_px + 255) / 256, d->n_levels), dim3(256), 0, st, ng);
    }
    if (d->n_lists == 0) {
        SM_LAUNCH_CHECK();
        return 0;
    }
    // ---- covers of every (list, level), in batches of 64 maps, then one concatenation launch
    static thread_local sm_cover_problem probs[SM_VIEW_MAX_LISTS * SM_VIEW_MAX_LEVELS];
    int32_t* counts_dev = static_cast<int32_t*>(d->ws);
    const int n = view_cover_problems(d, probs, counts_dev);
    if (n < 0) return (int)hipErrorInvalidValue;
    const size_t counts_bytes = ((size_t)n * 4 + 255) & ~(size_t)255;
    const size_t cover_bytes = view_cover_ws_bytes(d, probs, n);
    if (d->ws_bytes < counts_bytes + cover_bytes) return (int)hipErrorInvalidValue;
    void* cover_ws = static_cast<char*>(d->ws) + counts_bytes;
    for (int i = 0; i < n; i += SM_COVER_MAX) {
        const int nb = n - i < SM_COVER_MAX ? n - i : SM_COVER_MAX;
        if (int e = sm_cover_segments(probs + i, nb, cover_ws, cover_bytes, stream)) return e;
    }
    sm::ConcatGroup cg;
    cg.summary = d->summary;
    cg.counts = counts_dev;
    cg.n_levels = d->n_levels;
    for (int s = 0; s < d->n_lists; ++s) {
        const sm_view_list& L = d->lists[s];
        cg.l[s] = sm::ConcatList{L.staging, L.out, L.cap, L.group, L.staging_cap, 0};
    }
    hipLaunchKernelGGL(sm::list_concat_kernel, dim3(d->n_lists), dim3(256), 0, st, cg);
    SM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

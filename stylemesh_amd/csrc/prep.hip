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
__host__ __device__ inline int cover_words(int h, int w, int pair_w) {
    return pair_w > 0 ? h * cover_row_words(w) + 1 : (h * row_stride(w) + 31) / 32 + 2;
}
__host__ __device__ inline int cover_chunks(int h, int w, int pair_w) {
    return pair_w > 0 ? h : (h * row_stride(w) + SM_COVER_CHUNK_WORDS * 32 - 1) / (SM_COVER_CHUNK_WORDS * 32);
}

// (1) bits. One wave = 64 consecutive positions per ballot = two words; a block of 4 waves makes 64 words.
__global__ __launch_bounds__(256) void cover_pack_kernel(CoverGroup g) {
    const CoverProblem& P = g.p[blockIdx.y];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int first_word = blockIdx.x * 64 + wave * 16;
    const int n_words = cover_words(P.h, P.w, P.pair_w);
    if (first_word >= n_words) return;
    const int Wp = row_stride(P.w), n_pos = P.h * Wp, rw = cover_row_words(P.w);
    for (int k = 0; k < 8; ++k) {
        const int wd = first_word + 2 * k;
        if (wd >= n_words) break;
        bool bit = false;
        if (P.pair_w > 0) {             // word wd = row wd / rw, columns (wd % rw) * 32 ...; a row is a whole number of words
            const int wl = wd + (lane >> 5);
            const int Y = wl / rw, X = (wl - Y * rw) * 32 + (lane & 31);
            bit = Y < P.h && X < P.w && P.need[(size_t)Y * P.w + X] > 0.f;
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
template <bool PAIR, typename Emit>
__device__ __forceinline__ int cover_walk(const uint32_t* __restrict__ words, int n_bits, int e, Emit emit) {
    int cursor = e, end = e;     // end: one past the last position covered so far (skipping zeros covers nothing)
    while (cursor < n_bits) {
        const int wd = cursor >> 5, sh = cursor & 31;
        uint64_t win = ((uint64_t)words[wd] | ((uint64_t)words[wd + 1] << 32)) >> sh;   // >= 33 valid bits from cursor on
        const int valid = n_bits - cursor;                                              // bits of THIS chunk in the window
        if (valid < 64) win &= (1ull << valid) - 1ull;
        if (win == 0) { cursor += 64 - sh; continue; }
        const int p = cursor + __builtin_ctzll(win);
        const int st = PAIR ? p : (p & ~3);
        emit(st);
        cursor = end = st + (PAIR ? 16 : 32);
    }
    return end > n_bits ? end - n_bits : 0;
}

// (2) chunk tables. thread = (chunk, entry state); pair mode uses entry state 0 only.
__global__ __launch_bounds__(256) void cover_table_kernel(CoverGroup g) {
    const CoverProblem& P = g.p[blockIdx.y];
    const int chunk = blockIdx.x * 32 + (threadIdx.x >> 3), e8 = threadIdx.x & 7;
    if (chunk >= cover_chunks(P.h, P.w, P.pair_w)) return;
    int n = 0, ex = 0;
    if (P.pair_w > 0) {
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
    const int n_chunks = cover_chunks(P.h, P.w, P.pair_w);
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
    if (threadIdx.x == 0) *P.count = P.pair_w > 0 ? 2 * base : base;
}

// (4) every chunk once more, from its true entry state, writing its segments.
__global__ __launch_bounds__(256) void cover_emit_kernel(CoverGroup g) {
    const CoverProblem& P = g.p[blockIdx.y];
    const int chunk = blockIdx.x * 256 + threadIdx.x;
    if (chunk >= cover_chunks(P.h, P.w, P.pair_w)) return;
    const uint32_t s = g.state[P.chunk_base + chunk];
    int n = (int)(s >> 3);
    if (P.pair_w > 0) {
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

}  // namespace sm

extern "C" {

static void cover_extent(const sm_cover_problem& p, int* n_words, int* n_chunks) {
    *n_words = sm::cover_words(p.h, p.w, p.pair_w);
    *n_chunks = sm::cover_chunks(p.h, p.w, p.pair_w);
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
    sm::CoverGroup g;
    int words = 0, chunks = 0, max_words = 0, max_chunks = 0;
    for (int i = 0; i < n; ++i) {
        const sm_cover_problem& p = problems[i];
        if (p.h < 1 || p.w < 1 || p.tag < 0 || p.tag > 127) return (int)hipErrorInvalidValue;
        if (p.pair_w != 0 && p.pair_w / 2 != p.w) return (int)hipErrorInvalidValue;
        const long plane_rows = (p.pair_w > 0 ? 2l * p.h : (long)p.h) + 2;
        if (plane_rows * sm::row_stride(p.pair_w > 0 ? p.pair_w : p.w) >= 0xFFFFFFl)
            return (int)hipErrorInvalidValue;                      // a list entry holds the position in 24 bits
        int nw, nc;
        cover_extent(p, &nw, &nc);
        if (p.pair_w > 0 && (p.w + 15) / 16 + 1 > 8191) return (int)hipErrorInvalidValue;   // 13-bit counts per chunk
        g.p[i] = sm::CoverProblem{p.need, p.starts, p.count, p.h, p.w, p.tag, p.cap, p.pair_w, words, chunks};
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

}  // extern "C"

// K2p: the texture scatter-add as a SORTED GATHER. Which texels a view's pixels hit, and with which bilinear weights,
// depends only on the view (UV grids + pixel weights), not on the step - and a view is optimised for 20-100
// consecutive steps (RepeatingSampler). So once per view every (pixel, texture layer, tap) contribution is listed as
// (texel, pixel, weight) and the list is sorted by texel; every step then walks the sorted list: one thread per
// entry gathers its pixel's image gradient, a wave-wide segmented sum adds the runs of equal texels, and the run
// totals are written with plain stores - atomics only where a run crosses a wave boundary.
//
// Replaces grid_sampler_2d_backward + the gradient hooks exactly like tex_sample_bwd_tiled_kernel (texture.hip);
// same tap arithmetic (make_taps below mirrors texture.hip). The sort (per-view preparation, not a per-step kernel) is the
// library's OWN stable LSD radix sort since round 5 (rs_* kernels below; rounds 1-4 called rocPRIM's device radix sort -
// the last third-party device code on the product path).
#include <algorithm>
#include <cstring>

#include "common.h"

namespace sm {

struct PlanLevel {
    const float2* grid;
    const float* pixel_weight;   // may be NULL
    int h, w;
    int first_pixel;             // prefix sum of h * w over the levels
};
struct PlanLevels {
    PlanLevel lv[SM_MAX_TEX_LAYERS];
    int n;
    int total_pixels;
};
struct PlanLayers {
    unsigned base[SM_MAX_TEX_LAYERS];   // arena offset of the layer's channel 0
    int w[SM_MAX_TEX_LAYERS];
    int h[SM_MAX_TEX_LAYERS];
    int n;
};

// entry value: low word = q | level << 24 | layer << 27 (q: position in the level's padded image plane), high word = weight
__device__ __forceinline__ unsigned long long pack_value(int q, int level, int layer, float wgt) {
    return (unsigned long long)((unsigned)q | ((unsigned)level << 24) | ((unsigned)layer << 27)) |
           ((unsigned long long)__float_as_uint(wgt) << 32);
}

// one thread per (pixel, layer): its four tap entries at e = (pixel * n_layers + layer) * 4 + tap
__global__ __launch_bounds__(256) void scatter_entries_kernel(PlanLevels V, PlanLayers L, unsigned* __restrict__ keys,
                                                              unsigned long long* __restrict__ vals, unsigned invalid) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= V.total_pixels * L.n) return;
    const int pix = t / L.n, layer = t - pix * L.n;
    int level = 0;
#pragma unroll
    for (int k = 1; k < SM_MAX_TEX_LAYERS; ++k)
        if (k < V.n && pix >= V.lv[k].first_pixel) level = k;
    const PlanLevel P = V.lv[level];
    const int i = pix - P.first_pixel;
    const int y = i / P.w, x = i - y * P.w;
    const float pw = P.pixel_weight ? P.pixel_weight[i] : 1.f;
    const float2 g = P.grid[i];
    const int W = L.w[layer], H = L.h[layer];
    // ATen grid_sampler source index (align_corners=True) + border clip, as make_taps() of texture.hip
    float ix = ((g.x + 1.f) / 2.f) * (float)(W - 1);
    float iy = ((g.y + 1.f) / 2.f) * (float)(H - 1);
    ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
    iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float ex = fx + 1.f, ey = fy + 1.f;
    const float wt[4] = {(ex - ix) * (ey - iy), (ix - fx) * (ey - iy), (ex - ix) * (iy - fy), (ix - fx) * (iy - fy)};
    const bool x1_in = x0 + 1 <= W - 1, y1_in = y0 + 1 <= H - 1;
    const bool in[4] = {true, x1_in, y1_in, x1_in && y1_in};
    const int q = (y + 1) * row_stride(P.w) + x + 1;
    const size_t e = (size_t)t * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float wk = wt[k] * pw;
        const bool live = in[k] && wk != 0.f;
        keys[e + k] = live ? L.base[layer] + (unsigned)((y0 + (k >> 1)) * W + x0 + (k & 1)) : invalid;
        vals[e + k] = pack_value(q, level, layer, wk);
    }
}

// ---------------------------------------------------------------------------------------------------
// Stable LSD radix sort of (u32 key, u64 value) pairs, ceil(key_bits / 9) passes of <= 9 bits. A pass = three launches:
//   rs_hist_kernel     one block per TILE of 4096 entries: its digit histogram -> hist[bin][tile] (bin-major);
//   rs_scan_rows_kernel one block per bin: exclusive scan of the bin's row over the tiles + the bin's total;
//   rs_scatter_kernel  one block per tile: base[bin] (the scan of the totals, redone per block: 512 values) + the row scan
//                      = where the tile's first entry of every bin goes; the tile's entries are ranked STABLY in 16
//                      rounds of 256 (round-major, then thread order = the input order): within a wave by matching
//                      digits with ballots, across the four waves through tagged per-wave counts in LDS.
// The keys of a view are spatially coherent (neighbouring pixels hit neighbouring texels): the high-digit passes write
// whole tiles into one or two bins - contiguous; the low-digit pass scatters runs of ~8 entries.
// Deterministic: equal keys keep their input order (pixel, layer, tap), so the per-step sums add in a fixed order.
// ---------------------------------------------------------------------------------------------------
constexpr int RS_ITEMS = 16, RS_TILE = 256 * RS_ITEMS, RS_MAX_BITS = 9, RS_MAX_BINS = 1 << RS_MAX_BITS;

// lanes of the wave whose (valid) entry has the same digit as this lane's
__device__ __forceinline__ unsigned long long rs_match(unsigned d, bool valid, int bits) {
    unsigned long long m = __ballot(valid);
    for (int b = 0; b < bits; ++b) {
        const bool one = (d >> b) & 1u;
        const unsigned long long s = __ballot(one);
        m &= one ? s : ~s;
    }
    return m;
}

__global__ __launch_bounds__(256) void rs_hist_kernel(const unsigned* __restrict__ keys, size_t n, int shift, int bits,
                                                      int n_tiles, unsigned* __restrict__ hist) {
    __shared__ unsigned h[RS_MAX_BINS];
    const int bins = 1 << bits, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < bins; i += 256) h[i] = 0u;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RS_TILE;
    unsigned k[RS_ITEMS];
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const size_t e = base + r * 256 + threadIdx.x;
        k[r] = e < n ? keys[e] : 0u;
    }
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const size_t e = base + r * 256 + threadIdx.x;
        const bool valid = e < n;
        const unsigned d = (k[r] >> shift) & (unsigned)(bins - 1);
        const unsigned long long m = rs_match(d, valid, bits);
        if (valid && (m & ((1ull << lane) - 1ull)) == 0ull) atomicAdd(&h[d], (unsigned)__popcll(m));   // one add per wave and bin
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += 256) hist[(size_t)i * n_tiles + blockIdx.x] = h[i];
}

// block-wide inclusive scan of one value per thread (256 threads); returns the inclusive prefix, *total = block sum
__device__ __forceinline__ unsigned rs_block_scan(unsigned v, unsigned* wsum /*[4] shared*/, unsigned* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    __syncthreads();            // (wsum may still be read from the previous call)
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    unsigned off = 0u, tot = 0u;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wave) off += wsum[w];
        tot += wsum[w];
    }
    *total = tot;
    return x + off;
}

__global__ __launch_bounds__(256) void rs_scan_rows_kernel(unsigned* __restrict__ hist, int n_tiles, unsigned* __restrict__ totals) {
    __shared__ unsigned wsum[4];
    unsigned* row = hist + (size_t)blockIdx.x * n_tiles;
    unsigned carry = 0u;
    for (int i0 = 0; i0 < n_tiles; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const unsigned v = i < n_tiles ? row[i] : 0u;
        unsigned tot;
        const unsigned incl = rs_block_scan(v, wsum, &tot);
        if (i < n_tiles) row[i] = carry + incl - v;
        carry += tot;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

__global__ __launch_bounds__(256) void rs_scatter_kernel(const unsigned* __restrict__ keys_in,
                                                         const unsigned long long* __restrict__ vals_in,
                                                         unsigned* __restrict__ keys_out, unsigned long long* __restrict__ vals_out,
                                                         size_t n, int shift, int bits, int n_tiles,
                                                         const unsigned* __restrict__ hist, const unsigned* __restrict__ totals) {
    __shared__ unsigned run[RS_MAX_BINS];          // next output position of every bin for this tile
    __shared__ unsigned wcnt[4][RS_MAX_BINS];      // per-wave count of the current round, tagged with the round: (r + 1) << 16 | count
    __shared__ unsigned wsum[4];
    const int bins = 1 << bits, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {   // base[bin] = exclusive scan of the bins' totals (two bins per thread), + this tile's offset inside the bin
        const int b0 = 2 * threadIdx.x, b1 = b0 + 1;
        const unsigned t0 = b0 < bins ? totals[b0] : 0u, t1 = b1 < bins ? totals[b1] : 0u;
        unsigned tot;
        const unsigned incl = rs_block_scan(t0 + t1, wsum, &tot);
        const unsigned ex = incl - (t0 + t1);
        if (b0 < bins) run[b0] = ex + hist[(size_t)b0 * n_tiles + blockIdx.x];
        if (b1 < bins) run[b1] = ex + t0 + hist[(size_t)b1 * n_tiles + blockIdx.x];
        for (int i = threadIdx.x; i < 4 * RS_MAX_BINS; i += 256) (&wcnt[0][0])[i] = 0u;
    }
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RS_TILE;
    unsigned k[RS_ITEMS];
    unsigned long long v[RS_ITEMS];
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {   // all loads of the tile in flight together
        const size_t e = base + r * 256 + threadIdx.x;
        k[r] = e < n ? keys_in[e] : 0u;
        v[r] = e < n ? vals_in[e] : 0ull;
    }
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const size_t e = base + r * 256 + threadIdx.x;
        const bool valid = e < n;
        const unsigned d = (k[r] >> shift) & (unsigned)(bins - 1);
        const unsigned long long m = rs_match(d, valid, bits);
        const unsigned before = (unsigned)__popcll(m & ((1ull << lane) - 1ull)), cnt = (unsigned)__popcll(m);
        const unsigned tag = (unsigned)(r + 1) << 16;
        if (valid && before == 0u) wcnt[wave][d] = tag | cnt;
        __syncthreads();
        unsigned pos = 0u;
        if (valid) {
            pos = run[d] + before;
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const unsigned c = wcnt[w][d];
                if (w < wave && (c >> 16) == (unsigned)(r + 1)) pos += c & 0xffffu;
            }
        }
        __syncthreads();
        if (valid && before == 0u) atomicAdd(&run[d], cnt);      // (<= 4 adds per bin and round)
        if (valid) {
            keys_out[pos] = k[r];
            vals_out[pos] = v[r];
        }
        __syncthreads();
    }
}

struct RsPlan {
    int passes, bits, n_tiles;
    size_t temp_bytes;
};
static RsPlan rs_plan(size_t n, int key_bits) {
    RsPlan p;
    p.passes = (key_bits + RS_MAX_BITS - 1) / RS_MAX_BITS;
    p.bits = (key_bits + p.passes - 1) / p.passes;
    p.n_tiles = (int)((n + RS_TILE - 1) / RS_TILE);
    p.temp_bytes = ((size_t)(1 << p.bits) * (size_t)std::max(p.n_tiles, 1) + RS_MAX_BINS) * sizeof(unsigned);
    return p;
}
// sorts (keys0, vals0) by the low key_bits of the keys, ping-ponging with (keys1, vals1); *sorted_in = 0 / 1
static int rs_sort_pairs(unsigned* keys0, unsigned* keys1, unsigned long long* vals0, unsigned long long* vals1, size_t n,
                         int key_bits, void* temp, size_t temp_bytes, int* sorted_in, hipStream_t s) {
    const RsPlan p = rs_plan(n, key_bits);
    *sorted_in = 0;
    if (n == 0) return 0;
    if (temp == nullptr || temp_bytes < p.temp_bytes || n >= (1ull << 32)) return (int)hipErrorInvalidValue;
    unsigned* hist = reinterpret_cast<unsigned*>(temp);
    unsigned* totals = hist + (size_t)(1 << p.bits) * p.n_tiles;
    unsigned* k[2] = {keys0, keys1};
    unsigned long long* v[2] = {vals0, vals1};
    int cur = 0;
    for (int pass = 0; pass < p.passes; ++pass) {
        const int shift = pass * p.bits, bits = std::min(p.bits, key_bits - shift);
        hipLaunchKernelGGL(rs_hist_kernel, dim3(p.n_tiles), dim3(256), 0, s, k[cur], n, shift, bits, p.n_tiles, hist);
        hipLaunchKernelGGL(rs_scan_rows_kernel, dim3(1 << bits), dim3(256), 0, s, hist, p.n_tiles, totals);
        hipLaunchKernelGGL(rs_scatter_kernel, dim3(p.n_tiles), dim3(256), 0, s, k[cur], v[cur], k[1 - cur], v[1 - cur], n, shift,
                           bits, p.n_tiles, hist, totals);
        SM_LAUNCH_CHECK();
        cur = 1 - cur;
    }
    *sorted_in = cur;
    return 0;
}

struct GatherLevels {
    const float* gimg[SM_MAX_TEX_LAYERS];
    int plane[SM_MAX_TEX_LAYERS];
    int first[SM_MAX_TEX_LAYERS + 1];   // prefix sums of the planes: position of level k's pixels in the packed copy
    int n;
};

// The image gradients are planar ([3][plane] per level): gathering them per entry costs three scattered 4-byte
// requests. One pass packs them into one float4 per pixel (25 MB for a c3 view), so that an entry is ONE 16-byte request.
__global__ __launch_bounds__(256) void scatter_pack_kernel(GatherLevels G, f32x4* __restrict__ packed) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= G.first[G.n]) return;
    int level = 0;
#pragma unroll
    for (int k = 1; k < SM_MAX_TEX_LAYERS; ++k)
        if (k < G.n && i >= G.first[k]) level = k;
    const int q = i - G.first[level];
    const float* p = G.gimg[level] + q;
    const int plane = G.plane[level];
    packed[i] = f32x4{p[0], p[plane], p[2 * (size_t)plane], 0.f};
}

// Runs of equal texels that cross a 64-entry chunk boundary. Found once per view (cross_list_kernel); every step the
// chunks write their pieces of such runs to `part` and cross_fix_kernel - one thread per listed run - adds the pieces
// in chunk order and owns the texel: one writer per texel, bit-reproducible. Only monster runs (thousands of pixels
// clamped onto one border texel) are listed as several segments of at most CROSS_GROUP chunks, which then add
// atomically - a serial walk over such a run would take longer than the whole scatter.
constexpr unsigned CROSS_GROUP = 32;
struct CrossRun {
    unsigned key;
    unsigned first_chunk;   // first piece: chunk first_chunk's LAST run (part[2 c + 1]) if the run starts there,
                            // its FIRST run (part[2 c]) if this is a later segment of a monster run ...
                            // (a segment = the run's pieces in one aligned group of CROSS_GROUP chunks)
    unsigned pieces;        // bits 0-15: pieces (the further ones are the chunks' FIRST runs); bit 16: first piece
                            // is a FIRST run; bit 17: the run has several segments (add atomically)
};

__global__ __launch_bounds__(256) void cross_list_kernel(const unsigned* __restrict__ keys, size_t n, unsigned invalid,
                                                         CrossRun* __restrict__ list, unsigned* __restrict__ count) {
    const size_t b = (size_t)blockIdx.x * 256 + threadIdx.x + 1;   // boundary between chunks b - 1 and b
    bool emit = false;
    unsigned k = invalid, word = 0;
    size_t first = 0;
    if (b * 64 < n) {
        k = keys[b * 64];
        if (k != invalid && keys[b * 64 - 1] == k) {                 // a run crosses this boundary
            const bool starts_here = !(b >= 2 && keys[(b - 1) * 64 - 1] == k);   // ... and starts inside chunk b - 1
            const bool group_head = b % CROSS_GROUP == 0;            // ... or chunk b opens a group of chunks
            if (starts_here || group_head) {
                // segment: [chunk b - 1's last run, if the run starts there] + the first runs of chunks b, b + 1, ...
                // up to the end of chunk b's group
                emit = true;
                first = starts_here ? b - 1 : b;
                const size_t group_end = (b / CROSS_GROUP + 1) * CROSS_GROUP;   // first chunk of the next group
                unsigned pieces = starts_here ? 1 : 0;
                size_t j = b;
                for (; j < group_end && j * 64 < n && keys[j * 64] == k; ++j) ++pieces;
                const bool goes_on = j == group_end && j * 64 < n && keys[j * 64] == k;
                word = pieces | (starts_here ? 0u : 1u << 16) | ((!starts_here || goes_on) ? 1u << 17 : 0u);
            }
        }
    }
    // one counter update per wave (same-address atomics serialise)
    const unsigned long long m = __ballot(emit);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    unsigned slot0 = 0;
    if (lane == leader) slot0 = atomicAdd(count, (unsigned)__popcll(m));
    slot0 = __shfl(slot0, leader, 64);
    if (emit) list[slot0 + __popcll(m & ((1ull << lane) - 1ull))] = CrossRun{k, (unsigned)first, word};
}

__device__ __forceinline__ size_t channel_stride(const PlanLayers& L, unsigned key) {
    // static indexing of the argument arrays (a per-lane index would send the struct through scratch memory)
    size_t cs = (size_t)L.w[0] * L.h[0];
#pragma unroll
    for (int l = 1; l < SM_MAX_TEX_LAYERS; ++l)
        if (l < L.n && key >= L.base[l]) cs = (size_t)L.w[l] * L.h[l];
    return cs;
}

// One lane per sorted entry, four 64-entry chunks per wave: a chunk's work is a chain of dependent memory round trips
// (keys / values -> the pixel's gradient -> the texel), so the loads of all four chunks are issued together before
// anything waits. ACCUMULATE = false: the arena is known to be zero (the fused update zeroes it), runs that lie
// inside a chunk store their sum without reading the texel first.
template <bool ACCUMULATE>
__global__ __launch_bounds__(256) void scatter_sorted_kernel(const unsigned* __restrict__ keys,
                                                             const unsigned long long* __restrict__ vals, size_t n,
                                                             GatherLevels G, const f32x4* __restrict__ packed,
                                                             PlanLayers L, float* __restrict__ arena,
                                                             f32x4* __restrict__ part, unsigned invalid) {
    constexpr int U = 4;
    const int lane = threadIdx.x & 63;
    const size_t base = (((size_t)blockIdx.x * 256 + threadIdx.x) >> 6) * (64 * U);   // first entry of this wave
    if (base >= n) return;
    // sorted: the invalid entries (taps without weight: ~a third of a c3 view's list) are the tail - a wave whose first
    // entry is one of them has nothing to do and leaves on one wave-uniform load instead of streaming its keys and values
    if (keys[base] == invalid) return;
    unsigned key[U], kprev[U], knext[U];
    unsigned long long val[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t c0 = base + u * 64, e = c0 + lane;
        key[u] = e < n ? keys[e] : invalid;
        val[u] = e < n ? vals[e] : 0ull;
        // the keys next to the chunk's 64 entries (wave-uniform loads)
        kprev[u] = c0 > 0 && c0 - 1 < n ? keys[c0 - 1] : invalid;
        knext[u] = c0 + 64 < n ? keys[c0 + 64] : invalid;
    }
    f32x4 gv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned lo = (unsigned)val[u];
        const int q = lo & 0xFFFFFF, level = (lo >> 24) & 7;
        gv[u] = key[u] != invalid ? packed[G.first[level] + q] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned k = key[u];
        if (__ballot(k != invalid) == 0ull) break;   // sorted: the invalid entries are the tail
        const float wgt = __uint_as_float((unsigned)(val[u] >> 32));
        float g0 = wgt * gv[u][0], g1 = wgt * gv[u][1], g2 = wgt * gv[u][2];
        // inclusive segmented sum over runs of equal keys (sorted: equal keys are adjacent)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned ku = __shfl_up(k, d, 64);
            const float a0 = __shfl_up(g0, d, 64), a1 = __shfl_up(g1, d, 64), a2 = __shfl_up(g2, d, 64);
            if (lane >= d && ku == k) { g0 += a0; g1 += a1; g2 += a2; }
        }
        const unsigned kn = __shfl_down(k, 1, 64);
        const unsigned kfirst = __shfl(k, 0, 64);
        const bool tail = k != invalid && (lane == 63 || kn != k);
        if (tail) {
            const size_t chunk = base / 64 + u;
            if (k == kfirst && kprev[u] == k) {             // continues a run of the previous chunk
                part[2 * chunk] = f32x4{g0, g1, g2, 0.f};
            } else if (lane == 63 && knext[u] == k) {       // starts here, continues in the next chunk
                part[2 * chunk + 1] = f32x4{g0, g1, g2, 0.f};
            } else {                                         // the run lies inside the chunk: it owns its texel
                float* dst = arena + k;
                const size_t cs = channel_stride(L, k);
                if (ACCUMULATE) {
                    dst[0] += g0;
                    dst[cs] += g1;
                    dst[2 * cs] += g2;
                } else {
                    dst[0] = g0;
                    dst[cs] = g1;
                    dst[2 * cs] = g2;
                }
            }
        }
    }
}

template <bool ACCUMULATE>
__global__ __launch_bounds__(256) void cross_fix_kernel(const CrossRun* __restrict__ list, const unsigned* __restrict__ count,
                                                        const f32x4* __restrict__ part, PlanLayers L,
                                                        float* __restrict__ arena) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i >= *count) return;
    const CrossRun r = list[i];
    const unsigned pieces = r.pieces & 0xFFFF;
    f32x4 sum = part[2 * (size_t)r.first_chunk + ((r.pieces >> 16) & 1 ? 0 : 1)];
    for (unsigned j = 1; j < pieces; ++j) sum += part[2 * ((size_t)r.first_chunk + j)];
    float* dst = arena + r.key;
    const size_t cs = channel_stride(L, r.key);
    if ((r.pieces >> 17) & 1) {          // one of several segments of a monster run
        atomicAdd(dst, sum[0]);
        atomicAdd(dst + cs, sum[1]);
        atomicAdd(dst + 2 * cs, sum[2]);
    } else if (ACCUMULATE) {
        dst[0] += sum[0];
        dst[cs] += sum[1];
        dst[2 * cs] += sum[2];
    } else {
        dst[0] = sum[0];
        dst[cs] = sum[1];
        dst[2 * cs] = sum[2];
    }
}

}  // namespace sm

extern "C" {

size_t sm_tex_scatter_plan_temp_bytes(size_t n_entries, int key_bits) {
    if (key_bits < 1 || key_bits > 32) return 0;
    return sm::rs_plan(n_entries, key_bits).temp_bytes;
}

int sm_radix_sort_pairs(uint32_t* keys0, uint32_t* keys1, uint64_t* vals0, uint64_t* vals1, size_t n, int key_bits,
                        void* temp, size_t temp_bytes, int* sorted_in, void* stream) {
    if (key_bits < 1 || key_bits > 32 || sorted_in == nullptr) return (int)hipErrorInvalidValue;
    return sm::rs_sort_pairs(keys0, keys1, reinterpret_cast<unsigned long long*>(vals0),
                             reinterpret_cast<unsigned long long*>(vals1), n, key_bits, temp, temp_bytes, sorted_in,
                             (hipStream_t)stream);
}

/* cross buffer layout: [count u32, 3 pad][CrossRun list: n_chunks][part: 2 n_chunks float4] */
static size_t cross_chunks(size_t n_entries) { return (n_entries + 63) / 64; }
size_t sm_tex_scatter_plan_cross_bytes(size_t n_entries) {
    const size_t c = cross_chunks(n_entries);
    return 16 + ((c * sizeof(sm::CrossRun) + 15) / 16) * 16 + c * 2 * sizeof(sm::f32x4);
}
static sm::CrossRun* cross_list_ptr(void* cross) { return reinterpret_cast<sm::CrossRun*>(static_cast<char*>(cross) + 16); }
static sm::f32x4* cross_part_ptr(void* cross, size_t n_entries) {
    const size_t c = cross_chunks(n_entries);
    return reinterpret_cast<sm::f32x4*>(static_cast<char*>(cross) + 16 + ((c * sizeof(sm::CrossRun) + 15) / 16) * 16);
}

static int fill_layers(sm::PlanLayers& L, float* const* grad_layers, const int* layer_w, const int* layer_h, int n_layers,
                       const float* arena_base) {
    if (n_layers < 1 || n_layers > SM_MAX_TEX_LAYERS) return (int)hipErrorInvalidValue;
    L.n = n_layers;
    for (int l = 0; l < n_layers; ++l) {
        const ptrdiff_t off = grad_layers[l] - arena_base;
        if (off < 0 || off > 0x7fffffff) return (int)hipErrorInvalidValue;
        L.base[l] = (unsigned)off;
        L.w[l] = layer_w[l];
        L.h[l] = layer_h[l];
    }
    return 0;
}

int sm_tex_scatter_plan(float* const* grad_layers, const int* layer_w, const int* layer_h, int n_layers,
                        const float* arena_base, const float* const* grids, const float* const* pixel_weights,
                        const int* level_h, const int* level_w, int n_levels, uint32_t* keys0, uint32_t* keys1,
                        uint64_t* vals0, uint64_t* vals1, void* temp, size_t temp_bytes, void* cross, int key_bits,
                        int* sorted_in, void* stream) {
    if (n_levels < 1 || n_levels > SM_MAX_TEX_LAYERS || key_bits < 1 || key_bits > 32) return (int)hipErrorInvalidValue;
    sm::PlanLayers L;
    if (int e = fill_layers(L, grad_layers, layer_w, layer_h, n_layers, arena_base)) return e;
    sm::PlanLevels V;
    V.n = n_levels;
    int total = 0;
    for (int k = 0; k < n_levels; ++k) {
        if ((size_t)(level_h[k] + 2) * sm::row_stride(level_w[k]) >= (1u << 24)) return (int)hipErrorInvalidValue;
        V.lv[k] = sm::PlanLevel{reinterpret_cast<const float2*>(grids[k]), pixel_weights ? pixel_weights[k] : nullptr,
                                level_h[k], level_w[k], total};
        total += level_h[k] * level_w[k];
    }
    V.total_pixels = total;
    const unsigned invalid = key_bits == 32 ? 0xffffffffu : ((1u << key_bits) - 1u);
    hipStream_t s = (hipStream_t)stream;
    const size_t threads = (size_t)total * n_layers;
    hipLaunchKernelGGL(sm::scatter_entries_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, V, L, keys0,
                       reinterpret_cast<unsigned long long*>(vals0), invalid);
    SM_LAUNCH_CHECK();
    const size_t n = threads * 4;
    if (int e = sm::rs_sort_pairs(keys0, keys1, reinterpret_cast<unsigned long long*>(vals0),
                                  reinterpret_cast<unsigned long long*>(vals1), n, key_bits, temp, temp_bytes, sorted_in, s))
        return e;
    const unsigned* sorted_keys = *sorted_in ? keys1 : keys0;
    // runs that cross chunk boundaries (static per view)
    if (hipError_t e2 = hipMemsetAsync(cross, 0, 16, s); e2 != hipSuccess) return (int)e2;
    const size_t boundaries = cross_chunks(n);
    hipLaunchKernelGGL(sm::cross_list_kernel, dim3((unsigned)((boundaries + 255) / 256)), dim3(256), 0, s, sorted_keys, n,
                       invalid, cross_list_ptr(cross), reinterpret_cast<unsigned*>(cross));
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_tex_scatter_planned(const uint32_t* keys, const uint64_t* vals, size_t n_entries, const float* const* grad_imgs,
                           const int* level_h, const int* level_w, int n_levels, float* const* grad_layers,
                           const int* layer_w, const int* layer_h, int n_layers, float* arena_base, int key_bits,
                           float* packed_scratch, void* cross, int accumulate, void* stream) {
    if (n_levels < 1 || n_levels > SM_MAX_TEX_LAYERS || packed_scratch == nullptr || cross == nullptr)
        return (int)hipErrorInvalidValue;
    sm::PlanLayers L;
    if (int e = fill_layers(L, grad_layers, layer_w, layer_h, n_layers, arena_base)) return e;
    sm::GatherLevels G;
    G.n = n_levels;
    G.first[0] = 0;
    for (int k = 0; k < n_levels; ++k) {
        G.gimg[k] = grad_imgs[k];
        G.plane[k] = sm::plane_size(level_h[k], level_w[k]);
        G.first[k + 1] = G.first[k] + G.plane[k];
    }
    const unsigned invalid = key_bits == 32 ? 0xffffffffu : ((1u << key_bits) - 1u);
    sm::f32x4* packed = reinterpret_cast<sm::f32x4*>(packed_scratch);
    hipLaunchKernelGGL(sm::scatter_pack_kernel, dim3((unsigned)((G.first[n_levels] + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, G, packed);
    const dim3 grid((unsigned)((n_entries + 1023) / 1024));   // 4 chunks of 64 entries per wave
    const unsigned long long* v64 = reinterpret_cast<const unsigned long long*>(vals);
    sm::f32x4* part = cross_part_ptr(cross, n_entries);
    const dim3 fgrid((unsigned)((cross_chunks(n_entries) + 255) / 256));
    hipStream_t s = (hipStream_t)stream;
    if (accumulate) {
        hipLaunchKernelGGL(sm::scatter_sorted_kernel<true>, grid, dim3(256), 0, s, keys, v64, n_entries, G, packed, L,
                           arena_base, part, invalid);
        hipLaunchKernelGGL(sm::cross_fix_kernel<true>, fgrid, dim3(256), 0, s, cross_list_ptr(cross),
                           reinterpret_cast<const unsigned*>(cross), part, L, arena_base);
    } else {
        hipLaunchKernelGGL(sm::scatter_sorted_kernel<false>, grid, dim3(256), 0, s, keys, v64, n_entries, G, packed, L,
                           arena_base, part, invalid);
        hipLaunchKernelGGL(sm::cross_fix_kernel<false>, fgrid, dim3(256), 0, s, cross_list_ptr(cross),
                           reinterpret_cast<const unsigned*>(cross), part, L, arena_base);
    }
    SM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

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
// 0.659 -> 0.635 of dense; the exact need is 0.589). One block per need map: all threads pack the map into one bit per
// position in LDS (<= 103 KB for a 784 x 1045 plane), then ONE lane walks the bits - a chain of <= positions / 32 steps of
// LDS latency each, all maps of all levels in one launch, once per view (in prepare_view: beside the previous view's steps).
// ---------------------------------------------------------------------------------------------------
#define SM_COVER_MAX 64
struct CoverProblem {
    const float* need;     // [h][w] 0 / 1
    int32_t* starts;       // out: (tag << 24) | first position q of each segment (index into the padded plane)
    int32_t* count;        // out: number of segments
    int h, w, tag, cap;
    int pair_w;
};
struct CoverGroup {
    CoverProblem p[SM_COVER_MAX];
};

// PAIR mode (P.pair_w > 0): `need` is the need map of a POOLED plane; row Y of it is covered with runs of 16 windows
// (one thread per row: count, exclusive scan over the rows, write), each run emitted as the two 32-position segments of
// the full-resolution plane that hold its windows - rows 2Y and 2Y + 1, columns 2 X0 .. 2 X0 + 31.
__device__ void cover_pairs(const CoverProblem& P, uint32_t* bits) {
    const int rw = (P.w + 31) / 32 + 1;                 // words per pooled row (+ a zero word behind it)
    int* row_off = reinterpret_cast<int*>(bits + P.h * rw);
    for (int i = threadIdx.x; i < P.h * rw; i += 256) {
        const int Y = i / rw, wd = i - Y * rw;
        uint32_t v = 0;
        for (int b = 0; b < 32; ++b) {
            const int X = wd * 32 + b;
            if (X < P.w && P.need[(size_t)Y * P.w + X] > 0.f) v |= 1u << b;
        }
        bits[i] = v;
    }
    __syncthreads();
    for (int Y = threadIdx.x; Y < P.h; Y += 256) {      // segments of row Y
        int n = 0, cursor = 0;
        while (cursor < P.w) {
            const int wd = cursor >> 5, sh = cursor & 31;
            const uint64_t win = ((uint64_t)bits[Y * rw + wd] | ((uint64_t)(wd + 1 < rw ? bits[Y * rw + wd + 1] : 0u) << 32)) >> sh;
            if (win == 0) { cursor += 64 - sh; continue; }
            cursor += __builtin_ctzll(win) + 16;
            ++n;
        }
        row_off[Y] = n;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int Y = 0; Y < P.h; ++Y) { const int n = row_off[Y]; row_off[Y] = tot; tot += n; }
        *P.count = 2 * tot;
    }
    __syncthreads();
    const int Wp = row_stride(P.pair_w);
    for (int Y = threadIdx.x; Y < P.h; Y += 256) {
        int n = row_off[Y], cursor = 0;
        while (cursor < P.w) {
            const int wd = cursor >> 5, sh = cursor & 31;
            const uint64_t win = ((uint64_t)bits[Y * rw + wd] | ((uint64_t)(wd + 1 < rw ? bits[Y * rw + wd + 1] : 0u) << 32)) >> sh;
            if (win == 0) { cursor += 64 - sh; continue; }
            const int X0 = cursor + __builtin_ctzll(win);
            const int q = (2 * Y + 1) * Wp + 2 * X0 + 1;
            if (2 * n + 1 < P.cap) {
                P.starts[2 * n] = (P.tag << 24) | q;
                P.starts[2 * n + 1] = (P.tag << 24) | (q + Wp);
            }
            ++n;
            cursor = X0 + 16;
        }
    }
}

__global__ __launch_bounds__(256) void cover_segments_kernel(CoverGroup g) {
    extern __shared__ uint32_t bits[];
    const CoverProblem P = g.p[blockIdx.x];
    if (P.pair_w > 0) { cover_pairs(P, bits); return; }
    const int Wp = row_stride(P.w);
    const int n_pos = P.h * Wp;                       // positions of rows 1 .. h, relative to q = Wp
    const int n_words = (n_pos + 31) / 32 + 2;        // + a zero tail the 64-bit window may read
    for (int wd = threadIdx.x; wd < n_words; wd += 256) {
        uint32_t v = 0;
        const int base = wd * 32;
        if (base < n_pos) {
            int r = base / Wp, x = base - r * Wp;
#pragma unroll 4
            for (int b = 0; b < 32; ++b) {
                if (base + b < n_pos && x >= 1 && x <= P.w && P.need[(size_t)r * P.w + x - 1] > 0.f) v |= 1u << b;
                if (++x == Wp) { x = 0; ++r; }
            }
        }
        bits[wd] = v;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    int n = 0, cursor = 0;                            // cursor: a multiple of 4; everything before it is covered or dead
    while (cursor < n_pos) {
        const int wd = cursor >> 5, sh = cursor & 31;
        const uint64_t lo = bits[wd], mid = bits[wd + 1], hi = bits[wd + 2];
        const uint64_t win = sh ? (((lo | (mid << 32)) >> sh) | (hi << (64 - sh))) : (lo | (mid << 32));
        if (win == 0) { cursor += 64; continue; }
        const int p = cursor + __builtin_ctzll(win);
        const int st = p & ~3;                        // >= cursor
        if (n < P.cap) P.starts[n] = (P.tag << 24) | (Wp + st);
        ++n;
        cursor = st + 32;
    }
    *P.count = n;
}

}  // namespace sm

extern "C" {

int sm_cover_segments(const sm_cover_problem* problems, int n, void* stream) {
    if (n < 1 || n > SM_COVER_MAX) return (int)hipErrorInvalidValue;
    sm::CoverGroup g;
    size_t lds = 0;
    for (int i = 0; i < n; ++i) {
        if (problems[i].h < 1 || problems[i].w < 1 || problems[i].tag < 0 || problems[i].tag > 127) return (int)hipErrorInvalidValue;
        if (problems[i].pair_w != 0 && problems[i].pair_w / 2 != problems[i].w) return (int)hipErrorInvalidValue;
        g.p[i] = sm::CoverProblem{problems[i].need, problems[i].starts, problems[i].count, problems[i].h, problems[i].w,
                                  problems[i].tag, problems[i].cap, problems[i].pair_w};
        const size_t words = problems[i].pair_w > 0
            ? (size_t)problems[i].h * ((problems[i].w + 31) / 32 + 1) + problems[i].h     // bit rows + row offsets
            : ((size_t)problems[i].h * sm::row_stride(problems[i].w) + 31) / 32 + 2;
        lds = words * 4 > lds ? words * 4 : lds;
    }
    if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
    static size_t lds_set = 0;
    if (lds > lds_set) {   // > 64 KB of dynamic LDS needs the opt-in
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sm::cover_segments_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        lds_set = lds;
    }
    hipLaunchKernelGGL(sm::cover_segments_kernel, dim3(n), dim3(256), lds, (hipStream_t)stream, g);
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

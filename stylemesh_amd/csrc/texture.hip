// K1 / K2 / K7: texture sampling, its scatter-add backward, and the fused regulariser + Adam + clamp update.
//
// Reference operators replaced (lukasHoel/stylemesh): F.grid_sample(bilinear, border, align_corners=True) and
// its backward (model/texture/texture.py:46-54,96-100), the angle / depth gradient hooks
// (model/model.py:195-202,245-251), HierarchicalNeuralTexture.regularizer (texture.py:102-108),
// NeuralTexture.normalize (texture.py:41-44) and torch.optim.Adam.step (model/model.py:387-395).
// All HBM-bound: one pass over the view's pixels (K1, K2) or over the texture arena (K7).
#include "common.h"

namespace sm {

struct TexLayers {
    float* p[SM_MAX_TEX_LAYERS];
    int w[SM_MAX_TEX_LAYERS];
    int h[SM_MAX_TEX_LAYERS];
    int n;
};

// ATen grid_sampler source index (align_corners=True) + border clip; returns the 4 tap weights in ATen's
// order nw, ne, sw, se and the integer corner.
struct Taps {
    int x0, y0;
    float nw, ne, sw, se;
    bool x1_in, y1_in;
};

__device__ __forceinline__ Taps make_taps(float gx, float gy, int W, int H) {
    float ix = ((gx + 1.f) / 2.f) * (float)(W - 1);
    float iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
    ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
    iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
    const float fx = floorf(ix), fy = floorf(iy);
    Taps t;
    t.x0 = (int)fx;
    t.y0 = (int)fy;
    const float ex = fx + 1.f, ey = fy + 1.f;  // south-east corner
    t.nw = (ex - ix) * (ey - iy);
    t.ne = (ix - fx) * (ey - iy);
    t.sw = (ex - ix) * (iy - fy);
    t.se = (ix - fx) * (iy - fy);
    t.x1_in = t.x0 + 1 <= W - 1;
    t.y1_in = t.y0 + 1 <= H - 1;
    return t;
}

struct __attribute__((aligned(4))) pair4 { float x, y; };   // two adjacent floats at 4-byte alignment

#ifndef SM_TEX_FWD_BATCH
#define SM_TEX_FWD_BATCH 1      // 1: the gathers of up to four texture layers are issued together, branch-free (round 4)
#endif
// Layers l0 .. l0 + NB - 1 of one pixel with ALL their 6 NB tap loads in flight at once. The loop below issues a layer's
// six gathers, waits, and only then computes the next layer's addresses: four dependent memory round trips per pixel for
// the hierarchical texture. Branch-free: at the right / bottom border the east / south weights are exactly 0 (the
// source index is clamped to W - 1 / H - 1), so the pair is loaded one texel to the left / the north row again and the
// zero-weight products are added - the same sums as the branching form (a zero may change its sign).
template <int NB>
__device__ __forceinline__ void tex_sample_layers(const TexLayers& L, int l0, float2 g, float& acc0, float& acc1, float& acc2) {
    pair4 a[NB][3], b[NB][3];
    Taps t[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int W = L.w[l0 + k], H = L.h[l0 + k];
        t[k] = make_taps(g.x, g.y, W, H);
        const float* p = L.p[l0 + k] + (size_t)t[k].y0 * W + t[k].x0 - (t[k].x1_in ? 0 : 1);
        const size_t cs = (size_t)W * H;
        const int dy = t[k].y1_in ? W : 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a[k][c] = *reinterpret_cast<const pair4*>(p + c * cs);
            b[k][c] = *reinterpret_cast<const pair4*>(p + c * cs + dy);
        }
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float west = t[k].x1_in ? a[k][c].x : a[k][c].y, swest = t[k].x1_in ? b[k][c].x : b[k][c].y;
            v[c] = west * t[k].nw;
            v[c] += a[k][c].y * t[k].ne;
            v[c] += swest * t[k].sw;
            v[c] += b[k][c].y * t[k].se;
        }
        acc0 += v[0];
        acc1 += v[1];
        acc2 += v[2];
    }
}

__device__ __forceinline__ void tex_sample_fwd_body(const TexLayers& L, const float2* __restrict__ grid, int h, int w,
                                                    float* __restrict__ out, int Wp, int plane, int block_x) {
    const int i = block_x * 256 + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    const float2 g = grid[i];
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
    int l_first = 0;
#if SM_TEX_FWD_BATCH
    bool wide = true;                 // (a 1-texel-wide layer has no texel to the left of its border)
    for (int l = 0; l < L.n; ++l) wide = wide && L.w[l] >= 2;
    if (wide) {
        for (; l_first + 4 <= L.n; l_first += 4) tex_sample_layers<4>(L, l_first, g, acc0, acc1, acc2);
        if (l_first + 2 <= L.n) { tex_sample_layers<2>(L, l_first, g, acc0, acc1, acc2); l_first += 2; }
        if (l_first + 1 <= L.n) { tex_sample_layers<1>(L, l_first, g, acc0, acc1, acc2); l_first += 1; }
    }
#endif
    for (int l = l_first; l < L.n; ++l) {
        const int W = L.w[l], H = L.h[l];
        const Taps t = make_taps(g.x, g.y, W, H);
        const float* p = L.p[l] + (size_t)t.y0 * W + t.x0;
        const size_t cs = (size_t)W * H;
        // the west / east taps of a row are adjacent texels: one 8-byte load (4-byte aligned) instead of two gathers.
        // Same products and the same order of additions as the scalar form.
        float v0, v1, v2;
        if (t.x1_in) {
            const pair4 a0 = *reinterpret_cast<const pair4*>(p), a1 = *reinterpret_cast<const pair4*>(p + cs),
                        a2 = *reinterpret_cast<const pair4*>(p + 2 * cs);
            v0 = a0.x * t.nw; v1 = a1.x * t.nw; v2 = a2.x * t.nw;
            v0 += a0.y * t.ne; v1 += a1.y * t.ne; v2 += a2.y * t.ne;
            if (t.y1_in) {
                const pair4 b0 = *reinterpret_cast<const pair4*>(p + W), b1 = *reinterpret_cast<const pair4*>(p + cs + W),
                            b2 = *reinterpret_cast<const pair4*>(p + 2 * cs + W);
                v0 += b0.x * t.sw; v1 += b1.x * t.sw; v2 += b2.x * t.sw;
                v0 += b0.y * t.se; v1 += b1.y * t.se; v2 += b2.y * t.se;
            }
        } else {
            v0 = p[0] * t.nw;
            v1 = p[cs] * t.nw;
            v2 = p[2 * cs] * t.nw;
            if (t.y1_in) {
                v0 += p[W] * t.sw;
                v1 += p[cs + W] * t.sw;
                v2 += p[2 * cs + W] * t.sw;
            }
        }
        acc0 += v0;
        acc1 += v1;
        acc2 += v2;
    }
    const size_t q = (size_t)(y + 1) * Wp + x + 1;
    out[q] = acc0;
    out[plane + q] = acc1;
    out[2 * (size_t)plane + q] = acc2;
}

__global__ __launch_bounds__(256) void tex_sample_fwd_kernel(TexLayers L, const float2* __restrict__ grid, int h, int w,
                                                             float* __restrict__ out, int Wp, int plane) {
    tex_sample_fwd_body(L, grid, h, w, out, Wp, plane, blockIdx.x);
}

// The UV levels of a view in ONE launch (four gather-bound launches of 11-22 us each otherwise)
constexpr int TEX_MAX_GROUP = 8;
struct TexSampleGroup {
    const float2* grid[TEX_MAX_GROUP];
    float* out[TEX_MAX_GROUP];
    int h[TEX_MAX_GROUP], w[TEX_MAX_GROUP];
    int block_begin[TEX_MAX_GROUP + 1];
    int n;
};
__global__ __launch_bounds__(256) void tex_sample_fwd_group_kernel(TexLayers L, TexSampleGroup G) {
    int g = 0;
#pragma unroll
    for (int i = 1; i < TEX_MAX_GROUP; ++i)
        if (i < G.n && (int)blockIdx.x >= G.block_begin[i]) g = i;
    const float2* grid = G.grid[0];
    float* out = G.out[0];
    int h = G.h[0], w = G.w[0];
#pragma unroll
    for (int i = 1; i < TEX_MAX_GROUP; ++i)
        if (i == g) { grid = G.grid[i]; out = G.out[i]; h = G.h[i]; w = G.w[i]; }
    tex_sample_fwd_body(L, grid, h, w, out, row_stride(w), plane_size(h, w), blockIdx.x - G.block_begin[g]);
}

// K2, tiled: one block = a 16x16 pixel tile of the view. Neighbouring pixels hit neighbouring (coarse layers: the
// same) texels, and a device-scope atomic costs a fabric transaction whether or not it shares a cache line, so the
// tile's contributions to one layer are first summed in an LDS window of WS x WS texels (48: 28 KB) anchored at the tile's
// smallest (x0, y0) (ds_add_f32), and only the window's non-zero texels go to memory - as row-contiguous atomics.
// Pixels whose 2x2 taps fall outside the window (UV seams, strongly magnified views) use the direct path.
template <int WS>
__global__ __launch_bounds__(256) void tex_sample_bwd_tiled_kernel(TexLayers L, const float2* __restrict__ grid, int h,
                                                                   int w, const float* __restrict__ gimg,
                                                                   const float* __restrict__ pixel_weight, int Wp,
                                                                   int plane, int tiles_x) {
    __shared__ float win[3][WS][WS + 1];
    __shared__ int bb[2];
    const int tid = threadIdx.x;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int x = tx * 16 + (tid & 15), y = ty * 16 + (tid >> 4);
    bool valid = x < w && y < h;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    float2 g = make_float2(0.f, 0.f);
    if (valid) {
        const size_t q = (size_t)(y + 1) * Wp + x + 1;
        g0 = gimg[q]; g1 = gimg[plane + q]; g2 = gimg[2 * (size_t)plane + q];
        if (pixel_weight) {
            const float pw = pixel_weight[y * w + x];
            g0 *= pw; g1 *= pw; g2 *= pw;
        }
        valid = !(g0 == 0.f && g1 == 0.f && g2 == 0.f);   // adding zero is a no-op: skip the atomics
        if (valid) g = grid[y * w + x];
    }
    if (__syncthreads_or(valid) == 0) return;
    for (int i = tid; i < 3 * WS * (WS + 1); i += 256) (&win[0][0][0])[i] = 0.f;
    for (int l = 0; l < L.n; ++l) {
        const int W = L.w[l], H = L.h[l];
        const size_t cs = (size_t)W * H;
        const Taps t = make_taps(g.x, g.y, W, H);
        if (tid == 0) { bb[0] = 0x7fffffff; bb[1] = 0x7fffffff; }
        __syncthreads();   // also orders the previous layer's flush (window re-zeroed) before this layer's adds
        {   // block minimum of (x0, y0) over the contributing pixels: wave shuffle reduction, one LDS atomic per wave
            int mx = valid ? t.x0 : 0x7fffffff, my = valid ? t.y0 : 0x7fffffff;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mx = min(mx, __shfl_xor(mx, o, 64));
                my = min(my, __shfl_xor(my, o, 64));
            }
            if ((tid & 63) == 0) { atomicMin(&bb[0], mx); atomicMin(&bb[1], my); }
        }
        __syncthreads();
        const int wx = bb[0], wy = bb[1];
        if (valid) {
            const int dx = t.x0 - wx, dy = t.y0 - wy;
            if (dx + 1 < WS && dy + 1 < WS) {
                atomicAdd(&win[0][dy][dx], g0 * t.nw);
                atomicAdd(&win[1][dy][dx], g1 * t.nw);
                atomicAdd(&win[2][dy][dx], g2 * t.nw);
                if (t.x1_in) {
                    atomicAdd(&win[0][dy][dx + 1], g0 * t.ne);
                    atomicAdd(&win[1][dy][dx + 1], g1 * t.ne);
                    atomicAdd(&win[2][dy][dx + 1], g2 * t.ne);
                }
                if (t.y1_in) {
                    atomicAdd(&win[0][dy + 1][dx], g0 * t.sw);
                    atomicAdd(&win[1][dy + 1][dx], g1 * t.sw);
                    atomicAdd(&win[2][dy + 1][dx], g2 * t.sw);
                    if (t.x1_in) {
                        atomicAdd(&win[0][dy + 1][dx + 1], g0 * t.se);
                        atomicAdd(&win[1][dy + 1][dx + 1], g1 * t.se);
                        atomicAdd(&win[2][dy + 1][dx + 1], g2 * t.se);
                    }
                }
            } else {
                float* p = L.p[l] + (size_t)t.y0 * W + t.x0;
                atomicAdd(p, g0 * t.nw);
                atomicAdd(p + cs, g1 * t.nw);
                atomicAdd(p + 2 * cs, g2 * t.nw);
                if (t.x1_in) {
                    atomicAdd(p + 1, g0 * t.ne);
                    atomicAdd(p + cs + 1, g1 * t.ne);
                    atomicAdd(p + 2 * cs + 1, g2 * t.ne);
                }
                if (t.y1_in) {
                    atomicAdd(p + W, g0 * t.sw);
                    atomicAdd(p + cs + W, g1 * t.sw);
                    atomicAdd(p + 2 * cs + W, g2 * t.sw);
                    if (t.x1_in) {
                        atomicAdd(p + W + 1, g0 * t.se);
                        atomicAdd(p + cs + W + 1, g1 * t.se);
                        atomicAdd(p + 2 * cs + W + 1, g2 * t.se);
                    }
                }
            }
        }
        __syncthreads();
        // flush the window (and re-zero it for the next layer): consecutive threads -> consecutive texels of a row
        for (int i = tid; i < 3 * WS * WS; i += 256) {
            const int c = i / (WS * WS), r = i - c * WS * WS, yy = r / WS, xx = r - yy * WS;
            const float v = win[c][yy][xx];
            if (v != 0.f) {
                win[c][yy][xx] = 0.f;
                atomicAdd(L.p[l] + c * cs + (size_t)(wy + yy) * W + wx + xx, v);   // inside the texture: taps were
            }
        }
    }
}

// Which chunks of the gradient arena can a view's scatter write? One flag per 2^chunk_log2 floats, set for every
// texel any tap of a contributing pixel (pixel_weight != 0) lands on. Depends only on the view's UV grid and weights,
// so it is evaluated once per view; the multi-GPU path exchanges only the flagged chunks (runtime/distributed.py).
__global__ __launch_bounds__(256) void tex_touch_flags_kernel(TexLayers L, const float* base,
                                                              const float2* __restrict__ grid, int h, int w,
                                                              const float* __restrict__ pixel_weight, int* flags,
                                                              int chunk_log2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    if (pixel_weight && pixel_weight[i] == 0.f) return;
    const float2 g = grid[i];
    for (int l = 0; l < L.n; ++l) {
        const int W = L.w[l], H = L.h[l];
        const size_t cs = (size_t)W * H;
        const Taps t = make_taps(g.x, g.y, W, H);
        const size_t o = (size_t)(L.p[l] - base) + (size_t)t.y0 * W + t.x0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const size_t oc = o + c * cs;
            flags[oc >> chunk_log2] = 1;
            if (t.x1_in) flags[(oc + 1) >> chunk_log2] = 1;
            if (t.y1_in) {
                flags[(oc + W) >> chunk_log2] = 1;
                if (t.x1_in) flags[(oc + W + 1) >> chunk_log2] = 1;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// K7 fused update over the whole texture arena (all layers back to back), float4 per thread.
// ---------------------------------------------------------------------------------------------------
struct Segs {
    size_t end[SM_MAX_TEX_LAYERS];
    float reg[SM_MAX_TEX_LAYERS];
    int n;
};

__device__ __forceinline__ int seg_of(const Segs& s, size_t i) {
    int k = 0;
    while (k < s.n - 1 && i >= s.end[k]) ++k;
    return k;
}

__device__ __forceinline__ float block_sum(float v) {
    __shared__ float red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

#ifndef SM_ADAM_UNROLL
#define SM_ADAM_UNROLL 1        // tiles whose loads are in flight together. Measured in round 4 (profiles/r04/
                                // adam_variants.txt, 66.8 M floats, dense / 40 % / 16 % flagged): U = 1 432 / 199 / 78 us
                                // (4.34 TB/s dense), U = 2 450 / 198 / 83, U = 4 443 / 226 / 98, U = 8 587 / 329 / 158,
                                // U = 4 + non-temporal stores 429 / 211 / 96: the 8-stream update is not latency-bound -
                                // more loads in flight per thread only cost occupancy; it stays at 1
#endif
#ifndef SM_ADAM_STRIDED
#define SM_ADAM_STRIDED 2       // block b walks tiles b, b + G, b + 2 G, ... (G = blocks) instead of 16 consecutive ones:
                                // 0 never, 1 always, 2 for the DENSE update only (measured, profiles/r04/adam_strided.txt)
#endif
#ifndef SM_ADAM_NT
#define SM_ADAM_NT 0            // 1: non-temporal stores of p / m / v / zeroed g (A/B switch)
#endif
__device__ __forceinline__ void adam_store4(float* dst, const float* src) {
#if SM_ADAM_NT
    __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(src), reinterpret_cast<f32x4*>(dst));
#else
    *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
#endif
}

template <bool ADAM>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, Segs segs, float lr_over_bc1,
                                                   float one_minus_beta1, float beta2, float one_minus_beta2,
                                                   float eps, float inv_sqrt_bc2,
                                                   float grad_scale, float lo, float hi, int zero_grad,
                                                   float* __restrict__ sumsq, const float* __restrict__ dev_hyper,
                                                   int tiles_per_block, const int32_t* __restrict__ touched,
                                                   int touched_log2) {
    if (ADAM && dev_hyper) {   // step-dependent scalars from device memory: the launch can be replayed from a hipGraph
        lr_over_bc1 = dev_hyper[0];
        inv_sqrt_bc2 = dev_hyper[1];
    }
    // A block walks tiles_per_block consecutive tiles of 1024 elements and keeps sum(p^2) of the segment it is in in
    // registers: one atomic per block and segment instead of one per tile - tens of thousands of atomics on the same
    // few addresses serialise (~2 ns each) and cost more than the kernel's HBM time.
    // Round 4: the walk can go U tiles at a time - first the U flags, then the 4 U vector loads, then the arithmetic and
    // the stores tile by tile, in the same order as before (bit-identical sums). Built to test whether the update is
    // bound by loads in flight: it is not (see SM_ADAM_UNROLL) - U stays 1.
    constexpr int U = SM_ADAM_UNROLL;
    // Which tiles a block walks. Strided (block b: tiles b, b + G, ...; still ascending, so the running sum's segment
    // changes at most n_layers - 1 times) keeps all blocks of the chip on the same stretch of the arena: the DENSE update
    // runs 433 -> 370 us (4.3 -> 5.1 TB/s) and synthetic long runs of flagged chunks 96 -> 71 us. A real view's flags are
    // fragmented at the chunk level, and there 16 consecutive tiles per block are faster (closing update in situ
    // 146-152 against 171 us, c3 step 185.4 against 184.5 views/s): strided for the dense update only.
    const bool strided = SM_ADAM_STRIDED == 1 || (SM_ADAM_STRIDED == 2 && (!ADAM || touched == nullptr));
    const size_t tile0 = strided ? (size_t)blockIdx.x : (size_t)blockIdx.x * tiles_per_block;
    const size_t tile_step = strided ? (size_t)gridDim.x : 1;
    int k_cur = seg_of(segs, tile0 * 1024);
    float sq = 0.f;
    for (int t0 = 0; t0 < tiles_per_block; t0 += U) {
        size_t i0s[U];
        bool act[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t tile = tile0 + (size_t)(t0 + u) * tile_step;
            i0s[u] = (tile * 256 + threadIdx.x) * 4;
            const bool in = t0 + u < tiles_per_block && i0s[u] < n;
            // Ever-touched chunks (zero-initialised textures): a texel no view has ever reached has p = g = m = v = 0, so
            // its regulariser gradient, its update and its contribution to sum(p^2) are exactly zero - the chunk is
            // skipped without being read. (A chunk is 2^touched_log2 >= 4 floats: a thread's four elements share one flag.)
            act[u] = in && (!ADAM || touched == nullptr || touched[i0s[u] >> touched_log2] != 0);
        }
        float pv[U][4], gv[U][4], mv[U][4], vv[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i0 = i0s[u];
            if (!act[u]) continue;
            if (i0 + 4 <= n) {
                *reinterpret_cast<float4*>(pv[u]) = *reinterpret_cast<const float4*>(p + i0);
                if (ADAM) {
                    // g == NULL: the data-term gradient is known to be zero wherever this launch walks (the early half
                    // of the split update) - neither read nor zeroed: 6 instead of 8 streams
                    *reinterpret_cast<float4*>(gv[u]) = g ? *reinterpret_cast<const float4*>(g + i0) : make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(mv[u]) = *reinterpret_cast<const float4*>(m + i0);
                    *reinterpret_cast<float4*>(vv[u]) = *reinterpret_cast<const float4*>(v + i0);
                }
            } else {
                for (int j = 0; j < 4; ++j) {
                    const bool ok = i0 + j < n;
                    pv[u][j] = ok ? p[i0 + j] : 0.f;
                    if (ADAM) {
                        gv[u][j] = (ok && g) ? g[i0 + j] : 0.f;
                        mv[u][j] = ok ? m[i0 + j] : 0.f;
                        vv[u][j] = ok ? v[i0 + j] : 0.f;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t tile = tile0 + (size_t)(t0 + u) * tile_step;
            if (t0 + u >= tiles_per_block || tile * 1024 >= n) break;      // block-uniform
            const size_t i0 = i0s[u];
            // a tile almost always lies inside one segment (k_blk); elements of a tile that straddles a boundary and
            // belong to another segment are added one by one.
            const int k_blk = seg_of(segs, tile * 1024);
            if (k_blk != k_cur) {   // block-uniform
                if (sumsq) {
                    __syncthreads();
                    const float s = block_sum(sq);
                    if (threadIdx.x == 0 && s != 0.f) atomicAdd(sumsq + k_cur, s);
                }
                sq = 0.f;
                k_cur = k_blk;
            }
            if (!act[u]) continue;
            const bool full = i0 + 4 <= n;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = seg_of(segs, i0 + j);
                float x = pv[u][j];
                if (ADAM) {
                    const float gr = fmaf(gv[u][j], grad_scale, segs.reg[k] * x);  // data term (+ mean over ranks) + reg
                    mv[u][j] = mv[u][j] + (gr - mv[u][j]) * one_minus_beta1;       // exp_avg.lerp_(grad, 1 - beta1)
                    vv[u][j] = vv[u][j] * beta2 + (gr * gr) * one_minus_beta2;     // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
                    const float denom = sqrtf(vv[u][j]) * inv_sqrt_bc2 + eps;
                    x = x - lr_over_bc1 * (mv[u][j] / denom);
                }
                x = fminf(hi, fmaxf(x, lo));  // the clamp the next forward would start with
                pv[u][j] = x;
                if (sumsq && i0 + j < n) {
                    if (k == k_blk) sq += x * x;
                    else atomicAdd(sumsq + k, x * x);
                }
            }
            if (full) {
                adam_store4(p + i0, pv[u]);
                if (ADAM) {
                    adam_store4(m + i0, mv[u]);
                    adam_store4(v + i0, vv[u]);
                    if (zero_grad && g) {
                        const float z[4] = {0.f, 0.f, 0.f, 0.f};
                        adam_store4(g + i0, z);
                    }
                }
            } else {
                for (int j = 0; j < 4 && i0 + j < n; ++j) {
                    p[i0 + j] = pv[u][j];
                    if (ADAM) {
                        m[i0 + j] = mv[u][j];
                        v[i0 + j] = vv[u][j];
                        if (zero_grad && g) g[i0 + j] = 0.f;
                    }
                }
            }
        }
    }   // tiles
    if (sumsq) {
        __syncthreads();
        const float s = block_sum(sq);
        if (threadIdx.x == 0 && s != 0.f) atomicAdd(sumsq + k_cur, s);
    }
}

// ---------------------------------------------------------------------------------------------------
// K7s (round 6): the same update over FLAGGED 256-byte chunks, for views whose footprint is a small, fragmented share of
// the arena (closing half of the split update: 14 % of the chunks at c3; the early half: 25 %). adam_kernel walks every
// 1024-element tile of the arena and asks the flag of each thread's chunk first: a dependent flag -> data round trip per
// tile, sixteen tiles in a row per block - 157 us for 268 MB (1.7 TB/s), most of it waiting on flags of chunks that are not
// flagged. Here a block owns a SPAN of 256 chunks of ONE segment (layer): it reads the span's flags, one per thread,
// compacts the flagged chunk numbers into LDS (ballot + prefix counts), and then streams the
// listed chunks, 32 per iteration (sixteen lanes x float4 per chunk, two chunks per lane group: eight loads in flight per
// thread): no work and no latency for an unflagged chunk. Same arithmetic per element as adam_kernel (the same bits of
// p, m, v, g); sum(p^2) of the block's segment in registers, one atomic per block. Requires chunks of 64 floats and
// chunk-aligned segment boundaries (else sm_adam_fused takes adam_kernel).
// ---------------------------------------------------------------------------------------------------
constexpr int ADAM_SPAN = 256;           // chunks per block: one flag per thread = 16 K elements, the tile walk's block size
                                         // (measured at c3 / c2, closing half: 64 -> 185 / 79 us, 128 -> 112 / 76, 256 -> 91 / 82,
                                         // 1024 with 16-byte flag loads -> 127 / 109: a view's footprint is a few blobs, long
                                         // spans leave most blocks idle; the tile walk: 151 / 89 - profiles/r06/adam_span_ab.txt)
struct SpanTable {
    int first_block[SM_MAX_TEX_LAYERS + 1];   // blocks of segment k: [first_block[k], first_block[k + 1])
    size_t begin[SM_MAX_TEX_LAYERS];          // first element of segment k
};

__global__ __launch_bounds__(256) void adam_sparse_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, size_t n, Segs segs, SpanTable spans,
                                                          float lr_over_bc1, float one_minus_beta1, float beta2,
                                                          float one_minus_beta2, float eps, float inv_sqrt_bc2,
                                                          float grad_scale, float lo, float hi, int zero_grad,
                                                          float* __restrict__ sumsq, const float* __restrict__ dev_hyper,
                                                          const int32_t* __restrict__ touched) {
    if (dev_hyper) {
        lr_over_bc1 = dev_hyper[0];
        inv_sqrt_bc2 = dev_hyper[1];
    }
    __shared__ unsigned short list[ADAM_SPAN];
    __shared__ int wave_count[4];
    int k = 0;
#pragma unroll
    for (int i = 1; i < SM_MAX_TEX_LAYERS; ++i)
        if (i < segs.n && (int)blockIdx.x >= spans.first_block[i]) k = i;
    const size_t seg_begin = spans.begin[k], seg_end = segs.end[k];
    const size_t chunk0 = (seg_begin >> 6) + (size_t)((int)blockIdx.x - spans.first_block[k]) * ADAM_SPAN;   // first chunk of the span
    const size_t seg_chunks_end = (seg_end + 63) >> 6;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- the span's flags -> ascending list of its flagged chunks
    const size_t c = chunk0 + (size_t)tid;
    const bool flagged = c < seg_chunks_end && touched[c] != 0;
    const unsigned long long ball = __ballot(flagged);
    if (lane == 0) wave_count[wave] = __popcll(ball);
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wave) base += wave_count[w];
        total += wave_count[w];
    }
    if (flagged) list[base + __popcll(ball & ((1ull << lane) - 1ull))] = (unsigned short)tid;
    __syncthreads();
    // ---- the listed chunks: lane group tid / 16 takes entries it, it + 16 of every 32; lane tid % 16 one float4 of the chunk
    const float reg = segs.reg[k];
    float sq = 0.f;
    for (int it = 0; it < total; it += 32) {
        size_t i0[2];
        bool on[2];
        float pv[2][4], gv[2][4], mv[2][4], vv[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = it + 16 * u + (tid >> 4);
            on[u] = e < total;
            i0[u] = ((chunk0 + list[on[u] ? e : 0]) << 6) + (size_t)(tid & 15) * 4;
            on[u] = on[u] && i0[u] < seg_end;                          // (the segment's last chunk may be a partial one)
            if (on[u] && i0[u] + 4 <= seg_end) {
                *reinterpret_cast<float4*>(pv[u]) = *reinterpret_cast<const float4*>(p + i0[u]);
                *reinterpret_cast<float4*>(gv[u]) = g ? *reinterpret_cast<const float4*>(g + i0[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(mv[u]) = *reinterpret_cast<const float4*>(m + i0[u]);
                *reinterpret_cast<float4*>(vv[u]) = *reinterpret_cast<const float4*>(v + i0[u]);
            } else if (on[u]) {
                for (int j = 0; j < 4; ++j) {
                    const bool ok = i0[u] + j < seg_end;
                    pv[u][j] = ok ? p[i0[u] + j] : 0.f;
                    gv[u][j] = (ok && g) ? g[i0[u] + j] : 0.f;
                    mv[u][j] = ok ? m[i0[u] + j] : 0.f;
                    vv[u][j] = ok ? v[i0[u] + j] : 0.f;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!on[u]) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = pv[u][j];
                const float gr = fmaf(gv[u][j], grad_scale, reg * x);          // data term (+ mean over ranks) + reg
                mv[u][j] = mv[u][j] + (gr - mv[u][j]) * one_minus_beta1;       // exp_avg.lerp_(grad, 1 - beta1)
                vv[u][j] = vv[u][j] * beta2 + (gr * gr) * one_minus_beta2;     // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
                const float denom = sqrtf(vv[u][j]) * inv_sqrt_bc2 + eps;
                x = x - lr_over_bc1 * (mv[u][j] / denom);
                x = fminf(hi, fmaxf(x, lo));
                pv[u][j] = x;
                if (i0[u] + j < seg_end) sq += x * x;
            }
            if (i0[u] + 4 <= seg_end) {
                adam_store4(p + i0[u], pv[u]);
                adam_store4(m + i0[u], mv[u]);
                adam_store4(v + i0[u], vv[u]);
                if (zero_grad && g) {
                    const float z[4] = {0.f, 0.f, 0.f, 0.f};
                    adam_store4(g + i0[u], z);
                }
            } else {
                for (int j = 0; j < 4 && i0[u] + j < seg_end; ++j) {
                    p[i0[u] + j] = pv[u][j];
                    m[i0[u] + j] = mv[u][j];
                    v[i0[u] + j] = vv[u][j];
                    if (zero_grad && g) g[i0[u] + j] = 0.f;
                }
            }
        }
    }
    if (sumsq) {
        __syncthreads();
        const float s_ = block_sum(sq);
        if (threadIdx.x == 0 && s_ != 0.f) atomicAdd(sumsq + k, s_);
    }
}

// Step-dependent Adam scalars kept ON THE DEVICE (hipGraph replay: the host may run many steps ahead of the GPU, so
// nothing step-dependent may travel through a host buffer that a later step overwrites): state = {lr, step} in double;
// one thread advances the step and writes {lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)} - the doubles torch computes.
__global__ void adam_hyper_step_kernel(double* state, double beta1, double beta2, float* out2) {
    const double t = state[1] + 1.0;
    state[1] = t;
    out2[0] = (float)(state[0] / (1.0 - pow(beta1, t)));
    out2[1] = (float)(1.0 / sqrt(1.0 - pow(beta2, t)));
}

__global__ __launch_bounds__(256) void flags_or_kernel(int32_t* __restrict__ dst, const int32_t* __restrict__ src, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && src[i] != 0) dst[i] = 1;
}

// tiles of 1024 elements per block: enough blocks to fill the chip several times over, few enough atomics
static int adam_tiles_per_block(size_t n) {
    const size_t tiles = (n + 1023) / 1024;
    return (int)((tiles + 4095) / 4096);
}

static TexLayers make_layers(float* const* layers, const int* lw, const int* lh, int n) {
    TexLayers L;
    L.n = n;
    for (int i = 0; i < n; ++i) {
        L.p[i] = layers[i];
        L.w[i] = lw[i];
        L.h[i] = lh[i];
    }
    return L;
}

// Head of a step in one launch: the regulariser loss of the current texture (sum_l coef_l * sum p^2 of layer l, the
// sums left by the previous update) and the zero fill of everything the step accumulates into (loss values + operand
// bounds, Gram slabs) - instead of a multiply, a reduction and two fills.
__global__ __launch_bounds__(256) void step_begin_kernel(const float* __restrict__ sumsq, const float* __restrict__ coef,
                                                         int n_seg, float* __restrict__ reg_out, f32x4* __restrict__ za,
                                                         size_t na4, f32x4* __restrict__ zb, size_t nb4) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && reg_out != nullptr) {
        float r = 0.f;
        for (int l = 0; l < n_seg; ++l) r += sumsq[l] * coef[l];
        *reg_out = r;
    }
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < na4 + nb4; i += stride) {
        if (i < na4) za[i] = z; else zb[i - na4] = z;
    }
}

}  // namespace sm

extern "C" {

int sm_tex_sample_fwd_grouped(const float* const* layers, const int* layer_w, const int* layer_h, int n_layers,
                              const float* const* grids, const int* hs, const int* ws, float* const* outs, int n,
                              void* stream) {
    if (n_layers < 1 || n_layers > SM_MAX_TEX_LAYERS || n < 1 || n > sm::TEX_MAX_GROUP) return (int)hipErrorInvalidValue;
    sm::TexLayers L = sm::make_layers(const_cast<float* const*>(layers), layer_w, layer_h, n_layers);
    sm::TexSampleGroup G{};
    G.n = n;
    for (int i = 0; i < n; ++i) {
        G.grid[i] = reinterpret_cast<const float2*>(grids[i]);
        G.out[i] = outs[i];
        G.h[i] = hs[i];
        G.w[i] = ws[i];
        G.block_begin[i + 1] = G.block_begin[i] + (hs[i] * ws[i] + 255) / 256;
    }
    if (G.block_begin[n] == 0) return 0;
    hipLaunchKernelGGL(sm::tex_sample_fwd_group_kernel, dim3(G.block_begin[n]), dim3(256), 0, (hipStream_t)stream, L, G);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_tex_sample_fwd(const float* const* layers, const int* layer_w, const int* layer_h, int n_layers,
                      const float* grid, int h, int w, float* out, void* stream) {
    if (n_layers < 1 || n_layers > SM_MAX_TEX_LAYERS) return (int)hipErrorInvalidValue;
    sm::TexLayers L = sm::make_layers(const_cast<float* const*>(layers), layer_w, layer_h, n_layers);
    const int n = h * w;
    hipLaunchKernelGGL(sm::tex_sample_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, L,
                       reinterpret_cast<const float2*>(grid), h, w, out, sm::row_stride(w), sm::plane_size(h, w));
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_tex_sample_bwd(float* const* grad_layers, const int* layer_w, const int* layer_h, int n_layers,
                      const float* grid, int h, int w, const float* grad_img, const float* pixel_weight,
                      void* stream) {
    if (n_layers < 1 || n_layers > SM_MAX_TEX_LAYERS) return (int)hipErrorInvalidValue;
    sm::TexLayers L = sm::make_layers(grad_layers, layer_w, layer_h, n_layers);
    const int tiles_x = (w + 15) / 16, tiles_y = (h + 15) / 16;
    hipLaunchKernelGGL(sm::tex_sample_bwd_tiled_kernel<48>, dim3(tiles_x * tiles_y), dim3(256), 0, (hipStream_t)stream,
                       L, reinterpret_cast<const float2*>(grid), h, w, grad_img, pixel_weight, sm::row_stride(w),
                       sm::plane_size(h, w), tiles_x);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_tex_touch_flags(float* const* grad_layers, const int* layer_w, const int* layer_h, int n_layers,
                       const float* arena_base, const float* grid, int h, int w, const float* pixel_weight,
                       int32_t* flags, int chunk_log2, void* stream) {
    if (n_layers < 1 || n_layers > SM_MAX_TEX_LAYERS || chunk_log2 < 4 || chunk_log2 > 24) return (int)hipErrorInvalidValue;
    sm::TexLayers L = sm::make_layers(grad_layers, layer_w, layer_h, n_layers);
    const int n = h * w;
    hipLaunchKernelGGL(sm::tex_touch_flags_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, L,
                       arena_base, reinterpret_cast<const float2*>(grid), h, w, pixel_weight, flags, chunk_log2);
    SM_LAUNCH_CHECK();
    return 0;
}

static int make_segs(sm::Segs& s, size_t n, const size_t* seg_end, const float* reg_coef, int n_seg) {
    if (n_seg < 1 || n_seg > SM_MAX_TEX_LAYERS || seg_end[n_seg - 1] != n) return (int)hipErrorInvalidValue;
    s.n = n_seg;
    for (int i = 0; i < n_seg; ++i) {
        s.end[i] = seg_end[i];
        s.reg[i] = reg_coef ? reg_coef[i] : 0.f;
    }
    return 0;
}

int sm_adam_fused(float* p, float* g, float* m, float* v, size_t n, const size_t* seg_end, const float* reg_coef,
                  int n_seg, float lr, double beta1, double beta2, float eps, double bias_corr1, double bias_corr2,
                  float grad_scale, float clamp_lo, float clamp_hi, int zero_grad, float* sumsq_out,
                  const float* dev_hyper, const int32_t* touched, int touched_chunk_log2, void* stream) {
    if (touched != nullptr && (touched_chunk_log2 < 2 || touched_chunk_log2 > 24)) return (int)hipErrorInvalidValue;
    sm::Segs s;
    if (int e = make_segs(s, n, seg_end, reg_coef, n_seg)) return e;
    // step_size = lr / bias_correction1 and 1 / sqrt(bias_correction2) in double, as torch computes them
    const float lr_over_bc1 = (float)((double)lr / bias_corr1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bias_corr2));
    if (touched != nullptr && touched_chunk_log2 == 6) {
        // flagged chunks of 64 floats and every segment starting on a chunk boundary: the span kernel (adam_sparse_kernel)
        bool aligned = true;
        for (int k = 0; k + 1 < n_seg; ++k) aligned = aligned && seg_end[k] % 64 == 0;
        static const bool dense_walk = getenv("SM_ADAM_DENSE_WALK") != nullptr && atoi(getenv("SM_ADAM_DENSE_WALK")) != 0;
        if (aligned && !dense_walk) {
            constexpr int span = sm::ADAM_SPAN;
            sm::SpanTable t{};
            size_t begin = 0;
            t.first_block[0] = 0;
            for (int k = 0; k < n_seg; ++k) {
                t.begin[k] = begin;
                const size_t chunks = (seg_end[k] - begin + 63) / 64;
                t.first_block[k + 1] = t.first_block[k] + (int)((chunks + span - 1) / span);
                begin = seg_end[k];
            }
            if (t.first_block[n_seg] > 0)
                hipLaunchKernelGGL(sm::adam_sparse_kernel, dim3((unsigned)t.first_block[n_seg]), dim3(256), 0, (hipStream_t)stream,
                                   p, g, m, v, n, s, t, lr_over_bc1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), eps,
                                   inv_sqrt_bc2, grad_scale, clamp_lo, clamp_hi, zero_grad, sumsq_out, dev_hyper, touched);
            SM_LAUNCH_CHECK();
            return 0;
        }
    }
    const int tpb = sm::adam_tiles_per_block(n);
    const size_t blocks = ((n + 1023) / 1024 + tpb - 1) / tpb;
    hipLaunchKernelGGL(sm::adam_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                       s, lr_over_bc1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), eps, inv_sqrt_bc2,
                       grad_scale, clamp_lo, clamp_hi, zero_grad, sumsq_out, dev_hyper, tpb, touched, touched_chunk_log2);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_adam_hyper_step(double* state, double beta1, double beta2, float* dev_hyper, void* stream) {
    hipLaunchKernelGGL(sm::adam_hyper_step_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, beta1, beta2, dev_hyper);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_step_begin(const float* sumsq, const float* coef, int n_seg, float* reg_out, float* zero_a, size_t n_a,
                  float* zero_b, size_t n_b, void* stream) {
    if (n_seg < 0 || n_a % 4 != 0 || n_b % 4 != 0) return (int)hipErrorInvalidValue;
    const size_t work = (n_a + n_b) / 4;
    const unsigned blocks = (unsigned)std::min<size_t>(2048, std::max<size_t>(1, (work + 255) / 256));
    hipLaunchKernelGGL(sm::step_begin_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sumsq, coef, n_seg, reg_out,
                       reinterpret_cast<sm::f32x4*>(zero_a), n_a / 4, reinterpret_cast<sm::f32x4*>(zero_b), n_b / 4);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_flags_or(int32_t* dst, const int32_t* src, size_t n, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(sm::flags_or_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst, src, n);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_clamp_sumsq(float* p, size_t n, const size_t* seg_end, int n_seg, float clamp_lo, float clamp_hi,
                   float* sumsq_out, void* stream) {
    sm::Segs s;
    if (int e = make_segs(s, n, seg_end, nullptr, n_seg)) return e;
    const int tpb = sm::adam_tiles_per_block(n);
    const size_t blocks = ((n + 1023) / 1024 + tpb - 1) / tpb;
    hipLaunchKernelGGL(sm::adam_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, n, s, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, clamp_lo,
                       clamp_hi, 0, sumsq_out, (const float*)nullptr, tpb, (const int32_t*)nullptr, 0);
    SM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

// Shared helpers for the gfx950 kernels of libstylemesh_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/stylemesh_hip.h"

#define SM_LAUNCH_CHECK() \
    do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

namespace sm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// bf16x3 split of an fp32 number: x -> (h, m, l) bf16 with h + m + l == x to 24 significand bits; round-to-nearest
// conversions (v_cvt_pk_bf16_f32). Six partial products of two such triples reproduce the fp32 product to 2^-23.
__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}
// eight fp32 values -> three MFMA operand vectors
__device__ __forceinline__ void split3x8(const float (&x)[8], f32x4& vh, f32x4& vm, f32x4& vl) {
    bf16x8 h, m, l;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        __bf16 a, b, d;
        split3(x[c], a, b, d);
        h[c] = a; m[c] = b; l[c] = d;
    }
    vh = __builtin_bit_cast(f32x4, h);
    vm = __builtin_bit_cast(f32x4, m);
    vl = __builtin_bit_cast(f32x4, l);
}

// wave-wide max of a non-negative value -> at most one atomic max (bit patterns of non-negative floats order like
// uints). Same-address device atomics serialise at ~12 ns each (8 k of them cost a small kernel 100 us), so a wave first
// reads the current value - it only grows - and skips the atomic unless it would raise it: after the first few
// hundred waves of a launch almost none does.
__device__ __forceinline__ void record_amax(float* amax_out, float v) {
    if (amax_out == nullptr) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    if ((threadIdx.x & 63) == 0 && v > __hip_atomic_load(amax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(reinterpret_cast<unsigned*>(amax_out), __builtin_bit_cast(unsigned, v));
}

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
__host__ __device__ inline int row_stride(int W) { return round_up(W + 1, 4); }
__host__ __device__ inline int plane_size(int H, int W) { return round_up((H + 2) * row_stride(W), 64); }

// q (linear index inside a padded plane) -> is it an interior pixel? Rows 1..H, columns 1..W.
__device__ __forceinline__ bool interior(int q, int H, int W, int Wp) {
    int r = q / Wp;
    int x = q - r * Wp;
    return (r >= 1) & (r <= H) & (x >= 1) & (x <= W);
}

// XCD-aware 1-D block id -> (sequence index inside the XCD's share, xcd). Blocks are dispatched
// round-robin over the 8 XCDs (block b -> XCD b % 8, observed, speed only); consecutive sequence
// indices therefore share an L2.
__device__ __forceinline__ int xcd_linear(int bid, int nblocks) {
    // bijective remap: XCD x gets the contiguous chunk of work items [start_x, start_x + cnt_x)
    int xcd = bid & 7;
    int s = bid >> 3;
    int q = nblocks >> 3, r = nblocks & 7;
    int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + s;
}

}  // namespace sm

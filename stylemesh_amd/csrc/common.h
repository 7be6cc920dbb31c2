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

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
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- "amax" bounds: max |x| of what a launch writes = the operand bound of the fp16x2 kernels that read it ----------
// A bound is SM_AMAX_SLOTS words spaced SM_AMAX_STRIDE floats (256 bytes) apart; its value is the maximum over the
// slots. Writers: a wave-wide max, then at most one atomic max on the bit pattern (non-negative floats order like
// uints) into the slot of the wave's BLOCK (block id mod 64). Why slots: EVERY same-address access that reaches the L2 /
// fabric serialises there - an atomic costs ~12 ns, a coherent load ~2.5 ns - and a launch whose 16 k waves start
// together all see the same stale value: one word made a 6.5 us second-pass kernel take 47 us. Spread over 64 cache
// lines the same traffic is 64 short queues. A wave also peeks at its slot with a PLAIN vector load (served by the CU's
// L1; possibly stale, which only costs a redundant atomic - a bound only grows) beside the kernel's first loads and
// skips the atomic unless it would raise the slot. Readers (`amax_read`): lane l loads slot l, wave max - one vector
// load + six cross-lane steps per wave at kernel start.
#define SM_AMAX_SLOTS 64
#define SM_AMAX_STRIDE 64
__device__ __forceinline__ int amax_slot_offset() {
    return (int)((blockIdx.x + blockIdx.y * gridDim.x) & (SM_AMAX_SLOTS - 1)) * SM_AMAX_STRIDE;
}
__device__ __forceinline__ float amax_peek(const float* amax_out) {
    if (amax_out == nullptr) return 0.f;
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));   // opaque index: a vector (L1) load, not a scalar one
    return amax_out[amax_slot_offset() + zero];
}
__device__ __forceinline__ void record_amax(float* amax_out, float v, float seen) {
    if (amax_out == nullptr) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    if ((threadIdx.x & 63) == 0 && v > seen)
        atomicMax(reinterpret_cast<unsigned*>(amax_out + amax_slot_offset()), __builtin_bit_cast(unsigned, v));
}
// the bound's value, uniform over the wave (every lane returns the max over the slots)
__device__ __forceinline__ float amax_read(const float* amax) {
    float v = amax[(threadIdx.x & (SM_AMAX_SLOTS - 1)) * SM_AMAX_STRIDE];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Loads / stores of the HBM-bound kernels that run on side streams beside the conv trunk (style branches): with
// -DSM_SIDE_NT=1 they carry the non-temporal hint (streamed once: do not displace the trunk's operands from L2 / MALL).
#ifndef SM_SIDE_NT
#define SM_SIDE_NT 0
#endif
template <class T>
__device__ __forceinline__ T side_load(const T* p) {
#if SM_SIDE_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <class T>
__device__ __forceinline__ void side_store(T v, T* p) {
#if SM_SIDE_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
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

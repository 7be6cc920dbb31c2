// Multi-GPU gradient exchange, device side (SURVEY.md section 8 e; the reference is single-GPU): the compact form of
// the texture-gradient arena that travels over RCCL. Which 2^chunk_log2-float chunks of the arena the ranks' current
// views can write is known per view (sm_tex_touch_flags + the ranks' max-all-reduce); per step the flagged chunks are
// gathered into one contiguous buffer, all-reduced, and scattered back:
//   sm_flags_compact   flags -> ascending list of flagged chunk indices + their count, on the device, DETERMINISTIC
//                      (block counts -> one-block scan -> ordered write): every rank derives the identical list from
//                      the identical (all-reduced) flags, so the ranks' compact buffers line up element by element.
//                      The count stays on the device; the host reads it with the per-view read-back it does anyway.
//   sm_chunks_gather   dst[j][:] = arena[idx[j]][:]      (one 16-byte lane access per float4, 256 B per 64-float chunk)
//   sm_chunks_scatter  arena[idx[j]][:] = scale * src[j][:]
// All HBM-bound streaming kernels; nothing here depends on the step.
#include "common.h"

namespace sm {

constexpr int CP_PER_BLOCK = 1024;   // flags per block of the compaction passes (256 threads x 4)

__global__ __launch_bounds__(256) void flags_count_kernel(const int32_t* __restrict__ flags, size_t n, int* __restrict__ block_counts) {
    const size_t base = (size_t)blockIdx.x * CP_PER_BLOCK + threadIdx.x * 4;
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (base + k < n) c += flags[base + k] != 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    __shared__ int part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// exclusive scan of the block counts in place (one block; n_blocks is a few thousand at most), total -> *count_out
__global__ __launch_bounds__(1024) void flags_scan_kernel(int* __restrict__ block_counts, int n_blocks, int* __restrict__ count_out) {
    __shared__ int wave_sum[16];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < n_blocks; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < n_blocks ? block_counts[i] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        int before = carry_s;
        for (int w = 0; w < wave; ++w) before += wave_sum[w];
        if (i < n_blocks) block_counts[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count_out = carry_s;
}

__global__ __launch_bounds__(256) void flags_write_kernel(const int32_t* __restrict__ flags, size_t n,
                                                          const int* __restrict__ block_offsets, int32_t* __restrict__ idx) {
    const size_t base = (size_t)blockIdx.x * CP_PER_BLOCK + threadIdx.x * 4;
    int f[4], c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f[k] = (base + k < n) && flags[base + k] != 0;
        c += f[k];
    }
    // exclusive prefix of c over the block, in thread order (= ascending chunk index)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    __shared__ int wave_sum[4];
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int pos = block_offsets[blockIdx.x] + incl - c;
    for (int w = 0; w < wave; ++w) pos += wave_sum[w];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (f[k]) idx[pos++] = (int32_t)(base + k);
}

// one thread per float4 of the compact buffer; the chunk count comes from the device (n_idx_dev) or the host (n_idx)
template <bool GATHER>
__global__ __launch_bounds__(256) void chunks_move_kernel(float* __restrict__ arena, const int32_t* __restrict__ idx,
                                                          const int* __restrict__ n_idx_dev, size_t n_idx, int chunk_log2,
                                                          float* __restrict__ compact, float scale) {
    const size_t n = n_idx_dev ? (size_t)*n_idx_dev : n_idx;
    const int q_log2 = chunk_log2 - 2;                         // float4 per chunk, log2
    const size_t total = n << q_log2;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const size_t j = g >> q_log2;
        const size_t part = g - (j << q_log2);
        f32x4* a = reinterpret_cast<f32x4*>(arena + ((size_t)idx[j] << chunk_log2)) + part;
        f32x4* c = reinterpret_cast<f32x4*>(compact) + g;
        if (GATHER) {
            *c = *a;
        } else {
            f32x4 v = *c;
            v[0] *= scale; v[1] *= scale; v[2] *= scale; v[3] *= scale;
            *a = v;
        }
    }
}

}  // namespace sm

extern "C" {

size_t sm_flags_compact_ws_ints(size_t n_flags) { return (n_flags + sm::CP_PER_BLOCK - 1) / sm::CP_PER_BLOCK; }

int sm_flags_compact(const int32_t* flags, size_t n_flags, int32_t* idx_out, int32_t* count_out, int32_t* ws, void* stream) {
    if (n_flags == 0 || n_flags > ((size_t)1 << 31)) return (int)hipErrorInvalidValue;
    const int blocks = (int)sm_flags_compact_ws_ints(n_flags);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sm::flags_count_kernel, dim3(blocks), dim3(256), 0, s, flags, n_flags, ws);
    SM_LAUNCH_CHECK();
    hipLaunchKernelGGL(sm::flags_scan_kernel, dim3(1), dim3(1024), 0, s, ws, blocks, count_out);
    SM_LAUNCH_CHECK();
    hipLaunchKernelGGL(sm::flags_write_kernel, dim3(blocks), dim3(256), 0, s, flags, n_flags, ws, idx_out);
    SM_LAUNCH_CHECK();
    return 0;
}

static int chunks_move(bool gather, float* arena, const int32_t* idx, const int32_t* n_idx_dev, size_t n_idx,
                       int chunk_log2, float* compact, float scale, hipStream_t s) {
    if (chunk_log2 < 2 || chunk_log2 > 24) return (int)hipErrorInvalidValue;
    if (n_idx == 0) return 0;   // (with n_idx_dev, n_idx is the capacity the grid is sized for)
    const size_t total = n_idx << (chunk_log2 - 2);
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 32);
    if (gather)
        hipLaunchKernelGGL(sm::chunks_move_kernel<true>, dim3(blocks), dim3(256), 0, s, arena, idx, n_idx_dev, n_idx,
                           chunk_log2, compact, scale);
    else
        hipLaunchKernelGGL(sm::chunks_move_kernel<false>, dim3(blocks), dim3(256), 0, s, arena, idx, n_idx_dev, n_idx,
                           chunk_log2, compact, scale);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_chunks_gather(const float* arena, const int32_t* idx, const int32_t* n_idx_dev, size_t n_idx, int chunk_log2,
                     float* compact, void* stream) {
    return chunks_move(true, const_cast<float*>(arena), idx, n_idx_dev, n_idx, chunk_log2, compact, 1.f, (hipStream_t)stream);
}

int sm_chunks_scatter(float* arena, const int32_t* idx, const int32_t* n_idx_dev, size_t n_idx, int chunk_log2,
                      const float* compact, float scale, void* stream) {
    return chunks_move(false, arena, idx, n_idx_dev, n_idx, chunk_log2, const_cast<float*>(compact), scale, (hipStream_t)stream);
}

}  // extern "C"

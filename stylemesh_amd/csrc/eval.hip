// E1: reprojection warp + masked squared error of the multi-view consistency metric.
//
// Reference operators replaced (lukasHoel/stylemesh): reproject() of data/utils.py:73-194 (un-project the source
// pixels with the source depth, move them into the target camera, project, reject by depth hole / screen bounds /
// depth consistency, bilinear warp of the target image and validity mask: ~25 torch ops and 6 grid_sample calls)
// and the masked MSE accumulation of scripts/eval/eval_image_folders.py:286-305. One pass, one thread per source
// pixel, HBM / gather bound (5 + 3 + 1 planes read, 4 written).
#include "common.h"

namespace sm {

struct ReprojArgs {
    float m[12];               // rows 0..2 of src2tar = inverse(cam2world_tar) * cam2world_src
    float fx, fy, cx, cy;
    int H, W;
    float depth_tol;
};

// F.grid_sample(..., align_corners=True, padding_mode='border') source coordinate of the reference's pixel -> grid
// mapping g = 2 p / size - 1 (data/utils.py:140-149): p (size - 1) / size, clipped to the image
__device__ __forceinline__ float src_coord(float p, int size) {
    const float g = 2.0f * p / (float)size - 1.0f;
    const float i = (g + 1.0f) / 2.0f * (float)(size - 1);
    return fminf((float)(size - 1), fmaxf(i, 0.f));
}

__global__ __launch_bounds__(256) void reproject_kernel(ReprojArgs a, const float* __restrict__ depth_src,
                                                        const float* __restrict__ depth_tar,
                                                        const float* __restrict__ color_tar,
                                                        const float* __restrict__ mask_tar,
                                                        const float* __restrict__ styled_src, float* __restrict__ color_out,
                                                        uint8_t* __restrict__ mask_out, double* __restrict__ partial) {
    __shared__ float red[2][4];
    const int H = a.H, W = a.W, n = H * W;
    const int i = blockIdx.x * 256 + threadIdx.x;
    float err = 0.f, cnt = 0.f;
    if (i < n) {
        const int py_i = i / W, px_i = i - py_i * W;
        const float d = depth_src[i];
        const float X = ((float)px_i - a.cx) / a.fx * d, Y = ((float)py_i - a.cy) / a.fy * d;
        const float tx = X * a.m[0] + Y * a.m[1] + d * a.m[2] + a.m[3];
        const float ty = X * a.m[4] + Y * a.m[5] + d * a.m[6] + a.m[7];
        const float tz = X * a.m[8] + Y * a.m[9] + d * a.m[10] + a.m[11];
        const float px = tx / (1e-8f + tz) * a.fx + a.cx;
        const float py = ty / (1e-8f + tz) * a.fy + a.cy;
        bool ok = d != 0.f && px >= 0.f && py >= 0.f && px < (float)(W - 1) && py < (float)(H - 1);   // NaN -> false
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
        if (ok) {
            // depth consistency: nearest target depth at the 4 integer neighbours (round half to even, as ATen)
            const float lx = floorf(px), ly = floorf(py);
            float dzmin = 3.0e38f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int xi = (int)rintf(src_coord(lx + (float)(k >> 1), W));
                const int yi = (int)rintf(src_coord(ly + (float)(k & 1), H));
                dzmin = fminf(dzmin, fabsf(tz - depth_tar[yi * W + xi]));
            }
            ok = !(dzmin > a.depth_tol);
        }
        if (ok) {
            const float ix = src_coord(px, W), iy = src_coord(py, H);
            const float fx0 = floorf(ix), fy0 = floorf(iy);
            const int x0 = (int)fx0, y0 = (int)fy0;
            const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = (fx0 + 1.f) - ix, wy0 = (fy0 + 1.f) - iy;
            const bool x1_in = x0 + 1 <= W - 1, y1_in = y0 + 1 <= H - 1;
            const float w00 = wx0 * wy0, w10 = x1_in ? wx1 * wy0 : 0.f, w01 = y1_in ? wx0 * wy1 : 0.f,
                        w11 = (x1_in && y1_in) ? wx1 * wy1 : 0.f;
            const int o00 = y0 * W + x0, o10 = o00 + (x1_in ? 1 : 0), o01 = o00 + (y1_in ? W : 0),
                      o11 = o01 + (x1_in ? 1 : 0);
            const float m = mask_tar[o00] * w00 + mask_tar[o10] * w10 + mask_tar[o01] * w01 + mask_tar[o11] * w11;
            ok = m > 0.99f;
            if (ok) {
                c0 = color_tar[o00] * w00 + color_tar[o10] * w10 + color_tar[o01] * w01 + color_tar[o11] * w11;
                c1 = color_tar[n + o00] * w00 + color_tar[n + o10] * w10 + color_tar[n + o01] * w01 + color_tar[n + o11] * w11;
                c2 = color_tar[2 * n + o00] * w00 + color_tar[2 * n + o10] * w10 + color_tar[2 * n + o01] * w01 +
                     color_tar[2 * n + o11] * w11;
            }
        }
        color_out[i] = c0;
        color_out[n + i] = c1;
        color_out[2 * n + i] = c2;
        mask_out[i] = ok ? 1 : 0;
        if (ok && styled_src) {
            const float e0 = styled_src[i] - c0, e1 = styled_src[n + i] - c1, e2 = styled_src[2 * n + i] - c2;
            err = e0 * e0 + e1 * e1 + e2 * e2;
            cnt = 3.f;
        }
    }
    // per-block partial sums (plain stores: deterministic); the host adds the few hundred doubles
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        err += __shfl_down(err, o, 64);
        cnt += __shfl_down(cnt, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = err; red[1][threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0 && partial) {
        partial[2 * blockIdx.x] = (double)red[0][0] + red[0][1] + red[0][2] + red[0][3];
        partial[2 * blockIdx.x + 1] = (double)red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

}  // namespace sm

extern "C" {

int sm_reproject_blocks(int H, int W) { return (H * W + 255) / 256; }

int sm_reproject(const float* src2tar, const float* intrinsics, int H, int W, const float* depth_src,
                 const float* depth_tar, const float* color_tar, const float* mask_tar, const float* styled_src,
                 float* color_out, uint8_t* mask_out, double* partial, float depth_tol, void* stream) {
    if (H < 2 || W < 2) return (int)hipErrorInvalidValue;
    sm::ReprojArgs a;
    for (int k = 0; k < 12; ++k) a.m[k] = src2tar[k];
    a.fx = intrinsics[0]; a.fy = intrinsics[1]; a.cx = intrinsics[2]; a.cy = intrinsics[3];
    a.H = H; a.W = W; a.depth_tol = depth_tol;
    hipLaunchKernelGGL(sm::reproject_kernel, dim3(sm_reproject_blocks(H, W)), dim3(256), 0, (hipStream_t)stream, a,
                       depth_src, depth_tar, color_tar, mask_tar, styled_src, color_out, mask_out, partial);
    SM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

// K3/K4: VGG 3x3 convolutions (forward and data gradient) as implicit-im2col GEMMs on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32), plus 2x2 max-pool forward / backward and the 64->3 first-layer data gradient.
//
// Reference operators replaced (lukasHoel/stylemesh): nn.Conv2d + F.relu + nn.MaxPool2d of
// model/losses/content_and_style_losses.py:11-32,49-69 and their autograd backward.
//
// GEMM view of one conv:  out[co][q] = sum_{tap,ci} Wt[tap][ci][co] * in[ci][q + d(tap)],
//   M = C_out, N = linear positions q of the padded plane (rows 1..H), K = 9 * C_in,
//   d(tap) = (ky-1)*Wp + (kx-1). Because the plane carries its own zero border, the im2col shift is a pure
//   offset in q: no per-tap bounds checks, and a tile of BN consecutive q needs just three contiguous input
//   segments per channel (one per ky) of BN+8 floats, staged in LDS once per K-chunk.
// Block = 256 threads = 4 waves; every wave owns a 64x64 output tile = 2x2 MFMA 32x32 tiles (64 accumulator
// VGPRs). A K-chunk = KC input channels x 9 taps. LDS: weights [9][KC][BM] + inputs [3][KC][BN+8] fp32
// (49.9 KB for 128x128x8 -> 3 blocks per CU, whose MFMA phases cover each other's staging phases).
#include "common.h"

namespace sm {

struct ConvArgs {
    const float* in;
    const float* wt;
    const float* bias;
    float* out;
    const float* gate;
    int Cin_pad, Cout, H, W, Wp, plane, n_tiles, m_tiles;
};

template <int BM, int BN, int KC, int WGM, int WGN, int FLAGS>
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(ConvArgs a) {
    static_assert(BM / WGM == 64 && BN / WGN == 64 && WGM * WGN == 4, "wave tile is 64x64");
    constexpr int BNP = BN + 8;  // 4 floats of halo on each side keeps every segment 16-byte aligned
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [9][KC][BM]
    float* Bs = smem + 9 * KC * BM;   // [3][KC][BNP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int lhi = lane >> 5;
    const int wm = (wave / WGN) * 64;
    const int wn = (wave % WGN) * 64;

    const int item = xcd_linear(blockIdx.x, a.n_tiles * a.m_tiles);
    const int m_tile = item / a.n_tiles;
    const int n_tile = item - m_tile * a.n_tiles;
    const int m0 = m_tile * BM;
    const int q0 = a.Wp + n_tile * BN;  // first computed position = start of row 1

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int c0 = 0; c0 < a.Cin_pad; c0 += KC) {
        // ---- stage weights: rows (tap, ci) of BM contiguous floats
        constexpr int A_F4_PER_ROW = BM / 4;
        constexpr int A_F4 = 9 * KC * A_F4_PER_ROW;
        for (int i = tid; i < A_F4; i += 256) {
            int row = i / A_F4_PER_ROW, c4 = i - row * A_F4_PER_ROW;
            int tap = row / KC, ci = row - tap * KC;
            const float4 v = *reinterpret_cast<const float4*>(
                a.wt + ((size_t)(tap * a.Cin_pad + c0 + ci) * a.Cout + m0 + c4 * 4));
            *reinterpret_cast<float4*>(As + row * BM + c4 * 4) = v;
        }
        // ---- stage inputs: rows (ky, ci) of BNP contiguous floats starting at q0 + (ky-1)*Wp - 4
        constexpr int B_F4_PER_ROW = BNP / 4;
        constexpr int B_F4 = 3 * KC * B_F4_PER_ROW;
        for (int i = tid; i < B_F4; i += 256) {
            int row = i / B_F4_PER_ROW, c4 = i - row * B_F4_PER_ROW;
            int ky = row / KC, ci = row - ky * KC;
            const float4 v = *reinterpret_cast<const float4*>(
                a.in + ((size_t)(c0 + ci) * a.plane + q0 + (ky - 1) * a.Wp - 4 + c4 * 4));
            *reinterpret_cast<float4*>(Bs + row * BNP + c4 * 4) = v;
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int kp = 0; kp < KC / 2; ++kp) {
                const int ci = kp * 2 + lhi;
                const float* ap = As + (tap * KC + ci) * BM + wm + l31;
                const float* bp = Bs + (ky * KC + ci) * BNP + 3 + kx + wn + l31;
                const float a0 = ap[0], a1 = ap[32];
                const float b0 = bp[0], b1 = bp[32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue. C/D layout of the 32x32 MFMA: column (pixel) = lane & 31,
    //      row (channel) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
    const int q_end = (a.H + 1) * a.Wp;
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
        const int q = q0 + wn + nj * 32 + l31;
        if (q >= q_end) continue;
        const bool inside = interior(q, a.H, a.W, a.Wp);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int co_base = m0 + wm + mi * 32 + 4 * lhi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_base + (r & 3) + 8 * (r >> 2);
                const size_t o = (size_t)co * a.plane + q;
                float v = acc[mi][nj][r];
                if (FLAGS & SM_EPI_BIAS_RELU) v = fmaxf(v + a.bias[co], 0.f);
                if (FLAGS & SM_EPI_ADD) v += a.out[o];
                if (FLAGS & SM_EPI_RELU_MASK) v = (a.gate[o] > 0.f) ? v : 0.f;
                a.out[o] = inside ? v : 0.f;
            }
        }
    }
}

template <int BM, int BN, int KC, int WGM, int WGN, int FLAGS>
static int launch_conv(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    a.m_tiles = a.Cout / BM;
    a.n_tiles = (a.H * a.Wp + BN - 1) / BN;
    constexpr size_t lds = (size_t)(9 * KC * BM + 3 * KC * (BN + 8)) * sizeof(float);
    auto k = conv3x3_mfma_kernel<BM, BN, KC, WGM, WGN, FLAGS>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k, dim3(a.m_tiles * a.n_tiles), dim3(256), lds, s, a);
    SM_LAUNCH_CHECK();
    return 0;
}

template <int FLAGS>
static int dispatch_conv(const ConvArgs& a, hipStream_t s) {
    if (a.Cin_pad == 4) return launch_conv<64, 256, 4, 1, 4, FLAGS>(a, s);
    if (a.Cout % 128 != 0) return launch_conv<64, 256, 8, 1, 4, FLAGS>(a, s);
    return launch_conv<128, 128, 8, 2, 2, FLAGS>(a, s);
}

// ---------------------------------------------------------------------------------------------------
// First-layer data gradient (64 -> 3 channels): far too thin for the matrix cores (M = 3), so a VALU kernel:
// one thread per position q, 3 accumulators, weights read through the scalar cache (wave-uniform).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv3x3_dgrad_c3_kernel(const float* __restrict__ dz,
                                                               const float* __restrict__ wd, float* out, int Cin,
                                                               int H, int W, int Wp, int plane) {
    const int q = Wp + blockIdx.x * 256 + threadIdx.x;
    if (q >= (H + 1) * Wp) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int ci = 0; ci < Cin; ++ci) {
        const float* p = dz + (size_t)ci * plane + q;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float x = p[(ky - 1) * Wp + (kx - 1)];
                const float* w = wd + ((ky * 3 + kx) * Cin + ci) * 4;
                a0 = fmaf(w[0], x, a0);
                a1 = fmaf(w[1], x, a1);
                a2 = fmaf(w[2], x, a2);
            }
    }
    const bool inside = interior(q, H, W, Wp);
    out[q] = inside ? a0 : 0.f;
    out[(size_t)plane + q] = inside ? a1 : 0.f;
    out[(size_t)2 * plane + q] = inside ? a2 : 0.f;
}

// ---------------------------------------------------------------------------------------------------
// 2x2 max-pool, floor output size.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          int H, int W, int Wp, int plane, int Ho, int Wo, int Wpo,
                                                          int plane_o) {
    const int c = blockIdx.y;
    const int q = Wpo + blockIdx.x * 256 + threadIdx.x;  // output position, rows 1..Ho
    if (q >= (Ho + 1) * Wpo) return;
    const int r = q / Wpo, x = q - r * Wpo;
    float v = 0.f;
    if (x >= 1 && x <= Wo) {
        const float* p = in + (size_t)c * plane + (2 * (r - 1) + 1) * Wp + 2 * (x - 1) + 1;
        v = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[Wp], p[Wp + 1]));
    }
    out[(size_t)c * plane_o + q] = v;
}

__global__ __launch_bounds__(256) void maxpool_bwd_relu_kernel(const float* __restrict__ act,
                                                               const float* __restrict__ pooled,
                                                               const float* __restrict__ dpooled,
                                                               float* __restrict__ dact, int H, int W, int Wp, int plane,
                                                               int Ho, int Wo, int Wpo, int plane_o) {
    const int c = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;  // window index
    if (i >= Ho * Wo) return;
    const int yo = i / Wo, xo = i - yo * Wo;
    const size_t qo = (size_t)c * plane_o + (yo + 1) * Wpo + xo + 1;
    const float pm = pooled[qo];
    const float g = (pm > 0.f) ? dpooled[qo] : 0.f;  // max <= 0 -> every ReLU gate in the window is closed
    const size_t base = (size_t)c * plane + (2 * yo + 1) * Wp + 2 * xo + 1;
    const float v00 = act[base], v01 = act[base + 1], v10 = act[base + Wp], v11 = act[base + Wp + 1];
    // first maximum in row-major order gets the gradient (ATen max_pool2d_with_indices)
    const bool s00 = v00 == pm;
    const bool s01 = !s00 && v01 == pm;
    const bool s10 = !s00 && !s01 && v10 == pm;
    const bool s11 = !s00 && !s01 && !s10;
    dact[base] = s00 ? g : 0.f;
    dact[base + 1] = s01 ? g : 0.f;
    dact[base + Wp] = s10 ? g : 0.f;
    dact[base + Wp + 1] = s11 ? g : 0.f;
}

}  // namespace sm

extern "C" {

int sm_fmap_row_stride(int W) { return sm::row_stride(W); }
int sm_fmap_plane(int H, int W) { return sm::plane_size(H, W); }
int sm_abi_version(void) { return 1; }

int sm_conv3x3(const float* in, const float* wt, const float* bias, float* out, const float* gate, int Cin_pad,
               int Cout, int H, int W, int flags, void* stream) {
    if (Cout % 64 != 0 || Cin_pad % 4 != 0 || (Cin_pad > 4 && Cin_pad % 8 != 0)) return (int)hipErrorInvalidValue;
    sm::ConvArgs a{in, wt, bias, out, gate, Cin_pad, Cout, H, W, sm::row_stride(W), sm::plane_size(H, W), 0, 0};
    hipStream_t s = (hipStream_t)stream;
    switch (flags) {
        case SM_EPI_BIAS_RELU: return sm::dispatch_conv<SM_EPI_BIAS_RELU>(a, s);
        case 0: return sm::dispatch_conv<0>(a, s);
        case SM_EPI_RELU_MASK: return sm::dispatch_conv<SM_EPI_RELU_MASK>(a, s);
        case SM_EPI_RELU_MASK | SM_EPI_ADD: return sm::dispatch_conv<SM_EPI_RELU_MASK | SM_EPI_ADD>(a, s);
        case SM_EPI_ADD: return sm::dispatch_conv<SM_EPI_ADD>(a, s);
        default: return (int)hipErrorInvalidValue;
    }
}

int sm_conv3x3_dgrad_c3(const float* dz, const float* wd, float* out, int Cin, int H, int W, void* stream) {
    const int Wp = sm::row_stride(W), plane = sm::plane_size(H, W);
    const int n = H * Wp;
    hipLaunchKernelGGL(sm::conv3x3_dgrad_c3_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dz, wd,
                       out, Cin, H, W, Wp, plane);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_maxpool2x2_fwd(const float* in, float* out, int C, int H, int W, void* stream) {
    const int Ho = H / 2, Wo = W / 2;
    if (Ho < 1 || Wo < 1) return (int)hipErrorInvalidValue;
    const int Wp = sm::row_stride(W), plane = sm::plane_size(H, W);
    const int Wpo = sm::row_stride(Wo), plane_o = sm::plane_size(Ho, Wo);
    const int n = Ho * Wpo;
    hipLaunchKernelGGL(sm::maxpool_fwd_kernel, dim3((n + 255) / 256, C), dim3(256), 0, (hipStream_t)stream, in, out, H,
                       W, Wp, plane, Ho, Wo, Wpo, plane_o);
    SM_LAUNCH_CHECK();
    return 0;
}

int sm_maxpool2x2_bwd_relu(const float* act, const float* pooled, const float* dpooled, float* dact, int C, int H,
                           int W, void* stream) {
    const int Ho = H / 2, Wo = W / 2;
    if (Ho < 1 || Wo < 1) return (int)hipErrorInvalidValue;
    const int Wp = sm::row_stride(W), plane = sm::plane_size(H, W);
    const int Wpo = sm::row_stride(Wo), plane_o = sm::plane_size(Ho, Wo);
    const int n = Ho * Wo;
    hipLaunchKernelGGL(sm::maxpool_bwd_relu_kernel, dim3((n + 255) / 256, C), dim3(256), 0, (hipStream_t)stream, act,
                       pooled, dpooled, dact, H, W, Wp, plane, Ho, Wo, Wpo, plane_o);
    SM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

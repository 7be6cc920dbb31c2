// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 as a function of the number of independent accumulators
// and of the waves per SIMD. Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters, long long* cyc) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    long long t0 = __builtin_readcyclecounter();
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(threadIdx.x * 3 + e); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 24 / NACC; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC>
void run(int threads, float* out, long long* cyc) {
    const int iters = 2000, blocks = 256 * (512 / threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, threads>>>(out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, threads>>>(out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfmas = (double)blocks * (threads / 64) * iters * 24;
    double tf = mfmas * 32768.0 / (ms * 1e-3) / 1e12;
    // cycles per MFMA per SIMD assuming 2.4 GHz: SIMD executes (blocks*waves/1024) waves' worth
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("  wave0 of block0: %lld memtime ticks for %d MFMAs = %.1f ticks/MFMA; ticks/elapsed = %.2f GHz-equivalent\n", c, iters * 24, (double)c / (iters * 24), c / (ms * 1e-3) / 1e9 / (blocks / 256.0 / (512 / threads) > 1 ? 1 : 1));
    printf("NACC=%d threads=%d: %.3f ms  %.0f TF/s bf16 (%.1f%% of 2500)\n", NACC, threads, ms, tf, tf / 25.0);
}

int main() {
    float* out; (void)hipMalloc(&out, 1024 * 512 * 4 * sizeof(float)); long long* cyc; (void)hipMalloc(&cyc, 64);
    for (int t : {256, 512}) { run<1>(t, out, cyc); run<4>(t, out, cyc); }
    return 0;
}

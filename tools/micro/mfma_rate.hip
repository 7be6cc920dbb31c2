// Micro-benchmark: what a PURE v_mfma_f32_32x32x16_bf16 loop sustains on this part as a function of the OPERAND DATA
// (zeros / small integers / uniform random [-1, 1)), the waves per SIMD and the number of independent accumulators.
// No memory traffic, no LDS, no VALU inside the loop: this is the matrix pipes alone under the chip's power management
// (MI355X_MICROARCH.md "DVFS give-back"; cdna_hip_programming.md 5.4 rule 25: zero-filled operands run ~20 % faster
// than random ones at the same instruction stream). The conv kernel's roofline fraction is quoted against the
// nominal 2500 TFLOP/s; this table says how much of that a real-data instruction stream can get at all.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip       Run: ./mfma_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512) void k(const bf16x8* __restrict__ ops, float* out, int iters, long long* cyc) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // four A and four B fragments per lane, different in every lane (rotated through the loop: operand toggling)
    bf16x8 a[4], b[4];
    const int lane_slot = (blockIdx.x * blockDim.x + threadIdx.x) % 4096;
    for (int j = 0; j < 4; ++j) {
        a[j] = ops[lane_slot * 8 + j];
        b[j] = ops[lane_slot * 8 + 4 + j];
    }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 24 / NACC; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + rep) & 3], b[(i * 3 + rep + 1) & 3], acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

static unsigned short f2bf(float x) {
    unsigned u;
    memcpy(&u, &x, 4);
    return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
}

template <int NACC>
void run(const char* fill, const bf16x8* ops, int threads, int blocks_per_cu, float* out, long long* cyc) {
    const int iters = 200000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<NACC><<<blocks, threads>>>(ops, out, 2000, cyc);   // warm-up: clocks settle
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, threads>>>(ops, out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * (threads / 64) * iters * 24;
    const double tf = mfmas * 32768.0 / (ms * 1e-3) / 1e12;
    long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double waves_per_simd = (double)blocks_per_cu * (threads / 64) / 4.0;
    // shader cycles one wave spent / wall time = effective shader clock; cycles per MFMA per SIMD = c / (its MFMAs * waves sharing the SIMD)
    printf("%-8s acc=%d waves/SIMD=%.0f: %7.1f ms  %6.0f TFLOP/s bf16 = %5.1f %% of 2500   clock %.2f GHz   %.1f cyc/MFMA/SIMD\n",
           fill, NACC, waves_per_simd, ms, tf, tf / 25.0, c / (ms * 1e-3) / 1e9, (double)c / ((double)iters * 24 * waves_per_simd));
}

int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 4 * 512 * sizeof(float));
    long long* cyc;
    (void)hipMalloc(&cyc, 64);
    const size_t n = 4096 * 8 * 8;   // bf16 elements
    bf16x8* ops;
    (void)hipMalloc(&ops, n * 2);
    srand(1);
    for (int mode = 0; mode < 3; ++mode) {
        const char* name = mode == 0 ? "zeros" : mode == 1 ? "smallint" : "random";
        std::vector<unsigned short> h(n);
        for (size_t i = 0; i < n; ++i) {
            float v = 0.f;
            if (mode == 1) v = (float)(rand() % 7 - 3);
            if (mode == 2) v = 2.f * (float)rand() / (float)RAND_MAX - 1.f;
            h[i] = f2bf(v);
        }
        (void)hipMemcpy(ops, h.data(), n * 2, hipMemcpyHostToDevice);
        run<4>(name, ops, 256, 1, out, cyc);    // one wave per SIMD, 4 independent accumulators
        run<8>(name, ops, 256, 1, out, cyc);    // one wave per SIMD, 8
        run<4>(name, ops, 256, 2, out, cyc);    // two waves per SIMD (the conv kernel's residency)
        run<8>(name, ops, 256, 2, out, cyc);
    }
    return 0;
}

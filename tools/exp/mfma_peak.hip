// Micro-benchmark: sustained v_mfma_f32_16x16x4_f32 rate and shader clock under full-chip MFMA load.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, unsigned long long* clk) {
    f32x4 acc[NACC];
#pragma unroll
    for (int f = 0; f < NACC; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int f = 0; f < NACC; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[f], 0, 0, 0);
        a += 1e-6f;
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < NACC; ++f) s += acc[f][0] + acc[f][1] + acc[f][2] + acc[f][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma32_loop(float* out, int iters, unsigned long long* clk) {
    f32x16 acc[NACC];
#pragma unroll
    for (int f = 0; f < NACC; ++f)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[f][q] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int f = 0; f < NACC; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[f], 0, 0, 0);
        a += 1e-6f;
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < NACC; ++f)
#pragma unroll
        for (int q = 0; q < 16; ++q) s += acc[f][q];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int NACC>
static int run32(float* out, unsigned long long* clk, hipEvent_t e0, hipEvent_t e1) {
    for (int G : {256, 512, 1024}) {
        const int iters = 4096;
        float best = 1e9f; unsigned long long h[2] = {0, 0};
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(mfma32_loop<NACC>, dim3(G), dim3(256), 0, 0, out, iters, clk);
            CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        }
        const double flops = (double)G * 4 * iters * NACC * 4096.0;
        printf("32x32x2 NACC=%d G=%d: %.3f ms, %.1f TFLOP/s; %.0f MHz; cycles per MFMA per wave = %.2f\n", NACC, G, best,
               flops / best * 1e-9, (double)h[0] / ((double)h[1] / 100.0), (double)h[0] / (iters * (double)NACC));
    }
    return 0;
}

int main() {
    float* out; unsigned long long* clk;
    CK(hipMalloc(&out, 4096 * 256 * sizeof(float))); CK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (run32<1>(out, clk, e0, e1) || run32<2>(out, clk, e0, e1) || run32<4>(out, clk, e0, e1)) return 1;
    for (int G : {256, 512, 1024, 2048}) {
        const int iters = 4096;
        float best = 1e9f; unsigned long long h[2] = {0, 0};
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(mfma_loop<5>, dim3(G), dim3(256), 0, 0, out, iters, clk);
            CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        }
        const double flops = (double)G * 4 * iters * 5 * 2048.0;
        printf("G=%d: %.3f ms, %.1f TFLOP/s; WG0: %llu shader clocks / %llu ref ticks (100 MHz) => %.0f MHz; cycles per MFMA per wave = %.2f\n", G, best,
               flops / best * 1e-9, h[0], h[1], (double)h[0] / ((double)h[1] / 100.0), (double)h[0] / (iters * 5.0));
    }
    return 0;
}

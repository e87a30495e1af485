// v_mfma_f32_32x32x16_bf16 with ONE wave per SIMD: cycles per MFMA as a function of the dependence pattern.
// nc_fwd_x3q_kernel issues 12 MFMAs in a row into the same accumulator tile (6 products x 2 k blocks); is a dependent chain slower than
// the 32 cycles the instruction occupies the pipe?   CHAIN = consecutive MFMAs into one accumulator before moving to the next;
// NACC = accumulators cycled through.  (NACC = 5, CHAIN = 12 is the kernel's pattern; CHAIN = 1 is full interleave.)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip && ./mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int CHAIN, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters, unsigned long long* clk) {
    f32x16 acc[NACC];
#pragma unroll
    for (int f = 0; f < NACC; ++f)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[f][q] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int q = 0; q < 8; ++q) { a[q] = (__bf16)(threadIdx.x * 1e-3f + q); b[q] = (__bf16)(blockIdx.x * 1e-4f + q); }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int f = 0; f < NACC; ++f)
#pragma unroll
            for (int c = 0; c < CHAIN; ++c) acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[f], 0, 0, 0);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < NACC; ++f)
#pragma unroll
        for (int q = 0; q < 16; ++q) s += acc[f][q];
    out[(size_t)blockIdx.x * 64 * WAVES + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

// interleaved: round-robin over NACC accumulators, CHAIN rounds per iteration (consecutive MFMAs always independent for NACC >= 2)
template <int NACC, int ROUNDS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void kr(float* out, int iters, unsigned long long* clk) {
    f32x16 acc[NACC];
#pragma unroll
    for (int f = 0; f < NACC; ++f)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[f][q] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int q = 0; q < 8; ++q) { a[q] = (__bf16)(threadIdx.x * 1e-3f + q); b[q] = (__bf16)(blockIdx.x * 1e-4f + q); }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < ROUNDS; ++c)
#pragma unroll
            for (int f = 0; f < NACC; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[f], 0, 0, 0);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < NACC; ++f)
#pragma unroll
        for (int q = 0; q < 16; ++q) s += acc[f][q];
    out[(size_t)blockIdx.x * 64 * WAVES + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <typename F>
static int timeit(const char* label, F launch, int per_iter, int waves_total, float* out, unsigned long long* clk) {
    const int iters = 2048;
    unsigned long long h[2];
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) { CK(hipEventRecord(e0, 0)); launch(iters); CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1)); }
    CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    printf("%-58s %6.2f cycles per MFMA per wave   (%4.0f MHz)  %7.1f TFLOP/s per wave-slot x waves\n", label, (double)h[0] / ((double)iters * per_iter),
           (double)h[0] / ((double)h[1] / 100.0), (double)iters * per_iter * 32768.0 * waves_total / (ms * 1e-3) * 1e-12);
    return 0;
}

int main() {
    float* out; unsigned long long* clk;
    CK(hipMalloc(&out, 4096 * 512 * sizeof(float))); CK(hipMalloc(&clk, 16));
#define RUN(K, NACC, C, W, G, label) timeit(label, [&](int it) { hipLaunchKernelGGL((K<NACC, C, W>), dim3(G), dim3(64 * W), 0, 0, out, it, clk); }, NACC * C, G * W, out, clk)
    RUN(k, 1, 12, 4, 256, "1 wave/SIMD, one accumulator, fully dependent");
    RUN(k, 5, 12, 4, 256, "1 wave/SIMD, 5 accumulators, chains of 12 (kernel's pattern)");
    RUN(k, 5, 6, 4, 256, "1 wave/SIMD, 5 accumulators, chains of 6");
    RUN(k, 5, 2, 4, 256, "1 wave/SIMD, 5 accumulators, chains of 2");
    RUN(kr, 2, 6, 4, 256, "1 wave/SIMD, 2 accumulators round-robin");
    RUN(kr, 5, 12, 4, 256, "1 wave/SIMD, 5 accumulators round-robin");
    RUN(k, 5, 12, 8, 256, "2 waves/SIMD, 5 accumulators, chains of 12");
    RUN(kr, 5, 12, 8, 256, "2 waves/SIMD, 5 accumulators round-robin");
    return 0;
}

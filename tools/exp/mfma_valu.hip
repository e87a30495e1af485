// One wave per SIMD: how many independent VALU instructions hide behind a v_mfma_f32_32x32x16_bf16 of the same wave?
// Loop body: 1 MFMA (dependent chain on one accumulator) followed by NV fp32 VALU ops on registers the MFMA does not touch.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip && ./mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// KIND 0: v_fma_f32 (independent chains over 8 registers); 1: v_cvt_pk_bf16_f32; 2: v_and_b32; 3: v_pk_add_f32; 4: ds_read_b128 (LDS)
template <int NV, int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
    __shared__ float lds[4096];
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int q = 0; q < 8; ++q) { a[q] = (__bf16)(threadIdx.x * 1e-3f + q); b[q] = (__bf16)(blockIdx.x * 1e-4f + q); }
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = threadIdx.x * 0.5f + q;
    lds[threadIdx.x] = v[0];
    __syncthreads();
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                float& x = v[n & 7];
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(v[(n + 1) & 7]));
                else if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(v[(n + 1) & 7]));
                else if (KIND == 2) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(x));
                else if (KIND == 3) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(v[(n + 1) & 7]));
            }
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += acc[q];
#pragma unroll
    for (int q = 0; q < 8; ++q) s += v[q];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s + lds[(threadIdx.x * 7) & 4095];
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int NV, int KIND>
static int run(const char* kind, float* out, unsigned long long* clk) {
    const int iters = 1024;
    unsigned long long h[2];
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL((k<NV, KIND>), dim3(256), dim3(256), 0, 0, out, iters, clk); CK(hipDeviceSynchronize()); }
    CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    printf("%-20s %2d per MFMA: %6.2f cycles per (MFMA + VALU group)\n", kind, NV, (double)h[0] / ((double)iters * 8));
    return 0;
}

int main() {
    float* out; unsigned long long* clk;
    CK(hipMalloc(&out, 4096 * 256 * sizeof(float))); CK(hipMalloc(&clk, 16));
    run<0, 0>("v_fma_f32", out, clk); run<2, 0>("v_fma_f32", out, clk); run<4, 0>("v_fma_f32", out, clk); run<5, 0>("v_fma_f32", out, clk);
    run<6, 0>("v_fma_f32", out, clk); run<8, 0>("v_fma_f32", out, clk); run<12, 0>("v_fma_f32", out, clk);
    run<4, 1>("v_cvt_pk_bf16_f32", out, clk); run<6, 1>("v_cvt_pk_bf16_f32", out, clk); run<8, 1>("v_cvt_pk_bf16_f32", out, clk);
    run<4, 2>("v_and_b32", out, clk); run<6, 2>("v_and_b32", out, clk); run<8, 2>("v_and_b32", out, clk);
    run<4, 3>("v_sub_f32", out, clk); run<6, 3>("v_sub_f32", out, clk); run<8, 3>("v_sub_f32", out, clk);
    return 0;
}

// One wave per SIMD, four per CU: v_mfma_f32_32x32x16_bf16 fed by ds_read_b128 fragments, the way nc_fwd_x3q_kernel's inner loop is:
// per fragment three 16-byte LDS reads (one fragment ahead, claimed with s_waitcnt lgkmcnt(3)) and MPF MFMAs that use them.
// Variants: row stride of the LDS image (80 bytes as in the kernel / 144), reads per fragment, MFMAs per fragment.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_lds mfma_lds.hip && ./mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ unsigned char L[];

template <int OFF> __device__ __forceinline__ void rd(u32x4& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }

// RS = row stride in bytes, NRD = LDS reads per fragment (0..3), MPF = MFMAs per fragment, SYNC = barrier every 10 fragments
template <int RS, int NRD, int MPF, bool SYNC>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < 40000 / 4; e += 256) reinterpret_cast<unsigned*>(L)[e] = 0x3f803f80u;
    __syncthreads();
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)L + (unsigned)((lane & 31) * RS + (lane >> 5) * 16);
    f32x16 acc[5];
#pragma unroll
    for (int f = 0; f < 5; ++f)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[f][q] = 0.f;
    bf16x8 b;
#pragma unroll
    for (int q = 0; q < 8; ++q) b[q] = (__bf16)(1.0f + q);
    u32x4 fa[2][3];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 3; ++q) fa[s][q] = (u32x4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (NRD >= 1) rd<0>(fa[0][0], addr);
        if (NRD >= 2) rd<12800>(fa[0][1], addr);
        if (NRD >= 3) rd<25600>(fa[0][2], addr);
#define FRAG(J)                                                                                                              \
        {                                                                                                                    \
            if ((J) + 1 < 10) {                                                                                              \
                if (NRD >= 1) rd<(((J) + 1) / 2) * 32 * RS + (((J) + 1) % 2) * 32>(fa[((J) + 1) & 1][0], addr);              \
                if (NRD >= 2) rd<(((J) + 1) / 2) * 32 * RS + (((J) + 1) % 2) * 32 + 12800>(fa[((J) + 1) & 1][1], addr);      \
                if (NRD >= 3) rd<(((J) + 1) / 2) * 32 * RS + (((J) + 1) % 2) * 32 + 25600>(fa[((J) + 1) & 1][2], addr);      \
                if (NRD == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa[(J) & 1][0]), "+v"(fa[(J) & 1][1]), "+v"(fa[(J) & 1][2]));   \
                else if (NRD == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[(J) & 1][0]), "+v"(fa[(J) & 1][1]));        \
                else if (NRD == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fa[(J) & 1][0]));                              \
            } else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[(J) & 1][0]), "+v"(fa[(J) & 1][1]), "+v"(fa[(J) & 1][2]));  \
            f32x16 c = acc[(J) / 2];                                                                                         \
            _Pragma("unroll") for (int m = 0; m < MPF; ++m)                                                                  \
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[(J) & 1][m % 3]), b, c, 0, 0, 0);  \
            acc[(J) / 2] = c;                                                                                                \
        }
        FRAG(0) FRAG(1) FRAG(2) FRAG(3) FRAG(4) FRAG(5) FRAG(6) FRAG(7) FRAG(8) FRAG(9)
#undef FRAG
        if (SYNC) __syncthreads();
    }
    const unsigned long long c1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < 5; ++f)
#pragma unroll
        for (int q = 0; q < 16; ++q) s += acc[f][q];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}

template <int RS, int NRD, int MPF, bool SYNC>
static int run(const char* label, float* out, unsigned long long* clk) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<RS, NRD, MPF, SYNC>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    const int iters = 512;
    unsigned long long h;
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL((k<RS, NRD, MPF, SYNC>), dim3(256), dim3(256), 80 * 1024, 0, out, iters, clk); CK(hipDeviceSynchronize()); }
    CK(hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost));
    printf("%-64s %6.2f cycles per MFMA\n", label, (double)h / ((double)iters * 10 * MPF));
    return 0;
}

int main() {
    float* out; unsigned long long* clk;
    CK(hipMalloc(&out, 4096 * 256 * sizeof(float))); CK(hipMalloc(&clk, 16));
    run<80, 0, 6, false>("no LDS reads, 6 MFMAs per fragment", out, clk);
    run<80, 3, 6, false>("3 reads per fragment (stride 80 B), 6 MFMAs", out, clk);
    run<80, 3, 6, true>("3 reads per fragment (stride 80 B), 6 MFMAs, barrier per 10", out, clk);
    run<144, 3, 6, false>("3 reads per fragment (stride 144 B), 6 MFMAs", out, clk);
    run<80, 1, 6, false>("1 read per fragment (stride 80 B), 6 MFMAs", out, clk);
    run<80, 2, 6, false>("2 reads per fragment (stride 80 B), 6 MFMAs", out, clk);
    run<80, 3, 3, false>("3 reads per fragment (stride 80 B), 3 MFMAs", out, clk);
    run<80, 3, 12, false>("3 reads per fragment (stride 80 B), 12 MFMAs", out, clk);
    return 0;
}

// Micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 rate (and shader clock) with every SIMD of the chip issuing: the practical ceiling of the bf16x3
// tiles (six such MFMAs per 32x32x16 fp32 product: ceiling = rate / 6).  Variants: waves per SIMD 1 / 2 / 4, accumulators per wave 1 / 4; with the
// fragment reads re-read from LDS or not; operands = small regular values, or (last block) random bf16 -- the chip holds a lower clock for those.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_bf16 tools/exp/mfma_bf16.hip && /tmp/mfma_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS, bool RAND = false>
__global__ __launch_bounds__(256) void loop(float* out, int iters, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) {
        if (RAND) {          // two bf16 per word: random sign and mantissa, exponents 124..131 (|x| in 0.125 .. 16): the operands of a real product
            unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
            const unsigned lo = (h & 0x807fu) | ((124u + ((h >> 20) & 7u)) << 7), hi = ((h >> 16) & 0x807fu) | ((124u + ((h >> 8) & 7u)) << 7);
            lds[i] = __builtin_bit_cast(float, lo | (hi << 16));
        } else lds[i] = 1e-3f * i;
    }
    __syncthreads();
    f32x16 acc[NACC];
#pragma unroll
    for (int f = 0; f < NACC; ++f)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[f][q] = 0.f;
    bf16x8 a[3], b[3];
    const f32x4* L = reinterpret_cast<const f32x4*>(lds) + (threadIdx.x & 63);
#pragma unroll
    for (int m = 0; m < 3; ++m) { a[m] = __builtin_bit_cast(bf16x8, L[64 * m]); b[m] = __builtin_bit_cast(bf16x8, L[64 * (m + 3)]); }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                a[m] = __builtin_bit_cast(bf16x8, *(volatile const f32x4*)(L + 64 * m + 512 * (it & 3)));
                b[m] = __builtin_bit_cast(bf16x8, *(volatile const f32x4*)(L + 64 * (m + 3) + 512 * ((it >> 1) & 3)));
            }
        }
#pragma unroll
        for (int f = 0; f < NACC; ++f) {
            f32x16 v = acc[f];
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], v, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], v, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], v, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], v, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], v, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], v, 0, 0, 0);
            acc[f] = v;
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < NACC; ++f)
#pragma unroll
        for (int q = 0; q < 16; ++q) s += acc[f][q];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int NACC, bool LDS, bool RAND = false>
static int run(float* out, unsigned long long* clk, hipEvent_t e0, hipEvent_t e1, int cus) {
    for (int wps : {1, 2, 4}) {                   // waves per SIMD: blocks of 256 threads (4 waves = one per SIMD), wps blocks per CU
        const int G = cus * wps, iters = 2048;
        float best = 1e9f; unsigned long long h[2] = {0, 0};
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((loop<NACC, LDS, RAND>), dim3(G), dim3(256), 0, 0, out, iters, clk);
            CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) { best = ms; CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost)); }
        }
        const double fl = 2.0 * 32 * 32 * 16 * 6.0 * NACC * iters * (double)G * 4;
        printf("acc %d lds %d random operands %d waves/SIMD %d: %8.3f ms  %7.1f TF bf16 = %6.1f TF as bf16x3   shader clock %.0f MHz (clock64 %llu / wall %llu x 100 MHz), cycles per MFMA per SIMD %.1f\n",
               NACC, (int)LDS, (int)RAND, wps, best, fl / best / 1e9, fl / best / 1e9 / 6.0, 100.0 * h[0] / (double)h[1], h[0], h[1], (double)h[0] / (6.0 * NACC * iters * wps));
    }
    return 0;
}

// the same products on v_mfma_f32_16x16x32_bf16: four 16x16 accumulators per 32x32 one (same flops per loop iteration, same operand bytes)
template <int NACC, bool RAND>
__global__ __launch_bounds__(256) void loop16(float* out, int iters, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) {
        if (RAND) {
            unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
            const unsigned lo = (h & 0x807fu) | ((124u + ((h >> 20) & 7u)) << 7), hi = ((h >> 16) & 0x807fu) | ((124u + ((h >> 8) & 7u)) << 7);
            lds[i] = __builtin_bit_cast(float, lo | (hi << 16));
        } else lds[i] = 1e-3f * i;
    }
    __syncthreads();
    f32x4 acc[NACC][4];
#pragma unroll
    for (int f = 0; f < NACC; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[f][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a[2][3], b[2][3];
    const f32x4* L = reinterpret_cast<const f32x4*>(lds) + (threadIdx.x & 63);
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                a[h][m] = __builtin_bit_cast(bf16x8, *(volatile const f32x4*)(L + 64 * m + 256 * h + 512 * (it & 1)));
                b[h][m] = __builtin_bit_cast(bf16x8, *(volatile const f32x4*)(L + 64 * (m + 3) + 256 * h + 512 * ((it >> 1) & 1)));
            }
#pragma unroll
        for (int f = 0; f < NACC; ++f)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = acc[f][q];
                const int i = q >> 1, y = q & 1;
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[y][2], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][2], b[y][0], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][1], b[y][1], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[y][1], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][1], b[y][0], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[y][0], v, 0, 0, 0);
                acc[f][q] = v;
            }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < NACC; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) s += acc[f][q][0] + acc[f][q][1] + acc[f][q][2] + acc[f][q][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
template <int NACC, bool RAND>
static int run16(float* out, unsigned long long* clk, hipEvent_t e0, hipEvent_t e1, int cus) {
    for (int wps : {1, 2, 4}) {
        const int G = cus * wps, iters = 2048;
        float best = 1e9f; unsigned long long h[2] = {0, 0};
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((loop16<NACC, RAND>), dim3(G), dim3(256), 0, 0, out, iters, clk);
            CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) { best = ms; CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost)); }
        }
        const double fl = 2.0 * 16 * 16 * 32 * 6.0 * 4 * NACC * iters * (double)G * 4;
        printf("16x16x32: acc %d x 4, lds 1 random operands %d waves/SIMD %d: %8.3f ms  %7.1f TF bf16 = %6.1f TF as bf16x3   shader clock %.0f MHz\n",
               NACC, (int)RAND, wps, best, fl / best / 1e9, fl / best / 1e9 / 6.0, 100.0 * h[0] / (double)h[1]);
    }
    return 0;
}

int main() {
    int dev = 0; hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, dev));
    const int cus = pr.multiProcessorCount;
    printf("%s, %d CUs, clockRate %d kHz\n", pr.name, cus, pr.clockRate);
    float* out; unsigned long long* clk;
    CK(hipMalloc(&out, sizeof(float) * 256 * cus * 4)); CK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (run<1, false>(out, clk, e0, e1, cus)) return 1;
    if (run<4, false>(out, clk, e0, e1, cus)) return 1;
    if (run<4, true>(out, clk, e0, e1, cus)) return 1;
    if (run<4, true, true>(out, clk, e0, e1, cus)) return 1;
    if (run16<4, false>(out, clk, e0, e1, cus)) return 1;
    if (run16<4, true>(out, clk, e0, e1, cus)) return 1;
    return 0;
}

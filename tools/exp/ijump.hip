// Micro-benchmark: cost of TAKEN BRANCHES into code that has not been fetched yet in this launch.  The kernel walks through
// NJ blocks of code that sit 4 KB apart (each: a few VALU ops, then a jump to the next); first pass (cold instruction
// cache) vs second pass, in a graph chain of alternating kernels as in the step programs.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define PAD4(x) asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0" : "+v"(x));
#define PAD16(x) PAD4(x) PAD4(x) PAD4(x) PAD4(x)
#define PAD64(x) PAD16(x) PAD16(x) PAD16(x) PAD16(x)
#define PAD256(x) PAD64(x) PAD64(x) PAD64(x) PAD64(x)
#define PAD1K(x) PAD256(x) PAD256(x) PAD256(x) PAD256(x)     // 4 KB of never-executed filler

template <int NJ>
__global__ __launch_bounds__(256) void jump_kernel(float* out, long long* rec, int sel) {
    float x = threadIdx.x;
    long long d[2];
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
        const long long t0 = clock64();
        // `sel` is always 0 at run time: the filler blocks are skipped by taken branches
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (sel == j + 1) { PAD1K(x) }
            PAD4(x)
        }
        d[it] = clock64() - t0;
    }
    out[blockIdx.x * 256 + threadIdx.x] = x;
    if (threadIdx.x == 0) { rec[blockIdx.x * 2] = d[0]; rec[blockIdx.x * 2 + 1] = d[1]; }
}
__global__ __launch_bounds__(256) void other_kernel(float* out) {
    float x = threadIdx.x; PAD1K(x) PAD1K(x) PAD1K(x) PAD1K(x)
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

template <int NJ> int run(float* out, long long* rec, hipStream_t st) {
    const int G = 512, L = 20;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int p = 0; p < L; ++p) {
        hipLaunchKernelGGL(jump_kernel<NJ>, dim3(G), dim3(256), 0, st, out, rec, 0);
        hipLaunchKernelGGL(other_kernel, dim3(G), dim3(256), 0, st, out);
    }
    hipLaunchKernelGGL(jump_kernel<NJ>, dim3(G), dim3(256), 0, st, out, rec, 0);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); }
    long long h[1024];
    CK(hipMemcpy(h, rec, sizeof(h), hipMemcpyDeviceToHost));
    double a = 0, b = 0;
    for (int i = 0; i < G; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
    printf("%2d taken branches over 4 KB gaps: first pass %.0f cycles, second pass %.0f cycles -> %.0f cycles (%.2f us) per cold branch target\n",
           NJ, a / G, b / G, (a - b) / G / NJ, (a - b) / G / NJ / 2200.0);
    return 0;
}

int main() {
    float* out; long long* rec;
    CK(hipMalloc(&out, 512 * 256 * 4)); CK(hipMalloc(&rec, 8192 * 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    if (run<1>(out, rec, st)) return 1;
    if (run<2>(out, rec, st)) return 1;
    if (run<4>(out, rec, st)) return 1;
    if (run<8>(out, rec, st)) return 1;
    return 0;
}

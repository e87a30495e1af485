// Micro-benchmark: cost of executing straight-line code for the first time in a launch (cold instruction cache)
// versus the second time.  Each launch is a fresh kernel in a graph chain, as in the step programs.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define OP4(x) asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0" : "+v"(x));
#define OP16(x) OP4(x) OP4(x) OP4(x) OP4(x)
#define OP64(x) OP16(x) OP16(x) OP16(x) OP16(x)
#define OP256(x) OP64(x) OP64(x) OP64(x) OP64(x)        // 1 KB of code
template <int KB>
__global__ __launch_bounds__(256) void code_kernel(float* out, long long* rec) {
    float x = threadIdx.x;
    long long d[2];
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
        const long long t0 = clock64();
        if (KB >= 1) { OP256(x) }
        if (KB >= 2) { OP256(x) }
        if (KB >= 4) { OP256(x) OP256(x) }
        if (KB >= 8) { OP256(x) OP256(x) OP256(x) OP256(x) }
        if (KB >= 16) { OP256(x) OP256(x) OP256(x) OP256(x) OP256(x) OP256(x) OP256(x) OP256(x) }
        d[it] = clock64() - t0;
    }
    out[blockIdx.x * 256 + threadIdx.x] = x;
    if (threadIdx.x == 0) { rec[blockIdx.x * 2] = d[0]; rec[blockIdx.x * 2 + 1] = d[1]; }
}

template <int KB> int run(float* out, long long* rec, hipStream_t st) {
    const int G = 512, L = 20;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int p = 0; p < L; ++p) {
        // alternate with other kernels so that consecutive launches never share code
        hipLaunchKernelGGL(code_kernel<KB>, dim3(G), dim3(256), 0, st, out, rec);
        hipLaunchKernelGGL(code_kernel<(KB == 16 ? 8 : 16)>, dim3(G), dim3(256), 0, st, out, rec + 4096);
    }
    hipLaunchKernelGGL(code_kernel<KB>, dim3(G), dim3(256), 0, st, out, rec);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); }
    long long h[1024];
    CK(hipMemcpy(h, rec, sizeof(h), hipMemcpyDeviceToHost));
    double a = 0, b = 0; long long mx = 0;
    for (int i = 0; i < G; ++i) { a += h[2 * i]; b += h[2 * i + 1]; mx = h[2 * i] > mx ? h[2 * i] : mx; }
    printf("%2d KB straight-line code: first pass %.0f cycles (max %lld), second pass %.0f cycles -> cold penalty %.2f us (%.0f ns per KB)\n",
           KB, a / G, mx, b / G, (a - b) / G / 2200.0, (a - b) / G / 2.2 / KB);
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
    if (run<16>(out, rec, st)) return 1;
    return 0;
}

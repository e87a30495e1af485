// Micro-benchmark: is workgroup -> XCD placement a deterministic round-robin on blockIdx.x, and what does a load of the
// previous kernel's output cost when producer and consumer workgroups sit on the same XCD vs. different ones?
// Producer kernel: workgroup b writes line b.  Consumer kernel: workgroup b reads the line written by workgroup b + shift.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void produce(float* buf, int G, float v) {
    buf[(size_t)blockIdx.x * 1024 + threadIdx.x] = v + threadIdx.x;          // 4 KB apart: one private set of lines per workgroup
}
__global__ __launch_bounds__(256) void consume(const float* buf, float* sink, long long* rec, int G, int shift, int slot) {
    const int src = (blockIdx.x + shift) % G;
    const long long t0 = clock64();
    const float a = buf[(size_t)src * 1024 + threadIdx.x];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    sink[(size_t)blockIdx.x * 256 + threadIdx.x] = a;
    if (threadIdx.x == 0) rec[(size_t)slot * 1024 + blockIdx.x] = t1 - t0;
}

int main() {
    const int G = 512, L = 30;
    float *buf, *sink; long long* rec;
    CK(hipMalloc(&buf, (size_t)G * 4096)); CK(hipMalloc(&sink, (size_t)G * 1024)); CK(hipMalloc(&rec, (size_t)L * 1024 * 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int shift : {0, 8, 16, 256, 1, 2, 4, 7, 9}) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int p = 0; p < L; ++p) {
            hipLaunchKernelGGL(produce, dim3(G), dim3(256), 0, st, buf, G, (float)p);
            hipLaunchKernelGGL(consume, dim3(G), dim3(256), 0, st, buf, sink, rec, G, shift, p);
        }
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); }
        static long long h[30 * 1024];
        CK(hipMemcpy(h, rec, sizeof(h), hipMemcpyDeviceToHost));
        double s = 0; int n = 0, fast = 0;
        for (int p = 5; p < L; ++p) for (int b = 0; b < G; ++b) { const long long v = h[(size_t)p * 1024 + b]; s += v; ++n; fast += v < 600; }
        printf("consumer b reads producer b+%-3d: mean %.0f cycles, %.0f %% of loads under 600 cycles\n", shift, s / n, 100.0 * fast / n);
    }
    return 0;
}

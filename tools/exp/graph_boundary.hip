// What does the boundary between two hipGraphLaunch calls on one stream cost on the DEVICE, beyond a kernel-to-kernel boundary inside a graph?
// (the pipelined train() launches one 34-kernel feature graph per call: chain stamps put 11.9 us between the tail of one and the head of the next)
//   hipcc --offload-arch=gfx950 -O3 -o graph_boundary graph_boundary.hip && ./graph_boundary
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256) void work(float* buf, int spin) {
    float v = buf[blockIdx.x * 256 + threadIdx.x];
    for (int k = 0; k < spin; ++k) v = v * 1.0001f + 0.5f;
    buf[blockIdx.x * 256 + threadIdx.x] = v;
}
int main() {
    float* a; CK(hipMalloc(&a, 256 * 256 * 4)); CK(hipMemset(a, 0, 256 * 256 * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int K = 34, N = 30;
    for (int spin : {0, 2000}) {
        auto capture = [&](int nk, hipGraphExec_t* x) -> int {
            hipGraph_t g;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int k = 0; k < nk; ++k) hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, st, a, spin);
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(x, g, nullptr, nullptr, 0));
            return 0;
        };
        hipGraphExec_t big, one[3];
        if (capture(K * N, &big)) return 1;
        for (int q = 0; q < 3; ++q) if (capture(K, &one[q])) return 1;
        float tb = 1e9f, ts = 1e9f, t3 = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            float ms;
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(big, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < tb) tb = ms;
            CK(hipEventRecord(e0, st)); for (int q = 0; q < N; ++q) CK(hipGraphLaunch(one[0], st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < ts) ts = ms;
            CK(hipEventRecord(e0, st)); for (int q = 0; q < N; ++q) CK(hipGraphLaunch(one[q % 3], st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < t3) t3 = ms;
        }
        printf("spin %4d: ONE graph of %d kernels %.1f us (%.2f us / kernel) | %d launches of a %d-kernel graph %.1f us: +%.2f us per graph launch | three execs in rotation %.1f us: +%.2f us per graph launch\n",
               spin, K * N, tb * 1e3f, tb * 1e3f / (K * N), N, K, ts * 1e3f, (ts - tb) * 1e3f / N, t3 * 1e3f, (t3 - tb) * 1e3f / N);
    }
    return 0;
}

// Micro-benchmark: do two INDEPENDENT chains of small dependent kernels overlap when they sit on two streams (or two branches
// of one hipGraph)?  Decides what the deferred critic/actor pipeline (two concurrent launch chains per train()) can gain.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void work(float* buf, int n, int spin) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v = buf[i % n];
    for (int k = 0; k < spin; ++k) v = v * 1.0001f + 0.5f;
    buf[i % n] = v;
}

int main() {
    float *a, *b; const int n = 512 * 256;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1, f, j; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f)); CK(hipEventCreate(&j));
    const int L = 200;
    for (int G : {64, 256, 512}) for (int spin : {0, 400, 2000}) {
        hipGraph_t g1, g2; hipGraphExec_t x1, x2;
        // one chain
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
        for (int k = 0; k < L; ++k) hipLaunchKernelGGL(work, dim3(G), dim3(256), 0, s1, a, n, spin);
        CK(hipStreamEndCapture(s1, &g1)); CK(hipGraphInstantiate(&x1, g1, nullptr, nullptr, 0));
        // two chains as branches of one graph
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
        CK(hipEventRecord(f, s1)); CK(hipStreamWaitEvent(s2, f, 0));
        for (int k = 0; k < L; ++k) hipLaunchKernelGGL(work, dim3(G), dim3(256), 0, s1, a, n, spin);
        for (int k = 0; k < L; ++k) hipLaunchKernelGGL(work, dim3(G), dim3(256), 0, s2, b, n, spin);
        CK(hipEventRecord(j, s2)); CK(hipStreamWaitEvent(s1, j, 0));
        CK(hipStreamEndCapture(s1, &g2)); CK(hipGraphInstantiate(&x2, g2, nullptr, nullptr, 0));
        float t1 = 1e9f, t2 = 1e9f, t3 = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            float ms;
            CK(hipEventRecord(e0, s1)); CK(hipGraphLaunch(x1, s1)); CK(hipEventRecord(e1, s1)); CK(hipStreamSynchronize(s1));
            CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < t1) t1 = ms;
            CK(hipEventRecord(e0, s1)); CK(hipGraphLaunch(x2, s1)); CK(hipEventRecord(e1, s1)); CK(hipStreamSynchronize(s1));
            CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < t2) t2 = ms;
            // two chains as two graphs on two streams
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s1)); CK(hipStreamWaitEvent(s2, e0, 0));
            CK(hipGraphLaunch(x1, s1)); CK(hipGraphLaunch(x1, s2));
            CK(hipEventRecord(j, s2)); CK(hipStreamWaitEvent(s1, j, 0)); CK(hipEventRecord(e1, s1)); CK(hipStreamSynchronize(s1));
            CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < t3) t3 = ms;
        }
        printf("G=%3d spin=%4d: one chain %.2f us/launch | two branches in one graph %.2f us per launch pair | two graphs on two streams %.2f\n",
               G, spin, t1 * 1000.f / L, t2 * 1000.f / L, t3 * 1000.f / L);
    }
    return 0;
}

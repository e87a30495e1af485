// Micro-benchmark: latency of the FIRST vector load of a freshly launched kernel (graph chain), for data written by the
// previous kernel vs. data that never changes, and of a second dependent load right behind it.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void probe(const float* constant_buf, const float* prev_out, float* my_out, long long* rec, int slot) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const long long t0 = clock64();
    const float a = __builtin_nontemporal_load(constant_buf + tid);           // never written
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    const float b = __builtin_nontemporal_load(prev_out + ((tid * 7) & 65535)); // written by the previous launch
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t2 = clock64();
    const float c = constant_buf[65536 + tid];                                  // second touch of the constant buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t3 = clock64();
    const float d = constant_buf[tid];                                          // same line again: L1/L2 hit
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t4 = clock64();
    my_out[tid & 65535] = a + b + c + d;
    if (threadIdx.x == 0) { long long* r = rec + ((size_t)slot * 512 + blockIdx.x) * 4; r[0] = t1 - t0; r[1] = t2 - t1; r[2] = t3 - t2; r[3] = t4 - t3; }
}

int main() {
    float *cb, *o0, *o1; long long* rec;
    const int L = 40;
    CK(hipMalloc(&cb, 1 << 22)); CK(hipMalloc(&o0, 65536 * 4)); CK(hipMalloc(&o1, 65536 * 4)); CK(hipMalloc(&rec, (size_t)L * 512 * 4 * 8));
    CK(hipMemset(cb, 0, 1 << 22)); CK(hipMemset(o0, 0, 65536 * 4)); CK(hipMemset(o1, 0, 65536 * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int G : {1, 256, 512}) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int p = 0; p < L; ++p) hipLaunchKernelGGL(probe, dim3(G), dim3(256), 0, st, cb, (p & 1) ? o0 : o1, (p & 1) ? o1 : o0, rec, p);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); }
        static long long h[40 * 512 * 4];
        CK(hipMemcpy(h, rec, sizeof(h), hipMemcpyDeviceToHost));
        double s[4] = {0, 0, 0, 0}; int n = 0;
        for (int p = 5; p < L; ++p) for (int b = 0; b < G; ++b) { for (int q = 0; q < 4; ++q) s[q] += h[((size_t)p * 512 + b) * 4 + q]; ++n; }
        printf("G=%3d WGs: first load (constant data) %.0f cyc | data of previous kernel %.0f cyc | constant, other line %.0f cyc | same line again %.0f cyc   (2200 cyc = 1 us)\n",
               G, s[0] / n, s[1] / n, s[2] / n, s[3] / n);
    }
    return 0;
}

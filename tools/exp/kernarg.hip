// Micro-benchmark: what do dependent scalar loads from the kernel-argument segment cost per launch (graph replay)?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct Rec { const float* a; float* c; int base; int pad[73]; };     // 304 bytes, like GemmTask
struct Table { int n; int pad; Rec r[8]; };

// two dependent kernarg round trips: scan the bases, then read the chosen record
__global__ __launch_bounds__(256) void k_two(Table tb) {
    int ti = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q) if (q < tb.n && (int)blockIdx.x >= tb.r[q].base) ti = q;
    const Rec& r = tb.r[ti];
    r.c[blockIdx.x * 256 + threadIdx.x] = r.a[threadIdx.x] + 1.f;
}
// one round trip: the record index is blockIdx.y
__global__ __launch_bounds__(256) void k_one(Table tb) {
    const Rec& r = tb.r[blockIdx.y];
    r.c[(blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = r.a[threadIdx.x] + 1.f;
}
// no table: plain pointer arguments
__global__ __launch_bounds__(256) void k_zero(const float* a, float* c) {
    c[blockIdx.x * 256 + threadIdx.x] = a[threadIdx.x] + 1.f;
}
// table in device memory behind one pointer argument (stays in L2 across replays)
__global__ __launch_bounds__(256) void k_dev(const Table* tb) {
    int ti = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q) if (q < tb->n && (int)blockIdx.x >= tb->r[q].base) ti = q;
    const Rec& r = tb->r[ti];
    r.c[blockIdx.x * 256 + threadIdx.x] = r.a[threadIdx.x] + 1.f;
}

int main() {
    float *a, *c; CK(hipMalloc(&a, 4096)); CK(hipMalloc(&c, 1024 * 256 * 4)); CK(hipMemset(a, 0, 4096));
    Table tb{}; tb.n = 8;
    for (int q = 0; q < 8; ++q) { tb.r[q].a = a; tb.r[q].c = c; tb.r[q].base = 32 * q; }
    Table* dtb; CK(hipMalloc(&dtb, sizeof(Table))); CK(hipMemcpy(dtb, &tb, sizeof(Table), hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int L = 200;
    for (int mode = 0; mode < 4; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int p = 0; p < L; ++p) {
            if (mode == 0) hipLaunchKernelGGL(k_zero, dim3(256), dim3(256), 0, st, a, c);
            else if (mode == 1) hipLaunchKernelGGL(k_one, dim3(32, 8), dim3(256), 0, st, tb);
            else if (mode == 2) hipLaunchKernelGGL(k_two, dim3(256), dim3(256), 0, st, tb);
            else hipLaunchKernelGGL(k_dev, dim3(256), dim3(256), 0, st, dtb);
        }
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const char* names[4] = {"plain pointer args", "by-value table, index = blockIdx.y (1 trip)", "by-value table, scan + record (2 trips)", "device-memory table (2 trips)"};
        printf("%-48s %.3f us per launch\n", names[mode], best * 1000.f / L);
    }
    return 0;
}

// Micro-benchmark: what does a workgroup's first scalar-load burst (its 256-byte task record) cost when the record comes
//   (a) from the kernel-argument segment of a graph node (by-value table, 3.5 KB, as gemm16_kernel reads it),
//   (b) from a table in device memory that nobody touched since the last replay (cold: 64 MB were streamed in between),
//   (c) from the same table after the PREVIOUS kernel of the chain touched its lines from every XCD (one wave-wide load per XCD)?
// hipcc --offload-arch=gfx950 -O3 -o recfetch recfetch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct Rec { const float* a; float* c; int v[60]; };          // 256 bytes
struct Table { int hdr[16]; Rec r[8]; int pad[80]; };          // ~2.4 KB by value
typedef __attribute__((address_space(4))) const Table CTable;

__device__ __forceinline__ void consume(const Rec& r, unsigned long long t0, unsigned long long* out) {
    // the whole record as one burst, pinned
    int s = 0;
#pragma unroll
    for (int q = 0; q < 60; ++q) s += r.v[q];
    const float* a = r.a; float* c = r.c;
    asm volatile("" :: "s"(s), "s"(a), "s"(c));
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    c[blockIdx.x * 256 + threadIdx.x] = a[threadIdx.x] + (float)s;
}
__global__ __launch_bounds__(256) void k_byvalue(Table tb, unsigned long long* out) {
    const unsigned long long t0 = clock64();
    consume(tb.r[blockIdx.x & 7], t0, out);
}
__global__ __launch_bounds__(256) void k_table(CTable* tb, unsigned long long* out) {
    const unsigned long long t0 = clock64();
    consume(*(const Rec*)&tb->r[blockIdx.x & 7], t0, out);
}
// stands for the previous kernel of the chain: some work + (optionally) one wave-wide touch of the next table per XCD-local block index
__global__ __launch_bounds__(256) void k_prev(const float* x, float* y, const int* next, int lines, int touch) {
    float v = x[blockIdx.x * 256 + threadIdx.x];
    int pf = 0;
    if (touch && threadIdx.x < 64) {
        const int line = (int)(blockIdx.x >> 3) * 64 + (int)threadIdx.x;
        if (line < lines) pf = next[line * 16];
    }
    asm volatile("" ::: "memory");
    for (int i = 0; i < 64; ++i) v = v * 1.0001f + 0.5f;
    y[blockIdx.x * 256 + threadIdx.x] = v;
    if (pf == 0x7fc12345) y[0] = 1.f;
}
__global__ __launch_bounds__(256) void k_stream(const float4* x, float4* y, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 v = x[i]; v.x += 1.f; y[i] = v; }
}
int main() {
    float *a, *c, *x, *y; CK(hipMalloc(&a, 4096)); CK(hipMalloc(&c, 1024 * 256 * 4)); CK(hipMemset(a, 0, 4096));
    CK(hipMalloc(&x, 1024 * 256 * 4)); CK(hipMalloc(&y, 1024 * 256 * 4)); CK(hipMemset(x, 0, 1024 * 256 * 4));
    const size_t big = 64ull << 20; float4 *bx, *by; CK(hipMalloc(&bx, big)); CK(hipMalloc(&by, big)); CK(hipMemset(bx, 0, big));
    Table tb{}; for (int q = 0; q < 8; ++q) { tb.r[q].a = a; tb.r[q].c = c; for (int k = 0; k < 60; ++k) tb.r[q].v[k] = k; }
    Table* dtb; CK(hipMalloc(&dtb, sizeof(Table))); CK(hipMemcpy(dtb, &tb, sizeof(Table), hipMemcpyHostToDevice));
    unsigned long long* out; CK(hipMalloc(&out, 1024 * 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    const int lines = (int)((sizeof(Table) + 63) / 64);
    const char* names[3] = {"by-value table in the kernarg segment", "table in device memory, cold", "table in device memory, touched from every XCD by the previous kernel"};
    for (int grid : {256, 512}) for (int mode = 0; mode < 3; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, st, (const float4*)bx, by, big / 16);
        hipLaunchKernelGGL(k_prev, dim3(grid), dim3(256), 0, st, (const float*)x, y, (const int*)dtb, lines, mode == 2 ? 1 : 0);
        if (mode == 0) hipLaunchKernelGGL(k_byvalue, dim3(grid), dim3(256), 0, st, tb, out);
        else hipLaunchKernelGGL(k_table, dim3(grid), dim3(256), 0, st, (CTable*)dtb, out);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        std::vector<unsigned long long> all;
        for (int rep = 0; rep < 30; ++rep) {
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
            std::vector<unsigned long long> h(grid); CK(hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost));
            if (rep >= 5) all.insert(all.end(), h.begin(), h.end());
        }
        std::sort(all.begin(), all.end());
        printf("%4d workgroups, %-75s: record burst median %5llu cycles, p10 %5llu, p90 %5llu\n", grid, names[mode], all[all.size() / 2], all[all.size() / 10], all[all.size() * 9 / 10]);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}

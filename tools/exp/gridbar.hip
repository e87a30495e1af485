// Micro-benchmark: cost of a device-wide barrier between dependent phases INSIDE one persistent kernel,
// versus one kernel launch per phase (graph-replayed).  Decides whether a persistent step program pays.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target) {
    __shared__ int ok_s;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && spins < 2000000) { ++spins; __builtin_amdgcn_s_sleep(1); }
        __threadfence();
        ok_s = spins < 2000000;
    }
    __syncthreads();
    return ok_s != 0;
}

// per-XCD two-level variant: WG b belongs to group b % NG; group counters then a top counter
__device__ __forceinline__ bool grid_barrier2(unsigned* ctr, int ng, unsigned phase, unsigned per_group) {
    __shared__ int ok_s;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const int g = blockIdx.x % ng;
        unsigned* gc = ctr + 64 * (1 + g);
        const unsigned old = __hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (phase + 1) * per_group - 1) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        const unsigned target = (phase + 1) * ng;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && spins < 2000000) { ++spins; __builtin_amdgcn_s_sleep(1); }
        __threadfence();
        ok_s = spins < 2000000;
    }
    __syncthreads();
    return ok_s != 0;
}

__global__ __launch_bounds__(256) void persistent(unsigned* ctr, float* buf, int G, int phases, int mode, int* err) {
    const int b = blockIdx.x;
    for (int p = 0; p < phases; ++p) {
        float* w = buf + (size_t)(p & 1) * G * 256;
        w[(size_t)b * 256 + threadIdx.x] = (float)(p * 7 + b + threadIdx.x);
        bool ok = mode == 0 ? grid_barrier(ctr, (unsigned)(p + 1) * G) : grid_barrier2(ctr, 8, p, G / 8);
        if (!ok) { if (threadIdx.x == 0) atomicAdd(err, 1000000); return; }
        const int o = (b + 37) % G;
        const float v = w[(size_t)o * 256 + threadIdx.x];
        if (v != (float)(p * 7 + o + threadIdx.x)) atomicAdd(err, 1);
    }
}

__global__ __launch_bounds__(256) void phase_kernel(float* buf, int G, int p, int* err) {
    const int b = blockIdx.x;
    float* w = buf + (size_t)(p & 1) * G * 256;
    const float* rd = buf + (size_t)((p + 1) & 1) * G * 256;
    const int o = (b + 37) % G;
    if (p > 0) { const float v = rd[(size_t)o * 256 + threadIdx.x]; if (v != (float)((p - 1) * 7 + o + threadIdx.x)) atomicAdd(err, 1); }
    w[(size_t)b * 256 + threadIdx.x] = (float)(p * 7 + b + threadIdx.x);
}

int main() {
    const int phases = 200;
    unsigned* ctr; float* buf; int* err;
    CK(hipMalloc(&ctr, 64 * 16 * sizeof(unsigned))); CK(hipMalloc(&buf, 2 * 1024 * 256 * sizeof(float))); CK(hipMalloc(&err, sizeof(int)));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode)
        for (int G : {64, 256, 512}) {
            float best = 1e9f; int herr = 0;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemsetAsync(ctr, 0, 64 * 16 * sizeof(unsigned), st)); CK(hipMemsetAsync(err, 0, sizeof(int), st));
                CK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(persistent, dim3(G), dim3(256), 0, st, ctr, buf, G, phases, mode, err);
                CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
                CK(hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost));
            }
            printf("persistent mode=%d G=%d: %.3f us per phase (errors=%d)\n", mode, G, best * 1000.f / phases, herr);
        }
    for (int G : {64, 256, 512}) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipMemset(err, 0, sizeof(int)));
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(256), 0, st, buf, G, p, err);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        int herr; CK(hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost));
        printf("graph of launches G=%d: %.3f us per phase (errors=%d)\n", G, best * 1000.f / phases, herr);
    }
    return 0;
}

// What does a dependent launch cost as a function of its SHAPE, with (almost) no work inside?  nc_fwd_x3q_kernel's waves live 13.6 us
// (tools/exp/nc_timeline.py) but the launch costs 22-24 us in a dependent chain: which resource makes the other 8-10 us?
//   hipcc --offload-arch=gfx950 -O3 -o launch_shape launch_shape.hip && ./launch_shape
// Each line: a hipGraph of 50 dependent launches of one kernel, replayed; microseconds per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
extern __shared__ float dyn[];

struct Arg { float* out; const float* in; int write_floats_per_thread; int read_floats_per_thread; int spin; int pad[64]; };

// VG = number of live VGPR-resident values kept across the body (forces the register footprint)
template <int VG>
__global__ __launch_bounds__(256) void shape_kernel(Arg a) {
    float v[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) v[i] = (float)(threadIdx.x + i);
    const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.f;
    for (int i = 0; i < a.read_floats_per_thread; ++i) s += a.in[gtid + (size_t)i * gridDim.x * blockDim.x];
    // busy wait (s_sleep-free) to emulate a body of `spin` x ~64 cycles
    for (int i = 0; i < a.spin; ++i) {
#pragma unroll
        for (int q = 0; q < VG; ++q) v[q] = v[q] * 1.0001f + s;
    }
    if (dyn && threadIdx.x == 1023) dyn[0] = v[0];
    float r = 0.f;
#pragma unroll
    for (int q = 0; q < VG; ++q) r += v[q];
    for (int i = 0; i < a.write_floats_per_thread; ++i) a.out[gtid + (size_t)i * gridDim.x * blockDim.x] = r + (float)i;
    if (r == 12345.f) a.out[0] = r;
}

template <int VG>
static int run(const char* label, int blocks, int lds, int wr, int rd, int spin, float* out, const float* in) {
    Arg a; a.out = out; a.in = in; a.write_floats_per_thread = wr; a.read_floats_per_thread = rd; a.spin = spin;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(shape_kernel<VG>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(shape_kernel<VG>, dim3(blocks), dim3(256), lds, st, a);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 4; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-64s %7.2f us per launch\n", label, best * 1e3f / 200.f);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(st));
    return 0;
}

int main() {
    float *out, *in;
    const size_t n = (size_t)64 << 20;
    CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&in, n * 4)); CK(hipMemset(in, 0, n * 4));
    // 256 workgroups x 256 threads throughout (65,536 threads): 44 floats per thread = 11.5 MB
    run<8>("empty, 8 values, no LDS", 256, 0, 0, 0, 0, out, in);
    run<8>("empty, 8 values, 77 KB LDS", 256, 77 * 1024, 0, 0, 0, out, in);
    run<160>("empty, 160 values (VGPR-heavy), 77 KB LDS", 256, 77 * 1024, 0, 0, 0, out, in);
    run<8>("writes 11.5 MB (44 floats/thread), no LDS", 256, 0, 44, 0, 0, out, in);
    run<8>("writes 11.5 MB, 77 KB LDS", 256, 77 * 1024, 44, 0, 0, out, in);
    run<8>("writes 1 MB (4 floats/thread)", 256, 0, 4, 0, 0, out, in);
    run<8>("reads 3.3 MB (12 floats/thread)", 256, 0, 0, 12, 0, out, in);
    run<8>("body ~10 us (spin), no memory", 256, 0, 0, 0, 3000, out, in);
    run<8>("body ~10 us (spin) + writes 11.5 MB", 256, 0, 44, 0, 3000, out, in);
    run<8>("body ~10 us (spin) + writes 11.5 MB, 77 KB LDS", 256, 77 * 1024, 44, 0, 3000, out, in);
    run<8>("body ~10 us (spin) + 1024 workgroups, writes 11.5 MB", 1024, 0, 11, 0, 3000, out, in);
    return 0;
}

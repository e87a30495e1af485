// Per-CU global store throughput by store width: nc_fwd_x3q's live-head workgroups spend ~4 us storing 82 KB each with 4-byte stores
// (32 consecutive lanes = one 128-byte line).  Is that the memory system or the per-CU store path?
//   hipcc --offload-arch=gfx950 -O3 -o store_width store_width.hip && ./store_width
// Each workgroup (256 threads) writes BYTES_PER_WG bytes of its own contiguous region; time = kernel duration minus an empty launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int W>   // floats per lane per store
__global__ __launch_bounds__(256) void k(float* out, int floats_per_wg, unsigned long long* clk) {
    float* base = out + (size_t)blockIdx.x * floats_per_wg;
    const float v = (float)threadIdx.x;
    const unsigned long long c0 = clock64();
    for (int o = threadIdx.x * W; o < floats_per_wg; o += 256 * W) {
        if (W == 1) base[o] = v;
        else if (W == 2) *reinterpret_cast<f32x2*>(base + o) = (f32x2){v, v};
        else *reinterpret_cast<f32x4*>(base + o) = (f32x4){v, v, v, v};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long c1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}

template <int W>
static int run(int wgs, int bytes_per_wg, float* out, unsigned long long* clk) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f; unsigned long long h = 0;
    for (int rep = 0; rep < 20; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<W>, dim3(wgs), dim3(256), 0, 0, out, bytes_per_wg / 4, clk);
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    CK(hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost));
    printf("%3d workgroups x %3d KB, %2d-byte stores: kernel %6.2f us; workgroup 0 issued + drained its stores in %6llu cycles = %5.1f bytes per cycle per CU\n",
           wgs, bytes_per_wg / 1024, 4 * W, best * 1e3f, h, (double)bytes_per_wg / (double)h);
    return 0;
}

int main() {
    float* out; unsigned long long* clk;
    CK(hipMalloc(&out, (size_t)256 << 20)); CK(hipMalloc(&clk, 16));
    for (int wgs : {1, 128, 256}) {
        run<1>(wgs, 80 * 1024, out, clk); run<2>(wgs, 80 * 1024, out, clk); run<4>(wgs, 80 * 1024, out, clk);
    }
    run<1>(256, 320 * 1024, out, clk); run<4>(256, 320 * 1024, out, clk);
    return 0;
}

#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void trivial(int* c) { if (threadIdx.x == 0 && blockIdx.x == 0) *c += 1; }
__global__ void trivial256(float* c) { c[blockIdx.x * 256 + threadIdx.x] += 1.f; }
extern "C" int run_chain(int n, int use_graph, int big) {
  int* c; float* f; hipMalloc(&c, 4); hipMalloc(&f, 256*256*4); hipMemset(c, 0, 4); hipMemset(f,0,256*256*4);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&]() { if (big) hipLaunchKernelGGL(trivial256, dim3(256), dim3(256), 0, st, f); else hipLaunchKernelGGL(trivial, dim3(1), dim3(64), 0, st, c); };
  for (int i = 0; i < 100; ++i) launch();
  hipStreamSynchronize(st);
  float ms = 0;
  if (!use_graph) {
    auto t0 = std::chrono::high_resolution_clock::now();
    hipEventRecord(e0, st);
    for (int i = 0; i < n; ++i) launch();
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    auto t1 = std::chrono::high_resolution_clock::now();
    hipEventElapsedTime(&ms, e0, e1);
    printf("eager  big=%d n=%d: %.3f us/kernel (events), %.3f us/kernel (host wall)\n", big, n, ms * 1e3 / n, std::chrono::duration<double, std::micro>(t1 - t0).count() / n);
  } else {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) launch();
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    auto t0 = std::chrono::high_resolution_clock::now();
    hipEventRecord(e0, st);
    for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    auto t1 = std::chrono::high_resolution_clock::now();
    hipEventElapsedTime(&ms, e0, e1);
    printf("graph  big=%d n=%d: %.3f us/kernel (events), %.3f us/kernel (host wall)\n", big, n, ms * 1e3 / (5.0 * n), std::chrono::duration<double, std::micro>(t1 - t0).count() / (5.0 * n));
  }
  return 0;
}
#ifdef MAIN
int main() { int v; hipRuntimeGetVersion(&v); printf("runtime %d\n", v); run_chain(2000,0,0); run_chain(2000,1,0); run_chain(2000,0,1); run_chain(2000,1,1); return 0; }
#endif

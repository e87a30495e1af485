// Micro-benchmark: can the dependent-launch boundary be HIDDEN while every layer keeps its chip-wide launch?
//
// Today a chain of 256 x 256 x 256 layers costs one launch boundary per layer (last wave of k -> first wave of k + 1: 1.5 - 3.8 us in
// situ) plus the front of k + 1 (task record, cold operands).  Early dispatch: consecutive launches alternate between TWO hardware
// queues, so launch k + 1 is dispatched while k runs (its queue's predecessor is k - 1, long finished).  Its workgroups fetch what
// does not depend on k (weights, bias), then wait on the row-block flags k's tiles publish -- tile (tr, tc) of k + 1 needs only the 16
// column tiles of row block tr of k -- and read k's output with sc1 loads (k stores write-through: sc1 stores, vmcnt(0), barrier, one
// sc1 flag store per workgroup; MI355X guide, visibility table, first row).
//
// Liveness: queue order makes k + 1 start only after k - 1 has COMPLETED, so at any time at most one launch (k + 1) is resident and
// waiting while the launch it waits for (k) is either resident or queued in front of free slots: the grids here (256 / 512
// workgroups of 256 threads, 4 KB LDS, < 64 VGPRs) fit the chip several times over, so waiting workgroups can never hold the
// slots k needs.  Every wait is bounded (SPIN_LIMIT polls with s_sleep): a timeout sets the error word and the workgroup proceeds, so
// every launch drains whatever happens.
//
// Forms timed (L layers, results must be bit-identical to the plain chain):
//   0  plain kernels, one stream, hipGraph                      (what the engine does today)
//   1  flag kernels, one stream, hipGraph                       (price of write-through + flags without any overlap)
//   2  flag kernels, two alternating branches of ONE hipGraph   (early dispatch inside a graph)
//   3  flag kernels, two hipGraphs (even / odd layers) on two streams
// each alone and beside a chip-filling streaming kernel on a third stream; 256 and 512 workgroups per layer.
//
//   hipcc --offload-arch=gfx950 -O3 -o early_dispatch early_dispatch.hip && ./early_dispatch [L]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int H = 256, B = 256;
#define SPIN_LIMIT (1 << 19)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x27000);
}
__device__ __forceinline__ f32x4 ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);       // aux 16 = sc1
    return (f32x4){__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}

// plain layer: one launch per layer, stream-ordered (the reference)
__global__ __launch_bounds__(256) void layer_kernel(const float* A, const float* Wl, const float* bl, float* C, float* C2) {
    __shared__ float red[4][4][64];
    const int b = blockIdx.x & 255, task = blockIdx.x >> 8;
    const int r0 = (b >> 4) * 16, c0 = (b & 15) * 16;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 15, kq = lane >> 4;
    f32x4 a[4], bw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        a[u] = *reinterpret_cast<const f32x4*>(A + (size_t)(r0 + i) * H + w * 16 + 64 * u + 4 * kq);
        bw[u] = *reinterpret_cast<const f32x4*>(Wl + (size_t)(c0 + i) * H + w * 16 + 64 * u + 4 * kq);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][s], bw[u][s], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) red[w][q][lane] = acc[q];
    __syncthreads();
    const int ol = threadIdx.x & 63, oreg = threadIdx.x >> 6;
    const int r = r0 + (ol >> 4) * 4 + oreg, c = c0 + (ol & 15);
    const float v = ((red[0][oreg][ol] + red[1][oreg][ol]) + red[2][oreg][ol]) + red[3][oreg][ol];
    (task ? C2 : C)[(size_t)r * H + c] = fmaxf(v + bl[c], 0.f) + 0.01f * v;
}

struct EdArgs {
    const float* A; const float* Wl; const float* bl; float* C; float* C2;
    const unsigned* in_flags;      // [16 row blocks][16] tags of the producing layer (nullptr: nothing to wait for)
    unsigned* out_flags;           // [16][16] (task 0 only publishes)
    const unsigned* epoch;         // device word, bumped once per replay in front of the chain
    unsigned* err;
    unsigned long long* stamp;     // [L][4] wall clock: first entry / last exit per layer (min / max by atomics), optional
    int layer;                     // tag = epoch * 64 + layer
};

__global__ __launch_bounds__(256) void layer_ed_kernel(EdArgs p) {
    __shared__ float red[4][4][64];
    __shared__ int dead_s;
    const int b = blockIdx.x & 255, task = blockIdx.x >> 8;
    const int tr = b >> 4, tc = b & 15, r0 = tr * 16, c0 = tc * 16;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 15, kq = lane >> 4;
    if (p.stamp && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 255)) p.stamp[8 * p.layer + (blockIdx.x ? 4 : 0)] = wall_clock64();
    // what does not depend on the previous layer: in flight before the wait
    f32x4 a[4], bw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) bw[u] = *reinterpret_cast<const f32x4*>(p.Wl + (size_t)(c0 + i) * H + w * 16 + 64 * u + 4 * kq);
    const int ol = threadIdx.x & 63, oreg = threadIdx.x >> 6;
    const int r = r0 + (ol >> 4) * 4 + oreg, c = c0 + (ol & 15);
    const float bias = p.bl[c];
    const unsigned ep = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (threadIdx.x == 0) dead_s = 0;
    if (p.in_flags) {
        if (w == 0) {
            const unsigned target = ep * 64u + (unsigned)(p.layer - 1);
            int spins = 0; bool ok;
            while (true) {
                const unsigned v = lane < 16 ? __hip_atomic_load(p.in_flags + tr * 16 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
                ok = __all((int)(v - target) >= 0);
                if (ok || ++spins > SPIN_LIMIT) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (!ok && lane == 0) { atomicOr(p.err, 1u); dead_s = 1; }
        }
        __syncthreads();
    }
    if (p.stamp && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 255)) p.stamp[8 * p.layer + (blockIdx.x ? 4 : 0) + 1] = wall_clock64();      // past its wait
    const __amdgpu_buffer_rsrc_t ra = rsrc_of(p.A);
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = ld16_sc1(ra, (unsigned)(((r0 + i) * H + w * 16 + 64 * u + 4 * kq) * 4));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][s], bw[u][s], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) red[w][q][lane] = acc[q];
    __syncthreads();
    const float v = ((red[0][oreg][ol] + red[1][oreg][ol]) + red[2][oreg][ol]) + red[3][oreg][ol];
    const float y = fmaxf(v + bias, 0.f) + 0.01f * v;
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), rsrc_of(task ? p.C2 : p.C), (unsigned)((r * H + c) * 4), 0, 16);      // sc1: write-through
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (task == 0) __hip_atomic_store(p.out_flags + tr * 16 + tc, ep * 64u + (unsigned)p.layer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p.stamp && (blockIdx.x == 0 || blockIdx.x == 255)) p.stamp[8 * p.layer + (blockIdx.x ? 4 : 0) + 2] = wall_clock64();
    }
}

__global__ void bump_kernel(unsigned* epoch) { if (threadIdx.x == 0) *epoch = *epoch + 1u; }

__global__ __launch_bounds__(256) void hog_kernel(const float* src, float* dst, size_t n, int iters) {
    float s = 0.f;
    for (int it = 0; it < iters; ++it)
        for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) s += src[k] * 1.0001f;
    dst[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 40;
    if (L < 2 || L > 62 || (L & 1)) { printf("L must be even, 2..62\n"); return 1; }
    std::vector<float> hW((size_t)L * H * H), hb((size_t)L * H), hX((size_t)B * H);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto& v : hW) v = rnd() * 0.108f;
    for (auto& v : hb) v = rnd() * 0.05f;
    for (auto& v : hX) v = rnd();
    float *dW, *db, *dX, *dScr;
    unsigned *flags, *epoch, *err; unsigned long long* stamp;
    CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&db, hb.size() * 4));
    CK(hipMalloc(&dX, (size_t)2 * B * H * 4)); CK(hipMalloc(&dScr, (size_t)2 * B * H * 4));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&flags, 2 * 256 * 4)); CK(hipMemset(flags, 0, 2 * 256 * 4));
    CK(hipMalloc(&epoch, 256)); CK(hipMemset(epoch, 0, 256));
    CK(hipMalloc(&err, 256)); CK(hipMemset(err, 0, 256));
    CK(hipMalloc(&stamp, 64 * 8 * 8)); CK(hipMemset(stamp, 0, 64 * 8 * 8));
    hipStream_t sa, sb, sh, sq[4]; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb)); CK(hipStreamCreate(&sh)); sq[0] = sa; sq[1] = sb; CK(hipStreamCreate(&sq[2])); CK(hipStreamCreate(&sq[3]));
    hipEvent_t ejq[4]; for (int q = 0; q < 4; ++q) CK(hipEventCreate(&ejq[q]));
    hipEvent_t e0, e1, ef, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ef)); CK(hipEventCreate(&ej));
    const size_t hogn = (size_t)64 << 20; float *hsrc, *hdst;
    CK(hipMalloc(&hsrc, hogn * 4)); CK(hipMalloc(&hdst, 1024 * 256 * 4)); CK(hipMemset(hsrc, 0, hogn * 4));

    auto ed_args = [&](int l, bool stamps) {
        EdArgs p; memset(&p, 0, sizeof(p));
        p.A = dX + (size_t)(l & 1) * B * H; p.Wl = dW + (size_t)l * H * H; p.bl = db + (size_t)l * H; p.C = dX + (size_t)((l + 1) & 1) * B * H;
        p.C2 = dScr + (size_t)(l & 1) * B * H; p.in_flags = l > 0 ? flags + ((l - 1) & 1) * 256 : nullptr; p.out_flags = flags + (l & 1) * 256;
        p.epoch = epoch; p.err = err; p.stamp = stamps ? stamp : nullptr; p.layer = l;
        return p;
    };
    std::vector<float> ref((size_t)B * H), out((size_t)B * H);
    for (int G : {256, 512}) {
        hipGraph_t g0, g1, g2, g3a, g3b, g4a, g4b; hipGraphExec_t x0, x1, x2, x3a, x3b, x4a, x4b;
        // form 0
        CK(hipStreamBeginCapture(sa, hipStreamCaptureModeGlobal));
        for (int l = 0; l < L; ++l)
            hipLaunchKernelGGL(layer_kernel, dim3(G), dim3(256), 0, sa, dX + (size_t)(l & 1) * B * H, dW + (size_t)l * H * H, db + (size_t)l * H, dX + (size_t)((l + 1) & 1) * B * H, dScr + (size_t)(l & 1) * B * H);
        CK(hipStreamEndCapture(sa, &g0)); CK(hipGraphInstantiate(&x0, g0, nullptr, nullptr, 0));
        // form 1
        CK(hipStreamBeginCapture(sa, hipStreamCaptureModeGlobal));
        hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(64), 0, sa, epoch);
        for (int l = 0; l < L; ++l) hipLaunchKernelGGL(layer_ed_kernel, dim3(G), dim3(256), 0, sa, ed_args(l, false));
        CK(hipStreamEndCapture(sa, &g1)); CK(hipGraphInstantiate(&x1, g1, nullptr, nullptr, 0));
        // form 2: one graph, two alternating branches
        CK(hipStreamBeginCapture(sa, hipStreamCaptureModeGlobal));
        hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(64), 0, sa, epoch);
        CK(hipEventRecord(ef, sa)); CK(hipStreamWaitEvent(sb, ef, 0));
        for (int l = 0; l < L; ++l) hipLaunchKernelGGL(layer_ed_kernel, dim3(G), dim3(256), 0, (l & 1) ? sb : sa, ed_args(l, false));
        CK(hipEventRecord(ej, sb)); CK(hipStreamWaitEvent(sa, ej, 0));
        CK(hipStreamEndCapture(sa, &g2)); CK(hipGraphInstantiate(&x2, g2, nullptr, nullptr, 0));
        // form 3: two graphs (even / odd layers) on two streams; the epoch bump leads the even graph, the odd stream waits for it by event
        CK(hipStreamBeginCapture(sa, hipStreamCaptureModeGlobal));
        for (int l = 0; l < L; l += 2) hipLaunchKernelGGL(layer_ed_kernel, dim3(G), dim3(256), 0, sa, ed_args(l, false));
        CK(hipStreamEndCapture(sa, &g3a)); CK(hipGraphInstantiate(&x3a, g3a, nullptr, nullptr, 0));
        CK(hipStreamBeginCapture(sb, hipStreamCaptureModeGlobal));
        for (int l = 1; l < L; l += 2) hipLaunchKernelGGL(layer_ed_kernel, dim3(G), dim3(256), 0, sb, ed_args(l, false));
        CK(hipStreamEndCapture(sb, &g3b)); CK(hipGraphInstantiate(&x3b, g3b, nullptr, nullptr, 0));

        CK(hipStreamBeginCapture(sa, hipStreamCaptureModeGlobal));
        for (int l = 0; l < L; l += 2) hipLaunchKernelGGL(layer_ed_kernel, dim3(G), dim3(256), 0, sa, ed_args(l, true));
        CK(hipStreamEndCapture(sa, &g4a)); CK(hipGraphInstantiate(&x4a, g4a, nullptr, nullptr, 0));
        CK(hipStreamBeginCapture(sb, hipStreamCaptureModeGlobal));
        for (int l = 1; l < L; l += 2) hipLaunchKernelGGL(layer_ed_kernel, dim3(G), dim3(256), 0, sb, ed_args(l, true));
        CK(hipStreamEndCapture(sb, &g4b)); CK(hipGraphInstantiate(&x4b, g4b, nullptr, nullptr, 0));
        hipGraph_t gq[2][4]; hipGraphExec_t xq[2][4];          // forms 5 / 6: three / four alternating queues (launch k + 2 / k + 3 dispatched while k runs)
        for (int v = 0; v < 2; ++v) for (int q = 0; q < 3 + v; ++q) {
            CK(hipStreamBeginCapture(sq[q], hipStreamCaptureModeGlobal));
            for (int l = q; l < L; l += 3 + v) hipLaunchKernelGGL(layer_ed_kernel, dim3(G), dim3(256), 0, sq[q], ed_args(l, false));
            CK(hipStreamEndCapture(sq[q], &gq[v][q])); CK(hipGraphInstantiate(&xq[v][q], gq[v][q], nullptr, nullptr, 0));
        }
        for (int hog = 0; hog < 2; ++hog) {
            for (int form = 0; form < 7; ++form) {
                std::vector<float> ts; size_t bad = 0; unsigned herr = 0;
                double first_gap = 0, span_sum = 0; int nspan = 0;
                for (int rep = 0; rep < 14; ++rep) {
                    CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
                    
                    CK(hipDeviceSynchronize());
                    if (false) {           // stamps: min slots start at all-ones, the max slot at zero
                        std::vector<unsigned long long> z(64 * 4, ~0ull); for (int l = 0; l < 64; ++l) z[4 * l + 2] = 0;
                        CK(hipMemcpy(stamp, z.data(), z.size() * 8, hipMemcpyHostToDevice));
                    }
                    if (hog) hipLaunchKernelGGL(hog_kernel, dim3(1024), dim3(256), 0, sh, hsrc, hdst, hogn, 6);
                    CK(hipEventRecord(e0, sa));
                    if (form == 0) CK(hipGraphLaunch(x0, sa));
                    else if (form == 1) CK(hipGraphLaunch(x1, sa));
                    else if (form == 2) CK(hipGraphLaunch(x2, sa));
                    else if (form >= 5) {
                        const int nq = form - 2;
                        hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(64), 0, sa, epoch);
                        CK(hipEventRecord(ef, sa));
                        for (int q = 1; q < nq; ++q) CK(hipStreamWaitEvent(sq[q], ef, 0));
                        for (int q = 0; q < nq; ++q) CK(hipGraphLaunch(xq[form - 5][q], sq[q]));
                        for (int q = 1; q < nq; ++q) { CK(hipEventRecord(ejq[q], sq[q])); CK(hipStreamWaitEvent(sa, ejq[q], 0)); }
                    } else {
                        hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(64), 0, sa, epoch);
                        CK(hipEventRecord(ef, sa)); CK(hipStreamWaitEvent(sb, ef, 0));
                        CK(hipGraphLaunch(form == 3 ? x3a : x4a, sa)); CK(hipGraphLaunch(form == 3 ? x3b : x4b, sb));
                        CK(hipEventRecord(ej, sb)); CK(hipStreamWaitEvent(sa, ej, 0));
                    }
                    CK(hipEventRecord(e1, sa)); CK(hipDeviceSynchronize());
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep >= 2) ts.push_back(ms * 1000.f);
                    CK(hipMemcpy(out.data(), dX + (size_t)(L & 1) * B * H, out.size() * 4, hipMemcpyDeviceToHost));
                    if (form == 0 && rep == 0 && hog == 0) ref = out;
                    for (size_t q = 0; q < out.size(); ++q) bad += memcmp(&out[q], &ref[q], 4) != 0;
                    unsigned e = 0; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost)); herr |= e;
                    if (false) {
                        std::vector<unsigned long long> z(64 * 4); CK(hipMemcpy(z.data(), stamp, z.size() * 8, hipMemcpyDeviceToHost));
                        span_sum += (double)(z[4 * (L - 1) + 2] - z[0]) / 100.0; ++nspan;         // first entry of layer 0 -> last exit of layer L-1
                        double fg = 0; for (int l = 1; l < L; ++l) fg += (double)((long long)(z[4 * l + 1] - z[4 * (l - 1) + 2])) / 100.0;   // first tile of l past its wait, relative to last exit of l-1 (negative = overlap)
                        first_gap += fg / (L - 1);
                    }
                }
                std::sort(ts.begin(), ts.end());
                const float med = ts[ts.size() / 2];
                static const char* names[7] = {"", "", "", "", "", "5 flag kernels, three graphs / streams ", "6 flag kernels, four graphs / streams  "};
                static const char* names5[5] = {"0 plain kernels, one stream            ", "1 flag kernels, one stream             ", "2 flag kernels, two branches, one graph", "3 flag kernels, two graphs, two streams", "4 = 3 with stamps by two workgroups    "};
                printf("G=%3d %s %s: %8.2f us = %6.3f us per layer (min %6.3f)  mismatches %zu  err %u", G, hog ? "[beside a streaming kernel]" : "[alone]                    ", form < 5 ? names5[form] : names[form], med, med / L, ts[0] / L, bad, herr);
                if (false) printf("   in-kernel span %6.3f us per layer, first-tile-past-wait minus predecessor's last exit %+.2f us", span_sum / nspan / L, first_gap / nspan);
                printf("\n");
                if (form == 4) {
                    std::vector<unsigned long long> z(64 * 8); CK(hipMemcpy(z.data(), stamp, z.size() * 8, hipMemcpyDeviceToHost));
                    printf("      last replay, 100 MHz wall clock relative to layer 0's entry, us: layer: wg0 entry / past wait / exit | wg255 entry / past wait / exit\n");
                    for (int l = 0; l < std::min(L, 12); ++l)
                        printf("      %2d: %7.2f %7.2f %7.2f | %7.2f %7.2f %7.2f\n", l, (double)(long long)(z[8 * l] - z[0]) / 100., (double)(long long)(z[8 * l + 1] - z[0]) / 100., (double)(long long)(z[8 * l + 2] - z[0]) / 100.,
                               (double)(long long)(z[8 * l + 4] - z[0]) / 100., (double)(long long)(z[8 * l + 5] - z[0]) / 100., (double)(long long)(z[8 * l + 6] - z[0]) / 100.);
                }
                fflush(stdout);
            }
        }
        CK(hipGraphExecDestroy(x0)); CK(hipGraphExecDestroy(x1)); CK(hipGraphExecDestroy(x2)); CK(hipGraphExecDestroy(x3a)); CK(hipGraphExecDestroy(x3b)); CK(hipGraphExecDestroy(x4a)); CK(hipGraphExecDestroy(x4b));
    }
    return 0;
}

// xchain: a CHAIN of dependent, row-local stages of a step program as ONE persistent launch (gfx950).
//
// Why.  At batch 256 every MLP layer of the update path is a 256 x 256 x 256 product: 64 - 256 tiny tiles, and the step programs are
// chains of 8 - 10 such launches, each costing 6 - 7 us when it has to wait for its predecessor: ~1.5 us of launch
// boundary, a cold task record, operands that arrive from another XCD's L2 through the fabric.  Forward and dX chains are ROW-LOCAL
// -- a block of minibatch rows flows through the layers on its own -- so the chain needs no device-wide synchronisation at all:
//   * group g (the workgroups with blockIdx.x % 8 == g: ONE XCD under the hardware's round-robin dealing of workgroups) owns the 16-row
//     blocks g * rbg ... of the minibatch in EVERY phase, so a group only ever consumes what the same group produced;
//   * inside an XCD a hand-off needs no write-through, no release fence and no acquire (MI355X guide, visibility table: plain stores KEEP
//     the line in the XCD's L2, sc1 loads bypass the reader's L1 and are L2-served):
//       producer: plain stores -> s_waitcnt vmcnt(0) -> workgroup barrier -> ONE flag store (sc0: stays in L2)
//       consumer: one wave polls the flags of its group (sc1 loads of one or two 128-byte lines) -> workgroup barrier -> sc1 loads;
//   * a phase = what used to be one launch: the same GemmTask records, the same tile body (gemm16_tile.h, COH = true), the same epilogues.
// tools/exp/xcd_chain.hip is the prototype: 2.1 us per 256-wide layer inside the launch against 3.3 us for the graph of launches (both
// alone on the chip, results bit-identical), 2.8 / 3.2 us with write-through stores on same-XCD / cross-XCD groups.
//
// What is NOT assumed.  Which XCD a workgroup lands on is read from HW_REG_XCC_ID: every flag carries its writer's XCC id, and a reader
// that meets a foreign id sets bit 1 of the error word (the hand-off protocol is only valid inside one L2); a wait that does not
// complete within ~2^18 polls sets bit 0, and the workgroup then stops waiting, so that the launch always drains.  The host reads the
// word at its next synchronisation point and raises (rlrep_chain_status).  All 8 * mpg workgroups must be co-resident (256 or 512
// workgroups of 256 threads, <= 128 VGPRs... checked against the device's CU count by the launcher).
#include "common.h"
#include "kparams.h"
#include "gemm16_tile.h"
#include "heads_vae_tile.h"

#define XC_SPIN_LIMIT (1 << 18)

#ifdef RL_TIMING_XC
// Instrumented build (tools/exp/xc_timeline.py; EXTRA_FLAGS="-DRL_TIMING -DRL_TIMING_XC"): thread 0 of every workgroup records, per phase, the
// 100 MHz wall clock at phase entry / wait passed / tiles done / flag published, and the shader clock inside its first tile (gemm16_tile's
// TIMB hooks: record in registers, MFMAs issued, reduction barrier passed).  Slot = ((launch * 32 + phase) * 512 + block) * 12.
__device__ unsigned long long* g_xct = nullptr;
__device__ unsigned g_xct_cap = 0, g_xct_launch = 0;
extern "C" int rl_xc_timing_buffer(void* buf, unsigned cap) {
    unsigned zero = 0;
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_xct), &buf, sizeof(buf));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_xct_launch), &zero, sizeof(zero));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_xct_cap), &cap, sizeof(cap));
    return (int)e;
}
extern "C" unsigned rl_xc_timing_count() { unsigned n = 0; (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_xct_launch), sizeof(n)); return n; }
#define XCT(k) do { if (threadIdx.x == 0) xs[k] = wall_clock64(); } while (0)
#else
#define XCT(k) do {} while (0)
#endif

__device__ __forceinline__ bool xc_reached(unsigned flag, unsigned target) {      // 28-bit counters, wrap-safe
    return (((flag >> 4) - target) & 0x0fffffffu) < 0x08000000u;
}

// the launch's tables (phases, tasks, heads_vae records) are written by the host only: read them through the constant address space
typedef RL_CONST_AS GemmTask XcTask;
typedef RL_CONST_AS XcPhase XcPh;
typedef RL_CONST_AS HeadsVae XcHv;

template <int LA, int LB, bool VA, bool VB, bool PRE>
__device__ __forceinline__ void xc_gemm_tile(const XcTask& t, int tr, int tc, float* smem, float (&bsum)[4][16], const float* const* dyn, unsigned long long* timc) {
    auto& red = *reinterpret_cast<float (*)[4][1][4][64]>(smem);
#ifdef RL_TIMING
    gemm16_tile<LA, LB, 1, VA, VB, PRE, true, XcTask>(t, tr, tc, red, bsum, dyn, timc);
#else
    (void)timc;
    gemm16_tile<LA, LB, 1, VA, VB, PRE, true, XcTask>(t, tr, tc, red, bsum, dyn);
#endif
}
#define XC_RUN(...) xc_gemm_tile<__VA_ARGS__>(t, tr, tc, smem, bsum, L.dyn, timp)

// tile index of a GEMM phase -> (task, row tile, column tile); false: the tile lies beyond the task's rows
__device__ __forceinline__ bool xc_decode(const XcPh& ph, const XcTask* tasks, int g, int tile, int& ti, int& tr, int& tc) {
    ti = ph.task0;
    for (int q = 1; q < ph.ntasks; ++q) if (tile >= ph.tb[q]) ti = ph.task0 + q;
    const int local = tile - ph.tb[ti - ph.task0];
    const int tiles_c = ph.tcs[ti - ph.task0];
    const int rb = local / tiles_c; tc = local - rb * tiles_c;
    tr = g * ph.rbg + rb;
    return tr * 16 < ph.R;
}

__global__ __launch_bounds__(256) void xchain_kernel(XcLaunch L) {
    __shared__ float smem[4 * 4 * 256];      // gemm16: red[4][1][4][64]; heads_vae: red[4][4][256]
    __shared__ float bsum[4][16];
    __shared__ float sh4[4];
    __shared__ int dead_s;
    if (!L.low_prio) __builtin_amdgcn_s_setprio(3);
    const int g = blockIdx.x & (XC_GROUPS - 1), m = blockIdx.x >> 3;
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u;      // HW_REG_XCC_ID[3:0]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned* const gf = L.flags + g * XC_FLAG_STRIDE;
    // the counter this launch starts from: my own flag as my previous launch left it (every member of every group advances by nph per launch)
    const unsigned c0 = __builtin_amdgcn_readfirstlane(__hip_atomic_load(gf + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 4;
    if (threadIdx.x == 0) dead_s = 0;
    __syncthreads();
#ifdef RL_TIMING_XC
    unsigned xlaunch = 0;
    if (threadIdx.x == 0) xlaunch = *(volatile unsigned*)&g_xct_launch;
#endif
    for (int p = 0; p < L.nph; ++p) {
        const XcPh& ph = ((const XcPh*)L.ph)[p];
#ifdef RL_TIMING_XC
        unsigned long long xs[4] = {0, 0, 0, 0}, tcs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        bool first_tile = true;
#endif
        XCT(0);
        const XcTask* const tasks = (const XcTask*)L.tasks;
        if (p > 0) {
            // every member of my group has finished phase p - 1
            if (w == 0 && !dead_s) {
                const unsigned target = (c0 + (unsigned)p) & 0x0fffffffu;
                int spins = 0; bool ok; unsigned v;
                while (true) {
                    v = lane < L.mpg ? __hip_atomic_load(gf + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    ok = __all(lane >= L.mpg || xc_reached(v, target));
                    if (ok || ++spins > XC_SPIN_LIMIT) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!ok) { if (lane == 0) { atomicOr(L.err, 1u); dead_s = 1; } }
                else if (__any(lane < L.mpg && (v & 15u) != xcc)) { if (lane == 0) atomicOr(L.err, 2u); }
            }
            __syncthreads();
        }
        XCT(1);
        bool first = true;
        for (int tile = m; tile < ph.tiles; tile += L.mpg) {
            if (!first) __syncthreads();          // the reduction buffers of the previous tile are still being read
            if (ph.kind == XC_GEMM) {
                int ti, tr, tc;
                if (!xc_decode(ph, tasks, g, tile, ti, tr, tc)) { first = false; continue; }
                const XcTask& t = tasks[ti];
#ifdef RL_TIMING_XC
                unsigned long long* const timp = first_tile ? tcs : nullptr;
                if (first_tile && threadIdx.x == 0) tcs[0] = clock64();
#else
                unsigned long long* const timp = nullptr;
#endif
                if (ph.lb == LD_ROW) {
                    if (ph.vecA && ph.vecB) XC_RUN(LD_ROW, LD_ROW, true, true, false);
                    else XC_RUN(LD_ROW, LD_ROW, false, false, false);
                } else if (ph.pre) XC_RUN(LD_ROW, LD_COL, false, false, true);
                else if (ph.vecA) XC_RUN(LD_ROW, LD_COL, true, false, false);
                else XC_RUN(LD_ROW, LD_COL, false, false, false);
#ifdef RL_TIMING_XC
                if (first_tile && threadIdx.x == 0) tcs[4] = clock64();
                first_tile = false;
#endif
            } else if (ph.kind == XC_HEADS_VAE) {
                const XcHv& hv = ((const XcHv*)L.hv)[ph.aux];
                const int rb = tile / hv.tiles_c, tc = tile - rb * hv.tiles_c;
                const int tr = g * L.rbg + rb;
                if (tr * 16 < hv.B) {
                    auto& red = *reinterpret_cast<float (*)[4][4][256]>(smem);
                    heads_vae_tile<true, XcHv>(hv, tr, tc, L.dyn[ph.dyn], red, sh4);
                }
            }
            first = false;
        }
        XCT(2);
        // publish: every store of this workgroup has reached L2, then ONE flag (it stays in this XCD's L2: workgroup-scope store = sc0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(gf + m, (((c0 + (unsigned)p + 1u) & 0x0fffffffu) << 4) | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        XCT(3);
#ifdef RL_TIMING_XC
        if (threadIdx.x == 0 && g_xct) {
            const size_t slot = (((size_t)xlaunch * 32 + p) * 512 + blockIdx.x) * 12;
            if (slot + 12 <= g_xct_cap) {
                for (int q = 0; q < 4; ++q) g_xct[slot + q] = xs[q];
                for (int q = 0; q < 5; ++q) g_xct[slot + 4 + q] = tcs[q];
                g_xct[slot + 9] = ((unsigned long long)ph.kind << 32) | (unsigned)ph.tiles;
                g_xct[slot + 10] = tcs[5]; g_xct[slot + 11] = tcs[6];
            }
        }
#endif
    }
#ifdef RL_TIMING_XC
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&g_xct_launch, 1u);
#endif
}

extern "C" int rl_launch_xchain(const XcLaunch* L, hipStream_t st) {
    if (L->nph <= 0) return 0;
    if (L->mpg < 1 || L->mpg > XC_FLAG_STRIDE) return -1;
    hipLaunchKernelGGL(xchain_kernel, dim3(XC_GROUPS * L->mpg), dim3(256), 0, st, *L);
    return (int)hipGetLastError();
}

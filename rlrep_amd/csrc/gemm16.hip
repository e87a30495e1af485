// gemm16: the small-matrix fp32 tile engine of the update path (gfx950).
//
// Every MLP layer of the hot path at batch 256 is a GEMM whose output is 256 x {1..512}: too small to
// fill 256 CUs with large tiles, and bounded by dependent-launch latency rather than FLOPs.  The engine
// therefore gives every workgroup ONE 16 x (16*NF) output tile and splits the inner dimension over its four
// waves (one wave per SIMD), each wave streaming its operand fragments straight from L2 into VGPRs
// (no LDS round trip, no barrier in the main loop) and issuing v_mfma_f32_16x16x4_f32 -- exact fp32,
// k-ordered fma chain, same peak as the VALU but one VGPR per operand.  The four partial tiles are
// summed through LDS in fixed wave order (bitwise reproducible), after which each thread owns NF output
// elements and applies the fused epilogue (bias+activation, activation derivative, reparameterisation
// backward, weight/bias gradient (+Adam), mse loss, tanh-Gaussian policy forward/backward).
// NF (column fragments per workgroup) is chosen per launch so that a launch stays within about two
// workgroups per CU: wide layers (N = 512) and the 8-task weight-gradient launch use NF = 2 or 4.
//
// One launch executes a TABLE of independent GEMMs: independent layers of one stage of a step program
// (e.g. encoder.l1 and f.l1) share a launch, so the number of dependent launches is the depth of the
// network graph, not its size.  The table travels BY VALUE in the kernel-argument segment.
//
// Operand fragment maps (MI355X guide, section 3): A: lane l holds A[i=l&15][k=l>>4];
// B: lane l holds B[k=l>>4][j=l&15]; C/D: col=l&15, row=4*(l>>4)+reg.  The inner index may be permuted
// freely as long as A and B agree, so each lane takes FOUR CONSECUTIVE inner indices (one 16-byte load in
// the row-contiguous case) and feeds them to four successive MFMAs.
#include "common.h"
#include "kparams.h"

#ifdef RL_TIMING
// Instrumented build (tools/exp/gemm_timeline.py): thread 0 of every 8th workgroup records the 100 MHz wall clock at
// entry and exit and the shader clock at entry, once its task record is in registers, after its MFMAs, after the
// reduction barrier and at exit.  Slot = launch sequence number (bumped by workgroup 0 at exit) * 2048 + blockIdx.x.
struct RlTimRec { unsigned long long w0, w4, c[5]; int grid, bid; unsigned tag, valid; };
__device__ RlTimRec* g_tim = nullptr;
__device__ unsigned g_tim_launch = 0, g_tim_cap = 0;
extern "C" int rl_timing_buffer(void* buf, unsigned cap) {
    unsigned zero = 0;
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_tim), &buf, sizeof(buf));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_tim_launch), &zero, sizeof(zero));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_tim_cap), &cap, sizeof(cap));
    return (int)e;
}
extern "C" unsigned rl_timing_count() { unsigned n = 0; (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_tim_launch), sizeof(n)); return n; }
#define TIM_ON (threadIdx.x == 0 && (blockIdx.x & 7) == 0)
#define TIM(k) do { if (TIM_ON) tim_c[k] = clock64(); } while (0)
#define TIM_FIN() do { if (threadIdx.x == 0 && g_tim) { if (TIM_ON) { tim_c[4] = clock64(); const unsigned long long w4 = wall_clock64(); \
    const unsigned slot = tim_lid * 2048u + blockIdx.x; \
    if (slot < g_tim_cap) { RlTimRec r; r.w0 = tim_w0; r.w4 = w4; for (int q = 0; q < 5; ++q) r.c[q] = tim_c[q]; r.grid = gridDim.x; r.bid = blockIdx.x; \
    r.tag = (unsigned)(((uintptr_t)pC) >> 4) ^ ((unsigned)epi << 28); r.valid = 1; g_tim[slot] = r; } } \
    if (blockIdx.x == 0) atomicAdd(&g_tim_launch, 1u); } } while (0)
#else
#define TIM(k) do {} while (0)
#define TIM_FIN() do {} while (0)
#endif

// BRANCH-FREE operand fetch.  Out-of-range rows / inner indices are handled by CLAMPING the address into the
// matrix and zeroing the value with a select: no exec-masked branch around any load.  (With `if (in_range) load`
// hipcc wraps every load in s_cbranch_execz + s_waitcnt vmcnt(0): 147 branches and 21 full drains in a kernel
// with 16 MFMAs, i.e. the eight 16-byte loads a wave needs were serialised instead of overlapped.)
template <int LOADER, bool VEC>
__device__ __forceinline__ void load_raw(const float* __restrict__ P, int ld, int base, int lim,
                                         int i, int k0, int K, float (&v)[4]) {
    const int idx = min(base + i, lim - 1);
    if (LOADER == LD_ROW) {
        if (VEC) {                       // K % 4 == 0, 16-byte aligned rows: the 4 indices are valid together
            const f32x4 x = *reinterpret_cast<const f32x4*>(P + (size_t)idx * ld + min(k0, K - 4));
            v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
        } else {
            const float* p = P + (size_t)idx * ld;
#pragma unroll
            for (int s = 0; s < 4; ++s) v[s] = p[min(k0 + s, K - 1)];
        }
    } else {  // LD_COL
        const float* p = P + idx;
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = p[(size_t)min(k0 + s, K - 1) * ld];
    }
}
// zero what the clamped load fetched from outside the matrix
__device__ __forceinline__ void mask_frag(int base, int lim, int i, int k0, int K, float (&v)[4]) {
    const bool rok = (base + i) < lim;
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = (rok && k0 + s < K) ? v[s] : 0.f;
}

// NU 16-wide inner chunks of wave w (NU = 4 covers K <= 256 in one go): EVERY load of the group is issued before
// anything consumes one (hipcc otherwise sinks each load next to its MFMA and drains vmcnt(0) in between: eight
// serialised L2 round trips instead of one)
template <int LA, int LB, int NF, bool VA, bool VB, int NU>
__device__ __forceinline__ void mac_group(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                          int r0, int R, int c0, int Cn, int i, int k0, int K,
                                          f32x4 (&acc)[NF], float& asum, bool want_bias) {
    float a[NU][4], b[NU][NF][4];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        load_raw<LA, VA>(A, lda, r0, R, i, k0 + 64 * u, K, a[u]);
#pragma unroll
        for (int f = 0; f < NF; ++f) load_raw<LB, VB>(B, ldb, c0 + 16 * f, Cn, i, k0 + 64 * u, K, b[u][f]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        mask_frag(r0, R, i, k0 + 64 * u, K, a[u]);
#pragma unroll
        for (int f = 0; f < NF; ++f) mask_frag(c0 + 16 * f, Cn, i, k0 + 64 * u, K, b[u][f]);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][s], b[u][f][s], acc[f], 0, 0, 0);
        if (want_bias) asum += (a[u][0] + a[u][1]) + (a[u][2] + a[u][3]);
    }
}

// ---- fused short product (FLAG_PRE) ---------------------------------------------------------------------------------------------
// By MFMA with the operand roles swapped: the row operand is a Wt fragment (16 values of k for one inner index j: consecutive addresses
// across the 16 lanes), the column operand the X fragment (16 batch rows).  The result tile D has col = lane & 15 = batch row and
// row = 4 (lane >> 4) + reg = k offset 4 kq + reg inside the wave's 16-wide chunk -- exactly the "four consecutive inner indices per
// lane" layout in which the main loop wants its A operand: no lane movement, no LDS.  (The same construction for the FIRST layers of the
// MLPs, whose weights are stored [k][j] with 92-byte rows, made every fragment load touch 16-23 cache lines and lost: DESIGN.md 5.0.)
struct PreSrc { const float* X; int ldx; const float* Wt; int ldw; int K1; const float* M; int ldm; float* out; int ldo; };

template <int LB, int NF, bool VB, int NU, int NJ, bool FWD>
__device__ __forceinline__ void mac_group_pre(const PreSrc& ps, const float* __restrict__ B, int ldb, int r0, int R, int c0, int Cn,
                                              int i, int kq, int kb, int K, bool store, f32x4 (&acc)[NF]) {
    float xf[NJ][4], wf[NU][NJ][4], mk[NU][4], b[NU][NF][4];
    const int K1 = ps.K1;
    const float* xrow = ps.X + (size_t)min(r0 + i, R - 1) * ps.ldx;
    const float* mrow = ps.M + (size_t)min(r0 + i, R - 1) * ps.ldm;
#pragma unroll
    for (int jc = 0; jc < NJ; ++jc)
#pragma unroll
        for (int m = 0; m < 4; ++m) xf[jc][m] = xrow[min(16 * jc + 4 * kq + m, K1 - 1)];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int kcol = min(kb + 64 * u + i, K - 1);
#pragma unroll
        for (int jc = 0; jc < NJ; ++jc)
#pragma unroll
            for (int m = 0; m < 4; ++m) wf[u][jc][m] = ps.Wt[(size_t)min(16 * jc + 4 * kq + m, K1 - 1) * ps.ldw + kcol];
#pragma unroll
        for (int m = 0; m < 4; ++m) mk[u][m] = mrow[min(kb + 64 * u + 4 * kq + m, K - 1)];
#pragma unroll
        for (int f = 0; f < NF; ++f) load_raw<LB, VB>(B, ldb, c0 + 16 * f, Cn, i, kb + 4 * kq + 64 * u, K, b[u][f]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // zero what the clamped loads fetched beyond K1 (inner index); rows beyond R / k beyond K are masked when A is formed
#pragma unroll
    for (int jc = 0; jc < NJ; ++jc)
#pragma unroll
        for (int m = 0; m < 4; ++m) xf[jc][m] = (16 * jc + 4 * kq + m) < K1 ? xf[jc][m] : 0.f;
    f32x4 D[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) D[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jc = 0; jc < NJ; ++jc) {
        if (16 * jc >= K1) break;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int u = 0; u < NU; ++u) D[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][jc][m], xf[jc][m], D[u], 0, 0, 0);
    }
    const bool rok = (r0 + i) < R;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        float a[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            // dX form: mk = saved ReLU output (mask); forward form: mk = the layer's bias
            const float g = FWD ? fmaxf(D[u][m] + mk[u][m], 0.f) : (mk[u][m] > 0.f ? D[u][m] : 0.f);
            a[m] = (rok && (kb + 64 * u + 4 * kq + m) < K) ? g : 0.f;
        }
        if (store && rok) {
            float* op = ps.out + (size_t)(r0 + i) * ps.ldo + kb + 64 * u + 4 * kq;
#pragma unroll
            for (int m = 0; m < 4; ++m) if (kb + 64 * u + 4 * kq + m < K) op[m] = a[m];
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) mask_frag(c0 + 16 * f, Cn, i, kb + 4 * kq + 64 * u, K, b[u][f]);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[u][f][m], acc[f], 0, 0, 0);
    }
}

template <int LA, int LB, int NF, bool VA, bool VB, bool PRE = false>
__global__ __launch_bounds__(256) void gemm16_kernel(GemmBatch gb) {
    __shared__ float red[4][NF][4][64];
    __shared__ float bsum[4][16];

#ifdef RL_TIMING
    unsigned long long tim_c[5] = {0, 0, 0, 0, 0}, tim_w0 = 0; unsigned tim_lid = 0;
    if (threadIdx.x == 0) { tim_lid = *(volatile unsigned*)&g_tim_launch; if (TIM_ON) tim_w0 = wall_clock64(); }
#endif
    TIM(0);
    // latency-critical launch: win the issue arbitration against the waves of a noise-critic launch that may be running on the
    // other stream of the deferred pipeline (405.8 vs 429.7 us per train(); alone on the chip it changes nothing)
    if (!gb.low_prio) __builtin_amdgcn_s_setprio(3);
    const int bid = blockIdx.x;
    if (gb.nfin > 0 && bid == (int)gridDim.x - 1) {      // trailing workgroup: metric finalisation / temperature update
        if (threadIdx.x < 64) finalize_tasks(gb.fin, gb.nfin, threadIdx.x);
        return;
    }
    int ti = 0;
#pragma unroll
    for (int q = 1; q < GEMM_MAX_TASKS; ++q) if (q < gb.ntasks && bid >= gb.t[q].tile_base) ti = q;
    const GemmTask& t = gb.t[ti];
    // the hot block of the task record, fetched as one burst of scalar loads
    const float* const pA = t.A; const float* const pB = t.B; float* const pC = t.C;
    const float* const pbias = t.bias; const float* const paux = t.aux;
    const float* const pr1u = t.r1u; const float* const pr1v = t.r1v;
    const int lda = t.lda, ldb = t.ldb, ldc = t.ldc, ldaux = t.ldaux;
    const int R = t.R, Cn = t.Cn, K = t.K, tiles_c = t.tiles_c, tile_base = t.tile_base;
    const int epi = t.epi, act = t.act, flags = t.flags, n0 = t.n0;
    const float scale = t.scale;
    float* const pout2 = t.out2; const int ldout2 = t.ldout2;

#ifdef RL_TIMING
    asm volatile("" :: "s"(R), "s"(epi));
#endif
    TIM(1);
    const int local = bid - tile_base;
    const int tr = local / tiles_c, tc = local - tr * tiles_c;
    const int r0 = tr * 16, c0 = tc * 16 * NF;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;

    f32x4 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float asum = 0.f;
    const bool want_bias = (epi == EPI_DW) && (flags & FLAG_BIASGRAD) && (tc == 0);

    // This thread's output elements are known up front, so the epilogue's operands (bias, saved activation,
    // accumulate-into value, ...) are fetched NOW and their latency overlaps the operand stream.  The epilogue
    // kind only selects up to four SLOT descriptors (base, row stride, column stride, offset, column window) in
    // scalar code; the loads themselves are generic and branch-free (a lane outside its window reads the slot's
    // base address and the value is discarded), so nothing waits on them before the epilogue.
    const float* sp[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int srs[5] = {0, 0, 0, 0, 0}, scs[5] = {1, 1, 1, 1, 1}, sof[5] = {0, 0, 0, 0, 0}, slo[5] = {0, 0, 0, 0, 0}, shi[5] = {Cn, Cn, Cn, Cn, Cn};
    switch (epi) {
    case EPI_FWD: sp[0] = pbias; break;
    case EPI_DX:
        if (act != ACT_NONE) { sp[0] = paux; srs[0] = ldaux; }
        if (flags & FLAG_ACCUM) { sp[1] = pC; srs[1] = ldc; }
        if (pr1u) { sp[2] = pr1u; srs[2] = 1; scs[2] = 0; sp[3] = pr1v; }
        break;
    case EPI_FWD_MSE:
        sp[0] = pbias;
        sp[1] = t.x0; srs[1] = t.ldx0; shi[1] = n0;
        sp[2] = t.x1; srs[2] = 1; scs[2] = 0; slo[2] = n0;
        break;
    case EPI_FWD_POLICY:
        sp[0] = pbias;
        sp[1] = t.x2; srs[1] = n0; shi[1] = n0;
        break;
    case EPI_DX_POLICYBWD:
        sp[0] = t.x0; srs[0] = 2 * n0; sof[0] = n0;
        sp[1] = t.x2; srs[1] = n0;
        sp[2] = t.x1; srs[2] = t.ldx1;
        break;
    case EPI_DX_REPARAM:
        sp[0] = t.aux3; srs[0] = t.ldaux3;
        sp[1] = pC; srs[1] = ldc;
        sp[2] = pC; srs[2] = ldc; sof[2] = t.F;
        break;
    default:   // EPI_DW
        if (flags & FLAG_ACCUM) { sp[1] = pC; srs[1] = ldc; }
        if (t.ad_p) {      // optimizer fused in: the tile of the parameter, its Adam moments (and its Polyak target)
            sp[0] = t.ad_p; sp[2] = t.ad_m; sp[3] = t.ad_v; sp[4] = t.ad_t;
            srs[0] = srs[2] = srs[3] = srs[4] = ldc;
        }
    }
    const int ol = threadIdx.x & 63, oreg = threadIdx.x >> 6;
    const int r = r0 + (ol >> 4) * 4 + oreg;
    // The loads are volatile inline asm: as plain C++ loads hipcc sinks them below the reduction barrier, next to
    // their first use (a serialised L2 round trip in the epilogue).  The compiler does not count asm loads in its
    // vmcnt bookkeeping; they are OLDER than every operand load of the main loop and vmcnt retires in order, so the
    // compiler's own counted waits stay correct, and the values are claimed by the explicit wait after the loop.
    float ev[5][NF];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
#pragma unroll
        for (int f = 0; f < NF; ++f) ev[q][f] = 0.f;
        if (sp[q]) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int c = c0 + 16 * f + (ol & 15);
                const bool ok = (r < R) && (c >= slo[q]) && (c < shi[q]);
                const float* p = sp[q] + (ok ? (size_t)r * srs[q] + (size_t)(c * scs[q] + sof[q]) : (size_t)0);
                asm volatile("global_load_dword %0, %1, off" : "+v"(ev[q][f]) : "v"(p));
            }
        }
    }

    // fused optimizer: Adam scalars of the group, and (column-tile 0 only) the bias element this thread will update
    AdamScal adsc;
    const bool fuse_opt = (epi == EPI_DW) && t.ad_p;
    if (fuse_opt) adsc = t.ad_grp->sc;
    float bpv = 0.f, bmv = 0.f, bvv = 0.f, btv = 0.f;
    const bool bias_opt = want_bias && fuse_opt && t.ad_pb && threadIdx.x < 16 && r0 + (int)threadIdx.x < R;
    if (bias_opt) {
        const int o = r0 + threadIdx.x;
        asm volatile("global_load_dword %0, %1, off" : "+v"(bpv) : "v"(t.ad_pb + o));
        asm volatile("global_load_dword %0, %1, off" : "+v"(bmv) : "v"(t.ad_mb + o));
        asm volatile("global_load_dword %0, %1, off" : "+v"(bvv) : "v"(t.ad_vb + o));
        if (t.ad_tb) asm volatile("global_load_dword %0, %1, off" : "+v"(btv) : "v"(t.ad_tb + o));
    }

    // wave w owns the 16-wide inner chunks w, w+4, w+8, ...
    if constexpr (PRE) {
        PreSrc ps; ps.X = t.x0; ps.ldx = t.ldx0; ps.Wt = t.x1; ps.ldw = t.ldx1; ps.K1 = n0; ps.M = t.x2; ps.ldm = t.ldaux2; ps.out = t.y0; ps.ldo = t.ldout2;
        const bool store = (tc == 0) && ps.out;
        // forward form (LB = LD_ROW): K1 <= 48, bias + ReLU; dX form (LB = LD_COL): K1 <= 32, ReLU mask
        constexpr int NJ = (LB == LD_ROW) ? 3 : 2;
        constexpr bool FW = (LB == LD_ROW);
        for (int kb = w * 16; kb < K; kb += 256) {
            const int nu = (K - kb + 63) >> 6;
            if (nu >= 4) mac_group_pre<LB, NF, VB, 4, NJ, FW>(ps, pB, ldb, r0, R, c0, Cn, i, kq, kb, K, store, acc);
            else if (nu == 1) mac_group_pre<LB, NF, VB, 1, NJ, FW>(ps, pB, ldb, r0, R, c0, Cn, i, kq, kb, K, store, acc);
            else if (nu == 2) mac_group_pre<LB, NF, VB, 2, NJ, FW>(ps, pB, ldb, r0, R, c0, Cn, i, kq, kb, K, store, acc);
            else mac_group_pre<LB, NF, VB, 3, NJ, FW>(ps, pB, ldb, r0, R, c0, Cn, i, kq, kb, K, store, acc);
        }
    } else
    for (int kb = w * 16; kb < K; kb += 256) {
        const int k0 = kb + 4 * kq;
        const int nu = (K - kb + 63) >> 6;           // chunks of this group that touch the matrix (uniform per wave)
        if (nu >= 4) mac_group<LA, LB, NF, VA, VB, 4>(pA, lda, pB, ldb, r0, R, c0, Cn, i, k0, K, acc, asum, want_bias);
        else if (nu == 1) mac_group<LA, LB, NF, VA, VB, 1>(pA, lda, pB, ldb, r0, R, c0, Cn, i, k0, K, acc, asum, want_bias);
        else if (nu == 2) mac_group<LA, LB, NF, VA, VB, 2>(pA, lda, pB, ldb, r0, R, c0, Cn, i, k0, K, acc, asum, want_bias);
        else mac_group<LA, LB, NF, VA, VB, 3>(pA, lda, pB, ldb, r0, R, c0, Cn, i, k0, K, acc, asum, want_bias);
    }

    TIM(2);
    // claim the epilogue operands (long since arrived: the operand stream behind them has been consumed) and
    // discard what out-of-window lanes fetched
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int f = 0; f < NF; ++f) asm volatile("" : "+v"(ev[q][f]));
    asm volatile("" : "+v"(bpv), "+v"(bmv), "+v"(bvv), "+v"(btv));
    float e0[NF], cold[NF], cold2[NF], cold3[NF], cold4[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int c = c0 + 16 * f + (ol & 15);
        float m[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) m[q] = ((r < R) && (c >= slo[q]) && (c < shi[q])) ? ev[q][f] : 0.f;
        e0[f] = m[0];
        cold[f] = (epi == EPI_FWD_MSE) ? m[1] + m[2] : m[1];
        cold2[f] = (epi == EPI_DX) ? m[2] * m[3] : m[2];
        cold3[f] = m[3]; cold4[f] = m[4];
    }

    // cross-wave reduction in fixed order
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[w][f][q][lane] = acc[f][q];
    if (want_bias) {
        asum += __shfl_xor(asum, 16, 64);
        asum += __shfl_xor(asum, 32, 64);
        if (lane < 16) bsum[w][lane] = asum;
    }
    __syncthreads();
    TIM(3);

    if (want_bias && threadIdx.x < 16 && r0 + (int)threadIdx.x < R) {
        const int q = threadIdx.x;
        const float gbv = ((bsum[0][q] + bsum[1][q]) + bsum[2][q]) + bsum[3][q];
        pout2[r0 + q] = gbv;
        if (bias_opt) {
            adam_elem(adsc, gbv, &bpv, &bmv, &bvv, t.ad_tb ? &btv : nullptr);
            t.ad_pb[r0 + q] = bpv; t.ad_mb[r0 + q] = bmv; t.ad_vb[r0 + q] = bvv;
            if (t.ad_tb) t.ad_tb[r0 + q] = btv;
        }
    }

    if (epi == EPI_FWD_MSE) {
        // decoder heads of the vlsac ELBO (vlsac_agent.py:137-140): the gradient of 0.5*mse replaces the prediction
        float es = 0.f, er = 0.f;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int c = c0 + 16 * f + (ol & 15);
            if (r >= R || c >= Cn) continue;
            const float v = (((red[0][f][oreg][ol] + red[1][f][oreg][ol]) + red[2][f][oreg][ol]) + red[3][f][oreg][ol]) * scale;
            const float d = (v + e0[f]) - cold[f];
            float* cp = pC + (size_t)r * ldc + c;
            if (c < n0) { es += d * d; *cp = d * t.s0; } else { er += d * d; *cp = d * t.s1; }
        }
        es = wave_sum(es); er = wave_sum(er);
        __syncthreads();
        if (lane == 0) { red[0][0][0][w] = es; red[0][0][1][w] = er; }
        __syncthreads();
        if (threadIdx.x == 0) {
            t.y0[2 * local] = ((red[0][0][0][0] + red[0][0][0][1]) + red[0][0][0][2]) + red[0][0][0][3];
            t.y0[2 * local + 1] = ((red[0][0][1][0] + red[0][0][1][1]) + red[0][0][1][2]) + red[0][0][1][3];
        }
        TIM_FIN(); return;
    }
    if (epi == EPI_FWD_POLICY) {
        // (NF == 1 launches only) the whole [mu | rho] row sits in this one 16-column tile: rho_j is A lanes to the right
        const int A = n0;
        const int c = c0 + (ol & 15);
        const bool inb = (r < R) && (c < Cn);
        float lp = 0.f;
        if (inb) {
            const float v = (((red[0][0][oreg][ol] + red[1][0][oreg][ol]) + red[2][0][oreg][ol]) + red[3][0][oreg][ol]) * scale;
            const float x0v = v + e0[0];
            pC[(size_t)r * ldc + c] = x0v;
            if (c < A) {
                const int pl = ol + A;
                const float rho = (((red[0][0][oreg][pl] + red[1][0][oreg][pl]) + red[2][0][oreg][pl]) + red[3][0][oreg][pl]) * scale + pbias[c + A];
                const float tt = tanhf(rho);
                const float l = -5.f + 3.5f * (tt + 1.f);
                const float sg = expf(l);
                const float x = x0v + cold[0] * sg;
                t.y0[(size_t)r * t.ldx0 + c] = tanhf(x);
                lp = -0.5f * cold[0] * cold[0] - l - 0.91893853320467274f - 2.f * (0.69314718055994531f - x - softplus_f(-2.f * x));
            }
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) lp += __shfl_xor(lp, o, 64);
        if (inb && c == 0 && t.y1) t.y1[r] = lp;
        TIM_FIN(); return;
    }

#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int c = c0 + 16 * f + (ol & 15);
        if (r >= R || c >= Cn) continue;
        const float v = (((red[0][f][oreg][ol] + red[1][f][oreg][ol]) + red[2][f][oreg][ol]) + red[3][f][oreg][ol]) * scale;
        float* cp = pC + (size_t)r * ldc + c;
        switch (epi) {
        case EPI_FWD: {
            const float x = v + e0[f];
            float y;
            switch (act) {
            case ACT_RELU: y = fmaxf(x, 0.f); break;
            case ACT_ELU: y = elu_f(x); break;
            case ACT_SIN: y = sinf(x); pout2[(size_t)r * ldout2 + c] = x; break;
            case ACT_TANH: y = tanhf(x); break;
            default: y = x;
            }
            *cp = y;
        } break;
        case EPI_DX: {
            float g = v + cold2[f];
            switch (act) {
            case ACT_RELU: g = e0[f] > 0.f ? g : 0.f; break;
            case ACT_ELU: g *= elu_grad_from_out(e0[f]); break;
            case ACT_SIN: g *= cosf(e0[f]); break;
            case ACT_TANH: g *= (1.f - e0[f] * e0[f]); break;
            default: break;
            }
            *cp = cold[f] + g;
        } break;
        case EPI_DX_POLICYBWD: {
            // v = dL/da_c from the critic path; e0 = rho, cold = eps, cold2 = a = tanh(x)
            const int A = n0;
            const float g = (float)exp(t.dptr[0]) * t.s0;            // dL/dlogpi = alpha / B
            const float tt = tanhf(e0[f]);
            const float sg = expf(-5.f + 3.5f * (tt + 1.f));
            const float y = cold2[f], e = cold[f];
            const float h = v * (1.f - y * y);
            t.y0[(size_t)r * 2 * A + c] = g * 2.f * y + h;
            t.y0[(size_t)r * 2 * A + A + c] = (g * (-1.f + 2.f * y * e * sg) + h * e * sg) * 3.5f * (1.f - tt * tt);
        } break;
        case EPI_DX_REPARAM:
            // e0 = eps * exp(log_std) * clamp-mask, written by vae_mid_kernel
            *cp = cold[f] + v;
            cp[t.F] = cold2[f] + v * e0[f];
            break;
        case EPI_DW:
        default: {
            const float g = cold[f] + v;
            *cp = g;
            if (fuse_opt) {      // e0 / cold2 / cold3 / cold4 = parameter, exp_avg, exp_avg_sq, Polyak target (prefetched)
                const size_t o = (size_t)r * ldc + c;
                float pv = e0[f], mv = cold2[f], vv = cold3[f], tv = cold4[f];
                adam_elem(adsc, g, &pv, &mv, &vv, t.ad_t ? &tv : nullptr);
                t.ad_p[o] = pv; t.ad_m[o] = mv; t.ad_v[o] = vv;
                if (t.ad_t) t.ad_t[o] = tv;
            }
        } break;
        }
    }
    TIM_FIN();
}

// ------------------------------------------------------------------------------------------------
// host launcher
// ------------------------------------------------------------------------------------------------
template <int LA, int LB, bool VA, bool VB>
static void launch_nf(int nf, dim3 g, hipStream_t st, const GemmBatch& gb) {
    if (nf == 1) hipLaunchKernelGGL((gemm16_kernel<LA, LB, 1, VA, VB>), g, dim3(256), 0, st, gb);
    else if (nf == 2) hipLaunchKernelGGL((gemm16_kernel<LA, LB, 2, VA, VB>), g, dim3(256), 0, st, gb);
    else hipLaunchKernelGGL((gemm16_kernel<LA, LB, 4, VA, VB>), g, dim3(256), 0, st, gb);
}

// 16-byte operand loads are legal for a launch only if EVERY task of it has 4-float-aligned rows and inner length
static bool all_vec(const GemmBatch& gb, bool opB) {
    for (int q = 0; q < gb.ntasks; ++q) {
        const GemmTask& t = gb.t[q];
        const float* p = opB ? t.B : t.A;
        const int ld = opB ? t.ldb : t.lda;
        if ((ld & 3) || (t.K & 3) || (((uintptr_t)p) & 15)) return false;
    }
    return true;
}

extern "C" int rl_launch_gemm16(int la, int lb, int nf, const GemmBatch* gb, int total_tiles, hipStream_t st) {
    if (total_tiles <= 0) return 0;
    dim3 g(total_tiles + (gb->nfin > 0 ? 1 : 0));
    if (gb->ntasks > 0 && (gb->t[0].flags & FLAG_PRE)) {       // fused-short-product launch: every task carries FLAG_PRE (and agrees on the form)
        const int fw = gb->t[0].flags & FLAG_PRE_FWD;
        for (int q = 0; q < gb->ntasks; ++q) {
            if (!(gb->t[q].flags & FLAG_PRE) || (gb->t[q].flags & FLAG_PRE_FWD) != fw) return -3;
            if (gb->t[q].n0 > (fw ? 48 : 32)) return -3;
        }
        if (la != LD_ROW || nf != 1) return -3;
        if (fw) {
            if (lb != LD_ROW) return -3;
            if (all_vec(*gb, true)) hipLaunchKernelGGL((gemm16_kernel<LD_ROW, LD_ROW, 1, false, true, true>), g, dim3(256), 0, st, *gb);
            else hipLaunchKernelGGL((gemm16_kernel<LD_ROW, LD_ROW, 1, false, false, true>), g, dim3(256), 0, st, *gb);
        } else {
            if (lb != LD_COL) return -3;
            hipLaunchKernelGGL((gemm16_kernel<LD_ROW, LD_COL, 1, false, false, true>), g, dim3(256), 0, st, *gb);
        }
        return (int)hipGetLastError();
    }
    if (la == LD_ROW && lb == LD_ROW) {
        if (all_vec(*gb, false) && all_vec(*gb, true)) launch_nf<LD_ROW, LD_ROW, true, true>(nf, g, st, *gb);
        else launch_nf<LD_ROW, LD_ROW, false, false>(nf, g, st, *gb);
    } else if (la == LD_ROW && lb == LD_COL) {
        if (all_vec(*gb, false)) launch_nf<LD_ROW, LD_COL, true, false>(nf, g, st, *gb);
        else launch_nf<LD_ROW, LD_COL, false, false>(nf, g, st, *gb);
    } else if (la == LD_COL && lb == LD_COL) launch_nf<LD_COL, LD_COL, false, false>(nf, g, st, *gb);
    else return -1;
    return (int)hipGetLastError();
}

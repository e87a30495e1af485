// gemm16_tile: the body of the small-matrix fp32 tile engine -- ONE 16 x (16*NF) output tile of ONE GemmTask, computed by a 256-thread
// workgroup (inner dimension split over its four waves, v_mfma_f32_16x16x4_f32, fixed-order LDS reduction, fused epilogue).
//
// Two callers:
//   gemm16.hip  gemm16_kernel   one tile per workgroup, one launch per stage of a step program (COH = false);
//   xchain.hip  xchain_kernel   a persistent launch that walks SEVERAL dependent stages, the workgroups of one XCD handing their tiles to
//                               each other through that XCD's L2 (COH = true: everything another workgroup of this launch may have
//                               written is read with sc1 loads, which bypass the reader's L1 and are served by L2).
//
// Operand fragment maps (MI355X guide, section 3): A: lane l holds A[i=l&15][k=l>>4]; B: lane l holds B[k=l>>4][j=l&15];
// C/D: col=l&15, row=4*(l>>4)+reg.  The inner index may be permuted freely as long as A and B agree, so each lane takes FOUR
// CONSECUTIVE inner indices (one 16-byte load in the row-contiguous case) and feeds them to four successive MFMAs.
#pragma once
#include "common.h"
#include "kparams.h"

typedef unsigned rl_u32x4 __attribute__((ext_vector_type(4)));

// sc1 loads (L1 bypass, L2-served) through a raw buffer descriptor over the operand's matrix: the compiler sees them (waits, scheduling)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rl_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x27000);
}
template <bool COH> __device__ __forceinline__ float rl_ld(const float* base, size_t idx) {
    if (COH) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rl_rsrc(base), (unsigned)(idx * 4), 0, 16));
    return base[idx];
}
template <bool COH> __device__ __forceinline__ f32x4 rl_ld4(const float* base, size_t idx) {
    if (COH) {
        const rl_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rl_rsrc(base), (unsigned)(idx * 4), 0, 16);
        return (f32x4){__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
    }
    return *reinterpret_cast<const f32x4*>(base + idx);
}

// BRANCH-FREE operand fetch.  Out-of-range rows / inner indices are handled by CLAMPING the address into the
// matrix and zeroing the value with a select: no exec-masked branch around any load.  (With `if (in_range) load`
// hipcc wraps every load in s_cbranch_execz + s_waitcnt vmcnt(0): 147 branches and 21 full drains in a kernel
// with 16 MFMAs, i.e. the eight 16-byte loads a wave needs were serialised instead of overlapped.)
template <int LOADER, bool VEC, bool COH = false>
__device__ __forceinline__ void load_raw(const float* __restrict__ P, int ld, int base, int lim,
                                         int i, int k0, int K, float (&v)[4]) {
    const int idx = min(base + i, lim - 1);
    if (LOADER == LD_ROW) {
        if (VEC) {                       // K % 4 == 0, 16-byte aligned rows: the 4 indices are valid together
            const f32x4 x = rl_ld4<COH>(P, (size_t)idx * ld + min(k0, K - 4));
            v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) v[s] = rl_ld<COH>(P, (size_t)idx * ld + min(k0 + s, K - 1));
        }
    } else {  // LD_COL
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = rl_ld<COH>(P, (size_t)min(k0 + s, K - 1) * ld + idx);
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
template <int LA, int LB, int NF, bool VA, bool VB, int NU, bool COH = false>
__device__ __forceinline__ void mac_group(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                          int r0, int R, int c0, int Cn, int i, int k0, int K,
                                          f32x4 (&acc)[NF], float& asum, bool want_bias) {
    float a[NU][4], b[NU][NF][4];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        load_raw<LA, VA, COH>(A, lda, r0, R, i, k0 + 64 * u, K, a[u]);
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

template <int LB, int NF, bool VB, int NU, int NJ, bool FWD, bool COH = false>
__device__ __forceinline__ void mac_group_pre(const PreSrc& ps, const float* __restrict__ B, int ldb, int r0, int R, int c0, int Cn,
                                              int i, int kq, int kb, int K, bool store, f32x4 (&acc)[NF]) {
    float xf[NJ][4], wf[NU][NJ][4], mk[NU][4], b[NU][NF][4];
    const int K1 = ps.K1;
    const size_t xrow = (size_t)min(r0 + i, R - 1) * ps.ldx;
    // dX form: M = the saved ReLU output of this launch's forward half (row of the minibatch); forward form: M = the layer's bias (ldm = 0)
    const size_t mrow = (size_t)min(r0 + i, R - 1) * ps.ldm;
#pragma unroll
    for (int jc = 0; jc < NJ; ++jc)
#pragma unroll
        for (int m = 0; m < 4; ++m) xf[jc][m] = rl_ld<COH>(ps.X, xrow + min(16 * jc + 4 * kq + m, K1 - 1));
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int kcol = min(kb + 64 * u + i, K - 1);
#pragma unroll
        for (int jc = 0; jc < NJ; ++jc)
#pragma unroll
            for (int m = 0; m < 4; ++m) wf[u][jc][m] = ps.Wt[(size_t)min(16 * jc + 4 * kq + m, K1 - 1) * ps.ldw + kcol];
#pragma unroll
        for (int m = 0; m < 4; ++m) mk[u][m] = rl_ld<COH && !FWD>(ps.M, mrow + min(kb + 64 * u + 4 * kq + m, K - 1));
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

// instrumented build (RL_TIMING, gemm16.hip): the caller's stamp array travels as an extra argument
#ifdef RL_TIMING
#define RL_TIM_PARAM , unsigned long long* tim_c
#define TIMB(k) do { if (tim_c && threadIdx.x == 0 && (blockIdx.x & 7) == 0) tim_c[k] = clock64(); } while (0)
#else
#define RL_TIM_PARAM
#define TIMB(k) do {} while (0)
#endif

// TaskT: GemmTask, or GemmTask qualified with the CONSTANT address space (a table in device memory that no kernel writes: scalar loads;
// through a generic pointer hipcc reads every field with a vector load + s_waitcnt vmcnt(0), one L2 round trip each).
#define RL_CONST_AS __attribute__((address_space(4)))

// Everything a tile needs that is UNIFORM and known from its task record alone: the hot block of the record and the epilogue's operand
// slots.  gemm16_prep fills it (scalar loads + scalar code only); a persistent caller runs it BEFORE it waits for the tile's inputs, so
// that the record's two dependent L2 round trips (~1 800 cycles, xc_timeline "record" + "slots") hide behind the wait.
struct G16Plan {
    const float *pA, *pB, *pbias, *paux, *pr1u, *pr1v, *px2; float *pC, *pout2;
    int lda, ldb, ldc, ldaux, ldout2, R, Cn, K, epi, act, flags, n0, local, r0, c0, tc;
    float scale;
    // The epilogue kind only selects up to five SLOT descriptors (base, row stride, column stride, offset, column window); the loads
    // themselves are generic and branch-free (a lane outside its window reads the slot's base address and the value is discarded)
    const float* sp[5]; int srs[5], scs[5], sof[5], slo[5], shi[5];
};

// The record fields the plan is made of, as raw values: gemm16_load only ISSUES the scalar loads (nothing uses a value), so a caller that
// has something to wait for in between -- xchain's poll of its group's flags -- overlaps the record's L2 round trip with that wait;
// gemm16_slots then derives the plan (scalar code only).
struct G16Raw {
    const float *A, *B, *bias, *aux, *r1u, *r1v, *x0, *x1, *x2, *aux3; float *C, *out2;
    int lda, ldb, ldc, ldaux, ldout2, R, Cn, K, epi, act, flags, n0, tiles_c, ldx0, ldx1, ldaux3, F;
    float scale;
};
template <class TaskT>
__device__ __forceinline__ void gemm16_load(const TaskT& t, G16Raw& r) {
    r.A = t.A; r.B = t.B; r.C = t.C; r.bias = t.bias; r.aux = t.aux; r.r1u = t.r1u; r.r1v = t.r1v;
    r.lda = t.lda; r.ldb = t.ldb; r.ldc = t.ldc; r.ldaux = t.ldaux; r.R = t.R; r.Cn = t.Cn; r.K = t.K; r.tiles_c = t.tiles_c;
    r.epi = t.epi; r.act = t.act; r.flags = t.flags; r.scale = t.scale; r.n0 = t.n0; r.out2 = t.out2; r.ldout2 = t.ldout2;
    r.x0 = t.x0; r.x1 = t.x1; r.x2 = t.x2; r.aux3 = t.aux3; r.ldx0 = t.ldx0; r.ldx1 = t.ldx1; r.ldaux3 = t.ldaux3; r.F = t.F;
}
template <int NF, bool COH, class TaskT>
__device__ __forceinline__ void gemm16_slots(const TaskT& t, const G16Raw& r, const int tr, const int tc, const float* const* dyn, G16Plan& P) {
    P.pA = r.A; P.pB = r.B; P.pC = r.C; P.pbias = r.bias; P.paux = r.aux; P.pr1u = r.r1u; P.pr1v = r.r1v;
    P.lda = r.lda; P.ldb = r.ldb; P.ldc = r.ldc; P.ldaux = r.ldaux;
    P.R = r.R; P.Cn = r.Cn; P.K = r.K;
    P.epi = r.epi; P.act = r.act; P.flags = r.flags; P.n0 = r.n0; P.scale = r.scale;
    P.pout2 = r.out2; P.ldout2 = r.ldout2;
    P.px2 = r.x2;
    if (COH && dyn) P.px2 = (P.flags & FLAG_DYN_EPS) ? dyn[0] : (P.flags & FLAG_DYN_EPS2) ? dyn[1] : (P.flags & FLAG_DYN_EPS3) ? dyn[2] : P.px2;
    P.local = tr * r.tiles_c + tc; P.tc = tc;
    P.r0 = tr * 16; P.c0 = tc * 16 * NF;
#pragma unroll
    for (int q = 0; q < 5; ++q) { P.sp[q] = nullptr; P.srs[q] = 0; P.scs[q] = 1; P.sof[q] = 0; P.slo[q] = 0; P.shi[q] = P.Cn; }
    switch (P.epi) {
    case EPI_FWD: P.sp[0] = P.pbias; break;
    case EPI_DX:
        if (P.act != ACT_NONE) { P.sp[0] = P.paux; P.srs[0] = P.ldaux; }
        if (P.flags & FLAG_ACCUM) { P.sp[1] = P.pC; P.srs[1] = P.ldc; }
        if (P.pr1u) { P.sp[2] = P.pr1u; P.srs[2] = 1; P.scs[2] = 0; P.sp[3] = P.pr1v; }
        break;
    case EPI_FWD_MSE:
        P.sp[0] = P.pbias;
        P.sp[1] = r.x0; P.srs[1] = r.ldx0; P.shi[1] = P.n0;
        P.sp[2] = r.x1; P.srs[2] = 1; P.scs[2] = 0; P.slo[2] = P.n0;
        break;
    case EPI_FWD_POLICY:
        P.sp[0] = P.pbias;
        P.sp[1] = P.px2; P.srs[1] = P.n0; P.shi[1] = P.n0;
        break;
    case EPI_DX_POLICYBWD:
        P.sp[0] = r.x0; P.srs[0] = 2 * P.n0; P.sof[0] = P.n0;
        P.sp[1] = P.px2; P.srs[1] = P.n0;
        P.sp[2] = r.x1; P.srs[2] = r.ldx1;
        break;
    case EPI_DX_REPARAM:
        P.sp[0] = r.aux3; P.srs[0] = r.ldaux3;
        P.sp[1] = P.pC; P.srs[1] = P.ldc;
        P.sp[2] = P.pC; P.srs[2] = P.ldc; P.sof[2] = r.F;
        break;
    default:   // EPI_DW
        if (P.flags & FLAG_ACCUM) { P.sp[1] = P.pC; P.srs[1] = P.ldc; }
        if constexpr (!COH) {
            if (t.ad_p) {      // optimizer fused in: the tile of the parameter, its Adam moments (and its Polyak target)
                P.sp[0] = t.ad_p; P.sp[2] = t.ad_m; P.sp[3] = t.ad_v; P.sp[4] = t.ad_t;
                P.srs[0] = P.srs[2] = P.srs[3] = P.srs[4] = P.ldc;
            }
        }
    }
}
template <int NF, bool COH, class TaskT>
__device__ __forceinline__ void gemm16_prep(const TaskT& t, const int tr, const int tc, const float* const* dyn, G16Plan& P) {
    G16Raw r;
    gemm16_load<TaskT>(t, r);
    gemm16_slots<NF, COH, TaskT>(t, r, tr, tc, dyn, P);
}

// One output tile of task t (plan P from gemm16_prep).
template <int LA, int LB, int NF, bool VA, bool VB, bool PRE, bool COH, class TaskT = GemmTask>
__device__ __forceinline__ void gemm16_run(const TaskT& t, const G16Plan& P, float (&red)[4][NF][4][64], float (&bsum)[4][16] RL_TIM_PARAM) {
    const float* const pA = P.pA; const float* const pB = P.pB; float* const pC = P.pC; const float* const pbias = P.pbias;
    const int lda = P.lda, ldb = P.ldb, ldc = P.ldc;
    const int R = P.R, Cn = P.Cn, K = P.K;
    const int epi = P.epi, act = P.act, flags = P.flags, n0 = P.n0;
    const float scale = P.scale;
    float* const pout2 = P.pout2; const int ldout2 = P.ldout2;
    const int local = P.local, tc = P.tc, r0 = P.r0, c0 = P.c0;
    const float* sp[5]; int srs[5], scs[5], sof[5], slo[5], shi[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) { sp[q] = P.sp[q]; srs[q] = P.srs[q]; scs[q] = P.scs[q]; sof[q] = P.sof[q]; slo[q] = P.slo[q]; shi[q] = P.shi[q]; }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;

    f32x4 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float asum = 0.f;
    const bool want_bias = !COH && (epi == EPI_DW) && (flags & FLAG_BIASGRAD) && (tc == 0);
    TIMB(5);
    const int ol = threadIdx.x & 63, oreg = threadIdx.x >> 6;
    const int r = r0 + (ol >> 4) * 4 + oreg;
    // Plain loads, PINNED above the operand stream by a memory-clobbering empty asm: left alone, hipcc sinks them below the reduction
    // barrier, next to their first use (a serialised L2 round trip in the epilogue).  The compiler counts them in its vmcnt bookkeeping,
    // so the wait that claims them (long after the operand stream behind them has been consumed) is its own.  (Rounds 1-2 issued them as
    // volatile asm loads claimed by an explicit s_waitcnt: invisible to the register allocator, which may copy or reuse a destination
    // register while the load is in flight -- tools/check_async_asm.py caught exactly that when this body moved into a header.)
    // ... and UNCONDITIONAL: a slot the epilogue does not use reads one word of operand A instead (discarded below).  An `if (slot in use)`
    // around the loads is a control-flow diamond, and hipcc drains vmcnt at every merge point: five exposed L2 round trips per tile.
    float ev[5][NF];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const bool used = sp[q] != nullptr;
        const float* const spq = used ? sp[q] : pA;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int c = c0 + 16 * f + (ol & 15);
            const bool ok = used && (r < R) && (c >= slo[q]) && (c < shi[q]);
            ev[q][f] = rl_ld<COH>(spq, ok ? (size_t)r * srs[q] + (size_t)(c * scs[q] + sof[q]) : (size_t)0);
        }
    }

    // fused optimizer: Adam scalars of the group, and (column-tile 0 only) the bias element this thread will update
    AdamScal adsc;
    const bool fuse_opt = !COH && (epi == EPI_DW) && t.ad_p;
    if constexpr (!COH) { if (fuse_opt) adsc = t.ad_grp->sc; }
    float bpv = 0.f, bmv = 0.f, bvv = 0.f, btv = 0.f;
    const bool bias_opt = want_bias && fuse_opt && t.ad_pb && threadIdx.x < 16 && r0 + (int)threadIdx.x < R;
    if (bias_opt) {
        const int o = r0 + threadIdx.x;
        bpv = t.ad_pb[o]; bmv = t.ad_mb[o]; bvv = t.ad_vb[o];
        if (t.ad_tb) btv = t.ad_tb[o];
    }
    asm volatile("" ::: "memory");      // the pin (see above)
    TIMB(6);

    // wave w owns the 16-wide inner chunks w, w+4, w+8, ...
    if constexpr (PRE) {
        PreSrc ps; ps.X = t.x0; ps.ldx = t.ldx0; ps.Wt = t.x1; ps.ldw = t.ldx1; ps.K1 = n0; ps.M = t.x2; ps.ldm = t.ldaux2; ps.out = t.y0; ps.ldo = t.ldout2;
        const bool store = (tc == 0) && ps.out;
        // forward form (LB = LD_ROW): K1 <= 48, bias + ReLU; dX form (LB = LD_COL): K1 <= 32, ReLU mask
        constexpr int NJ = (LB == LD_ROW) ? 3 : 2;
        constexpr bool FW = (LB == LD_ROW);
        for (int kb = w * 16; kb < K; kb += 256) {
            const int nu = (K - kb + 63) >> 6;
            if (nu >= 4) mac_group_pre<LB, NF, VB, 4, NJ, FW, COH>(ps, pB, ldb, r0, R, c0, Cn, i, kq, kb, K, store, acc);
            else if (nu == 1) mac_group_pre<LB, NF, VB, 1, NJ, FW, COH>(ps, pB, ldb, r0, R, c0, Cn, i, kq, kb, K, store, acc);
            else if (nu == 2) mac_group_pre<LB, NF, VB, 2, NJ, FW, COH>(ps, pB, ldb, r0, R, c0, Cn, i, kq, kb, K, store, acc);
            else mac_group_pre<LB, NF, VB, 3, NJ, FW, COH>(ps, pB, ldb, r0, R, c0, Cn, i, kq, kb, K, store, acc);
        }
    } else
    for (int kb = w * 16; kb < K; kb += 256) {
        const int k0 = kb + 4 * kq;
        const int nu = (K - kb + 63) >> 6;           // chunks of this group that touch the matrix (uniform per wave)
        if (nu >= 4) mac_group<LA, LB, NF, VA, VB, 4, COH>(pA, lda, pB, ldb, r0, R, c0, Cn, i, k0, K, acc, asum, want_bias);
        else if (nu == 1) mac_group<LA, LB, NF, VA, VB, 1, COH>(pA, lda, pB, ldb, r0, R, c0, Cn, i, k0, K, acc, asum, want_bias);
        else if (nu == 2) mac_group<LA, LB, NF, VA, VB, 2, COH>(pA, lda, pB, ldb, r0, R, c0, Cn, i, k0, K, acc, asum, want_bias);
        else mac_group<LA, LB, NF, VA, VB, 3, COH>(pA, lda, pB, ldb, r0, R, c0, Cn, i, k0, K, acc, asum, want_bias);
    }

    TIMB(2);
    // discard what out-of-window lanes fetched
    float e0[NF], cold[NF], cold2[NF], cold3[NF], cold4[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int c = c0 + 16 * f + (ol & 15);
        float m[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) m[q] = (sp[q] && (r < R) && (c >= slo[q]) && (c < shi[q])) ? ev[q][f] : 0.f;
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
    TIMB(3);

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
        return;
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
        return;
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
}

// prep + run: one tile, record read at the point of use (gemm16_kernel: the record is in the kernel-argument segment)
template <int LA, int LB, int NF, bool VA, bool VB, bool PRE, bool COH, class TaskT = GemmTask>
__device__ __forceinline__ void gemm16_tile(const TaskT& t, const int tr, const int tc, float (&red)[4][NF][4][64], float (&bsum)[4][16],
                                            const float* const* dyn RL_TIM_PARAM) {
    G16Plan P;
    gemm16_prep<NF, COH, TaskT>(t, tr, tc, dyn, P);
#ifdef RL_TIMING
    TIMB(1);
    gemm16_run<LA, LB, NF, VA, VB, PRE, COH, TaskT>(t, P, red, bsum, tim_c);
#else
    gemm16_run<LA, LB, NF, VA, VB, PRE, COH, TaskT>(t, P, red, bsum);
#endif
}

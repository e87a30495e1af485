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
#ifndef RL_UNIFORM_W
#define RL_UNIFORM_W 1
#endif
#include "common.h"
#include "kparams.h"
#include <type_traits>

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
// GA: A is this lane's ROW already (gathered rows: GemmTask::gidx), not the matrix base
template <int LA, int LB, int NF, bool VA, bool VB, int NU, bool COH = false, bool GA = false>
__device__ __forceinline__ void mac_group(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                          int r0, int R, int c0, int Cn, int i, int k0, int K,
                                          f32x4 (&acc)[NF], float& asum, bool want_bias) {
    float a[NU][4], b[NU][NF][4];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        if (GA) load_raw<LA, VA, COH>(A, lda, 0, 1, 0, k0 + 64 * u, K, a[u]);
        else load_raw<LA, VA, COH>(A, lda, r0, R, i, k0 + 64 * u, K, a[u]);
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
// st_lo / st_hi: the k range of the recomputed first-layer output that THIS tile stores (every tile of a row block recomputes all of it; the
// sixteen column tiles of the headline layers each store one 16-wide chunk instead of column tile 0 storing everything: the launch ends when its
// slowest workgroup does)
struct PreSrc { const float* X; int ldx; const float* Wt; int ldw; int K1; const float* M; int ldm; float* out; int ldo; int st_lo, st_hi; };

// XS (FLAG_PRE_MSE): the row operand X of the short product sits in LDS ([16][RL_XS_LD] floats, written by this tile's first phase) instead of memory
#define RL_XS_LD 36
// the global operands of one mac_group_pre block (everything but the row operand X): loaded by pre_load, consumed by mac_group_pre<.., PL = true>
template <int NU, int NJ, int NF> struct PreRegs { float wf[NU][NJ][4], mk[NU][4], b[NU][NF][4], xf[NJ][4]; };
template <int LB, int NF, bool VB, int NU, int NJ, bool FWD, bool COH = false, bool XLDS = false>
__device__ __forceinline__ void pre_load(const PreSrc& ps, const float* __restrict__ B, int ldb, int r0, int R, int c0, int Cn,
                                         int i, int kq, int kb, int K, PreRegs<NU, NJ, NF>& pr) {
    const int K1 = ps.K1;
    // dX form: M = the saved ReLU output of this launch's forward half (row of the minibatch); forward form: M = the layer's bias (ldm = 0)
    const size_t mrow = (size_t)min(r0 + i, R - 1) * ps.ldm;
    if constexpr (!XLDS) {          // the row operand X of the short product (from LDS in the mse form: read where it is consumed)
        const size_t xrow = (size_t)min(r0 + i, R - 1) * ps.ldx;
#pragma unroll
        for (int jc = 0; jc < NJ; ++jc)
#pragma unroll
            for (int m = 0; m < 4; ++m) pr.xf[jc][m] = rl_ld<COH>(ps.X, xrow + min(16 * jc + 4 * kq + m, K1 - 1));
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int kcol = min(kb + 64 * u + i, K - 1);
#pragma unroll
        for (int jc = 0; jc < NJ; ++jc)
#pragma unroll
            for (int m = 0; m < 4; ++m) pr.wf[u][jc][m] = ps.Wt[(size_t)min(16 * jc + 4 * kq + m, K1 - 1) * ps.ldw + kcol];
        if constexpr (XLDS) {       // (FLAG_PRE_MSE launches: the launcher has checked that M's rows are 16-byte regular and K % 4 == 0)
            const f32x4 mv = *reinterpret_cast<const f32x4*>(ps.M + mrow + min(kb + 64 * u + 4 * kq, K - 4));
            pr.mk[u][0] = mv[0]; pr.mk[u][1] = mv[1]; pr.mk[u][2] = mv[2]; pr.mk[u][3] = mv[3];
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) pr.mk[u][m] = rl_ld<COH && !FWD>(ps.M, mrow + min(kb + 64 * u + 4 * kq + m, K - 1));
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) load_raw<LB, VB>(B, ldb, c0 + 16 * f, Cn, i, kb + 4 * kq + 64 * u, K, pr.b[u][f]);
    }
}
// PL: the block's global operands were loaded earlier (pre_load into `pl`): only X is fetched here
template <int LB, int NF, bool VB, int NU, int NJ, bool FWD, bool COH = false, bool XLDS = false, bool PL = false>
__device__ __forceinline__ void mac_group_pre(const PreSrc& ps, const float* __restrict__ B, int ldb, int r0, int R, int c0, int Cn,
                                              int i, int kq, int kb, int K, bool store, bool elu, f32x4 (&acc)[NF], const float* xs = nullptr,
                                              PreRegs<NU, NJ, NF>* pl = nullptr) {
    PreRegs<NU, NJ, NF> own;
    const int K1 = ps.K1;
    if constexpr (!PL) pre_load<LB, NF, VB, NU, NJ, FWD, COH, XLDS>(ps, B, ldb, r0, R, c0, Cn, i, kq, kb, K, own);
    PreRegs<NU, NJ, NF>& pr = PL ? *pl : own;
    float (&wf)[NU][NJ][4] = pr.wf; float (&mk)[NU][4] = pr.mk; float (&b)[NU][NF][4] = pr.b; float (&xf)[NJ][4] = pr.xf;
    if constexpr (XLDS) {
#pragma unroll
        for (int jc = 0; jc < NJ; ++jc)
#pragma unroll
            for (int m = 0; m < 4; ++m) xf[jc][m] = xs[i * RL_XS_LD + 16 * jc + 4 * kq + m];
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
            // dX form: mk = saved ReLU / ELU output (mask; `elu` is uniform over the launch: FLAG_PRE_ELU); forward form: mk = the layer's bias
            const float g = FWD ? fmaxf(D[u][m] + mk[u][m], 0.f) : (mk[u][m] > 0.f ? D[u][m] : (elu ? D[u][m] * (mk[u][m] + 1.f) : 0.f));
            a[m] = (rok && (kb + 64 * u + 4 * kq + m) < K) ? g : 0.f;
        }
        if (store && rok && kb + 64 * u >= ps.st_lo && kb + 64 * u < ps.st_hi) {
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

// One output tile (row tile tr, column tile tc) of task t.  `dyn`: the per-call noise pointers of a launch whose task table lives in
// device memory (xchain: FLAG_DYN_EPS* select dyn[0..2] for the slots whose base is x2); nullptr where the caller patched t.x2 itself
// before it planned the record (rl_gemm16_plan).
// EPI_K / ACT_K: the epilogue kind / activation as COMPILE-TIME constants (-1: read from the record).  A launch whose tasks all share a
// plain forward or dX epilogue runs an instantiation that contains nothing else: the generic body is ~2 900 instructions, and its
// epilogue walks a ladder of far scalar branches through cold code (instruction-cache misses: 1 200 - 2 800 cycles between "reduction
// barrier passed" and "tile stored" in tools/exp/gemm_timeline.py, for five VALU instructions and a store).
// GATHER: operand-A rows come through the task's row-index table (gidx), e.g. straight out of the replay ring (forward launches that ride in
// the optimizer launch of the previous step: the minibatch slot is being gathered by other workgroups of the same launch).
// MSE (with PRE, dX form): FLAG_PRE_MSE launches -- the short product's row operand is computed by a first phase of the tile (common.h)
// FAST (gemm16_fast_kernel; 0 = off): what the operand LOADS need -- bases, row strides, extents -- arrives in preloaded SGPRs (FastOps) instead of
// the record, and the first block's loads are issued BEFORE the record is waited for: the record's scalar-load round trip (~900 cycles, needed by
// the epilogue only) runs under the operand loads instead of in front of them.  FAST = 4: K is a multiple of 256 (blocks of four 16-deep chunks
// per wave); FAST = 1: K <= 64 (the first layers: one chunk per wave is all there is).
struct FastOps { const float* pA; const float* pB; int lda, ldb, K, R, Cn, tiles_c;
                 const float* X; const float* Wt; const float* M; int ldx, ldw, K1, ldm; };       // (X .. ldm: the fused short product's operands, gemm16_fastpre_kernel)
template <int LA, int LB, int NF, bool VA, bool VB, bool PRE, bool COH, class TaskT = GemmTask, int EPI_K = -1, int ACT_K = -1, bool GATHER = false, bool MSE = false,
          int FAST = 0, int NJ_K = 0>
__device__ __forceinline__ void gemm16_tile(const TaskT& t, const int tr, const int tc, float (&red)[4][NF][4][64], float (&bsum)[4][16],
                                            const float* const* dyn RL_TIM_PARAM, const FastOps* fo = nullptr) {
    static_assert(!FAST || (!COH && !GATHER && (PRE || !MSE) && (!PRE || (FAST == 4 && LA == LD_ROW && LB == LD_COL && NF == 1))), "FAST: plain forward / dX tiles, or the dX form of the fused short product");
    // fused short product: 16-wide chunks of its inner length K1 -- forward form (LB = LD_ROW) K1 <= 48, dX form K1 <= 32; NJ_K = 1: the launcher has
    // checked K1 <= 16 (the policy head: 2 A columns), and the second chunk's loads and MFMAs (all masked) are not issued at all
    constexpr int NJ = NJ_K ? NJ_K : (LB == LD_ROW) ? 3 : 2;
    constexpr bool FW = (LB == LD_ROW);
    // the hot block of the task record and the epilogue's operand slots, fetched as ONE burst of scalar loads
    const float* const pA = FAST ? fo->pA : t.A; const float* const pB = FAST ? fo->pB : t.B; float* const pC = t.C; const float* const pbias = t.bias;
    const int lda = FAST ? fo->lda : t.lda, ldb = FAST ? fo->ldb : t.ldb, ldc = t.ldc;
    const int R = FAST ? fo->R : t.R, Cn = FAST ? fo->Cn : t.Cn, K = FAST ? fo->K : t.K, tiles_c = FAST ? fo->tiles_c : t.tiles_c;
    const int epi = EPI_K == EPI_DWA ? EPI_DW : EPI_K >= 0 ? EPI_K : t.epi, act = ACT_K >= 0 ? ACT_K : t.act, flags = t.flags, n0 = t.n0;
    const int* const pgidx = GATHER ? t.gidx : nullptr;
    const float scale = t.scale;
    float* const pout2 = t.out2; const int ldout2 = t.ldout2;
    const float* sp[5]; int srs[5], scs[5], sof[5], slo[5], shi[5];
    const int scs0 = t.scs0, spx2 = t.spx2;
#pragma unroll
    for (int q = 0; q < 5; ++q) { sp[q] = t.sp[q]; srs[q] = t.srs[q]; sof[q] = t.sof[q]; slo[q] = t.slo[q]; shi[q] = t.shi[q]; scs[q] = (scs0 >> q) & 1 ? 0 : 1; }
    if (COH && dyn && spx2) {
        const float* const px2 = (flags & FLAG_DYN_EPS) ? dyn[0] : (flags & FLAG_DYN_EPS2) ? dyn[1] : (flags & FLAG_DYN_EPS3) ? dyn[2] : sp[1];
        sp[1] = px2;               // (x2 is slot 1 wherever it is a slot)
    }
    // FAST: the first 256-deep block's operand loads go out now, from preloaded scalars (the reads of the record above are in flight)
    float fa[FAST ? FAST : 1][4], fb[FAST ? FAST : 1][NF][4];
    // the first block of the fused short product's global operands (and, mse form, of the first phase's product): loaded by the mse phase (which
    // requests them behind its own operands) or, FAST, right here from preloaded scalars
    PreRegs<4, NJ, NF> mse_pl;
    float a1[PRE && MSE ? 4 : 1][4], b1[PRE && MSE ? 4 : 1][2][4];
    if constexpr (FAST && PRE) {
        const int lane_ = threadIdx.x & 63, w_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int i_ = lane_ & 15, kq_ = lane_ >> 4;
        PreSrc pf; pf.X = fo->X; pf.ldx = fo->ldx; pf.Wt = fo->Wt; pf.ldw = fo->ldw; pf.K1 = fo->K1; pf.M = fo->M; pf.ldm = fo->ldm; pf.out = nullptr; pf.ldo = 0; pf.st_lo = pf.st_hi = 0;
        if constexpr (MSE) {
            const int k0_ = w_ * 16 + 4 * kq_;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                load_raw<LD_ROW, true>(pf.M, pf.ldm, tr * 16, R, i_, k0_ + 64 * u, K, a1[u]);
#pragma unroll
                for (int f = 0; f < 2; ++f) load_raw<LD_ROW, true>(pf.Wt, pf.ldw, 16 * f, pf.K1, i_, k0_ + 64 * u, K, b1[u][f]);
            }
            __builtin_amdgcn_sched_barrier(0);          // (the first phase's operands FIRST: the load counter retires in order)
        }
        pre_load<LB, NF, VB, 4, NJ, FW, false, MSE>(pf, pB, ldb, tr * 16, R, tc * 16 * NF, Cn, i_, kq_, w_ * 16, K, mse_pl);
        TIMB(5);
    } else if constexpr (FAST) {
        const int lane_ = threadIdx.x & 63, w_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int i_ = lane_ & 15, k_ = w_ * 16 + 4 * (lane_ >> 4);
#pragma unroll
        for (int u = 0; u < FAST; ++u) {
            load_raw<LA, VA>(pA, lda, tr * 16, R, i_, k_ + 64 * u, K, fa[u]);
#pragma unroll
            for (int f = 0; f < NF; ++f) load_raw<LB, VB>(pB, ldb, tc * 16 * NF + 16 * f, Cn, i_, k_ + 64 * u, K, fb[u][f]);
        }
        TIMB(5);                    // FAST: the first block's operand loads are issued (before the record is claimed)
    }
    // (materialise the whole record HERE: left to itself hipcc loads each slot field next to its first use, between the operand loads,
    // with a scalar-load round trip in front of every one of them)
    asm volatile("" :: "s"(pA), "s"(pB), "s"(pC), "s"(pbias), "s"(lda), "s"(ldb), "s"(ldc), "s"(R), "s"(Cn), "s"(K), "s"(epi), "s"(act), "s"(flags), "s"(n0), "s"(scale));
    if constexpr (GATHER) asm volatile("" :: "s"(pgidx));
    // (the fused short product's operands and the mse phase's: same burst -- each of them read where first used costs a round trip of its own in
    // front of that phase's loads)
    if constexpr (PRE) asm volatile("" :: "s"(t.x0), "s"(t.ldx0), "s"(t.x1), "s"(t.ldx1), "s"(t.x2), "s"(t.ldaux2), "s"(t.y0), "s"(ldout2));
    if constexpr (MSE) asm volatile("" :: "s"(t.tgs), "s"(t.tgr), "s"(t.ldtgs), "s"(t.pad_mse), "s"(t.s0), "s"(t.s1), "s"(t.mse_part));
    asm volatile("" :: "s"(sp[0]), "s"(sp[1]), "s"(sp[2]), "s"(sp[3]), "s"(sp[4]), "s"(srs[0]), "s"(srs[1]), "s"(srs[2]), "s"(srs[3]), "s"(srs[4]),
                 "s"(sof[0]), "s"(sof[1]), "s"(sof[2]), "s"(sof[3]), "s"(sof[4]));
    asm volatile("" :: "s"(slo[0]), "s"(slo[1]), "s"(slo[2]), "s"(slo[3]), "s"(slo[4]), "s"(shi[0]), "s"(shi[1]), "s"(shi[2]), "s"(shi[3]), "s"(shi[4]), "s"(scs0));
    TIMB(1);
    const int local = tr * tiles_c + tc;
    const int r0 = tr * 16, c0 = tc * 16 * NF;
    // (w is wave-uniform but a VGPR value to hipcc: without the readfirstlane the K loop and the `nu` dispatch below are exec-masked
    // divergent control flow -- saveexec ladders, accumulator copies through VGPRs, conservative waits at every merge)
    const int lane = threadIdx.x & 63, w = RL_UNIFORM_W ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)(threadIdx.x >> 6);
    const int i = lane & 15, kq = lane >> 4;
    // gathered rows: this lane's row of operand A (one dependent load, issued with the tile's first instructions)
    const float* pArow = pA;
    if constexpr (GATHER) pArow = pA + (size_t)pgidx[min(r0 + i, R - 1)] * lda;

    f32x4 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float asum = 0.f;
    // (weight gradients are the k-major / k-major launches: every other instantiation drops the bias-gradient code)
    constexpr bool DW = !COH && LA == LD_COL && LB == LD_COL;
    const bool want_bias = DW && (epi == EPI_DW) && (flags & FLAG_BIASGRAD) && (tc == 0);
    if constexpr (!FAST) TIMB(5);
    const int ol = threadIdx.x & 63, oreg = threadIdx.x >> 6;
    const int r = r0 + (ol >> 4) * 4 + oreg;
    // Plain loads, PINNED above the operand stream by a memory-clobbering empty asm: left alone, hipcc sinks them below the reduction
    // barrier, next to their first use (a serialised L2 round trip in the epilogue).  The compiler counts them in its vmcnt bookkeeping,
    // so the wait that claims them (long after the operand stream behind them has been consumed) is its own.  (Rounds 1-2 issued them as
    // volatile asm loads claimed by an explicit s_waitcnt: invisible to the register allocator, which may copy or reuse a destination
    // register while the load is in flight -- tools/check_async_asm.py caught exactly that when this body moved into a header.)
    // ... and UNCONDITIONAL: a slot the epilogue does not use reads one word of operand A instead (discarded below).  An `if (slot in use)`
    // around the loads is a control-flow diamond, and hipcc drains vmcnt at every merge point: five exposed L2 round trips per tile.
    // Slots that an instantiation with its epilogue compiled in can never use are not loaded at all: forward: slot 0 (bias); dX: 0-3 (saved
    // activation, accumulate-into, rank-1 pair); weight gradient: none (its accumulate form runs the generic kernel).
    float ev[5][NF];
    auto slot = [&](auto qtag) {
        constexpr int q = decltype(qtag)::value;
#pragma unroll
        for (int f = 0; f < NF; ++f) ev[q][f] = 0.f;
        constexpr bool possible = EPI_K < 0 || (EPI_K == EPI_FWD && q == 0) || (EPI_K == EPI_DX && q < 4) || (EPI_K == EPI_DWA && q != 1) ||
                                  ((EPI_K == EPI_FWD_MSE || EPI_K == EPI_DX_REPARAM || EPI_K == EPI_DX_POLICYBWD) && q < 3) || (EPI_K == EPI_FWD_POLICY && q < 2);
        if constexpr (possible) {
            const bool used = sp[q] != nullptr;
            const float* const spq = used ? sp[q] : pA;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int c = c0 + 16 * f + (ol & 15);
                const bool ok = used && (r < R) && (c >= slo[q]) && (c < shi[q]);
                ev[q][f] = rl_ld<COH>(spq, ok ? (size_t)r * srs[q] + (size_t)(c * scs[q] + sof[q]) : (size_t)0);
            }
        }
    };
    slot(std::integral_constant<int, 0>()); slot(std::integral_constant<int, 1>()); slot(std::integral_constant<int, 2>());
    slot(std::integral_constant<int, 3>()); slot(std::integral_constant<int, 4>());

    // optimizer in the epilogue (EPI_DWA launches, tasks flagged FLAG_ADAM): Adam scalars of the group, and (column tile 0 only) the bias
    // element this thread will update
    AdamScal adsc;
    bool fuse_opt = false, bias_opt = false;
    float bpv = 0.f, bmv = 0.f, bvv = 0.f, btv = 0.f;
    if constexpr (EPI_K == EPI_DWA) {
        fuse_opt = (flags & FLAG_ADAM) && t.ad_p;
        if (fuse_opt) adsc = t.ad_grp->sc;
        bias_opt = want_bias && fuse_opt && t.ad_pb && threadIdx.x < 16 && r0 + (int)threadIdx.x < R;
        if (bias_opt) {
            const int o = r0 + threadIdx.x;
            bpv = t.ad_pb[o]; bmv = t.ad_mb[o]; bvv = t.ad_vb[o];
            if (t.ad_tb) btv = t.ad_tb[o];
        }
    }
    // (the policy's backward needs log(alpha): fetched here, under the operand stream, when the epilogue kind is known at compile time)
    double log_alpha_pre = 0.0;
    if constexpr (EPI_K == EPI_DX_POLICYBWD) log_alpha_pre = t.dptr[0];
    asm volatile("" ::: "memory");      // the pin (see above)
    TIMB(6);

    // wave w owns the 16-wide inner chunks w, w+4, w+8, ...
    if constexpr (PRE) {
        PreSrc ps;
        if constexpr (FAST) { ps.X = fo->X; ps.ldx = fo->ldx; ps.Wt = fo->Wt; ps.ldw = fo->ldw; ps.K1 = fo->K1; ps.M = fo->M; ps.ldm = fo->ldm; }
        else { ps.X = t.x0; ps.ldx = t.ldx0; ps.Wt = t.x1; ps.ldw = t.ldx1; ps.K1 = n0; ps.M = t.x2; ps.ldm = t.ldaux2; }
        ps.out = t.y0; ps.ldo = t.ldout2;
        // who stores the recomputed layer: one 16-wide chunk per column tile when there are enough tiles, else column tile 0 everything
        const bool spread = tiles_c * 16 >= K;
        const bool store = ps.out && (spread || tc == 0);
        ps.st_lo = spread ? tc * 16 : 0; ps.st_hi = spread ? tc * 16 + 16 : K;
        const bool pre_elu = (flags & FLAG_PRE_ELU) != 0;
        const float* xs = nullptr;
        if constexpr (MSE) {
            // ---- first phase: X = dmse( M Wt^T + bias ; targets ) for this tile's 16 rows, K1 <= 32 columns (FLAG_PRE_MSE) ----
            // P[16 x 32] over the inner dimension K, split over the four waves as in the main loop (16-byte loads: the launcher checks alignment)
            f32x4 pacc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
            float dummy = 0.f;
            // this thread's two elements of [s_hat | r_hat]: bias and target fetched NOW, under the product (after the reduction they were two
            // dependent L2 round trips per fragment: 4 us of this launch in tools/exp/gemm_timeline.py)
            float pbias_[2], ptgt_[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int c = min(16 * f + (ol & 15), ps.K1 - 1), rc = min(r, R - 1);
                pbias_[f] = t.bias[c];
                const float* const tp = c < t.pad_mse ? t.tgs + (size_t)rc * t.ldtgs + c : t.tgr + rc;
                ptgt_[f] = *tp;
            }
            // this wave's FIRST block of the product (four 16-deep chunks; what lies beyond K is clamped and masked, so any K is handled) ...
            {
                const int k0 = w * 16 + 4 * kq;
                if constexpr (!FAST) {          // (FAST: these loads and the prefetch below went out at the top of the tile)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    load_raw<LD_ROW, true, COH>(ps.M, ps.ldm, r0, R, i, k0 + 64 * u, K, a1[u]);
#pragma unroll
                    for (int f = 0; f < 2; ++f) load_raw<LD_ROW, true>(ps.Wt, ps.ldw, 16 * f, ps.K1, i, k0 + 64 * u, K, b1[u][f]);
                }
                __builtin_amdgcn_sched_barrier(0);      // (the first phase's operands FIRST: the load counter retires in order)
                }
                // ... and BEHIND them the SECOND phase's global operands of the same block (weights of both products, the ReLU mask): none of
                // them depends on the first phase, and a launch's first touch of anything costs a trip beyond the XCD's L2 (~1 us) -- one such
                // trip instead of two.  Unconditional (clamped addresses); used below only if the block is a full one (K >= 256).  Issued AFTER
                // the first phase's loads: the load counter retires in order, so the product below waits for its own operands only.
                if constexpr (!FAST) pre_load<LB, NF, VB, 4, NJ, FW, COH, true>(ps, pB, ldb, r0, R, c0, Cn, i, kq, w * 16, K, mse_pl);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    mask_frag(r0, R, i, k0 + 64 * u, K, a1[u]);
#pragma unroll
                    for (int f = 0; f < 2; ++f) mask_frag(16 * f, ps.K1, i, k0 + 64 * u, K, b1[u][f]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                        for (int f = 0; f < 2; ++f) pacc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u][s4], b1[u][f][s4], pacc[f], 0, 0, 0);
            }
            for (int kb = w * 16 + 256; kb < K; kb += 256) {
                const int k0 = kb + 4 * kq;
                const int nu = (K - kb + 63) >> 6;
                if (nu >= 4) mac_group<LD_ROW, LD_ROW, 2, true, true, 4, COH>(ps.M, ps.ldm, ps.Wt, ps.ldw, r0, R, 0, ps.K1, i, k0, K, pacc, dummy, false);
                else if (nu == 1) mac_group<LD_ROW, LD_ROW, 2, true, true, 1, COH>(ps.M, ps.ldm, ps.Wt, ps.ldw, r0, R, 0, ps.K1, i, k0, K, pacc, dummy, false);
                else if (nu == 2) mac_group<LD_ROW, LD_ROW, 2, true, true, 2, COH>(ps.M, ps.ldm, ps.Wt, ps.ldw, r0, R, 0, ps.K1, i, k0, K, pacc, dummy, false);
                else mac_group<LD_ROW, LD_ROW, 2, true, true, 3, COH>(ps.M, ps.ldm, ps.Wt, ps.ldw, r0, R, 0, ps.K1, i, k0, K, pacc, dummy, false);
            }
#ifdef RL_TIMING_MSE
            TIMB(7);
#endif
            // fixed-order reduction over the waves, both fragments in ONE round through a patch of their own
            __shared__ float XS[16 * RL_XS_LD];
            __shared__ float red1[4][2][4][64];
            float es = 0.f, er = 0.f;
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int q = 0; q < 4; ++q) red1[w][f][q][lane] = pacc[f][q];
            __syncthreads();
#ifdef RL_TIMING_MSE
            TIMB(1);
#endif
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int c = 16 * f + (ol & 15);
                const float v = ((red1[0][f][oreg][ol] + red1[1][f][oreg][ol]) + red1[2][f][oreg][ol]) + red1[3][f][oreg][ol];
                float g = 0.f;
                if (r < R && c < ps.K1) {
                    const float d = (v + pbias_[f]) - ptgt_[f];
                    if (c < t.pad_mse) { es += d * d; g = d * t.s0; }
                    else { er += d * d; g = d * t.s1; }
                    if (tc == 0) const_cast<float*>(ps.X)[(size_t)r * ps.ldx + c] = g;          // the weight-gradient pass reads X
                }
                XS[((ol >> 4) * 4 + oreg) * RL_XS_LD + c] = g;
            }
            es = wave_sum(es); er = wave_sum(er);
            if (lane == 0) { red[0][0][0][w] = es; red[0][0][1][w] = er; }       // (the main product's patch is still free)
            __syncthreads();                                                      // XS and the four partial sums are visible
            if (tc == 0 && threadIdx.x == 0) {
                t.mse_part[2 * tr] = ((red[0][0][0][0] + red[0][0][0][1]) + red[0][0][0][2]) + red[0][0][0][3];
                t.mse_part[2 * tr + 1] = ((red[0][0][1][0] + red[0][0][1][1]) + red[0][0][1][2]) + red[0][0][1][3];
            }
            xs = XS;
            TIMB(5);                // (MSE: the first phase is done)
        }
        for (int kb = w * 16; kb < K; kb += 256) {
            const int nu = (K - kb + 63) >> 6;
            if ((MSE || FAST) && nu >= 4 && kb == w * 16) mac_group_pre<LB, NF, VB, 4, NJ, FW, COH, MSE, true>(ps, pB, ldb, r0, R, c0, Cn, i, kq, kb, K, store, pre_elu, acc, xs, &mse_pl);
            else if (nu >= 4) mac_group_pre<LB, NF, VB, 4, NJ, FW, COH, MSE>(ps, pB, ldb, r0, R, c0, Cn, i, kq, kb, K, store, pre_elu, acc, xs);
            else if (nu == 1) mac_group_pre<LB, NF, VB, 1, NJ, FW, COH, MSE>(ps, pB, ldb, r0, R, c0, Cn, i, kq, kb, K, store, pre_elu, acc, xs);
            else if (nu == 2) mac_group_pre<LB, NF, VB, 2, NJ, FW, COH, MSE>(ps, pB, ldb, r0, R, c0, Cn, i, kq, kb, K, store, pre_elu, acc, xs);
            else mac_group_pre<LB, NF, VB, 3, NJ, FW, COH, MSE>(ps, pB, ldb, r0, R, c0, Cn, i, kq, kb, K, store, pre_elu, acc, xs);
        }
    } else if constexpr (FAST) {
        // first block from the registers filled above, then whole 256-deep blocks (FAST = 4: K % 256 == 0; FAST = 1: K <= 64 -- the launcher checks)
        const int k0 = w * 16 + 4 * kq;
#pragma unroll
        for (int u = 0; u < FAST; ++u) {
            mask_frag(r0, R, i, k0 + 64 * u, K, fa[u]);
#pragma unroll
            for (int f = 0; f < NF; ++f) mask_frag(c0 + 16 * f, Cn, i, k0 + 64 * u, K, fb[u][f]);
        }
#pragma unroll
        for (int u = 0; u < FAST; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int f = 0; f < NF; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][s], fb[u][f][s], acc[f], 0, 0, 0);
        if constexpr (FAST == 4)
            for (int kb = w * 16 + 256; kb < K; kb += 256)
                mac_group<LA, LB, NF, VA, VB, 4, false, false>(pArow, lda, pB, ldb, r0, R, c0, Cn, i, kb + 4 * kq, K, acc, asum, false);
    } else
    for (int kb = w * 16; kb < K; kb += 256) {
        const int k0 = kb + 4 * kq;
        const int nu = (K - kb + 63) >> 6;           // chunks of this group that touch the matrix (uniform per wave)
        if (nu >= 4) mac_group<LA, LB, NF, VA, VB, 4, COH, GATHER>(pArow, lda, pB, ldb, r0, R, c0, Cn, i, k0, K, acc, asum, want_bias);
        else if (nu == 1) mac_group<LA, LB, NF, VA, VB, 1, COH, GATHER>(pArow, lda, pB, ldb, r0, R, c0, Cn, i, k0, K, acc, asum, want_bias);
        else if (nu == 2) mac_group<LA, LB, NF, VA, VB, 2, COH, GATHER>(pArow, lda, pB, ldb, r0, R, c0, Cn, i, k0, K, acc, asum, want_bias);
        else mac_group<LA, LB, NF, VA, VB, 3, COH, GATHER>(pArow, lda, pB, ldb, r0, R, c0, Cn, i, k0, K, acc, asum, want_bias);
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
            const float g = (float)exp(EPI_K == EPI_DX_POLICYBWD ? log_alpha_pre : t.dptr[0]) * t.s0;            // dL/dlogpi = alpha / B
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
            if (fuse_opt) {      // e0 / cold2 / cold3 / cold4 = parameter, exp_avg, exp_avg_sq, Polyak target (prefetched with the other slots)
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

// rowprog: row-block-resident layer chains (gfx950).  See rowprog.h for the model.
//
// One workgroup = 16 batch rows = one MFMA row tile, 8 waves.  A layer [16, K] x [K, N] is cut into 64-column chunks; chunk c is
// computed by the wave pair (c & 3, half 0 / half 1): the two waves sit on the same SIMD (waves w and w + 4 share one) and split the
// inner dimension, because one wave alone issues a v_mfma_f32_16x16x4_f32 only every 52 cycles against 32-34 for two.
// Activations are read from LDS as 16-byte A fragments (lane l: row l & 15, four consecutive inner indices 4 (l >> 4) ..), weights
// go global -> VGPR as 16-byte B fragments, three 16-deep groups in flight per wave:
//   forward  (W stored [N][K]):  lane (q, j) loads W[c0 + 16 u + j][k0 + 4 q .. + 3], u = 0..3  -> acc[u] = tile u of the chunk
//   dX       (W stored [K][N]):  lane (q, j) loads W[k0 + 4 q + m][c0 + 4 j .. + 3], m = 0..3   -> acc[s] = columns c0 + 4 j + s
// (the dX form feeds register s of every lane to MFMA s, so one accumulator holds a STRIDED set of 16 columns and a lane ends up with
// four consecutive columns of a row: 16-byte stores).  Both are exact fp32, k-ordered per wave, halves added in fixed order.
#include "common.h"
#include "kparams.h"
#include "rowprog.h"

#define RP_MAX_OPS 40
#define RP_OP_WORDS ((int)(sizeof(RpOp) / 4))

extern __shared__ __attribute__((aligned(16))) float rp_buf[];

static_assert(sizeof(float) * 7 * 16 * 64 + RP_MAX_OPS * sizeof(RpOp) + 64 + RP_LDS_DYN_MAX <= 160 * 1024, "static + dynamic LDS must fit the CU's 160 KB");
struct RpShared {
    float red[7][16][64];          // accumulators of the inner-dimension parts 1.. of the column groups: [(part - 1) * groups + group]
    int ops[RP_MAX_OPS * RP_OP_WORDS];
    float part[16];
};

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// The op records are staged through LDS, so the compiler no longer knows that their pointers are GLOBAL ones and would emit flat_load /
// flat_store (which count on vmcnt AND lgkmcnt and complete out of order: every use then drains both counters).  All global accesses of
// this file go through explicitly address-space-1 pointers.
#define RP_GAS __attribute__((address_space(1)))
#define RP_STAMP(tim, slot, val) (((RP_GAS unsigned long long*)(tim))[(size_t)blockIdx.x * 512 + (slot)] = (val))
typedef const RP_GAS float* gcf_t;
typedef RP_GAS float* gf_t;
typedef const RP_GAS f32x4* gcf4_t;
typedef RP_GAS f32x4* gf4_t;
__device__ __forceinline__ gcf_t G(const float* p) { return (gcf_t)p; }
__device__ __forceinline__ gf_t G(float* p) { return (gf_t)p; }

// the op record as scalars: ONE LDS read per lane (lane e holds word e), then a readlane per word
static_assert(sizeof(RpOp) / 4 <= 64, "an op record must fit one wave-wide LDS read");
__device__ __forceinline__ RpOp rp_fetch(const int* ops, int oi) {
    RpOp op;
    int* w = reinterpret_cast<int*>(&op);
    const int mine = ops[oi * RP_OP_WORDS + min((int)(threadIdx.x & 63), RP_OP_WORDS - 1)];
#pragma unroll
    for (int e = 0; e < RP_OP_WORDS; ++e) w[e] = __builtin_amdgcn_readlane(mine, e);
    return op;
}

__device__ __forceinline__ float rp_act(float x, int act) {
    switch (act) {
    case ACT_RELU: return fmaxf(x, 0.f);
    case ACT_ELU: return elu_f(x);
    case ACT_TANH: return tanhf(x);
    case ACT_SIN: return sinf(x);
    default: return x;
    }
}
// derivative of the activation expressed through its OUTPUT y (ReLU: y > 0; ELU: in-place form of the reference, utils/util.py:89-91)
__device__ __forceinline__ float rp_dact(float y, int act) {
    switch (act) {
    case ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case ACT_ELU: return elu_grad_from_out(y);
    case ACT_TANH: return 1.f - y * y;
    default: return 1.f;
    }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt: every barrier behind an epilogue would then wait
// for that epilogue's global stores to be acknowledged (~3 us per layer, measured).  Global data handed to another workgroup is
// drained explicitly (RP_SIGNAL); global data this workgroup re-reads itself (masks, RP_LOAD) is ordered by rp_sync_global().
__device__ __forceinline__ void rp_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void rp_sync_global() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- B-operand fragment loads --------------------------------------------------------------------------------------------------------
// forward form, one 32-deep group: lane (q, j) takes EIGHT consecutive inner indices 32 g + 8 q .. + 7 of row c0 + 16 u + j (two 16-byte
// loads = 32 contiguous bytes per lane, so the four lanes of a row cover one whole 128-byte line: with 16-byte pieces per 16-deep group
// half of every fetched line was evicted before its second half was asked for, and the per-CU L2 -> L1 path is the bound here)
template <bool VEC>
__device__ __forceinline__ void rp_load_fwd(gcf_t W, int ldw, int K, int N, int c0, int j, int q, int g, f32x4 (&b)[8]) {
    const int k0 = 32 * g + 8 * q;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        gcf_t p = W + (size_t)min(c0 + 16 * u + j, N - 1) * ldw;
        if (VEC) {
            b[2 * u] = *(gcf4_t)(p + min(k0, K - 4));
            b[2 * u + 1] = *(gcf4_t)(p + min(k0 + 4, K - 4));
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) { b[2 * u][s] = p[min(k0 + s, K - 1)]; b[2 * u + 1][s] = p[min(k0 + 4 + s, K - 1)]; }
        }
    }
}
// dX form, one 16-deep group: lane (q, j) takes columns c0 + 4 j .. + 3 of rows 16 g + 4 q + m, m = 0..3 (256 contiguous bytes per row)
template <bool VEC>
__device__ __forceinline__ void rp_load_dx(gcf_t W, int ldw, int K, int N, int c0, int j, int q, int g, f32x4 (&b)[4]) {
    const int k0 = 16 * g + 4 * q, col = c0 + 4 * j;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        gcf_t p = W + (size_t)min(k0 + m, K - 1) * ldw;
        if (VEC) b[m] = *(gcf4_t)(p + min(col, N - 4));
        else {
#pragma unroll
            for (int s = 0; s < 4; ++s) b[m][s] = p[min(col + s, N - 1)];
        }
    }
}
__device__ __forceinline__ void rp_mac_fwd(const f32x4& a0, const f32x4& a1, const f32x4 (&b)[8], f32x4 (&acc)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], b[2 * u][s], acc[u], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], b[2 * u + 1][s], acc[u], 0, 0, 0);
}
__device__ __forceinline__ void rp_mac_dx(const f32x4& a, const f32x4 (&b)[4], f32x4 (&acc)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[m][s], acc[s], 0, 0, 0);
}

// One layer.  The LDS source must be zero beyond column K up to the next multiple of 32 (RP_LOAD, RP_VAE_MID and the epilogue below
// keep it so).  Output element of accumulator u, register reg of lane (q, j): row 4 q + reg, column c0 + 16 u + j (forward) or
// c0 + 4 j + u (dX).
template <bool COL, bool VEC>
__device__ __forceinline__ void rp_gemm(const RpOp& op, int r0, int B, RpShared& sh, unsigned long long* tim, int oi) {
    const int lane = threadIdx.x & 63, w = rfl(threadIdx.x >> 6);      // wave index as a SCALAR: loop bounds and roles are uniform
    const int j = lane & 15, q = lane >> 4;
    const int K = op.K, N = op.N, ldw = op.ldw;
    gcf_t W = G(op.W);
    gcf_t bias = G(op.bias), gaux = G(op.gaux);
    gf_t gout = G(op.gout);
    const int nchunks = (N + 63) >> 6;
    // the 8 waves: cpp column chunks per pass x P parts of the inner dimension (narrow layers split the inner dimension further instead
    // of leaving wave pairs idle: N <= 64 -> 1 x 8, N <= 128 -> 2 x 4, else 4 x 2)
    const int cpp = nchunks >= 3 ? 4 : nchunks, P = 8 / cpp;
    const int cg = w % cpp, kh = w / cpp;
    const int ng = COL ? (K + 15) >> 4 : (K + 31) >> 5;        // groups of the inner dimension (16-deep dX, 32-deep forward)
    const int g_beg = (ng * kh) / P, g_end = (ng * (kh + 1)) / P;
    const float* a_base = rp_buf + op.src + j * op.lds + (COL ? 4 : 8) * q;
    for (int cb = 0; cb < nchunks; cb += cpp) {
        const int c = cb + cg;
        const bool live = c < nchunks;
        const int c0 = c * 64;
        f32x4 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // epilogue operands of this thread's 16 outputs (half 0 only), fetched before the inner loop so that their latency hides
        // behind it (and so that no load sits between the epilogue's stores: the compiler cannot move a load above a store that may alias)
        // (left alone hipcc sinks these loads below the loop and the barrier, next to their first use -- 1.7 us of exposed latency per layer,
        // measured -- hence the memory-clobbering asm behind them.  They are PLAIN loads on purpose: an asm load whose result is claimed by a
        // later s_waitcnt leaves the compiler free to copy or reuse the destination register while the load is in flight, and a change of
        // register allocation turned exactly that into run-to-run differences of the loss.)
        float mk[4][4], bv[4];
        const bool epi = kh == 0 && live;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bv[u] = 0.f;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) mk[u][reg] = 0.f;
        }
        if (epi && (op.flags & RPF_MASK_GLOBAL)) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                gcf_t p = gaux + (size_t)min(r0 + 4 * q + reg, B - 1) * op.ldgaux;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    mk[u][reg] = p[min(COL ? c0 + 4 * j + u : c0 + 16 * u + j, N - 1)];
                }
            }
        }
        if (epi && (op.flags & RPF_BIAS)) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                bv[u] = bias[min(COL ? c0 + 4 * j + u : c0 + 16 * u + j, N - 1)];
            }
        }
        // pin the loads above the inner loop: nothing may move across an asm that clobbers memory
        asm volatile("" ::: "memory");
        if (live && g_beg < g_end) {
            const int last = g_end - 1;
            if (!COL) {
                // two named register sets of one 32-deep group each, the next group in flight while this one is multiplied
                f32x4 bA[8], bB[8];
                rp_load_fwd<VEC>(W, ldw, K, N, c0, j, q, g_beg, bA);
                for (int g = g_beg; g < g_end; g += 2) {
                    rp_load_fwd<VEC>(W, ldw, K, N, c0, j, q, min(g + 1, last), bB);
                    {
                        const f32x4 a0 = *reinterpret_cast<const f32x4*>(a_base + 32 * g);
                        const f32x4 a1 = *reinterpret_cast<const f32x4*>(a_base + 32 * g + 4);
                        rp_mac_fwd(a0, a1, bA, acc);
                    }
                    rp_load_fwd<VEC>(W, ldw, K, N, c0, j, q, min(g + 2, last), bA);
                    if (g + 1 < g_end) {
                        const f32x4 a0 = *reinterpret_cast<const f32x4*>(a_base + 32 * (g + 1));
                        const f32x4 a1 = *reinterpret_cast<const f32x4*>(a_base + 32 * (g + 1) + 4);
                        rp_mac_fwd(a0, a1, bB, acc);
                    }
                }
            } else if (VEC) {
                // Three named register sets, two 16-deep groups in flight behind the one being multiplied.  The loads are volatile asm and
                // the waits explicit: with plain loads the compiler's wait-count pass puts vmcnt(0) at the loop head (it merges the
                // pre-header's and the back edge's pending-load sets conservatively), which drains the prefetch every iteration.  Every
                // iteration issues the same number of loads (clamped addresses past the end), so the counts below are exact:
                // at each wait the two younger sets (8 loads) may stay outstanding.
                f32x4 bA[4], bB[4], bC[4];
#define RP_LD4(set, grp) { const int k0_ = 16 * (grp) + 4 * q; _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_) { \
                    gcf_t p_ = W + (size_t)min(k0_ + m_, K - 1) * ldw + min(c0 + 4 * j, N - 4); \
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(set[m_]) : "v"(p_) : "memory"); } }
#define RP_WAIT8(set) asm volatile("s_waitcnt vmcnt(8)" : "+v"(set[0]), "+v"(set[1]), "+v"(set[2]), "+v"(set[3]) :: "memory")
                RP_LD4(bA, g_beg);
                RP_LD4(bB, min(g_beg + 1, last));
                for (int g = g_beg; g < g_end; g += 3) {
                    RP_LD4(bC, min(g + 2, last));
                    RP_WAIT8(bA);
                    rp_mac_dx(*reinterpret_cast<const f32x4*>(a_base + 16 * g), bA, acc);
                    RP_LD4(bA, min(g + 3, last));
                    RP_WAIT8(bB);
                    if (g + 1 < g_end) rp_mac_dx(*reinterpret_cast<const f32x4*>(a_base + 16 * (g + 1)), bB, acc);
                    RP_LD4(bB, min(g + 4, last));
                    RP_WAIT8(bC);
                    if (g + 2 < g_end) rp_mac_dx(*reinterpret_cast<const f32x4*>(a_base + 16 * (g + 2)), bC, acc);
                }
                // the clamped prefetches of the last iterations are still in flight and the compiler does not know it: their destination
                // registers are dead to it from here on, so they must have landed before anything else is allocated there
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef RP_LD4
#undef RP_WAIT8
            } else {
                f32x4 bA[4], bB[4], bC[4];
                rp_load_dx<VEC>(W, ldw, K, N, c0, j, q, g_beg, bA);
                rp_load_dx<VEC>(W, ldw, K, N, c0, j, q, min(g_beg + 1, last), bB);
                for (int g = g_beg; g < g_end; g += 3) {
                    rp_load_dx<VEC>(W, ldw, K, N, c0, j, q, min(g + 2, last), bC);
                    rp_mac_dx(*reinterpret_cast<const f32x4*>(a_base + 16 * g), bA, acc);
                    rp_load_dx<VEC>(W, ldw, K, N, c0, j, q, min(g + 3, last), bA);
                    if (g + 1 < g_end) rp_mac_dx(*reinterpret_cast<const f32x4*>(a_base + 16 * (g + 1)), bB, acc);
                    rp_load_dx<VEC>(W, ldw, K, N, c0, j, q, min(g + 4, last), bB);
                    if (g + 2 < g_end) rp_mac_dx(*reinterpret_cast<const f32x4*>(a_base + 16 * (g + 2)), bC, acc);
                }
            }
        }
        asm volatile("" ::: "memory");
        if (tim && threadIdx.x == 0) RP_STAMP(tim, 128 + oi, wall_clock64());
        if (kh >= 1 && live) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) sh.red[(kh - 1) * cpp + cg][u * 4 + reg][lane] = acc[u][reg];
        }
        rp_barrier();
        if (tim && threadIdx.x == 0) RP_STAMP(tim, 192 + oi, wall_clock64());
        if (epi) {
            // all 16 values first (LDS reads only), then the LDS stores, then the global stores
            float v[4][4];
            if (op.flags & RPF_MASK_LDS) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                    for (int u = 0; u < 4; ++u) mk[u][reg] = rp_buf[op.src2 + (4 * q + reg) * op.lds2 + min(COL ? c0 + 4 * j + u : c0 + 16 * u + j, N - 1)];
            }
            const bool masked = op.flags & (RPF_MASK_LDS | RPF_MASK_GLOBAL);
            const bool biased = op.flags & RPF_BIAS;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                for (int u = 0; u < 4; ++u) v[reg][u] = acc[u][reg];
            // the other parts' partial sums in fixed order: part-major, so that the 16 LDS reads of a part are in flight together (an
            // element-major loop with a run-time part count serialises 16 x (P - 1) dependent LDS round trips: 3.7 us per layer at P = 8)
            for (int part = 1; part < P; ++part) {
                float t[4][4];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                    for (int u = 0; u < 4; ++u) t[reg][u] = sh.red[(part - 1) * cpp + cg][u * 4 + reg][lane];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[reg][u] += t[reg][u];
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                for (int u = 0; u < 4; ++u) v[reg][u] += biased ? bv[u] : 0.f;
            // ONE uniform branch per layer on (masked, activation), the 16 elements inside it (a switch per element is a ladder of ~80
            // scalar branches per layer)
#define RP_EACH(expr) _Pragma("unroll") for (int reg = 0; reg < 4; ++reg) _Pragma("unroll") for (int u = 0; u < 4; ++u) { const float x = v[reg][u]; const float m = mk[u][reg]; (void)m; v[reg][u] = (expr); }
            if (masked) {
                if (op.act == ACT_RELU) { RP_EACH(m > 0.f ? x : 0.f) }
                else if (op.act == ACT_ELU) { RP_EACH(x * elu_grad_from_out(m)) }
                else { RP_EACH(x * rp_dact(m, op.act)) }
            } else {
                if (op.act == ACT_RELU) { RP_EACH(fmaxf(x, 0.f)) }
                else if (op.act == ACT_ELU) { RP_EACH(elu_f(x)) }
                else if (op.act != ACT_NONE) { RP_EACH(rp_act(x, op.act)) }
            }
#undef RP_EACH
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int col = COL ? c0 + 4 * j + u : c0 + 16 * u + j; if (col >= N) v[reg][u] = 0.f; }
            if (tim && threadIdx.x == 0) RP_STAMP(tim, 256 + oi, wall_clock64());
            if (op.dst >= 0) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    float* d = rp_buf + op.dst + (4 * q + reg) * op.ldd;
                    if (COL) {
                        const int col = c0 + 4 * j;
                        if (col + 3 < op.wpad) *reinterpret_cast<f32x4*>(d + col) = (f32x4){v[reg][0], v[reg][1], v[reg][2], v[reg][3]};
                        else {
#pragma unroll
                            for (int u = 0; u < 4; ++u) if (col + u < op.wpad) d[col + u] = v[reg][u];
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int col = c0 + 16 * u + j; if (col < op.wpad) d[col] = v[reg][u]; }
                    }
                }
            }
            if (tim && threadIdx.x == 0) RP_STAMP(tim, 320 + oi, wall_clock64());
            if (op.gout) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int row = 4 * q + reg;
                    if (r0 + row >= B) continue;
                    gf_t g = gout + (size_t)(r0 + row) * op.ldg;
                    if (COL) {
                        const int col = c0 + 4 * j;
                        if (VEC && !(op.ldg & 3) && col + 3 < N) *(gf4_t)(g + col) = (f32x4){v[reg][0], v[reg][1], v[reg][2], v[reg][3]};
                        else {
#pragma unroll
                            for (int u = 0; u < 4; ++u) if (col + u < N) g[col + u] = v[reg][u];
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int col = c0 + 16 * u + j; if (col < N) g[col] = v[reg][u]; }
                    }
                }
            }
            if (tim && threadIdx.x == 0) RP_STAMP(tim, 384 + oi, wall_clock64());
        }
        rp_barrier();
        if (tim && threadIdx.x == 0) RP_STAMP(tim, 448 + oi, wall_clock64());
    }
}

// sum over the workgroup's 512 threads; result valid in thread 0
__device__ __forceinline__ float rp_block_sum(float v, float* sh8) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    rp_barrier();
    if ((threadIdx.x & 63) == 0) sh8[w] = v;
    rp_barrier();
    float r = 0.f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < RP_THREADS / 64; ++q) r += sh8[q];
    }
    return r;
}

// Diagnostic stamps (tools/exp/rp_timeline.py): with a buffer registered, thread 0 of every workgroup records the 100 MHz wall clock at
// the start of every op and at exit: slot [block][op].  Null in normal operation (one scalar load + branch per op).
__device__ unsigned long long* g_rp_tim = nullptr;
extern "C" int rl_rowprog_timing(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_rp_tim), &buf, sizeof(buf));
}

__global__ __launch_bounds__(RP_THREADS) void rowprog_kernel(RpLaunch L) {
    __shared__ RpShared sh;
    unsigned long long* const tim = g_rp_tim;       // (stamps are stored through RP_STAMP: a flat store would count on lgkmcnt and stall the next LDS wait)
    if (!L.low_prio) __builtin_amdgcn_s_setprio(3);
    const int bid = blockIdx.x;
    int pi = 0;
#pragma unroll
    for (int p = 1; p < RP_MAX_PROGS; ++p) if (p < L.nprog && bid >= L.prog[p].block_base) pi = p;
    const RpProg pr = L.prog[pi];
    const int cs = pr.csize > 1 ? pr.csize : 1;
    const int local = bid - pr.block_base;
    const int rb = local / cs, member = local - rb * cs;       // cluster programs: csize consecutive workgroups share a row block
    const int r0 = rb * RP_ROWS;
    const unsigned epoch = L.epoch ? (unsigned)*(const RP_GAS int*)L.epoch : 0u;
    int rp_dead = 0;            // a wait of this thread timed out: stop waiting (the launch drains; the error word says the results are void)
    const int B = L.B;
    const int nrb = (B + RP_ROWS - 1) / RP_ROWS;
    const int nops = pr.op_end - pr.op_begin;
    {
        const int* __restrict__ src = reinterpret_cast<const int*>(L.ops + pr.op_begin);
        const int nw = nops * RP_OP_WORDS;
        for (int e = threadIdx.x; e < nw; e += RP_THREADS) sh.ops[e] = src[e];
    }
    __syncthreads();
    const int tid = threadIdx.x;
    for (int oi = 0; oi < nops; ++oi) {
        RpOp op = rp_fetch(sh.ops, oi);
        if (member) {                           // this member's column slice of the layer: shifted operand / result pointers (uniform scalars)
            // (null stays null: an absent result / operand must stay absent)
            if (op.W) op.W += member * op.m_w;
            if (op.bias) op.bias += member * op.m_b;
            if (op.gout) op.gout += member * op.m_g;
            if (op.gout2) op.gout2 += member * op.m_g2;
            if (op.gaux) op.gaux += member * op.m_gaux;
            if (op.gin) op.gin += member * op.m_gin;
            if (op.dst >= 0) op.dst += member * op.m_dst;
            op.src2 += member * op.m_s2; op.src += member * op.m_src; op.dst2 += member * op.m_dst2;
        }
        if (tim && threadIdx.x == 0) { RP_STAMP(tim, oi, wall_clock64()); RP_STAMP(tim, 64 + oi, clock64()); }
        if (op.kind == RP_LOAD || (op.kind == RP_GEMM && (op.flags & RPF_MASK_GLOBAL))) rp_sync_global();
        switch (op.kind) {
        case RP_LOAD: {
            // rows r0 .. r0+15 of gin[., K] -> LDS dst, zero beyond K up to wpad columns; rows beyond the batch repeat the last one.
            // 32 threads per row; 16-byte pieces when the source rows allow it
            const int wp = op.wpad, row = tid >> 5, l = tid & 31;
            gcf_t srow = G(op.gin) + (size_t)min(r0 + row, B - 1) * op.ldgin;
            float* drow = rp_buf + op.dst + row * op.ldd;
            if (!(op.ldgin & 3) && !(op.K & 3) && !((uintptr_t)op.gin & 15)) {
                for (int c = 4 * l; c < wp; c += 128)
                    *reinterpret_cast<f32x4*>(drow + c) = c < op.K ? *(gcf4_t)(srow + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
            } else {
                for (int c = l; c < wp; c += 32) drow[c] = c < op.K ? srow[c] : 0.f;
            }
        } break;
        case RP_GEMM: {
            const bool col = op.flags & RPF_COL;
            const bool vec = !(op.ldw & 3) && !((uintptr_t)op.W & 15) && (col ? !(op.N & 3) : !(op.K & 3));
            if (col) { if (vec) rp_gemm<true, true>(op, r0, B, sh, tim, oi); else rp_gemm<true, false>(op, r0, B, sh, tim, oi); }
            else { if (vec) rp_gemm<false, true>(op, r0, B, sh, tim, oi); else rp_gemm<false, false>(op, r0, B, sh, tim, oi); }
        } break;
        case RP_VAE_MID: {
            // src = encoder heads [16, 2F] (mean | log_std), src2 = f heads (read only).  src <- (dKL/dmean1 | dKL/dlog_std1) in place,
            // dst <- z (zero up to its padded width: it is the decoder's K operand), dst2 <- eps * sigma1 * clamp-mask;
            // gout <- z, gout2 <- (dKL/dmean2 | dKL/dlog_std2); KL partial -> part[rb]
            // cluster member: N = the slice's width, K = F (offset of the log-std halves, row stride of the noise), all pointers pre-shifted
            const int F = op.N, Fp = op.wpad, FO = op.K > 0 ? op.K : op.N;
            gcf_t eps = G(L.dyn[op.dyn]) + member * op.m_gin;
            float k = 0.f;
            {
                const int row = tid >> 5, l = tid & 31;
                const bool rok = r0 + row < B;
                float* eh = rp_buf + op.src + row * op.lds;
                const float* fh = rp_buf + op.src2 + row * op.lds2;
                float* zr = rp_buf + op.dst + row * op.ldd;
                float* ezr = rp_buf + op.dst2 + row * op.ldd2;
                gcf_t er = eps + (size_t)min(r0 + row, B - 1) * FO;
                gf_t gz = G(op.gout) + (size_t)min(r0 + row, B - 1) * op.ldg;
                gf_t g2 = G(op.gout2) + (size_t)min(r0 + row, B - 1) * op.ldg2;
                const float sc = op.s0;                                  // 1 / (B_global * F)
                for (int j0 = l; j0 < Fp; j0 += 128) {
                    // the noise of four elements first: no load behind a (possibly aliasing) store
                    float ev[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) ev[i] = er[min(j0 + 32 * i, F - 1)];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int jj = j0 + 32 * i;
                        if (jj >= Fp) break;
                        if (jj >= F) { zr[jj] = 0.f; continue; }
                        const float m1 = eh[jj], l1r = eh[FO + jj], m2 = fh[jj], l2r = fh[FO + jj];
                        const float l1 = clamp_lstd(l1r), l2 = clamp_lstd(l2r);
                        const float es = ev[i] * expf(l1);
                        const float z = m1 + es;
                        const float v1 = expf(2.f * l1), iv2 = expf(-2.f * l2), d = m1 - m2;
                        const float kk = l2 - l1 + 0.5f * (v1 + d * d) * iv2 - 0.5f;
                        const float dm1 = d * iv2 * sc;
                        eh[jj] = dm1;
                        eh[FO + jj] = (v1 * iv2 - 1.f) * sc * lstd_mask(l1r);
                        zr[jj] = z;
                        ezr[jj] = es * lstd_mask(l1r);
                        const float gl2 = (1.f - (v1 + d * d) * iv2) * sc * lstd_mask(l2r);
                        if (op.flags & RPF_FH_INPLACE) { float* fw = const_cast<float*>(fh); fw[jj] = -dm1; fw[FO + jj] = gl2; }
                        if (rok) {
                            k += kk;
                            gz[jj] = z;
                            g2[jj] = -dm1;
                            g2[FO + jj] = gl2;
                        }
                    }
                }
            }
            const float s = rp_block_sum(k, sh.part);
            if (tid == 0) {
                G(op.part)[local] = s;
                if ((op.flags & RPF_BUMP) && local == 0 && op.step) bump_group(op.step);
            }
        } break;
        case RP_MSE: {
            // src = decoder heads [16, n0 + 1] (s_hat | r_hat): in place <- d(0.5 mse_s + 0.5 mse_r); targets gin (next state), gin2 (reward)
            const int S = op.n0, Wd = S + 1;
            float es = 0.f, er = 0.f;
            for (int e = tid; e < RP_ROWS * Wd; e += RP_THREADS) {
                const int row = e / Wd, c = e - row * Wd;
                const int gr = min(r0 + row, B - 1);
                float* p = rp_buf + op.src + row * op.lds + c;
                const float tgt = c < S ? G(op.gin)[(size_t)gr * op.ldgin + c] : G(op.gin2)[gr];
                const float d = *p - tgt;
                const float g = d * (c < S ? op.s0 : op.s1);
                *p = g;
                if (r0 + row < B && member == 0) {          // (cluster: every member evaluates the heads, member 0 reports)
                    if (c < S) es += d * d; else er += d * d;
                    G(op.gout)[(size_t)(r0 + row) * op.ldg + c] = g;
                }
            }
            const float a = rp_block_sum(es, sh.part);
            const float b = rp_block_sum(er, sh.part);
            if (tid == 0 && member == 0) { G(op.part)[2 * rb] = a; G(op.part)[2 * rb + 1] = b; }
        } break;
        case RP_REPARAM: {
            // dst (dKL/dmean1 | dKL/dlog_std1) += (dz | dz * eps sigma mask);  src = dz [16, F], src2 = eps sigma mask [16, F]; gout <- dst
            const int F = op.N, FO = op.K > 0 ? op.K : op.N, row = tid >> 5;
            const bool rok = r0 + row < B;
            float* g = rp_buf + op.dst + row * op.ldd;
            gf_t go = G(op.gout) + (size_t)min(r0 + row, B - 1) * op.ldg;
            for (int jj = tid & 31; jj < F; jj += 32) {
                const float dz = rp_buf[op.src + row * op.lds + jj];
                const float ez = rp_buf[op.src2 + row * op.lds2 + jj];
                const float a = g[jj] + dz, b = g[FO + jj] + dz * ez;
                g[jj] = a; g[FO + jj] = b;
                if (rok) { go[jj] = a; go[FO + jj] = b; }
            }
        } break;
        case RP_STORE: {
            for (int e = tid; e < RP_ROWS * op.N; e += RP_THREADS) {
                const int row = e / op.N, c = e - row * op.N;
                if (r0 + row < B) G(op.gout)[(size_t)(r0 + row) * op.ldg + c] = rp_buf[op.src + row * op.lds + c];
            }
        } break;
        case RP_POLICY: {
            // src = actor head output [16, 2A] (mu | rho): gout <- tanh(mu + eps sigma) (row stride ldg), gout2 <- log pi [B]
            const int A = op.n0;
            gcf_t eps = G(L.dyn[op.dyn]);
            if (tid < RP_ROWS && r0 + tid < B) {
                const float* o = rp_buf + op.src + tid * op.lds;
                float lp = 0.f;
                for (int jj = 0; jj < A; ++jj) {
                    const float t = tanhf(o[A + jj]);
                    const float l = -5.f + 3.5f * (t + 1.f);
                    const float sg = expf(l);
                    const float e = eps[(size_t)(r0 + tid) * A + jj];
                    const float x = o[jj] + e * sg;
                    G(op.gout)[(size_t)(r0 + tid) * op.ldg + jj] = tanhf(x);
                    lp += -0.5f * e * e - l - 0.91893853320467274f - 2.f * (0.69314718055994531f - x - softplus_f(-2.f * x));
                }
                if (op.gout2) G(op.gout2)[r0 + tid] = lp;
            }
        } break;
        case RP_PUBLISH: case RP_GATHER: case RP_XCHG: {
            // slice geometry: width ws = N, pieces np = K (>= 1), piece stride ps = ldw; element e of a slice = (row, piece, c)
            const int ws = op.N, np = op.K > 0 ? op.K : 1, ps = op.ldw, per_row = ws * np, ne = RP_ROWS * per_row;
            const unsigned tag = epoch * 64u + (unsigned)op.flag;
            RP_GAS unsigned long long* const xb = (RP_GAS unsigned long long*)L.xbuf;
            auto slot = [&](int ctype, int mem) { return xb + ((((size_t)rb * 2 + ctype) * RP_MAX_HOPS + op.flag) * cs + mem) * RP_XSLOT; };
            auto col_of = [&](int mem, int r) { const int qq = r / ws; return mem * ws + qq * ps + (r - qq * ws); };
            if (op.kind != RP_GATHER) {
                RP_GAS unsigned long long* mine = slot(pr.ctype, member);
                for (int e = tid; e < ne; e += RP_THREADS) {
                    const int row = e / per_row, r = e - row * per_row;
                    const float v = rp_buf[op.src + row * op.lds + col_of(member, r)];
                    __hip_atomic_store(mine + e, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (op.kind != RP_PUBLISH && op.n0 != 2) {
                // the assembled vector is the next layer's inner operand: zero its padding up to the next multiple of 32 columns (the region
                // may have held a wider vector before)
                const int total = (np - 1) * ps + cs * ws, padw = ((total + 31) & ~31) - total;
                for (int e = tid; e < RP_ROWS * padw; e += RP_THREADS) { const int row = e / padw; rp_buf[op.src + row * op.lds + total + (e - row * padw)] = 0.f; }
            }
            if (op.kind != RP_PUBLISH) {
                const int from = op.n0 == 0 ? pr.ctype : (pr.ctype ^ 1);
                // the granules this thread is responsible for: position e (and e + 512) of every source member
                for (int e = tid; e < ne; e += RP_THREADS) {
                    const int row = e / per_row, r = e - row * per_row;
                    unsigned pending = 0;
                    for (int o = 0; o < cs; ++o) {
                        const bool want = op.n0 == 0 ? (o != member) : (op.n0 == 1 ? true : (o == member));
                        if (want) pending |= 1u << o;
                    }
                    int spins = 0;
                    while (pending && spins < (1 << 20) && !rp_dead) {
                        for (int o = 0; o < cs; ++o) {
                            if (!(pending & (1u << o))) continue;
                            const unsigned long long g = __hip_atomic_load(slot(from, o) + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((unsigned)(g >> 32) == tag) {
                                rp_buf[op.src + row * op.lds + col_of(o, r)] = __uint_as_float((unsigned)g);
                                pending &= ~(1u << o);
                            }
                        }
                        if (pending) __builtin_amdgcn_s_sleep(1);
                        ++spins;
                    }
                    // a granule that never arrived: raise the error word (rlrep_chain_status) and stop waiting in this launch, so that it
                    // drains; what was assembled from stale slots is reported, not used silently
                    if (pending && !rp_dead) { if (L.err) atomicOr(L.err, 4u); rp_dead = 1; }
                }
            }
        } break;
        case RP_SIGNAL: {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // the flag carries the launch epoch: a signal that comes after its waiter gave up cannot be mistaken for the next launch's
                __hip_atomic_store(L.flags + op.flag * nrb + rb, (int)(epoch + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } break;
        case RP_WAIT: {
            if (tid == 0) {
                int* f = L.flags + op.flag * nrb + rb;
                // bounded: the partner workgroup of this launch is co-resident or will be (the grid is far smaller than the chip)
                long long spins = 0;
                const int want = (int)(epoch + 1u);
                while (!rp_dead && __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want && spins < (1ll << 22)) { __builtin_amdgcn_s_sleep(1); ++spins; }
                if (!rp_dead && spins >= (1ll << 22)) { if (L.err) atomicOr(L.err, 4u); rp_dead = 1; }      // partner never signalled: report, stop waiting
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } break;
        default: break;
        }
        rp_barrier();
    }
    if (tim && threadIdx.x == 0) { RP_STAMP(tim, nops, wall_clock64()); RP_STAMP(tim, 64 + nops, clock64()); }
}

static int g_rp_lds_ok = 0;
extern "C" int rl_rowprog_init() {
    if (g_rp_lds_ok) return 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rowprog_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RP_LDS_DYN_MAX);
    if (e != hipSuccess) return (int)e;
    g_rp_lds_ok = 1;
    return 0;
}

extern "C" int rl_launch_rowprog(const RpLaunch* L, int total_blocks, hipStream_t st) {
    if (total_blocks <= 0) return 0;
    if ((size_t)L->lds_floats * 4 > RP_LDS_DYN_MAX) return -2;
    hipLaunchKernelGGL(rowprog_kernel, dim3(total_blocks), dim3(RP_THREADS), (size_t)L->lds_floats * 4, st, *L);
    return (int)hipGetLastError();
}

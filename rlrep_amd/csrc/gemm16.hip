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
// backward, weight/bias gradient, mse loss, tanh-Gaussian policy forward/backward).
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
#include <cstdio>
#include "common.h"
#include "kparams.h"
#include "gemm16_tile.h"

#ifdef RL_TIMING
// Instrumented build (tools/exp/gemm_timeline.py): thread 0 of every 8th workgroup records the 100 MHz wall clock at
// entry and exit and the shader clock at entry, once its task record is in registers, after its MFMAs, after the
// reduction barrier and at exit.  Slot = launch sequence number (bumped by workgroup 0 at exit) * 2048 + blockIdx.x.
struct RlTimRec { unsigned long long w0, w4, c[8]; int grid, bid; unsigned tag, valid; };
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
    if (slot < g_tim_cap) { RlTimRec r; r.w0 = tim_w0; r.w4 = w4; for (int q = 0; q < 8; ++q) r.c[q] = tim_c[q]; r.grid = gridDim.x; r.bid = blockIdx.x; \
    r.tag = (unsigned)(((uintptr_t)pC) >> 4) ^ ((unsigned)epi << 28); r.valid = 1; g_tim[slot] = r; } } \
    if (blockIdx.x == 0) atomicAdd(&g_tim_launch, 1u); } } while (0)
#else
#define TIM(k) do {} while (0)
#define TIM_FIN() do {} while (0)
#endif

// The launch header travels TWICE: inside the batch (gemm_lds shares the struct) and as 14 leading scalar arguments -- flags, tile count,
// the eight tasks' first tiles, their column-tile counts packed two to a word -- which the hardware PRELOADS into SGPRs at wave launch
// (build.sh compiles this file with -mllvm -amdgpu-kernarg-preload-count=14): a workgroup knows its task and tile without a single load,
// and its first scalar-load round trip is the task record itself.
template <int LA, int LB, int NF, bool VA, bool VB, bool PRE = false, int EPI_K = -1, int ACT_K = -1, bool MSE = false, int NJ_K = 0>
__global__ __launch_bounds__(256) void gemm16_kernel(int hdr, int total, int tb0, int tb1, int tb2, int tb3, int tb4, int tb5, int tb6, int tb7,
                                                     unsigned tc01, unsigned tc23, unsigned tc45, unsigned tc67, GemmBatch gb) {
    __shared__ float red[4][NF][4][64];
    __shared__ float bsum[4][16];

#ifdef RL_TIMING
    unsigned long long tim_c[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tim_w0 = 0; unsigned tim_lid = 0;
    if (threadIdx.x == 0) { tim_lid = *(volatile unsigned*)&g_tim_launch; if (TIM_ON) tim_w0 = wall_clock64(); }
#endif
    TIM(0);
    // latency-critical launch: win the issue arbitration against the waves of a noise-critic launch that may be running on the
    // other stream of the deferred pipeline (405.8 vs 429.7 us per train(); alone on the chip it changes nothing)
    // header + per-task tile ranges (unused entries of tb are INT_MAX): preloaded kernel arguments
    const int low_prio = hdr & 1;
    const int tb[GEMM_MAX_TASKS] = {tb0, tb1, tb2, tb3, tb4, tb5, tb6, tb7};
    const int tcs[GEMM_MAX_TASKS] = {(int)(tc01 & 0xffffu), (int)(tc01 >> 16), (int)(tc23 & 0xffffu), (int)(tc23 >> 16),
                                     (int)(tc45 & 0xffffu), (int)(tc45 >> 16), (int)(tc67 & 0xffffu), (int)(tc67 >> 16)};
    if (!low_prio) __builtin_amdgcn_s_setprio(3);
    int bid = blockIdx.x;
    if constexpr (LA == LD_COL && LB == LD_COL) {
        // Weight gradients (hdr bit 1): workgroup b runs on XCD b % 8, and BOTH operands of these products are activations that every XCD's L2
        // has to fetch -- in launch order each XCD saw an eighth of EVERY task (18.8 MB fetched for 4 MB of inputs at the headline dims,
        // profiles/r05_pmc_*).  Dealt as contiguous runs, XCD x owns tiles [x T / 8, (x + 1) T / 8): about one task, i.e. one G and one X.
        if (hdr & 2) {
            const int x = bid & 7, j = bid >> 3, q = total >> 3, r = total & 7;
            bid = x * q + min(x, r) + j;
        }
    }
    (void)total;
    int ti = 0, base = tb[0], tiles_c = tcs[0];
#pragma unroll
    for (int q = 1; q < GEMM_MAX_TASKS; ++q) if (bid >= tb[q]) { ti = q; base = tb[q]; tiles_c = tcs[q]; }
    const GemmTask& t = gb.t[ti];
    const int local = bid - base;
    const int tr = local / tiles_c, tc = local - tr * tiles_c;
#ifdef RL_TIMING
    asm volatile("" :: "s"(tr), "s"(tc), "s"(ti));
    TIM(7);                       // task and tile known (preloaded scalars only): what follows is the record's scalar-load round trip
    float* const pC = t.C; const int epi = t.epi;
    gemm16_tile<LA, LB, NF, VA, VB, PRE, false, GemmTask, EPI_K, ACT_K, false, MSE, 0, NJ_K>(t, tr, tc, red, bsum, nullptr, tim_c);
#else
    gemm16_tile<LA, LB, NF, VA, VB, PRE, false, GemmTask, EPI_K, ACT_K, false, MSE, 0, NJ_K>(t, tr, tc, red, bsum, nullptr);
#endif
    TIM_FIN();
}

// The launches of ONE or TWO plain forward / dX tasks whose inner length is a multiple of 256 (every 256- / 512-deep layer of the feature and
// critic / actor steps at the headline dimensions): the 14 preloaded scalars carry what the operand LOADS need instead of a task directory --
// a common base and, per task, the A / B offsets from it in floats, lda | ldb << 16, K | R << 16, Cn | log2(column tiles) << 16 -- so the
// first operand loads issue from preloaded SGPRs while the record's scalar loads are still in flight (gemm16_tile FAST).
template <int LA, int LB, int NF, bool VA, bool VB, int EPI_K, int ACT_K, int FU = 4>
__global__ __launch_bounds__(256) void gemm16_fast_kernel(int hdr, int tb1, const float* base, unsigned a0, unsigned b0, unsigned ld0, unsigned kr0, unsigned ct0,
                                                          unsigned a1, unsigned b1, unsigned ld1, unsigned kr1, unsigned ct1, GemmBatch gb) {
    __shared__ float red[4][NF][4][64];
    __shared__ float bsum[4][16];
#ifdef RL_TIMING
    unsigned long long tim_c[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tim_w0 = 0; unsigned tim_lid = 0;
    if (threadIdx.x == 0) { tim_lid = *(volatile unsigned*)&g_tim_launch; if (TIM_ON) tim_w0 = wall_clock64(); }
#endif
    TIM(0);
    if (!(hdr & 1)) __builtin_amdgcn_s_setprio(3);
    const int bid = blockIdx.x;
    const bool second = bid >= tb1;
    const unsigned ao = second ? a1 : a0, bo = second ? b1 : b0, ld = second ? ld1 : ld0, kr = second ? kr1 : kr0, ct = second ? ct1 : ct0;
    const int local = second ? bid - tb1 : bid, sh = (int)(ct >> 16);
    const int tr = local >> sh, tc = local & ((1 << sh) - 1);
    FastOps fo;
    fo.pA = base + (size_t)ao; fo.pB = base + (size_t)bo; fo.lda = (int)(ld & 0xffffu); fo.ldb = (int)(ld >> 16);
    fo.K = (int)(kr & 0xffffu); fo.R = (int)(kr >> 16); fo.Cn = (int)(ct & 0xffffu); fo.tiles_c = 1 << sh;
    int ti = __builtin_amdgcn_readfirstlane(second ? 1 : 0);
    asm volatile("" : "+s"(ti));          // (opaque: with a visible 0 / 1 hipcc loads BOTH records and selects field by field -- behind one s_waitcnt in front of the operand loads)
    const GemmTask& t = gb.t[ti];
#ifdef RL_TIMING
    asm volatile("" :: "s"(tr), "s"(tc));
    TIM(7);
    gemm16_tile<LA, LB, NF, VA, VB, false, false, GemmTask, EPI_K, ACT_K, false, false, FU>(t, tr, tc, red, bsum, nullptr, tim_c, &fo);
    float* const pC = t.C; const int epi = 0x8 | (EPI_K & 7);       // (tag bit 3: a fast-front-end launch)
    TIM_FIN();
#else
    gemm16_tile<LA, LB, NF, VA, VB, false, false, GemmTask, EPI_K, ACT_K, false, false, FU>(t, tr, tc, red, bsum, nullptr, &fo);
#endif
}
// ... and the launches of up to FOUR tasks of ONE shape (the same layer of sibling networks: f_target on three inputs, the policy beside them): the
// shape words once, then the A / B offsets of every task; hdr = low_prio | tiles per task << 8.
template <int LA, int LB, int NF, bool VA, bool VB, int EPI_K, int ACT_K, int FU = 4>
__global__ __launch_bounds__(256) void gemm16_fast4_kernel(int hdr, const float* base, unsigned ld, unsigned kr, unsigned ct, unsigned a0, unsigned b0, unsigned a1, unsigned b1,
                                                           unsigned a2, unsigned b2, unsigned a3, unsigned b3, GemmBatch gb) {
    __shared__ float red[4][NF][4][64];
    __shared__ float bsum[4][16];
#ifdef RL_TIMING
    unsigned long long tim_c[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tim_w0 = 0; unsigned tim_lid = 0;
    if (threadIdx.x == 0) { tim_lid = *(volatile unsigned*)&g_tim_launch; if (TIM_ON) tim_w0 = wall_clock64(); }
#endif
    TIM(0);
    if (!(hdr & 1)) __builtin_amdgcn_s_setprio(3);
    const int bid = blockIdx.x, nt = hdr >> 8;
    int ti = (bid >= nt ? 1 : 0) + (bid >= 2 * nt ? 1 : 0) + (bid >= 3 * nt ? 1 : 0);
    const unsigned ao = ti == 0 ? a0 : ti == 1 ? a1 : ti == 2 ? a2 : a3, bo = ti == 0 ? b0 : ti == 1 ? b1 : ti == 2 ? b2 : b3;
    const int local = bid - ti * nt, sh = (int)(ct >> 16);
    const int tr = local >> sh, tc = local & ((1 << sh) - 1);
    FastOps fo;
    fo.pA = base + (size_t)ao; fo.pB = base + (size_t)bo; fo.lda = (int)(ld & 0xffffu); fo.ldb = (int)(ld >> 16);
    fo.K = (int)(kr & 0xffffu); fo.R = (int)(kr >> 16); fo.Cn = (int)(ct & 0xffffu); fo.tiles_c = 1 << sh;
    ti = __builtin_amdgcn_readfirstlane(ti);
    asm volatile("" : "+s"(ti));
    const GemmTask& t = gb.t[ti];
#ifdef RL_TIMING
    asm volatile("" :: "s"(tr), "s"(tc));
    TIM(7);
    gemm16_tile<LA, LB, NF, VA, VB, false, false, GemmTask, EPI_K, ACT_K, false, false, FU>(t, tr, tc, red, bsum, nullptr, tim_c, &fo);
    float* const pC = t.C; const int epi = 0x8 | (EPI_K & 7);       // (tag bit 3: a fast-front-end launch)
    TIM_FIN();
#else
    gemm16_tile<LA, LB, NF, VA, VB, false, false, GemmTask, EPI_K, ACT_K, false, false, FU>(t, tr, tc, red, bsum, nullptr, &fo);
#endif
}
// ... and the single-task launches of the fused short product's dX form (FLAG_PRE; the vlsac decoder launch with its mse phase, the policy's
// last backward pair): one task's operands PLUS the short product's (X, Wt, M) fit the 14 scalars -- base, A / B / X / Wt / M offsets,
// lda | ldb << 16, K | R << 16, Cn | K1 << 16, ldx | ldw << 16, ldm; hdr = low_prio | log2(column tiles) << 8.
template <int EPI_K, int ACT_K, bool MSE, int NJ_K = 0>
__global__ __launch_bounds__(256) void gemm16_fastpre_kernel(int hdr, const float* base, unsigned ao, unsigned bo, unsigned ld, unsigned kr, unsigned ck, unsigned xo, unsigned wo, unsigned mo,
                                                             unsigned ldxw, unsigned ldm, GemmBatch gb) {
    __shared__ float red[4][1][4][64];
    __shared__ float bsum[4][16];
#ifdef RL_TIMING
    unsigned long long tim_c[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tim_w0 = 0; unsigned tim_lid = 0;
    if (threadIdx.x == 0) { tim_lid = *(volatile unsigned*)&g_tim_launch; if (TIM_ON) tim_w0 = wall_clock64(); }
#endif
    TIM(0);
    if (!(hdr & 1)) __builtin_amdgcn_s_setprio(3);
    const int bid = blockIdx.x, sh = hdr >> 8;
    const int tr = bid >> sh, tc = bid & ((1 << sh) - 1);
    FastOps fo;
    fo.pA = base + (size_t)ao; fo.pB = base + (size_t)bo; fo.lda = (int)(ld & 0xffffu); fo.ldb = (int)(ld >> 16);
    fo.K = (int)(kr & 0xffffu); fo.R = (int)(kr >> 16); fo.Cn = (int)(ck & 0xffffu); fo.tiles_c = 1 << sh;
    fo.X = base + (size_t)xo; fo.Wt = base + (size_t)wo; fo.M = base + (size_t)mo; fo.K1 = (int)(ck >> 16);
    fo.ldx = (int)(ldxw & 0xffffu); fo.ldw = (int)(ldxw >> 16); fo.ldm = (int)ldm;
    const GemmTask& t = gb.t[0];
#ifdef RL_TIMING
    asm volatile("" :: "s"(tr), "s"(tc));
    TIM(7);
    gemm16_tile<LD_ROW, LD_COL, 1, false, false, true, false, GemmTask, EPI_K, ACT_K, false, MSE, 4, NJ_K>(t, tr, tc, red, bsum, nullptr, tim_c, &fo);
    float* const pC = t.C; const int epi = 0x8 | (EPI_K & 7);
    TIM_FIN();
#else
    gemm16_tile<LD_ROW, LD_COL, 1, false, false, true, false, GemmTask, EPI_K, ACT_K, false, MSE, 4, NJ_K>(t, tr, tc, red, bsum, nullptr, &fo);
#endif
}
// launches per front end since the library was loaded (rlrep_front_end_counts): 0 = gemm16_fast_kernel, 1 = gemm16_fast4_kernel, 2 = gemm16_fastpre_kernel,
// 3 = the record front end (gemm16_kernel / gemm16_duo_kernel).  Which one a launch gets depends on its shapes and on every operand lying within
// 16 GiB of the lowest one (rlrep_amd/core.py carves all arenas out of one block for that); tests and bench.py read the counts.
long long g_rl_front[4] = {0, 0, 0, 0};
static int s_front = 3;
// diagnostic switches, read when an agent is created (rl_gemm16_read_env; no getenv on the per-launch path): RLREP_DISABLE=gemm16_fast = every launch on
// the record front end, RLREP_DISABLE=gemm16_spec = no compiled-in epilogues (both: bit-identical results, tests/test_default_mode.py)
static bool s_no_fast = false, s_generic = false, s_trace = false, s_dw_xcd = true;
extern "C" void rl_gemm16_read_env() {
    s_no_fast = rl_off("gemm16_fast"); s_generic = rl_off("gemm16_spec"); s_trace = rl_opt("gemm16_trace") != nullptr;
    s_dw_xcd = !rl_off("dw_xcd");              // weight-gradient tiles dealt to the XCDs as contiguous runs (same tiles, same arithmetic: bit-identical)
}
struct FastPreArgs { int hdr; const float* base; unsigned ao, bo, ld, kr, ck, xo, wo, mo, ldxw, ldm; };
static bool fastpre_args(const GemmBatch& gb, FastPreArgs& fa) {
    if (gb.ntasks != 1 || s_no_fast) return false;
    const GemmTask& t = gb.t[0];
    if (!(t.flags & FLAG_PRE) || (t.flags & FLAG_PRE_FWD) || !t.x0 || !t.x1 || !t.x2 || t.tile_base != 0) return false;
    const int tcn = t.tiles_c;
    if (t.K <= 0 || (t.K & 255) || t.K > 0xffff || t.R > 0xffff || t.Cn > 0xffff || t.lda > 0xffff || t.ldb > 0xffff || t.n0 <= 0 || t.n0 > 32 || t.ldx0 > 0xffff || t.ldx1 > 0xffff ||
        t.ldaux2 < 0 || tcn <= 0 || (tcn & (tcn - 1))) return false;
    const void* ptrs[5] = {t.A, t.B, t.x0, t.x1, t.x2};
    uintptr_t lo = ~(uintptr_t)0;
    for (const void* q : ptrs) lo = std::min(lo, (uintptr_t)q);
    lo &= ~(uintptr_t)15;
    unsigned off[5];
    for (int q = 0; q < 5; ++q) {
        const uintptr_t d = (uintptr_t)ptrs[q] - lo;
        if ((d & 3) || (d >> 2) > 0xffffffffull) return false;
        off[q] = (unsigned)(d >> 2);
    }
    int sh = 0; while ((1 << sh) < tcn) ++sh;
    fa.hdr = (gb.low_prio ? 1 : 0) | (sh << 8); fa.base = reinterpret_cast<const float*>(lo);
    fa.ao = off[0]; fa.bo = off[1]; fa.xo = off[2]; fa.wo = off[3]; fa.mo = off[4];
    fa.ld = (unsigned)t.lda | ((unsigned)t.ldb << 16); fa.kr = (unsigned)t.K | ((unsigned)t.R << 16); fa.ck = (unsigned)t.Cn | ((unsigned)t.n0 << 16);
    fa.ldxw = (unsigned)t.ldx0 | ((unsigned)t.ldx1 << 16); fa.ldm = (unsigned)t.ldaux2;
    return true;
}
#define G16_FASTPRE_ARGS(F, B) (F).hdr, (F).base, (F).ao, (F).bo, (F).ld, (F).kr, (F).ck, (F).xo, (F).wo, (F).mo, (F).ldxw, (F).ldm, (B)

// host side of the above: the 13 argument values, or false when the launch does not qualify
struct FastArgs { int hdr, tb1; const float* base; unsigned a[2], b[2], ld[2], kr[2], ct[2]; };
// short = false: every K a multiple of 256; short = true: every K <= 64 (the FAST = 1 instantiations)
static bool fast_args(const GemmBatch& gb, FastArgs& fa, bool short_k = false) {
    if (gb.ntasks < 1 || gb.ntasks > 2 || s_no_fast) return false;
    uintptr_t lo = ~(uintptr_t)0;
    for (int q = 0; q < gb.ntasks; ++q) { lo = std::min(lo, (uintptr_t)gb.t[q].A); lo = std::min(lo, (uintptr_t)gb.t[q].B); }
    lo &= ~(uintptr_t)15;
    fa.hdr = gb.low_prio ? 1 : 0; fa.tb1 = gb.ntasks > 1 ? gb.t[1].tile_base : 0x7fffffff; fa.base = reinterpret_cast<const float*>(lo);
    if (gb.t[0].tile_base != 0) return false;
    for (int q = 0; q < 2; ++q) {
        const GemmTask& t = gb.t[q < gb.ntasks ? q : 0];
        const int tcn = t.tiles_c;
        if (t.K <= 0 || (short_k ? t.K > 64 : (t.K & 255) != 0) || t.K > 0xffff || t.R > 0xffff || t.Cn > 0xffff || t.lda > 0xffff || t.ldb > 0xffff || tcn <= 0 || (tcn & (tcn - 1))) return false;
        const uintptr_t da = (uintptr_t)t.A - lo, db = (uintptr_t)t.B - lo;
        if ((da & 3) || (db & 3) || (da >> 2) > 0xffffffffull || (db >> 2) > 0xffffffffull) return false;
        int sh = 0; while ((1 << sh) < tcn) ++sh;
        fa.a[q] = (unsigned)(da >> 2); fa.b[q] = (unsigned)(db >> 2); fa.ld[q] = (unsigned)t.lda | ((unsigned)t.ldb << 16);
        fa.kr[q] = (unsigned)t.K | ((unsigned)t.R << 16); fa.ct[q] = (unsigned)t.Cn | ((unsigned)sh << 16);
    }
    return true;
}
struct Fast4Args { int hdr; const float* base; unsigned ld, kr, ct, a[4], b[4]; };
static bool fast4_args(const GemmBatch& gb, Fast4Args& fa, bool short_k = false) {
    if (gb.ntasks < 3 || gb.ntasks > 4 || s_no_fast) return false;
    const GemmTask& t0 = gb.t[0];
    uintptr_t lo = ~(uintptr_t)0;
    for (int q = 0; q < gb.ntasks; ++q) { lo = std::min(lo, (uintptr_t)gb.t[q].A); lo = std::min(lo, (uintptr_t)gb.t[q].B); }
    lo &= ~(uintptr_t)15;
    const int tcn = t0.tiles_c, nt = t0.ntiles;
    if (t0.K <= 0 || (short_k ? t0.K > 64 : (t0.K & 255) != 0) || t0.K > 0xffff || t0.R > 0xffff || t0.Cn > 0xffff || t0.lda > 0xffff || t0.ldb > 0xffff || tcn <= 0 || (tcn & (tcn - 1)) || nt <= 0 || nt >= (1 << 22)) return false;
    int sh = 0; while ((1 << sh) < tcn) ++sh;
    fa.hdr = (gb.low_prio ? 1 : 0) | (nt << 8); fa.base = reinterpret_cast<const float*>(lo);
    fa.ld = (unsigned)t0.lda | ((unsigned)t0.ldb << 16); fa.kr = (unsigned)t0.K | ((unsigned)t0.R << 16); fa.ct = (unsigned)t0.Cn | ((unsigned)sh << 16);
    for (int q = 0; q < 4; ++q) {
        const GemmTask& t = gb.t[q < gb.ntasks ? q : 0];
        if (t.K != t0.K || t.R != t0.R || t.Cn != t0.Cn || t.lda != t0.lda || t.ldb != t0.ldb || t.tiles_c != tcn || t.ntiles != nt || (q < gb.ntasks && t.tile_base != q * nt)) return false;
        const uintptr_t da = (uintptr_t)t.A - lo, db = (uintptr_t)t.B - lo;
        if ((da & 3) || (db & 3) || (da >> 2) > 0xffffffffull || (db >> 2) > 0xffffffffull) return false;
        fa.a[q] = (unsigned)(da >> 2); fa.b[q] = (unsigned)(db >> 2);
    }
    return true;
}
#define G16_FAST4_ARGS(F, B) (F).hdr, (F).base, (F).ld, (F).kr, (F).ct, (F).a[0], (F).b[0], (F).a[1], (F).b[1], (F).a[2], (F).b[2], (F).a[3], (F).b[3], (B)
#define G16_FAST_ARGS(F, B) (F).hdr, (F).tb1, (F).base, (F).a[0], (F).b[0], (F).ld[0], (F).kr[0], (F).ct[0], (F).a[1], (F).b[1], (F).ld[1], (F).kr[1], (F).ct[1], (B)

// TWO tile forms in one launch ("duo"): tasks [0, split) are row-major x k-major products (the dX form, NF = 1), tasks [split, ntasks) k-major
// x k-major ones (the weight-gradient form, NF = NF2); a workgroup runs the body of the form its task belongs to.  For INDEPENDENT stages of
// different forms that the step program would otherwise launch one after the other (ctrlsac: d(phi) = dS mu' and d(mu') = dS^T phi both
// consume the InfoNCE gradient dS): one dependent launch less.  Generic epilogues.  hdr = low_prio | split << 4.
template <bool VA1, int NF2>
__global__ __launch_bounds__(256) void gemm16_duo_kernel(int hdr, int total, int tb0, int tb1, int tb2, int tb3, int tb4, int tb5, int tb6, int tb7,
                                                         unsigned tc01, unsigned tc23, unsigned tc45, unsigned tc67, GemmBatch gb) {
    __shared__ float red[4][NF2][4][64];
    __shared__ float bsum[4][16];
    const int low_prio = hdr & 1, split = hdr >> 4;
    const int tb[GEMM_MAX_TASKS] = {tb0, tb1, tb2, tb3, tb4, tb5, tb6, tb7};
    const int tcs[GEMM_MAX_TASKS] = {(int)(tc01 & 0xffffu), (int)(tc01 >> 16), (int)(tc23 & 0xffffu), (int)(tc23 >> 16),
                                     (int)(tc45 & 0xffffu), (int)(tc45 >> 16), (int)(tc67 & 0xffffu), (int)(tc67 >> 16)};
    if (!low_prio) __builtin_amdgcn_s_setprio(3);
    const int bid = blockIdx.x;
    int ti = 0, base = tb[0], tiles_c = tcs[0];
#pragma unroll
    for (int q = 1; q < GEMM_MAX_TASKS; ++q) if (bid >= tb[q]) { ti = q; base = tb[q]; tiles_c = tcs[q]; }
    const GemmTask& t = gb.t[ti];
    const int local = bid - base;
    const int tr = local / tiles_c, tc = local - tr * tiles_c;
    (void)total;
#ifdef RL_TIMING
    unsigned long long* const tim_none = nullptr;
#define DUO_TIM , tim_none
#else
#define DUO_TIM
#endif
    if (ti < split) gemm16_tile<LD_ROW, LD_COL, 1, VA1, false, false, false, GemmTask, -1, -1>(t, tr, tc, reinterpret_cast<float (&)[4][1][4][64]>(red), bsum, nullptr DUO_TIM);
    else gemm16_tile<LD_COL, LD_COL, NF2, false, false, false, false, GemmTask, -1, -1>(t, tr, tc, red, bsum, nullptr DUO_TIM);
#undef DUO_TIM
}

// ------------------------------------------------------------------------------------------------
// host launcher
// ------------------------------------------------------------------------------------------------
// the 14 preloaded header scalars of a planned batch, then the batch
#define G16_ARGS(B) (((B).low_prio ? 1 : 0) | ((B).xcd_runs ? 2 : 0)), (B).total, (B).tb[0], (B).tb[1], (B).tb[2], (B).tb[3], (B).tb[4], (B).tb[5], (B).tb[6], (B).tb[7], \
    (unsigned)((B).tcs[0] | ((B).tcs[1] << 16)), (unsigned)((B).tcs[2] | ((B).tcs[3] << 16)), (unsigned)((B).tcs[4] | ((B).tcs[5] << 16)), (unsigned)((B).tcs[6] | ((B).tcs[7] << 16)), (B)

template <int LA, int LB, bool VA, bool VB>
static void launch_nf(int nf, dim3 g, hipStream_t st, const GemmBatch& gb) {
    if (nf == 1) hipLaunchKernelGGL((gemm16_kernel<LA, LB, 1, VA, VB>), g, dim3(256), 0, st, G16_ARGS(gb));
    else if (nf == 2) hipLaunchKernelGGL((gemm16_kernel<LA, LB, 2, VA, VB>), g, dim3(256), 0, st, G16_ARGS(gb));
    else hipLaunchKernelGGL((gemm16_kernel<LA, LB, 4, VA, VB>), g, dim3(256), 0, st, G16_ARGS(gb));
}

// NF = 1 launches whose tasks all share ONE plain epilogue (forward or dX; none / ReLU / ELU; no rank-1 term, no second output): the
// instantiation with that epilogue compiled in.  Returns false when the launch needs the generic kernel.
// the front end that loads from preloaded scalars (gemm16_fast_kernel / gemm16_fast4_kernel) with epilogue EPI_K / ACT_K compiled in (-1: from the record)
template <int LA, int LB, int NF, bool VA, bool VB, int EPI_K, int ACT_K>
static bool launch_fast(dim3 g, hipStream_t st, const GemmBatch& gb) {
    FastArgs fa; Fast4Args f4;
    if (fast_args(gb, fa)) { hipLaunchKernelGGL((gemm16_fast_kernel<LA, LB, NF, VA, VB, EPI_K, ACT_K>), g, dim3(256), 0, st, G16_FAST_ARGS(fa, gb)); s_front = 0; return true; }
    if (fast4_args(gb, f4)) { hipLaunchKernelGGL((gemm16_fast4_kernel<LA, LB, NF, VA, VB, EPI_K, ACT_K>), g, dim3(256), 0, st, G16_FAST4_ARGS(f4, gb)); s_front = 1; return true; }
    return false;
}
// ... and the first layers (K <= 64, forward form, rows of 17 / 23 / 40 floats: with or without 16-byte operand loads)
template <int LA, int LB, int NF, bool VA, bool VB, int ACT_K>
static bool launch_fast_short(dim3 g, hipStream_t st, const GemmBatch& gb) {
    FastArgs fa; Fast4Args f4;
    if (fast_args(gb, fa, true)) { hipLaunchKernelGGL((gemm16_fast_kernel<LA, LB, NF, VA, VB, EPI_FWD, ACT_K, 1>), g, dim3(256), 0, st, G16_FAST_ARGS(fa, gb)); s_front = 0; return true; }
    if (fast4_args(gb, f4, true)) { hipLaunchKernelGGL((gemm16_fast4_kernel<LA, LB, NF, VA, VB, EPI_FWD, ACT_K, 1>), g, dim3(256), 0, st, G16_FAST4_ARGS(f4, gb)); s_front = 1; return true; }
    return false;
}
// NF = 1 launches whose tasks all share ONE plain epilogue (forward or dX; none / ReLU / ELU; no rank-1 term, no second output): the
// instantiation with that epilogue compiled in.  Returns false when the launch needs the generic kernel.
template <int LA, int LB, int NF, bool VA, bool VB>
static bool launch_spec(dim3 g, hipStream_t st, const GemmBatch& gb) {
    if (s_generic) return false;
    const int epi = gb.t[0].epi;
    int act = gb.t[0].act;
    bool same_epi = true;
    for (int q = 0; q < gb.ntasks; ++q) {
        if (gb.t[q].flags & FLAG_PRE) return false;
        if (gb.t[q].epi != epi) same_epi = false;
        if (gb.t[q].act != act) act = -1;                   // mixed activations: read from the record
    }
    // first layers (K <= 64): one or two tasks, any alignment
    if constexpr (LA == LD_ROW && LB == LD_ROW) if (same_epi && epi == EPI_FWD) {
        if (act == ACT_RELU) { if (launch_fast_short<LA, LB, NF, VA, VB, ACT_RELU>(g, st, gb)) return true; }
        else if (act == ACT_ELU) { if (launch_fast_short<LA, LB, NF, VA, VB, ACT_ELU>(g, st, gb)) return true; }
        else if (act < 0) { if (launch_fast_short<LA, LB, NF, VA, VB, -1>(g, st, gb)) return true; }
    }
    // one to four tasks, K % 256 == 0, 16-byte operand A: the front end that loads from preloaded scalars
    if (VA) {
        if (same_epi && LB == LD_ROW && epi == EPI_FWD) {
            if (act == ACT_NONE) { if (launch_fast<LA, LB, NF, VA, VB, EPI_FWD, ACT_NONE>(g, st, gb)) return true; }
            else if (act == ACT_RELU) { if (launch_fast<LA, LB, NF, VA, VB, EPI_FWD, ACT_RELU>(g, st, gb)) return true; }
            else if (act == ACT_ELU) { if (launch_fast<LA, LB, NF, VA, VB, EPI_FWD, ACT_ELU>(g, st, gb)) return true; }
            else if (launch_fast<LA, LB, NF, VA, VB, EPI_FWD, -1>(g, st, gb)) return true;
        } else if (same_epi && LB == LD_COL && epi == EPI_DX) {
            if (act == ACT_NONE) { if (launch_fast<LA, LB, NF, VA, VB, EPI_DX, ACT_NONE>(g, st, gb)) return true; }
            else if (act == ACT_RELU) { if (launch_fast<LA, LB, NF, VA, VB, EPI_DX, ACT_RELU>(g, st, gb)) return true; }
            else if (act == ACT_ELU) { if (launch_fast<LA, LB, NF, VA, VB, EPI_DX, ACT_ELU>(g, st, gb)) return true; }
            else if (launch_fast<LA, LB, NF, VA, VB, EPI_DX, -1>(g, st, gb)) return true;
        } else if (same_epi && NF == 1 && LB == LD_COL && epi == EPI_DX_POLICYBWD) {
            if (launch_fast<LA, LB, NF, VA, VB, EPI_DX_POLICYBWD, ACT_NONE>(g, st, gb)) return true;     // (the policy's backward: 3 500 cycles of epilogue in the generic body)
        } else if (launch_fast<LA, LB, NF, VA, VB, -1, -1>(g, st, gb)) return true;       // any other epilogue: the generic body behind the fast front end
    }
    if (!same_epi) return false;
    // mixed activations (the policy's ELU layers beside f's ReLU layers in one launch): the epilogue KIND compiled in, the activation read from the
    // record -- the fully generic body spends 5 700 cycles between the reduction barrier and the store of such a tile (tools/exp/gemm_timeline.py)
    if (act < 0) {
        if (LB == LD_ROW && epi == EPI_FWD) { hipLaunchKernelGGL((gemm16_kernel<LA, LB, NF, VA, VB, false, EPI_FWD, -1>), g, dim3(256), 0, st, G16_ARGS(gb)); return true; }
        if (LB == LD_COL && epi == EPI_DX) { hipLaunchKernelGGL((gemm16_kernel<LA, LB, NF, VA, VB, false, EPI_DX, -1>), g, dim3(256), 0, st, G16_ARGS(gb)); return true; }
        return false;
    }
    if (LB == LD_ROW && epi == EPI_FWD) {
        if (act == ACT_NONE) hipLaunchKernelGGL((gemm16_kernel<LA, LB, NF, VA, VB, false, EPI_FWD, ACT_NONE>), g, dim3(256), 0, st, G16_ARGS(gb));
        else if (act == ACT_RELU) hipLaunchKernelGGL((gemm16_kernel<LA, LB, NF, VA, VB, false, EPI_FWD, ACT_RELU>), g, dim3(256), 0, st, G16_ARGS(gb));
        else if (act == ACT_ELU) hipLaunchKernelGGL((gemm16_kernel<LA, LB, NF, VA, VB, false, EPI_FWD, ACT_ELU>), g, dim3(256), 0, st, G16_ARGS(gb));
        else return false;
        return true;
    }
    if (LB == LD_COL && epi == EPI_DX) {
        if (act == ACT_NONE) hipLaunchKernelGGL((gemm16_kernel<LA, LB, NF, VA, VB, false, EPI_DX, ACT_NONE>), g, dim3(256), 0, st, G16_ARGS(gb));
        else if (act == ACT_RELU) hipLaunchKernelGGL((gemm16_kernel<LA, LB, NF, VA, VB, false, EPI_DX, ACT_RELU>), g, dim3(256), 0, st, G16_ARGS(gb));
        else if (act == ACT_ELU) hipLaunchKernelGGL((gemm16_kernel<LA, LB, NF, VA, VB, false, EPI_DX, ACT_ELU>), g, dim3(256), 0, st, G16_ARGS(gb));
        else return false;
        return true;
    }
    if (NF == 1 && LB == LD_COL && epi == EPI_DX_POLICYBWD) {  // the policy's backward (same epilogue instantiation as on the fast front end)
        hipLaunchKernelGGL((gemm16_kernel<LA, LB, 1, VA, VB, false, EPI_DX_POLICYBWD, ACT_NONE>), g, dim3(256), 0, st, G16_ARGS(gb));
        return true;
    }
    if (NF == 1 && LB == LD_ROW && epi == EPI_FWD_MSE) {       // vlsac decoder heads + mse
        hipLaunchKernelGGL((gemm16_kernel<LA, LB, 1, VA, VB, false, EPI_FWD_MSE, ACT_NONE>), g, dim3(256), 0, st, G16_ARGS(gb));
        return true;
    }
    return false;
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

static int launch_gemm16_impl(int la, int lb, int nf, const GemmBatch* gb_in, int total_tiles, hipStream_t st);
extern "C" int rl_launch_gemm16(int la, int lb, int nf, const GemmBatch* gb_in, int total_tiles, hipStream_t st) {
    if (total_tiles <= 0) return 0;
    s_front = 3;
    const int rc = launch_gemm16_impl(la, lb, nf, gb_in, total_tiles, st);
    if (rc == 0) ++g_rl_front[s_front];
    return rc;
}
static int launch_gemm16_impl(int la, int lb, int nf, const GemmBatch* gb_in, int total_tiles, hipStream_t st) {
    // the epilogue operand slots travel in the record (common.h rl_gemm16_plan): filled here, on the launcher's copy, after the caller's
    // per-call patches (noise pointers)
    GemmBatch planned = *gb_in;
    for (int q = 0; q < GEMM_MAX_TASKS; ++q) { planned.tb[q] = 0x7fffffff; planned.tcs[q] = 1; }
    for (int q = 0; q < planned.ntasks; ++q) {
        rl_gemm16_plan(planned.t[q]); planned.tb[q] = planned.t[q].tile_base; planned.tcs[q] = planned.t[q].tiles_c;
        if (planned.t[q].tiles_c <= 0 || planned.t[q].tiles_c > 0xffff) return -3;       // the preloaded header packs column-tile counts two to a word
    }
    planned.total = total_tiles;
    const GemmBatch* const gb = &planned;
    dim3 g(total_tiles);
    if (s_trace) {                                                            // one line per launch: which front end it gets (tools/exp/gemm16_trace.py)
        FastArgs fa; Fast4Args f4;
        const bool pre = gb->ntasks > 0 && (gb->t[0].flags & FLAG_PRE), vecA = all_vec(*gb, false), vecB = all_vec(*gb, true);
        const bool vec_ok = la == LD_ROW && vecA && (lb == LD_COL || vecB), nf_ok = nf == 1 || nf == 2;
        const char* front = "record";
        FastPreArgs fpt;
        if (pre && la == LD_ROW && lb == LD_COL && nf == 1 && !s_generic && fastpre_args(*gb, fpt)) front = "fast (fused short product)";
        if (!pre && la == LD_ROW && nf_ok && !s_generic) {
            if (lb == LD_ROW && (vecA == vecB) && (fast_args(*gb, fa, true) || fast4_args(*gb, f4, true)) && gb->t[0].epi == EPI_FWD) front = "fast (K <= 64)";
            else if (vec_ok && fast_args(*gb, fa)) front = "fast";
            else if (vec_ok && fast4_args(*gb, f4)) front = "fast4";
        }
        fprintf(stderr, "[gemm16] la=%d lb=%d nf=%d tasks=%d tiles=%d front=%s :", la, lb, nf, gb->ntasks, total_tiles, front);
        for (int q = 0; q < gb->ntasks; ++q) fprintf(stderr, " [R=%d Cn=%d K=%d lda=%d ldb=%d epi=%d act=%d fl=0x%x]", gb->t[q].R, gb->t[q].Cn, gb->t[q].K, gb->t[q].lda, gb->t[q].ldb, gb->t[q].epi, gb->t[q].act, gb->t[q].flags);
        fprintf(stderr, "\n");
    }
    if (gb->ntasks > 0 && (gb->t[0].flags & FLAG_PRE)) {       // fused-short-product launch: every task carries FLAG_PRE (and agrees on the form)
        const int fw = gb->t[0].flags & FLAG_PRE_FWD;
        for (int q = 0; q < gb->ntasks; ++q) {
            if (!(gb->t[q].flags & FLAG_PRE) || (gb->t[q].flags & FLAG_PRE_FWD) != fw) return -3;
            if (gb->t[q].n0 > (fw ? 48 : 32)) return -3;
        }
        if (la != LD_ROW || nf != 1) return -3;
        if (fw) {
#ifndef RL_EXPERIMENTS
            return -3;                                   // first layers fused into the second: measured slower, experiments build only
#else
            if (lb != LD_ROW) return -3;
            if (all_vec(*gb, true)) hipLaunchKernelGGL((gemm16_kernel<LD_ROW, LD_ROW, 1, false, true, true>), g, dim3(256), 0, st, G16_ARGS(*gb));
            else hipLaunchKernelGGL((gemm16_kernel<LD_ROW, LD_ROW, 1, false, false, true>), g, dim3(256), 0, st, G16_ARGS(*gb));
#endif
        } else {
            if (lb != LD_COL) return -3;
            bool rep = !s_generic, plain = rep, elu = rep;
            for (int q = 0; q < gb->ntasks; ++q) {
                rep = rep && gb->t[q].epi == EPI_DX_REPARAM; plain = plain && gb->t[q].epi == EPI_DX && gb->t[q].act == ACT_NONE;
                elu = elu && gb->t[q].epi == EPI_DX && gb->t[q].act == ACT_ELU;
            }
            // FLAG_PRE_MSE (vlsac decoder: the heads + mse launch rides here): one task, reparameterisation epilogue, 16-byte-aligned M / Wt rows
            bool mse = false;
            for (int q = 0; q < gb->ntasks; ++q) mse = mse || (gb->t[q].flags & FLAG_PRE_MSE);
            FastPreArgs fp;
            const bool fastpre = (rep || plain || elu) && fastpre_args(*gb, fp);          // one task, K % 256 == 0: the front end that loads from preloaded scalars
            if (fastpre) s_front = 2;
            if (mse) {
                const GemmTask& m0 = gb->t[0];
                if (gb->ntasks != 1 || !rep || (m0.ldaux2 & 3) || (m0.ldx1 & 3) || (m0.K & 3) || ((((uintptr_t)m0.x2) | ((uintptr_t)m0.x1)) & 15) || !m0.bias || !m0.tgs || !m0.tgr || !m0.mse_part || !m0.x0)
                    return -3;
                if (fastpre) hipLaunchKernelGGL((gemm16_fastpre_kernel<EPI_DX_REPARAM, ACT_NONE, true>), g, dim3(256), 0, st, G16_FASTPRE_ARGS(fp, *gb));
                else hipLaunchKernelGGL((gemm16_kernel<LD_ROW, LD_COL, 1, false, false, true, EPI_DX_REPARAM, ACT_NONE, true>), g, dim3(256), 0, st, G16_ARGS(*gb));
                return (int)hipGetLastError();
            }
            // short products of at most 16 inner indices (the policy head's 2 A <= 16 columns): instantiations without the second 16-wide chunk
            bool nj1 = !s_generic;
            for (int q = 0; q < gb->ntasks; ++q) nj1 = nj1 && gb->t[q].n0 <= 16;
            if (fastpre) {
                if (elu && nj1) hipLaunchKernelGGL((gemm16_fastpre_kernel<EPI_DX, ACT_ELU, false, 1>), g, dim3(256), 0, st, G16_FASTPRE_ARGS(fp, *gb));
                else if (elu) hipLaunchKernelGGL((gemm16_fastpre_kernel<EPI_DX, ACT_ELU, false>), g, dim3(256), 0, st, G16_FASTPRE_ARGS(fp, *gb));
                else if (rep) hipLaunchKernelGGL((gemm16_fastpre_kernel<EPI_DX_REPARAM, ACT_NONE, false>), g, dim3(256), 0, st, G16_FASTPRE_ARGS(fp, *gb));
                else if (nj1) hipLaunchKernelGGL((gemm16_fastpre_kernel<EPI_DX, ACT_NONE, false, 1>), g, dim3(256), 0, st, G16_FASTPRE_ARGS(fp, *gb));
                else hipLaunchKernelGGL((gemm16_fastpre_kernel<EPI_DX, ACT_NONE, false>), g, dim3(256), 0, st, G16_FASTPRE_ARGS(fp, *gb));
                return (int)hipGetLastError();
            }
            if (elu && nj1) hipLaunchKernelGGL((gemm16_kernel<LD_ROW, LD_COL, 1, false, false, true, EPI_DX, ACT_ELU, false, 1>), g, dim3(256), 0, st, G16_ARGS(*gb));
            else if (elu) hipLaunchKernelGGL((gemm16_kernel<LD_ROW, LD_COL, 1, false, false, true, EPI_DX, ACT_ELU>), g, dim3(256), 0, st, G16_ARGS(*gb));
            else if (rep) hipLaunchKernelGGL((gemm16_kernel<LD_ROW, LD_COL, 1, false, false, true, EPI_DX_REPARAM, ACT_NONE>), g, dim3(256), 0, st, G16_ARGS(*gb));
            else if (plain) hipLaunchKernelGGL((gemm16_kernel<LD_ROW, LD_COL, 1, false, false, true, EPI_DX, ACT_NONE>), g, dim3(256), 0, st, G16_ARGS(*gb));
            else hipLaunchKernelGGL((gemm16_kernel<LD_ROW, LD_COL, 1, false, false, true>), g, dim3(256), 0, st, G16_ARGS(*gb));
        }
        return (int)hipGetLastError();
    }
    if (la == LD_ROW && lb == LD_ROW) {
        if (all_vec(*gb, false) && all_vec(*gb, true)) { if (!(nf == 1 && launch_spec<LD_ROW, LD_ROW, 1, true, true>(g, st, *gb)) && !(nf == 2 && launch_spec<LD_ROW, LD_ROW, 2, true, true>(g, st, *gb))) launch_nf<LD_ROW, LD_ROW, true, true>(nf, g, st, *gb); }
        else if (!(nf == 1 && launch_spec<LD_ROW, LD_ROW, 1, false, false>(g, st, *gb)) && !(nf == 2 && launch_spec<LD_ROW, LD_ROW, 2, false, false>(g, st, *gb))) launch_nf<LD_ROW, LD_ROW, false, false>(nf, g, st, *gb);
    } else if (la == LD_ROW && lb == LD_COL) {
        if (all_vec(*gb, false)) { if (!(nf == 1 && launch_spec<LD_ROW, LD_COL, 1, true, false>(g, st, *gb)) && !(nf == 2 && launch_spec<LD_ROW, LD_COL, 2, true, false>(g, st, *gb))) launch_nf<LD_ROW, LD_COL, true, false>(nf, g, st, *gb); }
        else if (!(nf == 1 && launch_spec<LD_ROW, LD_COL, 1, false, false>(g, st, *gb))) launch_nf<LD_ROW, LD_COL, false, false>(nf, g, st, *gb);
    } else if (la == LD_COL && lb == LD_COL) {
        // weight gradients with nothing to accumulate into: the instantiation without slot loads; with the optimizer in some tasks'
        // epilogues (FLAG_ADAM): the instantiation that loads the parameter / moment slots (EPI_DWA)
        bool plain = !s_generic, opt = false;
        planned.xcd_runs = s_dw_xcd && total_tiles >= 64 ? 1 : 0;
        for (int q = 0; q < gb->ntasks; ++q) {
            plain = plain && gb->t[q].epi == EPI_DW && !(gb->t[q].flags & FLAG_ACCUM);
            if (gb->t[q].flags & FLAG_ADAM) { opt = true; if (!gb->t[q].ad_p || !gb->t[q].ad_grp) return -3; }
        }
        if (opt) {
            if (!plain) return -3;
            if (nf == 4) hipLaunchKernelGGL((gemm16_kernel<LD_COL, LD_COL, 4, false, false, false, EPI_DWA, ACT_NONE>), g, dim3(256), 0, st, G16_ARGS(*gb));
            else if (nf == 2) hipLaunchKernelGGL((gemm16_kernel<LD_COL, LD_COL, 2, false, false, false, EPI_DWA, ACT_NONE>), g, dim3(256), 0, st, G16_ARGS(*gb));
            else hipLaunchKernelGGL((gemm16_kernel<LD_COL, LD_COL, 1, false, false, false, EPI_DWA, ACT_NONE>), g, dim3(256), 0, st, G16_ARGS(*gb));
            return (int)hipGetLastError();
        }
        if (plain && nf == 4) hipLaunchKernelGGL((gemm16_kernel<LD_COL, LD_COL, 4, false, false, false, EPI_DW, ACT_NONE>), g, dim3(256), 0, st, G16_ARGS(*gb));
        else if (plain && nf == 2) hipLaunchKernelGGL((gemm16_kernel<LD_COL, LD_COL, 2, false, false, false, EPI_DW, ACT_NONE>), g, dim3(256), 0, st, G16_ARGS(*gb));
        else if (plain && nf == 1) hipLaunchKernelGGL((gemm16_kernel<LD_COL, LD_COL, 1, false, false, false, EPI_DW, ACT_NONE>), g, dim3(256), 0, st, G16_ARGS(*gb));
        else launch_nf<LD_COL, LD_COL, false, false>(nf, g, st, *gb);
    }
    else return -1;
    return (int)hipGetLastError();
}

// tasks [0, split): LD_ROW x LD_COL at NF = 1; tasks [split, ntasks): LD_COL x LD_COL at NF = nf2 (1 or 4); tile bases / column-tile counts set by the caller
extern "C" int rl_launch_gemm16_duo(int split, int nf2, const GemmBatch* gb_in, int total_tiles, hipStream_t st) {
    if (total_tiles <= 0) return 0;
    if (split <= 0 || split >= gb_in->ntasks || (nf2 != 1 && nf2 != 4)) return -4;
    GemmBatch planned = *gb_in;
    for (int q = 0; q < GEMM_MAX_TASKS; ++q) { planned.tb[q] = 0x7fffffff; planned.tcs[q] = 1; }
    for (int q = 0; q < planned.ntasks; ++q) {
        rl_gemm16_plan(planned.t[q]); planned.tb[q] = planned.t[q].tile_base; planned.tcs[q] = planned.t[q].tiles_c;
        if (planned.t[q].flags & FLAG_PRE) return -4;
        if (planned.t[q].tiles_c <= 0 || planned.t[q].tiles_c > 0xffff) return -3;
    }
    planned.total = total_tiles;
    bool va = true;
    for (int q = 0; q < split; ++q) { const GemmTask& t = planned.t[q]; if ((t.lda & 3) || (t.K & 3) || (((uintptr_t)t.A) & 15)) va = false; }
    const int hdr = (planned.low_prio ? 1 : 0) | (split << 4);
    dim3 g(total_tiles);
#define DUO_ARGS hdr, planned.total, planned.tb[0], planned.tb[1], planned.tb[2], planned.tb[3], planned.tb[4], planned.tb[5], planned.tb[6], planned.tb[7], \
    (unsigned)(planned.tcs[0] | (planned.tcs[1] << 16)), (unsigned)(planned.tcs[2] | (planned.tcs[3] << 16)), (unsigned)(planned.tcs[4] | (planned.tcs[5] << 16)), (unsigned)(planned.tcs[6] | (planned.tcs[7] << 16)), planned
    if (va && nf2 == 1) hipLaunchKernelGGL((gemm16_duo_kernel<true, 1>), g, dim3(256), 0, st, DUO_ARGS);
    else if (va) hipLaunchKernelGGL((gemm16_duo_kernel<true, 4>), g, dim3(256), 0, st, DUO_ARGS);
    else if (nf2 == 1) hipLaunchKernelGGL((gemm16_duo_kernel<false, 1>), g, dim3(256), 0, st, DUO_ARGS);
    else hipLaunchKernelGGL((gemm16_duo_kernel<false, 4>), g, dim3(256), 0, st, DUO_ARGS);
#undef DUO_ARGS
    ++g_rl_front[3];
    return (int)hipGetLastError();
}

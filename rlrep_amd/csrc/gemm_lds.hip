// gemm_lds: the LDS-tiled fp32-MFMA GEMM of the update path for the LARGE layers (gfx950).
//
// gemm16.hip gives every workgroup one 16 x 64 tile and streams operands L2 -> VGPR: right for the 256-wide layers of
// the headline configuration (latency-bound), wrong for ctrlsac at F = 2048 / H = 1024, spedersac at M = 2048 and
// diffsrsac's nabla-mu head (2048 x 512 x 96 256, 202 GFLOP per pass), where every operand element would be re-read
// from L2 by 16..128 workgroups.  Here a workgroup (4 waves, 2 x 2) owns a BT x BT output tile (BT = 128 or 64),
// the K loop walks 32-deep slices that are staged global -> VGPR -> LDS (double-buffered, ONE barrier per slice: the
// loads of slice t+1 are issued before the MFMAs of slice t and written to the other buffer after them), and each
// wave runs (BT/32)^2 accumulators of v_mfma_f32_16x16x4_f32 -- exact fp32, same peak as the VALU (157.3 TF).
//
// The same GemmTask table as gemm16 (common.h) drives it, with the three operand-layout combinations of the path:
//   forward   Y = X W^T        A row-major [R,K],  B row-major [Cn,K]   (LD_ROW, LD_ROW)
//   dX        dX = G W         A row-major [R,K],  B k-major   [K,Cn]   (LD_ROW, LD_COL)
//   dW        dW = G^T X       A k-major   [K,R],  B k-major   [K,Cn]   (LD_COL, LD_COL)
// LDS images: a row-major operand is kept [row][36] (k contiguous, +4 pad: the ds_read_b64 fragment reads of a
// half-wave hit 32 distinct bank pairs), a k-major operand [k][BT+8] (row contiguous, the two k-rows a half-wave reads
// sit 16 banks apart).  MFMA k-slots are permuted identically for A and B: MFMA m of an 8-deep group takes
// k = 2*(lane>>4) + (m&1), so a row-major fragment is one 8-byte LDS read feeding two MFMAs.
//
// Small outputs with a long inner dimension (dX of the nabla-mu head: 2048 x 512 over K = 96 256; every M = 256 layer
// of ctrlsac) are split along K over `splits` workgroups per tile; partial tiles go to a slab and a finishing launch
// adds them in split order (deterministic) and applies the epilogue.  No float atomics anywhere.
#include "common.h"
// Every kernel of this file finds its task from EIGHT LEADING SCALAR ARGUMENTS -- the tasks' first tiles (first finishing blocks for the finishing
// kernel), INT_MAX where there is none -- which the hardware preloads into SGPRs at wave launch (build.sh: kernarg preload for this file): the task
// search costs no load, and the first scalar-load round trip is the task's own record (it used to be the second).
#define GL_DIR_PARAMS int d0, int d1, int d2, int d3, int d4, int d5, int d6, int d7
#define GL_DIR_ARGS(D) (D)[0], (D)[1], (D)[2], (D)[3], (D)[4], (D)[5], (D)[6], (D)[7]
extern long long g_rl_launches;
#include "kparams.h"

#define GL_BK 32
#define GL_KCS 36

template <int BT, int LD> struct GlTile {
    static constexpr int RCS = BT + 8;
    static constexpr int FLOATS = (LD == LD_ROW) ? BT * GL_KCS : GL_BK * RCS;
    static constexpr int NL = BT / 32;          // 16-byte loads per thread per slice
};

// ---- staging: global -> registers (branch-free: clamped address, value zeroed by a select) ----------------------
// `vec`: 16-byte loads are legal for this operand (workgroup-uniform); otherwise four clamped 4-byte loads per slot
template <int BT, int LD>
__device__ __forceinline__ void gl_stage_load(const float* __restrict__ P, int ld, int base, int lim, int k0, int kend, bool vec,
                                              f32x4 (&v)[BT / 32]) {
    if (vec) {
#pragma unroll
        for (int j = 0; j < BT / 32; ++j) {
            const int idx = threadIdx.x + 256 * j;
            if (LD == LD_ROW) {
                const int row = idx >> 3, kc = (idx & 7) * 4;
                const int r = min(base + row, lim - 1), k = min(k0 + kc, kend - 4);
                const f32x4 x = *reinterpret_cast<const f32x4*>(P + (size_t)r * ld + k);
                const bool ok = (k0 + kc) < kend;
                v[j] = ok ? x : (f32x4){0.f, 0.f, 0.f, 0.f};
            } else {
                constexpr int Q = BT / 4;
                const int kk = idx / Q, i4 = (idx % Q) * 4;
                const int k = min(k0 + kk, kend - 1), i = min(base + i4, lim - 4);
                const f32x4 x = *reinterpret_cast<const f32x4*>(P + (size_t)k * ld + i);
                const bool ok = (k0 + kk < kend) && (base + i4 < lim);
                v[j] = ok ? x : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < BT / 32; ++j) {
            const int idx = threadIdx.x + 256 * j;
            f32x4 x;
            if (LD == LD_ROW) {
                const int row = idx >> 3, kc = (idx & 7) * 4;
                const float* p = P + (size_t)min(base + row, lim - 1) * ld;
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float y = p[min(k0 + kc + q, kend - 1)]; x[q] = (k0 + kc + q) < kend ? y : 0.f; }
            } else {
                constexpr int Q = BT / 4;
                const int kk = idx / Q, i4 = (idx % Q) * 4;
                const float* p = P + (size_t)min(k0 + kk, kend - 1) * ld;
                const bool kok = (k0 + kk) < kend;
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float y = p[min(base + i4 + q, lim - 1)]; x[q] = (kok && (base + i4 + q) < lim) ? y : 0.f; }
            }
            v[j] = x;
        }
    }
}

template <int BT, int LD>
__device__ __forceinline__ void gl_stage_write(float* __restrict__ S, const f32x4 (&v)[BT / 32]) {
#pragma unroll
    for (int j = 0; j < BT / 32; ++j) {
        const int idx = threadIdx.x + 256 * j;
        if (LD == LD_ROW) {
            const int row = idx >> 3, kc = (idx & 7) * 4;
            *reinterpret_cast<f32x4*>(S + row * GL_KCS + kc) = v[j];
        } else {
            constexpr int Q = BT / 4;
            const int kk = idx / Q, i4 = (idx % Q) * 4;
            *reinterpret_cast<f32x4*>(S + kk * (BT + 8) + i4) = v[j];
        }
    }
}

// fragments of the j-th 8-deep group: f[t][e] = operand(row wbase + 16 t + i, k = 8 j + 2 kq + e)
template <int BT, int LD, int T>
__device__ __forceinline__ void gl_frag(const float* __restrict__ S, int wbase, int i, int kq, int j, float (&f)[T][2]) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
        if (LD == LD_ROW) {
            const float2 x = *reinterpret_cast<const float2*>(S + (wbase + t * 16 + i) * GL_KCS + 8 * j + 2 * kq);
            f[t][0] = x.x; f[t][1] = x.y;
        } else {
            const float* p = S + (8 * j + 2 * kq) * (BT + 8) + wbase + t * 16 + i;
            f[t][0] = p[0]; f[t][1] = p[BT + 8];
        }
    }
}

// ---- epilogue of four consecutive output columns (shared by the main kernel and the split-K finisher) -----------
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// one output element (outputs whose row stride / width / pointers rule out 16-byte accesses)
__device__ __forceinline__ void gl_epilogue1(const GemmTask& t, int r, int c, float v) {
    v *= t.scale;
    float* cp = t.C + (size_t)r * t.ldc + c;
    if (t.epi == EPI_FWD) {
        if (t.bias) v += t.bias[c];
        float y;
        switch (t.act) {
        case ACT_RELU: y = fmaxf(v, 0.f); break;
        case ACT_ELU: y = elu_f(v); break;
        case ACT_SIN: y = sinf(v); t.out2[(size_t)r * t.ldout2 + c] = v; break;
        case ACT_TANH: y = tanhf(v); break;
        default: y = v;
        }
        *cp = y;
    } else if (t.epi == EPI_DX) {
        if (t.r1u) v += t.r1u[r] * t.r1v[c];
        if (t.act != ACT_NONE) {
            const float a = t.aux[(size_t)r * t.ldaux + c];
            switch (t.act) {
            case ACT_RELU: v = a > 0.f ? v : 0.f; break;
            case ACT_ELU: v *= elu_grad_from_out(a); break;
            case ACT_SIN: v *= cosf(a); break;
            case ACT_TANH: v *= (1.f - a * a); break;
            default: break;
            }
        }
        *cp = ((t.flags & FLAG_ACCUM) ? *cp : 0.f) + v;
    } else {
        *cp = ((t.flags & FLAG_ACCUM) ? *cp : 0.f) + v;
    }
}

// bias_pre: the four bias values of columns c .. c + 3, loaded by the caller BEFORE its store loop (a tile's store loop keeps its columns and walks the
// rows: loaded here, the bias was a dependent global round trip in every iteration -- eight per wave and 128-wide tile, ~15 us of a 60 us workgroup
// in the 788 MB forward of the nabla-mu head); nullptr: load it here
__device__ __forceinline__ f32x4 gl_bias4(const GemmTask& t, int c) {
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (t.epi == EPI_FWD && t.bias && !(t.flags & FLAG_SCALAR_C) && c + 4 <= t.Cn) b = ld4(t.bias + c);
    return b;
}
__device__ __forceinline__ void gl_epilogue4(const GemmTask& t, int r, int c, f32x4 v, const f32x4* bias_pre = nullptr) {
    if (t.flags & FLAG_SCALAR_C) {
#pragma unroll
        for (int q = 0; q < 4; ++q) if (c + q < t.Cn) gl_epilogue1(t, r, c + q, v[q]);
        return;
    }
    v *= t.scale;
    float* cp = t.C + (size_t)r * t.ldc + c;
    if (t.epi == EPI_FWD) {
        if (t.bias) v += bias_pre ? *bias_pre : ld4(t.bias + c);
        f32x4 y;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float x = v[q];
            switch (t.act) {
            case ACT_RELU: y[q] = fmaxf(x, 0.f); break;
            case ACT_ELU: y[q] = elu_f(x); break;
            case ACT_SIN: y[q] = sinf(x); break;
            case ACT_TANH: y[q] = tanhf(x); break;
            default: y[q] = x;
            }
        }
        if (t.act == ACT_SIN) st4(t.out2 + (size_t)r * t.ldout2 + c, v);
        st4(cp, y);
    } else if (t.epi == EPI_DX) {
        if (t.r1u) v += t.r1u[r] * ld4(t.r1v + c);
        if (t.act != ACT_NONE) {
            const f32x4 a = ld4(t.aux + (size_t)r * t.ldaux + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                switch (t.act) {
                case ACT_RELU: v[q] = a[q] > 0.f ? v[q] : 0.f; break;
                case ACT_ELU: v[q] *= elu_grad_from_out(a[q]); break;
                case ACT_SIN: v[q] *= cosf(a[q]); break;
                case ACT_TANH: v[q] *= (1.f - a[q] * a[q]); break;
                default: break;
                }
            }
        }
        if (t.flags & FLAG_ACCUM) v += ld4(cp);
        st4(cp, v);
    } else {            // EPI_DW
        if (t.flags & FLAG_ACCUM) v += ld4(cp);
        st4(cp, v);
    }
}

// bijective XCD remap: workgroups with equal (index % 8) share an XCD; give each class a contiguous run of tiles
__device__ __forceinline__ int gl_xcd_remap(int local, int n) {
    const int q = n >> 3, r = n & 7, x = local & 7, y = local >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
}

template <int BT, int LA, int LB>
__global__ __launch_bounds__(256, (BT == 128 ? 2 : 4)) void gemm_lds_kernel(GL_DIR_PARAMS, GemmBatch gb) {
    const int gdir[GEMM_MAX_TASKS] = {d0, d1, d2, d3, d4, d5, d6, d7};
    constexpr int WT = BT / 2, TT = WT / 16;
    constexpr int SA = GlTile<BT, LA>::FLOATS, SB = GlTile<BT, LB>::FLOATS;
    constexpr int EPF = 4 * WT * (WT + 4);
    constexpr int LDSF = (2 * (SA + SB) > EPF) ? 2 * (SA + SB) : EPF;
    __shared__ __attribute__((aligned(16))) float lds[LDSF];

    const int bid = blockIdx.x;
    int ti = 0;
#pragma unroll
    for (int q = 1; q < GEMM_MAX_TASKS; ++q) if (bid >= gdir[q]) ti = q;          // (preloaded directory: first tiles, INT_MAX beyond the last task)
    const GemmTask& t = gb.t[ti];
    const float* const pA = t.A; const float* const pB = t.B;
    const int lda = t.lda, ldb = t.ldb, R = t.R, Cn = t.Cn, K = t.K;
    const int tiles_c = t.tiles_c, splits = t.splits, kchunk = t.kchunk;
    const int tiles_r = (R + BT - 1) / BT;

    const int local = gl_xcd_remap(bid - t.tile_base, t.ntiles);
    const int per_split = tiles_r * tiles_c;
    const int split = local / per_split, rem = local - split * per_split;
    const int tc = rem / tiles_r, tr = rem - tc * tiles_r;
    const int r0 = tr * BT, c0 = tc * BT;
    const int kbeg = split * kchunk, kend = min(K, kbeg + kchunk);
    const int nk = (kend - kbeg + GL_BK - 1) / GL_BK;

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int i = lane & 15, kq = lane >> 4;

    f32x4 acc[TT][TT];
#pragma unroll
    for (int a = 0; a < TT; ++a)
#pragma unroll
        for (int b = 0; b < TT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float asum[TT];
#pragma unroll
    for (int a = 0; a < TT; ++a) asum[a] = 0.f;

    const bool vecA = !(t.flags & FLAG_SCALAR_A), vecB = !(t.flags & FLAG_SCALAR_B);
    f32x4 va[BT / 32], vb[BT / 32];
    gl_stage_load<BT, LA>(pA, lda, r0, R, kbeg, kend, vecA, va);
    gl_stage_load<BT, LB>(pB, ldb, c0, Cn, kbeg, kend, vecB, vb);
    gl_stage_write<BT, LA>(lds, va);
    gl_stage_write<BT, LB>(lds + SA, vb);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        // slice kt+1 (past the end: a clamped, zeroed re-read that nobody consumes) -- issued before the MFMAs of slice kt
        const int kn = kbeg + GL_BK * (kt + 1);
        gl_stage_load<BT, LA>(pA, lda, r0, R, kn, kend, vecA, va);
        gl_stage_load<BT, LB>(pB, ldb, c0, Cn, kn, kend, vecB, vb);
        const float* As = lds + cur * (SA + SB);
        const float* Bs = As + SA;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float fa[TT][2], fb[TT][2];
            gl_frag<BT, LA, TT>(As, wr * WT, i, kq, j, fa);
            gl_frag<BT, LB, TT>(Bs, wc * WT, i, kq, j, fb);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int a = 0; a < TT; ++a)
#pragma unroll
                    for (int b = 0; b < TT; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a][e], fb[b][e], acc[a][b], 0, 0, 0);
            if (LA == LD_COL) {
#pragma unroll
                for (int a = 0; a < TT; ++a) asum[a] += fa[a][0] + fa[a][1];
            }
        }
        float* Sn = lds + (cur ^ 1) * (SA + SB);
        gl_stage_write<BT, LA>(Sn, va);
        gl_stage_write<BT, LB>(Sn + SA, vb);
        __syncthreads();
    }

    const size_t C4w = (size_t)((Cn + 3) & ~3);
    // bias gradient (EPI_DW): row sums of operand A, taken from the fragments the column-0 waves of column-tile 0 consumed
    const bool has_bias = LA == LD_COL && t.epi == EPI_DW && (t.flags & FLAG_BIASGRAD) && tc == 0;
    if (has_bias && wc == 0) {
#pragma unroll
        for (int a = 0; a < TT; ++a) {
            float s = asum[a];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            const int r = r0 + wr * WT + a * 16 + lane;
            if (lane < 16 && r < R) {
                if (splits > 1) t.bslab[(size_t)split * R + r] = s;
                else t.out2[r] = s;
            }
        }
    }

    // accumulators -> this wave's LDS patch -> 16-byte row segments (coalesced stores, vector epilogue operands)
    float* E = lds + w * (WT * (WT + 4));
#pragma unroll
    for (int a = 0; a < TT; ++a)
#pragma unroll
        for (int b = 0; b < TT; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) E[(a * 16 + 4 * kq + q) * (WT + 4) + b * 16 + i] = acc[a][b][q];
    constexpr int LPR = WT / 4, RPI = 64 / LPR;
    const f32x4 bpre = splits > 1 ? (f32x4){0.f, 0.f, 0.f, 0.f} : gl_bias4(t, c0 + wc * WT + (lane % LPR) * 4);      // (this lane's columns: the same in every iteration)
#pragma unroll 4
    for (int it = 0; it < WT / RPI; ++it) {
        const int rr = it * RPI + lane / LPR, cc = (lane % LPR) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(E + rr * (WT + 4) + cc);
        const int r = r0 + wr * WT + rr, c = c0 + wc * WT + cc;
        if (r < R && c < Cn) {
            if (splits > 1) st4(t.slab + ((size_t)split * R + r) * C4w + c, v);       // partial tile: the finishing blocks add the slabs in split order
            else gl_epilogue4(t, r, c, v, &bpre);
        }
    }
}

// split-K finisher: out = epilogue(sum over splits, in split order); bias gradient likewise
__global__ __launch_bounds__(256) void gemm_lds_fin_kernel(GL_DIR_PARAMS, GemmBatch gb) {
    const int gdir[GEMM_MAX_TASKS] = {d0, d1, d2, d3, d4, d5, d6, d7};
    const int bid = blockIdx.x;
    int ti = -1;
#pragma unroll
    for (int q = 0; q < GEMM_MAX_TASKS; ++q) if (bid >= gdir[q]) ti = q;       // (preloaded directory: first finishing blocks; INT_MAX for tasks that have none)
    if (ti < 0) return;
    const GemmTask& t = gb.t[ti];
    const int lb = bid - t.fin_base;
    const int R = t.R, Cn = t.Cn, C4 = (Cn + 3) >> 2;          // slab rows are padded to a multiple of 4 floats
    const long long nvec = (long long)R * C4;
    const int nb_main = (int)((nvec + 255) / 256);
    if (lb < nb_main) {
        const long long e = (long long)lb * 256 + threadIdx.x;
        if (e >= nvec) return;
        const int r = (int)(e / C4), c = (int)(e - (long long)r * C4) * 4;
        const float* p = t.slab + ((size_t)r * C4) * 4 + c;
        const size_t stride = (size_t)R * C4 * 4;
        f32x4 v = ld4(p);
        for (int s = 1; s < t.splits; ++s) v += ld4(p + s * stride);
        gl_epilogue4(t, r, c, v);
    } else if (t.epi == EPI_DW && (t.flags & FLAG_BIASGRAD)) {
        const int r = (lb - nb_main) * 256 + threadIdx.x;
        if (r >= R) return;
        float s = t.bslab[r];
        for (int q = 1; q < t.splits; ++q) s += t.bslab[(size_t)q * R + r];
        t.out2[r] = s;
    }
}

// ================================================================================================
// bf16x3: the same 128 x 128 tile on the BF16 matrix pipe at fp32 accuracy, for the 200-GFLOP products of diffsrsac.
//
// gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the fp32-MFMA rate.  Every fp32 operand x is split EXACTLY into three
// bf16 pieces x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): 3 x 8 significand bits), and the
// product keeps the six partial products of weight >= 2^-16: a1b1 + (a1b2 + a2b1) + (a1b3 + a3b1 + a2b2).  Each is exact in
// the fp32 accumulator, the dropped ones are <= 2^-24 |a||b|: the result differs from the fp32-MFMA engine by a few ulp
// (tests/test_gemm_engines.py holds both engines to 1e-5 of a float64 product; typical 2e-7) at 6/16 of its matrix
// cycles.  The split runs once per staged element on the VALU (v_cvt_pk_bf16_f32 + shifts + subtracts, ~5.5 ops per
// element) while the co-resident workgroup's waves hold the matrix pipe: one LDS buffer (three bf16 images per operand,
// 80-byte rows: conflict-free 16-byte fragment reads), two barriers per 32-deep slice, two workgroups per CU.
// ================================================================================================
#include "x3.h"

#ifndef X3_STAGGER
#define X3_STAGGER 24                /* s_sleep units (64 cycles) */
#endif
#define X3_RSB 64                    /* image row stride in bytes: 32 bf16, no pad -- the four 16-byte chunks of a row are XOR-swizzled instead (x3r_off) */
#define X3_IMGB (128 * X3_RSB)       /* bytes per image */
// Byte offset of 16-byte chunk ch (eight consecutive k) of image row `row`.  A 256-byte LDS bank row holds four image rows; the chunk index is
// XORed with (row >> 2) & 3, so that the sixteen rows one ds_read_b128 lane group reads at one chunk index -- {0-3, 12-15, 20-27} or
// {4-11, 16-19, 28-31}: four rows of every residue mod 4, with four different (row >> 2) & 3 -- cover all sixteen 16-byte slots of the bank row,
// and the two rows a 16-lane ds_write_b64 group fills are two whole, adjacent 64-byte rows (the 32 write banks once each).  The padded form
// ([row][80 bytes]) read conflict-free too but every staging write was a 2-way conflict on four banks: SQ_LDS_BANK_CONFLICT 77.0 M cycles per
// launch of the Humanoid forward (profiles/r04_pmc_diffsrsac_humanoid_b2048.json); tools/lds_banks.py does the arithmetic for both.
__device__ __forceinline__ int x3r_off(int row, int ch) { return X3_RSB * row + 16 * (ch ^ ((row >> 2) & 3)); }

// 512 threads stage a 128 x 32 slice: every thread two row slots x four consecutive k (row-major operand) or four rows x
// two consecutive k (k-major operand): e[slot][..]
template <int LD>
__device__ __forceinline__ void x3_stage_load(const float* __restrict__ P, int ld, int base, int lim, int k0, int kend, float (&e)[8]) {
    if (LD == LD_ROW) {
        const int kc = (threadIdx.x & 7) * 4;
        const int k = min(k0 + kc, kend - 4);
        const bool ok = (k0 + kc) < kend;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = min(base + (int)(threadIdx.x >> 3) + 64 * j, lim - 1);
            const f32x4 x = *reinterpret_cast<const f32x4*>(P + (size_t)r * ld + k);
#pragma unroll
            for (int q = 0; q < 4; ++q) e[4 * j + q] = ok ? x[q] : 0.f;
        }
    } else {
        // k-major operand: one row, eight consecutive k per thread.  A wave-instruction reads 64 consecutive rows of one
        // k (256 contiguous bytes) and the row's eight k land in ONE 16-byte LDS write per image; the 4-rows-per-thread
        // form (16-byte global loads) scattered its LDS writes 320 bytes apart: 16-way bank conflicts, dW at 82 TF
        const int i = min(base + (int)(threadIdx.x & 127), lim - 1), kg = 8 * (threadIdx.x >> 7);
        const bool iok = (base + (int)(threadIdx.x & 127)) < lim;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = min(k0 + kg + q, kend - 1);
            const float x = P[(size_t)k * ld + i];
            e[q] = (iok && (k0 + kg + q) < kend) ? x : 0.f;
        }
    }
}
template <int LD>
__device__ __forceinline__ void x3_stage_write(unsigned char* __restrict__ img, const float (&e)[8]) {
    if (LD == LD_ROW) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (int)(threadIdx.x >> 3) + 64 * j, kc = (int)(threadIdx.x & 7) * 4;
            u32x2 hi, mid, lo;
            unsigned h, m, l;
            x3_split2(e[4 * j], e[4 * j + 1], h, m, l); hi[0] = h; mid[0] = m; lo[0] = l;
            x3_split2(e[4 * j + 2], e[4 * j + 3], h, m, l); hi[1] = h; mid[1] = m; lo[1] = l;
            unsigned char* p = img + x3r_off(row, kc >> 3) + 8 * ((kc >> 2) & 1);
            *reinterpret_cast<u32x2*>(p) = hi;
            *reinterpret_cast<u32x2*>(p + X3_IMGB) = mid;
            *reinterpret_cast<u32x2*>(p + 2 * X3_IMGB) = lo;
        }
    } else {
        const int row = (int)(threadIdx.x & 127), kc = 8 * (int)(threadIdx.x >> 7);
        u32x4 hi, mid, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned h, m, l;
            x3_split2(e[2 * q], e[2 * q + 1], h, m, l); hi[q] = h; mid[q] = m; lo[q] = l;
        }
        unsigned char* p = img + x3r_off(row, kc >> 3);
        *reinterpret_cast<u32x4*>(p) = hi;
        *reinterpret_cast<u32x4*>(p + X3_IMGB) = mid;
        *reinterpret_cast<u32x4*>(p + 2 * X3_IMGB) = lo;
    }
}

// 8 waves as 4 (rows) x 2 (columns): a wave owns 32 x 64 of the tile = two 32x32 accumulators; four waves per SIMD with
// two workgroups per CU, so split (VALU), fragment reads (LDS) and the matrix pipe overlap across waves
template <int LA, int LB>
__global__ __launch_bounds__(512, 4) void gemm_x3_kernel(GL_DIR_PARAMS, GemmBatch gb) {
    const int gdir[GEMM_MAX_TASKS] = {d0, d1, d2, d3, d4, d5, d6, d7};
    constexpr int BT = 128;
    constexpr int EPB = 8 * 32 * 68 * 4;                         // epilogue patches [32][68] per wave, bytes
    constexpr int STB = 6 * X3_IMGB;                             // six images
    constexpr int LDSB = EPB > STB ? EPB : STB;
    __shared__ __attribute__((aligned(16))) float lds[LDSB / 4];
    unsigned char* const L = reinterpret_cast<unsigned char*>(lds);

    const int bid = blockIdx.x;
    int ti = 0;
#pragma unroll
    for (int q = 1; q < GEMM_MAX_TASKS; ++q) if (bid >= gdir[q]) ti = q;          // (preloaded directory: first tiles, INT_MAX beyond the last task)
    const GemmTask& t = gb.t[ti];
    const float* const pA = t.A; const float* const pB = t.B;
    const int lda = t.lda, ldb = t.ldb, R = t.R, Cn = t.Cn, K = t.K;
    const int tiles_c = t.tiles_c, splits = t.splits, kchunk = t.kchunk;
    const int tiles_r = (R + BT - 1) / BT;
    const int local = gl_xcd_remap(bid - t.tile_base, t.ntiles);
    const int per_split = tiles_r * tiles_c;
    const int split = local / per_split, rem = local - split * per_split;
    const int tc = rem / tiles_r, tr = rem - tc * tiles_r;
    const int r0 = tr * BT, c0 = tc * BT;
    const int kbeg = split * kchunk, kend = min(K, kbeg + kchunk);
    const int nk = (kend - kbeg + GL_BK - 1) / GL_BK;

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int r32 = lane & 31, hh = lane >> 5;
    const bool want_bias = (LA == LD_COL) && t.epi == EPI_DW && (t.flags & FLAG_BIASGRAD) && tc == 0;

    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[b][q] = 0.f;
    float rs = 0.f;

    float ea[8], eb[8];
    x3_stage_load<LA>(pA, lda, r0, R, kbeg, kend, ea);
    x3_stage_load<LB>(pB, ldb, c0, Cn, kbeg, kend, eb);

    // The two workgroups of a CU alternate a VALU phase (split + LDS write) and a matrix phase; started together they
    // stay in step.  Workgroups are dealt one per CU before any CU gets its second (observed, speed only), so delaying
    // every second group of 256 by about one VALU phase starts the pair in anti-phase.
    if (X3_STAGGER && ((blockIdx.x >> 8) & 1)) __builtin_amdgcn_s_sleep(X3_STAGGER);

    // fragment of 16-deep block c: chunk 2 c + hh of this lane's row -- the swizzle term (row >> 2) & 3 is the same for rows r32, 32 + r32, ...
    const unsigned char* const fa = L + x3r_off(wr * 32 + r32, hh);
    const unsigned char* const fb = L + 3 * X3_IMGB + x3r_off(wc * 64 + r32, hh);
    const int fsw = x3r_off(r32, 2 + hh) - x3r_off(r32, hh);            // block 1 relative to block 0: +32 or -32 bytes

    for (int kt = 0; kt < nk; ++kt) {
        if (want_bias) rs += ((ea[0] + ea[1]) + (ea[2] + ea[3])) + ((ea[4] + ea[5]) + (ea[6] + ea[7]));
        x3_stage_write<LA>(L, ea);
        x3_stage_write<LB>(L + 3 * X3_IMGB, eb);
        __syncthreads();
        const int kn = kbeg + GL_BK * (kt + 1);
        x3_stage_load<LA>(pA, lda, r0, R, kn, kend, ea);
        x3_stage_load<LB>(pB, ldb, c0, Cn, kn, kend, eb);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bf16x8 a[3], b[2][3];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                a[m] = *reinterpret_cast<const bf16x8*>(fa + m * X3_IMGB + fsw * c);
                b[0][m] = *reinterpret_cast<const bf16x8*>(fb + m * X3_IMGB + fsw * c);
                b[1][m] = *reinterpret_cast<const bf16x8*>(fb + 32 * X3_RSB + m * X3_IMGB + fsw * c);
            }
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                f32x16 v = acc[y];
                v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[y][2], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[y][0], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[y][1], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[y][1], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[y][0], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[y][0], v, 0, 0, 0);
                acc[y] = v;
            }
        }
        __syncthreads();
    }

    // bias gradient: this thread's row (k-major A: row tid & 127, one of four k groups) -> LDS -> fixed-order sum
    if (want_bias) {
        float* part = lds;                                   // [128][4]
        part[(threadIdx.x & 127) * 4 + (threadIdx.x >> 7)] = rs;
        __syncthreads();
        if (threadIdx.x < 128) {
            const float* q = part + threadIdx.x * 4;
            const float s = (q[0] + q[1]) + (q[2] + q[3]);
            const int r = r0 + threadIdx.x;
            if (r < R) { if (splits > 1) t.bslab[(size_t)split * R + r] = s; else t.out2[r] = s; }
        }
        __syncthreads();
    }

    // accumulators (32x32 C/D map: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) -> LDS patch -> row segments
    float* E = lds + w * (32 * 68);
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int q = 0; q < 16; ++q) E[((q & 3) + 8 * (q >> 2) + 4 * hh) * 68 + y * 32 + r32] = acc[y][q];
    const f32x4 bpre = splits > 1 ? (f32x4){0.f, 0.f, 0.f, 0.f} : gl_bias4(t, c0 + wc * 64 + (lane & 15) * 4);      // (this lane's columns: the same in every iteration)
#pragma unroll 4
    for (int it = 0; it < 8; ++it) {
        const int rr = it * 4 + (lane >> 4), cc = (lane & 15) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(E + rr * 68 + cc);
        const int r = r0 + wr * 32 + rr, c = c0 + wc * 64 + cc;
        if (r < R && c < Cn) {
            if (splits > 1) st4(t.slab + ((size_t)split * R + r) * ((Cn + 3) & ~3) + c, v);
            else gl_epilogue4(t, r, c, v, &bpre);
        }
    }
}

// ================================================================================================
// bf16x3 for the k-major x k-major form (weight gradients: dW = G^T X, both operands stored [K][rows]) -- gemm_x3t_kernel.
//
// The MFMA wants, per lane, EIGHT CONSECUTIVE k of one row; a k-major operand has them K-strided in memory.  gemm_x3_kernel<LD_COL, LD_COL>
// transposed while staging (one row, eight k per thread: 4-byte global loads, 107 TF, no better than the fp32 tile -- which is why weight
// gradients stayed on fp32 MFMA at 111 TF).  gfx950 can transpose on the way OUT of LDS instead: ds_read_b64_tr_b16 hands every lane of a
// 16-lane group one COLUMN of a 4 (k) x 16 (row) block of 16-bit elements.  So the slice is staged as it lies in memory -- 16-byte global
// loads along the rows, split into three bf16 images [32 k][128 rows] (256-byte image rows, 16-byte chunks XOR-swizzled by k so that the
// 8-byte writes and the transposed reads are both conflict-free: off(k, ch) = 256 k + 16 (ch ^ (((k & 3) << 2) | ((k >> 2) & 3)))) -- and a
// fragment is two transposed reads per image.  Same tile (128 x 128, 8 waves 4 x 2, 32x32x16 MFMAs, six per product), same epilogue.
// ================================================================================================
typedef short x3t_s16x4 __attribute__((ext_vector_type(4)));
#define X3T_IMGB (32 * 256)          /* one image: 32 k x 128 rows x 2 bytes */
__device__ __forceinline__ unsigned x3t_off(int k, int ch) { return 256u * k + 16u * (ch ^ (((k & 3) << 2) | ((k >> 2) & 3))); }

// 512 threads stage a 32 (k) x 128 (rows) slice: thread -> (k = idx / 32, four rows 4 (idx % 32)), two slots
__device__ __forceinline__ void x3t_stage_load(const float* __restrict__ P, int ld, int base, int lim, int k0, int kend, f32x4 (&e)[2]) {
    const int c4 = (int)(threadIdx.x & 31) * 4;
    const int i = min(base + c4, lim - 4);
    const bool iok = (base + c4) < lim;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int kk = (int)(threadIdx.x >> 5) + 16 * j;
        const int k = min(k0 + kk, kend - 1);
        const f32x4 x = *reinterpret_cast<const f32x4*>(P + (size_t)k * ld + i);
        e[j] = (iok && (k0 + kk) < kend) ? x : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}
__device__ __forceinline__ void x3t_stage_write(unsigned char* __restrict__ img, const f32x4 (&e)[2]) {
    const int c4 = (int)(threadIdx.x & 31) * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int kk = (int)(threadIdx.x >> 5) + 16 * j;
        u32x2 hi, mid, lo;
        unsigned h, m, l;
        x3_split2(e[j][0], e[j][1], h, m, l); hi[0] = h; mid[0] = m; lo[0] = l;
        x3_split2(e[j][2], e[j][3], h, m, l); hi[1] = h; mid[1] = m; lo[1] = l;
        unsigned char* p = img + x3t_off(kk, c4 >> 3) + 8 * ((c4 >> 2) & 1);
        *reinterpret_cast<u32x2*>(p) = hi;
        *reinterpret_cast<u32x2*>(p + X3T_IMGB) = mid;
        *reinterpret_cast<u32x2*>(p + 2 * X3T_IMGB) = lo;
    }
}
// one transposed read: 4 consecutive k of this lane's row (LDS byte address `a` = this lane's share of the block: see x3t_addr)
template <int OFF> __device__ __forceinline__ x3t_s16x4 x3t_rd(unsigned a) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) x3t_s16x4*>((uintptr_t)(a + OFF)));
}
// the eight k (16 c + 8 kg ..) of this lane's row as one MFMA operand: k-blocks h = 0, 1 at addresses a0 / a1 (image and c by immediate offset)
template <int OFF> __device__ __forceinline__ bf16x8 x3t_frag(unsigned a0, unsigned a1) {
    const x3t_s16x4 lo = x3t_rd<OFF>(a0), hi = x3t_rd<OFF>(a1);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
// address a lane supplies for the block (k rows kq .. kq + 3, 16 matrix rows starting at 8 * chunk0): lane 4 q + p of its 16-lane group gives
// row kq + q, columns 4 p .. 4 p + 3 (cdna_hip_programming.md T10)
__device__ __forceinline__ unsigned x3t_addr(unsigned lds_base, int kq, int chunk0) {
    const int li = threadIdx.x & 15, q = li >> 2, pp = li & 3;
    return lds_base + x3t_off(kq + q, chunk0 + (pp >> 1)) + 8u * (pp & 1);
}

// LA = LD_COL: both operands k-major (weight gradients).  LA = LD_ROW: A row-major [R, K] -- staged and read as in gemm_x3_kernel ([row][80-byte]
// images, ds_read_b128 fragments) -- and only B k-major through the transposed reads (dX = G W with W stored [K = out features][Cn = in features]).
template <int LA>
__global__ __launch_bounds__(512, 4) void gemm_x3t_kernel(GL_DIR_PARAMS, GemmBatch gb) {
    const int gdir[GEMM_MAX_TASKS] = {d0, d1, d2, d3, d4, d5, d6, d7};
    constexpr int BT = 128;
    constexpr int EPB = 8 * 32 * 68 * 4;                         // epilogue patches [32][68] per wave, bytes
    constexpr int AIMG = LA == LD_ROW ? X3_IMGB : X3T_IMGB;      // bytes per A image
    constexpr int STB = 3 * AIMG + 3 * X3T_IMGB;                 // six images
    constexpr int LDSB = EPB > STB ? EPB : STB;
    __shared__ __attribute__((aligned(16))) float lds[LDSB / 4];
    unsigned char* const L = reinterpret_cast<unsigned char*>(lds);
    const unsigned Lb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)L;

    const int bid = blockIdx.x;
    int ti = 0;
#pragma unroll
    for (int q = 1; q < GEMM_MAX_TASKS; ++q) if (bid >= gdir[q]) ti = q;          // (preloaded directory: first tiles, INT_MAX beyond the last task)
    const GemmTask& t = gb.t[ti];
    const float* const pA = t.A; const float* const pB = t.B;
    const int lda = t.lda, ldb = t.ldb, R = t.R, Cn = t.Cn, K = t.K;
    const int tiles_c = t.tiles_c, splits = t.splits, kchunk = t.kchunk;
    const int tiles_r = (R + BT - 1) / BT;
    const int local = gl_xcd_remap(bid - t.tile_base, t.ntiles);
    const int per_split = tiles_r * tiles_c;
    const int split = local / per_split, rem = local - split * per_split;
    const int tc = rem / tiles_r, tr = rem - tc * tiles_r;
    const int r0 = tr * BT, c0 = tc * BT;
    const int kbeg = split * kchunk, kend = min(K, kbeg + kchunk);
    const int nk = (kend - kbeg + GL_BK - 1) / GL_BK;

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int r32 = lane & 31, hh = lane >> 5, g1 = (lane >> 4) & 1;
    const bool want_bias = LA == LD_COL && t.epi == EPI_DW && (t.flags & FLAG_BIASGRAD) && tc == 0;

    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[b][q] = 0.f;
    f32x4 rs = {0.f, 0.f, 0.f, 0.f};

    f32x4 ea[2], eb[2];
    float ear[8];                                               // (row-major A: gemm_x3_kernel's staging registers)
    if constexpr (LA == LD_ROW) x3_stage_load<LD_ROW>(pA, lda, r0, R, kbeg, kend, ear);
    else x3t_stage_load(pA, lda, r0, R, kbeg, kend, ea);
    x3t_stage_load(pB, ldb, c0, Cn, kbeg, kend, eb);
    if (X3_STAGGER && ((blockIdx.x >> 8) & 1)) __builtin_amdgcn_s_sleep(X3_STAGGER);        // (anti-phase start of a CU's two workgroups: gemm_x3_kernel)

    // transposed-read addresses: k-block h of this lane's k group (k = 8 hh + 4 h within a 16-deep block c; c and the image by immediate offset)
    const unsigned aA0 = x3t_addr(Lb, 8 * hh, wr * 4 + 2 * g1), aA1 = x3t_addr(Lb, 8 * hh + 4, wr * 4 + 2 * g1);
    const unsigned aB00 = x3t_addr(Lb + 3 * AIMG, 8 * hh, wc * 8 + 2 * g1), aB01 = x3t_addr(Lb + 3 * AIMG, 8 * hh + 4, wc * 8 + 2 * g1);
    const unsigned aB10 = x3t_addr(Lb + 3 * AIMG, 8 * hh, wc * 8 + 4 + 2 * g1), aB11 = x3t_addr(Lb + 3 * AIMG, 8 * hh + 4, wc * 8 + 4 + 2 * g1);
    const unsigned char* const far = L + x3r_off(wr * 32 + r32, hh);                       // row-major A fragments (ds_read_b128; swizzled chunks: x3r_off)
    const int fsw = x3r_off(r32, 2 + hh) - x3r_off(r32, hh);

    for (int kt = 0; kt < nk; ++kt) {
        if (want_bias) rs += ea[0] + ea[1];
        if constexpr (LA == LD_ROW) x3_stage_write<LD_ROW>(L, ear);
        else x3t_stage_write(L, ea);
        x3t_stage_write(L + 3 * AIMG, eb);
        __syncthreads();
        const int kn = kbeg + GL_BK * (kt + 1);
        if constexpr (LA == LD_ROW) x3_stage_load<LD_ROW>(pA, lda, r0, R, kn, kend, ear);
        else x3t_stage_load(pA, lda, r0, R, kn, kend, ea);
        x3t_stage_load(pB, ldb, c0, Cn, kn, kend, eb);
#define X3T_BLOCK(C)                                                                                                          \
        {                                                                                                                     \
            bf16x8 a[3], b[2][3];                                                                                             \
            if constexpr (LA == LD_ROW) {                                                                                     \
                _Pragma("unroll") for (int m = 0; m < 3; ++m) a[m] = *reinterpret_cast<const bf16x8*>(far + m * X3_IMGB + fsw * (C)); \
            } else {                                                                                                          \
                a[0] = x3t_frag<(C) * 4096>(aA0, aA1); a[1] = x3t_frag<(C) * 4096 + X3T_IMGB>(aA0, aA1);                       \
                a[2] = x3t_frag<(C) * 4096 + 2 * X3T_IMGB>(aA0, aA1);                                                         \
            }                                                                                                                 \
            b[0][0] = x3t_frag<(C) * 4096>(aB00, aB01); b[0][1] = x3t_frag<(C) * 4096 + X3T_IMGB>(aB00, aB01);                 \
            b[0][2] = x3t_frag<(C) * 4096 + 2 * X3T_IMGB>(aB00, aB01);                                                        \
            b[1][0] = x3t_frag<(C) * 4096>(aB10, aB11); b[1][1] = x3t_frag<(C) * 4096 + X3T_IMGB>(aB10, aB11);                 \
            b[1][2] = x3t_frag<(C) * 4096 + 2 * X3T_IMGB>(aB10, aB11);                                                        \
            _Pragma("unroll") for (int y = 0; y < 2; ++y) {                                                                   \
                f32x16 v = acc[y];                                                                                            \
                v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[y][2], v, 0, 0, 0);                                       \
                v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[y][0], v, 0, 0, 0);                                       \
                v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[y][1], v, 0, 0, 0);                                       \
                v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[y][1], v, 0, 0, 0);                                       \
                v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[y][0], v, 0, 0, 0);                                       \
                v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[y][0], v, 0, 0, 0);                                       \
                acc[y] = v;                                                                                                   \
            }                                                                                                                 \
        }
        X3T_BLOCK(0) X3T_BLOCK(1)
#undef X3T_BLOCK
        __syncthreads();
    }

    // bias gradient = row sums of operand A: this thread holds four rows (4 (tid % 32) ..) over its k slots -> LDS -> fixed-order sum over the 16 slots
    if (want_bias) {
        float* part = lds;                                   // [128 rows][16 k slots]
        const int c4 = (int)(threadIdx.x & 31) * 4, ks = (int)(threadIdx.x >> 5);
#pragma unroll
        for (int q = 0; q < 4; ++q) part[(c4 + q) * 16 + ks] = rs[q];
        __syncthreads();
        if (threadIdx.x < 128) {
            const float* q = part + threadIdx.x * 16;
            float s0 = 0.f;
#pragma unroll
            for (int z = 0; z < 16; ++z) s0 += q[z];
            const int r = r0 + threadIdx.x;
            if (r < R) { if (splits > 1) t.bslab[(size_t)split * R + r] = s0; else t.out2[r] = s0; }
        }
        __syncthreads();
    }

    // accumulators (32x32 C/D map: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) -> LDS patch -> row segments
    float* E = lds + w * (32 * 68);
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int q = 0; q < 16; ++q) E[((q & 3) + 8 * (q >> 2) + 4 * hh) * 68 + y * 32 + r32] = acc[y][q];
    const f32x4 bpre = splits > 1 ? (f32x4){0.f, 0.f, 0.f, 0.f} : gl_bias4(t, c0 + wc * 64 + (lane & 15) * 4);      // (this lane's columns: the same in every iteration)
#pragma unroll 4
    for (int it = 0; it < 8; ++it) {
        const int rr = it * 4 + (lane >> 4), cc = (lane & 15) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(E + rr * 68 + cc);
        const int r = r0 + wr * 32 + rr, c = c0 + wc * 64 + cc;
        if (r < R && c < Cn) {
            if (splits > 1) st4(t.slab + ((size_t)split * R + r) * ((Cn + 3) & ~3) + c, v);
            else gl_epilogue4(t, r, c, v, &bpre);
        }
    }
}

#include "gemm_x3w.h"

// ================================================================================================
// gemm_x3s_kernel: the 64 x 64 tile on the bf16 pipe (bf16x3) -- for the layers whose outputs are too small for 128-wide tiles (ctrlsac at
// M = 256, spedersac at M = 2048 x 512, diffsrsac at HalfCheetah dims) and that ran on the fp32 64-wide tile at 38-60 TF: one wave per SIMD
// there issues a 16x16x4 fp32 MFMA every 52 cycles, while a 32x32x16 bf16 MFMA does 16x the work in 32.  256 threads = 4 waves (2 x 2), a wave
// owns 32 x 32 = ONE f32x16 accumulator (<= 128 VGPRs: four workgroups per CU, whose split / read / MFMA phases overlap), 32-deep slices, the
// exact three-way split while staging.  Row-major operands as in gemm_x3_kernel ([row][80-byte] images, ds_read_b128 fragments); k-major
// operands as in gemm_x3t_kernel (staged as they lie, [32 k][64 rows] images of 128-byte rows, ds_read_b64_tr_b16) with the chunk swizzle
// ch ^ (((k >> 1) & 1) << 2): the four k-rows a 32-lane half reads then sit on four different 16-bank groups.
// ================================================================================================
#define X3S_RIMGB (64 * X3_RSB)      /* row-major image: 64 rows x 64 bytes, chunks swizzled (x3r_off) */
#define X3S_TIMGB (32 * 128)         /* k-major image: 32 k x 64 rows x 2 bytes */
__device__ __forceinline__ unsigned x3s_toff(int k, int ch) { return 128u * k + 16u * (ch ^ (((k >> 1) & 1) << 2)); }

// 256 threads stage a 64 x 32 slice.  Row-major: two row slots x four consecutive k; k-major: (k = tid / 16 + 16 j, four rows 4 (tid % 16))
// VEC (compile time: a run-time choice puts the prefetch loads into a control-flow diamond, and hipcc drains vmcnt at its merge -- the loads then no
// longer overlap the MFMAs: measured -10 % on every workload): 1 = 16-byte loads at 16-byte-aligned addresses, lengths multiples of four;
// 2 = 16-byte loads at ANY 4-byte-aligned address (rows of 119 floats, operands that start inside another buffer: global loads only need dword
// alignment on this target) with the tail of a length that is no multiple of four fetched as the LAST four elements and shifted into place --
// no access ever leaves the row; 0 = clamped 4-byte loads (four times the load instructions: slower than the fp32 tile, kept for reference)
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
// y[q] = x[q + d] (0 beyond the vector): d = how far the load was pulled back to stay inside the row (>= 4: nothing valid)
__device__ __forceinline__ f32x4 x3s_shift(const f32x4 x, int d) {
    // two select stages on the bits of d (plain v_cndmask: a chain of comparisons on d became a jump table, and a branch between the prefetch loads
    // and the MFMAs drains vmcnt)
    const bool b0 = (d & 1) != 0, b1 = (d & 2) != 0, z = d >= 4;
    const float t0 = b0 ? x[1] : x[0], t1 = b0 ? x[2] : x[1], t2 = b0 ? x[3] : x[2], t3 = b0 ? 0.f : x[3];
    f32x4 y;
    y[0] = b1 ? t2 : t0; y[1] = b1 ? t3 : t1; y[2] = b1 ? 0.f : t2; y[3] = b1 ? 0.f : t3;
    y[0] = z ? 0.f : y[0]; y[1] = z ? 0.f : y[1]; y[2] = z ? 0.f : y[2]; y[3] = z ? 0.f : y[3];
    return y;
}
// (16-byte load of a group that is only 4-byte aligned in the any-alignment forms)
template <int VEC> __device__ __forceinline__ void x3s_ld16(f32x4& dst, const float* p) {
    if constexpr (VEC == 2) dst = *reinterpret_cast<const f32x4u*>(p);
    else dst = *reinterpret_cast<const f32x4*>(p);
}
// PHASE 0: issue the loads, raw, into e.  PHASE 1: what depends on the loaded VALUES -- zero fill of the K tail / the edge, the shift of an unaligned
// group -- in place, with the same arguments, right before the split.  (Done at load time these selects sat between the loads and the second barrier:
// every slice waited for the loads it had just issued, whatever the prefetch depth.)
template <int VEC, int PHASE>
__device__ __forceinline__ void x3s_load_row(const float* __restrict__ P, int ld, int base, int lim, int k0, int kend, f32x4 (&e)[2]) {
    const int kc = (threadIdx.x & 7) * 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if constexpr (VEC == 2) {
        const int kl = min(k0 + kc, kend - 4), d = k0 + kc - kl;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (PHASE == 0) x3s_ld16<2>(e[j], P + (size_t)min(base + (int)(threadIdx.x >> 3) + 32 * j, lim - 1) * ld + kl);
            else e[j] = x3s_shift(e[j], d);
        }
    } else if constexpr (VEC == 1) {
        const int k = min(k0 + kc, kend - 4);
        const bool ok = (k0 + kc) < kend;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (PHASE == 0) x3s_ld16<1>(e[j], P + (size_t)min(base + (int)(threadIdx.x >> 3) + 32 * j, lim - 1) * ld + k);
            else e[j] = ok ? e[j] : zero;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float* p = P + (size_t)min(base + (int)(threadIdx.x >> 3) + 32 * j, lim - 1) * ld;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (PHASE == 0) e[j][q] = p[min(k0 + kc + q, kend - 1)];
                else e[j][q] = (k0 + kc + q) < kend ? e[j][q] : 0.f;
            }
        }
    }
}
template <int VEC, int PHASE>
__device__ __forceinline__ void x3s_load_col(const float* __restrict__ P, int ld, int base, int lim, int k0, int kend, f32x4 (&e)[2]) {
    const int c4 = (int)(threadIdx.x & 15) * 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if constexpr (VEC == 2) {
        const int il = min(base + c4, lim - 4), d = base + c4 - il;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kk = (int)(threadIdx.x >> 4) + 16 * j;
            if constexpr (PHASE == 0) {
                const int k = min(k0 + kk, kend - 1);
                x3s_ld16<2>(e[j], P + (size_t)k * ld + il);
            } else {
                const f32x4 x = x3s_shift(e[j], d);
                e[j] = (k0 + kk) < kend ? x : zero;
            }
        }
    } else if constexpr (VEC == 1) {
        const int i = min(base + c4, lim - 4);
        const bool iok = (base + c4) < lim;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kk = (int)(threadIdx.x >> 4) + 16 * j;
            if constexpr (PHASE == 0) {
                const int k = min(k0 + kk, kend - 1);
                x3s_ld16<1>(e[j], P + (size_t)k * ld + i);
            } else e[j] = (iok && (k0 + kk) < kend) ? e[j] : zero;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kk = (int)(threadIdx.x >> 4) + 16 * j;
            const float* p = P + (size_t)min(k0 + kk, kend - 1) * ld;
            const bool kok = (k0 + kk) < kend;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (PHASE == 0) e[j][q] = p[min(base + c4 + q, lim - 1)];
                else e[j][q] = (kok && (base + c4 + q) < lim) ? e[j][q] : 0.f;
            }
        }
    }
}
__device__ __forceinline__ void x3s_write_row(unsigned char* __restrict__ img, const f32x4 (&e)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (int)(threadIdx.x >> 3) + 32 * j, kc = (int)(threadIdx.x & 7) * 4;
        u32x2 hi, mid, lo;
        unsigned h, m, l;
        x3_split2(e[j][0], e[j][1], h, m, l); hi[0] = h; mid[0] = m; lo[0] = l;
        x3_split2(e[j][2], e[j][3], h, m, l); hi[1] = h; mid[1] = m; lo[1] = l;
        unsigned char* p = img + x3r_off(row, kc >> 3) + 8 * ((kc >> 2) & 1);
        *reinterpret_cast<u32x2*>(p) = hi;
        *reinterpret_cast<u32x2*>(p + X3S_RIMGB) = mid;
        *reinterpret_cast<u32x2*>(p + 2 * X3S_RIMGB) = lo;
    }
}
template <int VEC>
__device__ __forceinline__ void x3s_load_col(const float* __restrict__ P, int ld, int base, int lim, int k0, int kend, f32x4 (&e)[2]) {
    const int c4 = (int)(threadIdx.x & 15) * 4;
    if constexpr (VEC == 2) {
        const int il = min(base + c4, lim - 4), d = base + c4 - il;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kk = (int)(threadIdx.x >> 4) + 16 * j;
            const int k = min(k0 + kk, kend - 1);
            const f32x4 x = x3s_shift(*reinterpret_cast<const f32x4u*>(P + (size_t)k * ld + il), d);
            e[j] = (k0 + kk) < kend ? x : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    } else if constexpr (VEC == 1) {
        const int i = min(base + c4, lim - 4);
        const bool iok = (base + c4) < lim;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kk = (int)(threadIdx.x >> 4) + 16 * j;
            const int k = min(k0 + kk, kend - 1);
            const f32x4 x = *reinterpret_cast<const f32x4*>(P + (size_t)k * ld + i);
            e[j] = (iok && (k0 + kk) < kend) ? x : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kk = (int)(threadIdx.x >> 4) + 16 * j;
            const float* p = P + (size_t)min(k0 + kk, kend - 1) * ld;
            const bool kok = (k0 + kk) < kend;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float y = p[min(base + c4 + q, lim - 1)]; e[j][q] = (kok && (base + c4 + q) < lim) ? y : 0.f; }
        }
    }
}
__device__ __forceinline__ void x3s_write_col(unsigned char* __restrict__ img, const f32x4 (&e)[2]) {
    const int c4 = (int)(threadIdx.x & 15) * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int kk = (int)(threadIdx.x >> 4) + 16 * j;
        u32x2 hi, mid, lo;
        unsigned h, m, l;
        x3_split2(e[j][0], e[j][1], h, m, l); hi[0] = h; mid[0] = m; lo[0] = l;
        x3_split2(e[j][2], e[j][3], h, m, l); hi[1] = h; mid[1] = m; lo[1] = l;
        unsigned char* p = img + x3s_toff(kk, c4 >> 3) + 8 * ((c4 >> 2) & 1);
        *reinterpret_cast<u32x2*>(p) = hi;
        *reinterpret_cast<u32x2*>(p + X3S_TIMGB) = mid;
        *reinterpret_cast<u32x2*>(p + 2 * X3S_TIMGB) = lo;
    }
}
__device__ __forceinline__ unsigned x3s_taddr(unsigned lds_base, int kq, int chunk0) {
    const int li = threadIdx.x & 15, q = li >> 2, pp = li & 3;
    return lds_base + x3s_toff(kq + q, chunk0 + (pp >> 1)) + 8u * (pp & 1);
}

template <int LA, int LB, int VEC>
__global__ __launch_bounds__(256, 3) void gemm_x3s_kernel(GL_DIR_PARAMS, GemmBatch gb) {      // (48 KB of LDS: three workgroups per CU)
    const int gdir[GEMM_MAX_TASKS] = {d0, d1, d2, d3, d4, d5, d6, d7};
    constexpr int BT = 64;
    constexpr int AIMG = LA == LD_ROW ? X3S_RIMGB : X3S_TIMGB, BIMG = LB == LD_ROW ? X3S_RIMGB : X3S_TIMGB;
    constexpr int EPB = 4 * 32 * 36 * 4;                         // epilogue patches [32][36] per wave, bytes
    constexpr int STB = 3 * AIMG + 3 * BIMG;
    static_assert(EPB <= STB, "the epilogue patches live in stage 0");
    // TWO LDS stages (two arrays: the compiler then knows that the fragment reads of one and the staging writes of the other do not alias), ONE
    // barrier per slice: slice kt is multiplied out of one stage while slice kt + 1 is split into the other.
    __shared__ __attribute__((aligned(16))) float lds[STB / 4], lds1[STB / 4];
    unsigned char* const L0 = reinterpret_cast<unsigned char*>(lds);
    unsigned char* const L1 = reinterpret_cast<unsigned char*>(lds1);
    const unsigned Lb0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)L0;
    const unsigned Lb1 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)L1;

    const int bid = blockIdx.x;
    int ti = 0;
#pragma unroll
    for (int q = 1; q < GEMM_MAX_TASKS; ++q) if (bid >= gdir[q]) ti = q;          // (preloaded directory: first tiles, INT_MAX beyond the last task)
    const GemmTask& t = gb.t[ti];
    const float* const pA = t.A; const float* const pB = t.B;
    const int lda = t.lda, ldb = t.ldb, R = t.R, Cn = t.Cn, K = t.K;
    const int tiles_c = t.tiles_c, splits = t.splits, kchunk = t.kchunk;
    const int tiles_r = (R + BT - 1) / BT;
    const int local = gl_xcd_remap(bid - t.tile_base, t.ntiles);
    const int per_split = tiles_r * tiles_c;
    const int split = local / per_split, rem = local - split * per_split;
    const int tc = rem / tiles_r, tr = rem - tc * tiles_r;
    const int r0 = tr * BT, c0 = tc * BT;
    const int kbeg = split * kchunk, kend = min(K, kbeg + kchunk);
    const int nk = (kend - kbeg + GL_BK - 1) / GL_BK;

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int r32 = lane & 31, hh = lane >> 5, g1 = (lane >> 4) & 1;
    const bool want_bias = LA == LD_COL && t.epi == EPI_DW && (t.flags & FLAG_BIASGRAD) && tc == 0;

    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    f32x4 rs = {0.f, 0.f, 0.f, 0.f};

    // TWO sets of staging registers, loads two slices ahead of their split: set z holds the slices of parity z.  (One set, loaded behind the barrier
    // and split before the next one, left a load's whole latency in every slice wherever fewer than four workgroups share a CU.)
    f32x4 ear[2][2], ebr[2][2];
    f32x4 eac[2][2], ebc[2][2];
#define X3S_LOAD(Z, KS)                                                                                                       \
    {                                                                                                                         \
        const int kz_ = kbeg + GL_BK * (KS);                                                                                  \
        if constexpr (LA == LD_ROW) x3s_load_row<VEC, 0>(pA, lda, r0, R, kz_, kend, ear[Z]); else x3s_load_col<VEC, 0>(pA, lda, r0, R, kz_, kend, eac[Z]); \
        if constexpr (LB == LD_ROW) x3s_load_row<VEC, 0>(pB, ldb, c0, Cn, kz_, kend, ebr[Z]); else x3s_load_col<VEC, 0>(pB, ldb, c0, Cn, kz_, kend, ebc[Z]); \
    }
    // what depends on the loaded values (K tail / edge zero fill, unaligned shift), then split set Z = slice KS into the stage at LW
#define X3S_SPLIT(Z, KS, LW)                                                                                                  \
    {                                                                                                                         \
        const int kz_ = kbeg + GL_BK * (KS);                                                                                  \
        if constexpr (LA == LD_ROW) x3s_load_row<VEC, 1>(pA, lda, r0, R, kz_, kend, ear[Z]); else x3s_load_col<VEC, 1>(pA, lda, r0, R, kz_, kend, eac[Z]); \
        if constexpr (LB == LD_ROW) x3s_load_row<VEC, 1>(pB, ldb, c0, Cn, kz_, kend, ebr[Z]); else x3s_load_col<VEC, 1>(pB, ldb, c0, Cn, kz_, kend, ebc[Z]); \
        if (want_bias) rs += eac[Z][0] + eac[Z][1];                                                                           \
        if constexpr (LA == LD_ROW) x3s_write_row(LW, ear[Z]); else x3s_write_col(LW, eac[Z]);                                \
        if constexpr (LB == LD_ROW) x3s_write_row((LW) + 3 * AIMG, ebr[Z]); else x3s_write_col((LW) + 3 * AIMG, ebc[Z]);      \
    }
    X3S_LOAD(0, 0) X3S_LOAD(1, 1)
    X3S_SPLIT(0, 0, L0)
    X3S_LOAD(0, 2)

    const int foA = x3r_off(wr * 32 + r32, hh), foB = 3 * AIMG + x3r_off(wc * 32 + r32, hh);
    const int fsw = x3r_off(r32, 2 + hh) - x3r_off(r32, hh);            // block 1 relative to block 0 (swizzled chunks)
    const unsigned tA0 = x3s_taddr(0, 8 * hh, wr * 4 + 2 * g1), tA1 = x3s_taddr(0, 8 * hh + 4, wr * 4 + 2 * g1);
    const unsigned tB0 = x3s_taddr(3 * AIMG, 8 * hh, wc * 4 + 2 * g1), tB1 = x3s_taddr(3 * AIMG, 8 * hh + 4, wc * 4 + 2 * g1);

#define X3S_BLOCK(C, LR, LBR)                                                                                                 \
        {                                                                                                                     \
            bf16x8 a[3], b[3];                                                                                                \
            if constexpr (LA == LD_ROW) {                                                                                     \
                _Pragma("unroll") for (int m = 0; m < 3; ++m) a[m] = *reinterpret_cast<const bf16x8*>((LR) + foA + m * X3S_RIMGB + fsw * (C)); \
            } else {                                                                                                          \
                a[0] = x3t_frag<(C) * 2048>((LBR) + tA0, (LBR) + tA1); a[1] = x3t_frag<(C) * 2048 + X3S_TIMGB>((LBR) + tA0, (LBR) + tA1); \
                a[2] = x3t_frag<(C) * 2048 + 2 * X3S_TIMGB>((LBR) + tA0, (LBR) + tA1);                                         \
            }                                                                                                                 \
            if constexpr (LB == LD_ROW) {                                                                                     \
                _Pragma("unroll") for (int m = 0; m < 3; ++m) b[m] = *reinterpret_cast<const bf16x8*>((LR) + foB + m * X3S_RIMGB + fsw * (C)); \
            } else {                                                                                                          \
                b[0] = x3t_frag<(C) * 2048>((LBR) + tB0, (LBR) + tB1); b[1] = x3t_frag<(C) * 2048 + X3S_TIMGB>((LBR) + tB0, (LBR) + tB1); \
                b[2] = x3t_frag<(C) * 2048 + 2 * X3S_TIMGB>((LBR) + tB0, (LBR) + tB1);                                         \
            }                                                                                                                 \
            f32x16 v = acc;                                                                                                   \
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], v, 0, 0, 0);                                              \
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], v, 0, 0, 0);                                              \
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], v, 0, 0, 0);                                              \
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], v, 0, 0, 0);                                              \
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], v, 0, 0, 0);                                              \
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], v, 0, 0, 0);                                              \
            acc = v;                                                                                                          \
        }
    // one slice: multiply slice KT out of the stage LR while set Z (slice KT + 1) is split into LW and refilled with slice KT + 3
#define X3S_ITER(Z, KT, LR, LBR, LW)                                                                                          \
    {                                                                                                                         \
        __syncthreads();                                                                                                      \
        X3S_SPLIT(Z, (KT) + 1, LW)                                                                                            \
        X3S_LOAD(Z, (KT) + 3)                                                                                                 \
        X3S_BLOCK(0, LR, LBR) X3S_BLOCK(1, LR, LBR)                                                                           \
    }
    // (ONE basic block per pair of slices: with a branch between the two the compiler's wait-count pass waited with vmcnt(0) in the second -- for the
    // loads the first had just issued.  A slice past the end of an odd chunk multiplies the zeros its own K-tail fill produces.)
    for (int kt = 0; kt < nk; kt += 2) {
        X3S_ITER(1, kt, L0, Lb0, L1)
        X3S_ITER(0, kt + 1, L1, Lb1, L0)
    }
#undef X3S_ITER
#undef X3S_BLOCK
#undef X3S_SPLIT
#undef X3S_LOAD
    __syncthreads();                                             // (the epilogue's patches and the bias sums reuse stage 0)

    // split-K without a finishing launch (FLAG_FIN_INLINE): every split workgroup writes its partial tile THROUGH to memory (the XCDs' L2s are
    // not coherent with each other inside a launch), takes a ticket on the tile's counter, and the LAST one to arrive sums the slabs IN SPLIT ORDER
    // (its own included, from memory: the arithmetic of gemm_lds_fin_kernel, bit for bit) and runs the epilogue.  The counters sit behind the
    // task's slabs, zero between launches (the last arrival resets its own).
    const bool inl = splits > 1 && (t.flags & FLAG_FIN_INLINE);
    const int C4p = (Cn + 3) & ~3;
    if (want_bias) {       // row sums of the k-major A: this thread holds rows 4 (tid % 16) .. over its 16 k slots
        float* part = lds;                                   // [64 rows][16 k slots]
        const int c4 = (int)(threadIdx.x & 15) * 4, ks = (int)(threadIdx.x >> 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) part[(c4 + q) * 16 + ks] = rs[q];
        __syncthreads();
        if (threadIdx.x < 64) {
            const float* q = part + threadIdx.x * 16;
            float s0 = 0.f;
#pragma unroll
            for (int z = 0; z < 16; ++z) s0 += q[z];
            const int r = r0 + threadIdx.x;
            if (r < R) { if (inl) dp_store1(t.bslab + (size_t)split * R + r, s0); else if (splits > 1) t.bslab[(size_t)split * R + r] = s0; else t.out2[r] = s0; }
        }
        __syncthreads();
    }

    // accumulator (32x32 C/D map: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) -> LDS patch -> row segments
    float* E = lds + w * (32 * 36);
#pragma unroll
    for (int q = 0; q < 16; ++q) E[((q & 3) + 8 * (q >> 2) + 4 * hh) * 36 + r32] = acc[q];
    const f32x4 bpre = splits > 1 ? (f32x4){0.f, 0.f, 0.f, 0.f} : gl_bias4(t, c0 + wc * 32 + (lane & 7) * 4);       // (this lane's columns: the same in every iteration)
#pragma unroll 4
    for (int it = 0; it < 4; ++it) {
        const int rr = it * 8 + (lane >> 3), cc = (lane & 7) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(E + rr * 36 + cc);
        const int r = r0 + wr * 32 + rr, c = c0 + wc * 32 + cc;
        if (r < R && c < Cn) {
            if (inl) dp_store4(t.slab + ((size_t)split * R + r) * C4p + c, v);
            else if (splits > 1) st4(t.slab + ((size_t)split * R + r) * C4p + c, v);
            else gl_epilogue4(t, r, c, v, &bpre);
        }
    }
    if (!inl) return;
    __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0): this thread's write-through stores have reached memory
    __syncthreads();
    int* const tick = reinterpret_cast<int*>(t.slab + (size_t)splits * R * C4p) + rem;
    if (threadIdx.x == 0) {
        const int old = atomicAdd(tick, 1);
        const int last = old == splits - 1;
        if (last) atomicExch(tick, 0);
        reinterpret_cast<volatile int*>(lds)[0] = last;          // (stage 0 is free: the patches above were consumed before the barrier)
    }
    __syncthreads();
    if (reinterpret_cast<volatile int*>(lds)[0] == 0) return;
#pragma unroll 2
    for (int it = 0; it < 4; ++it) {
        const int rr = it * 8 + (lane >> 3), cc = (lane & 7) * 4;
        const int r = r0 + wr * 32 + rr, c = c0 + wc * 32 + cc;
        if (r < R && c < Cn) {
            const float* p = t.slab + (size_t)r * C4p + c;
            const size_t stride = (size_t)R * C4p;
            f32x4 v = dp_load4(p);
            for (int s = 1; s < splits; ++s) v += dp_load4(p + s * stride);
            gl_epilogue4(t, r, c, v);
        }
    }
    if (want_bias && threadIdx.x < 64) {
        const int r = r0 + threadIdx.x;
        if (r < R) {
            float s0 = dp_load1(t.bslab + r);
            for (int q = 1; q < splits; ++q) s0 += dp_load1(t.bslab + (size_t)q * R + r);
            t.out2[r] = s0;
        }
    }
}

#include "gemm_x3q.h"

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int BT>
static int launch_bt(int la, int lb, dim3 g, hipStream_t st, const GemmBatch& gb, const int* dir) {
    if (la == LD_ROW && lb == LD_ROW) hipLaunchKernelGGL((gemm_lds_kernel<BT, LD_ROW, LD_ROW>), g, dim3(256), 0, st, GL_DIR_ARGS(dir), gb);
    else if (la == LD_ROW && lb == LD_COL) hipLaunchKernelGGL((gemm_lds_kernel<BT, LD_ROW, LD_COL>), g, dim3(256), 0, st, GL_DIR_ARGS(dir), gb);
    else if (la == LD_COL && lb == LD_COL) hipLaunchKernelGGL((gemm_lds_kernel<BT, LD_COL, LD_COL>), g, dim3(256), 0, st, GL_DIR_ARGS(dir), gb);
    else return -1;
    return (int)hipGetLastError();
}

static int launch_x3(int la, int lb, dim3 g, hipStream_t st, const GemmBatch& gb, const int* dir) {
    if (la == LD_ROW && lb == LD_ROW) hipLaunchKernelGGL((gemm_x3_kernel<LD_ROW, LD_ROW>), g, dim3(512), 0, st, GL_DIR_ARGS(dir), gb);
    // dX form: the k-major B operand through the transposed LDS reads; its 16-byte loads along the rows need Cn % 4 == 0 and an aligned B -- which the
    // routing guarantees for every task of the 128-wide bf16x3 tile
    else if (la == LD_ROW && lb == LD_COL) hipLaunchKernelGGL((gemm_x3t_kernel<LD_ROW>), g, dim3(512), 0, st, GL_DIR_ARGS(dir), gb);
    // weight-gradient form: both operands staged as they lie in memory
    else if (la == LD_COL && lb == LD_COL) hipLaunchKernelGGL((gemm_x3t_kernel<LD_COL>), g, dim3(512), 0, st, GL_DIR_ARGS(dir), gb);
    else return -1;
    return (int)hipGetLastError();
}

// 256 x 128 tile, persistent: one workgroup per CU walks tiles blockIdx, blockIdx + grid, ...
static int x3w_grid(int total_tiles) {
    static int cus = 0;
    if (!cus) {
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess || pr.multiProcessorCount <= 0) return total_tiles < 256 ? total_tiles : 256;
        cus = pr.multiProcessorCount;
    }
    return total_tiles < cus ? total_tiles : cus;
}
static int launch_x3w(int la, int lb, int total_tiles, hipStream_t st, const GemmBatch& gb, const int* dir) {
    const dim3 g(x3w_grid(total_tiles));
    if (la == LD_ROW && lb == LD_ROW) hipLaunchKernelGGL((gemm_x3w_kernel<LD_ROW, LD_ROW>), g, dim3(512), 0, st, GL_DIR_ARGS(dir), gb, total_tiles);
    else if (la == LD_ROW && lb == LD_COL) hipLaunchKernelGGL((gemm_x3w_kernel<LD_ROW, LD_COL>), g, dim3(512), 0, st, GL_DIR_ARGS(dir), gb, total_tiles);
    else if (la == LD_COL && lb == LD_COL) hipLaunchKernelGGL((gemm_x3w_kernel<LD_COL, LD_COL>), g, dim3(512), 0, st, GL_DIR_ARGS(dir), gb, total_tiles);
    else return -1;
    return (int)hipGetLastError();
}

static bool x3s_unaligned_ok(const GemmTask* t) { return t->R >= 4 && t->Cn >= 4 && t->K >= 4; }       // (the pulled-back tail load needs four elements to exist)
static int launch_x3s(int la, int lb, dim3 g, hipStream_t st, const GemmBatch& gb, const int* dir) {
    // one instantiation per launch: the any-alignment loaders as soon as ONE task of the stage has an operand that is not 16-byte regular
    bool unal = false;
    for (int q = 0; q < gb.ntasks; ++q) if (gb.t[q].flags & (FLAG_SCALAR_A | FLAG_SCALAR_B)) { unal = true; if (!x3s_unaligned_ok(&gb.t[q])) return -2; }
    if (unal) {
        if (la == LD_ROW && lb == LD_ROW) hipLaunchKernelGGL((gemm_x3s_kernel<LD_ROW, LD_ROW, 2>), g, dim3(256), 0, st, GL_DIR_ARGS(dir), gb);
        else if (la == LD_ROW && lb == LD_COL) hipLaunchKernelGGL((gemm_x3s_kernel<LD_ROW, LD_COL, 2>), g, dim3(256), 0, st, GL_DIR_ARGS(dir), gb);
        else if (la == LD_COL && lb == LD_COL) hipLaunchKernelGGL((gemm_x3s_kernel<LD_COL, LD_COL, 2>), g, dim3(256), 0, st, GL_DIR_ARGS(dir), gb);
        else return -1;
        return (int)hipGetLastError();
    }
    if (la == LD_ROW && lb == LD_ROW) hipLaunchKernelGGL((gemm_x3s_kernel<LD_ROW, LD_ROW, 1>), g, dim3(256), 0, st, GL_DIR_ARGS(dir), gb);
    else if (la == LD_ROW && lb == LD_COL) hipLaunchKernelGGL((gemm_x3s_kernel<LD_ROW, LD_COL, 1>), g, dim3(256), 0, st, GL_DIR_ARGS(dir), gb);
    else if (la == LD_COL && lb == LD_COL) hipLaunchKernelGGL((gemm_x3s_kernel<LD_COL, LD_COL, 1>), g, dim3(256), 0, st, GL_DIR_ARGS(dir), gb);
    else return -1;
    return (int)hipGetLastError();
}

static int launch_x3q(int la, int lb, dim3 g, hipStream_t st, const GemmBatch& gb, const int* dir) {
    if (la != LD_ROW) return -1;
    for (int q = 0; q < gb.ntasks; ++q) {
        const GemmTask& t = gb.t[q];
        if ((t.flags & (FLAG_SCALAR_A | FLAG_SCALAR_B)) || (t.K & 15) || t.K < 32 || t.splits != 1 || (lb == LD_COL && ((t.Cn & 7) || t.Cn < 8))) return -2;
    }
    if (lb == LD_ROW) hipLaunchKernelGGL((gemm_x3q_kernel<LD_ROW>), g, dim3(256), 0, st, GL_DIR_ARGS(dir), gb);
    else if (lb == LD_COL) hipLaunchKernelGGL((gemm_x3q_kernel<LD_COL>), g, dim3(256), 0, st, GL_DIR_ARGS(dir), gb);
    else return -1;
    return (int)hipGetLastError();
}

// bt: 64 / 128 = fp32-MFMA tiles; 129 = the 128-wide tile on the bf16 pipe (bf16x3); 65 = the 64-wide tile on the bf16 pipe; 257 = the 256 x 128 tile on the
// bf16 pipe (persistent workgroups: gemm_x3w.h); 33 = the 32 x 32 tile on the bf16 pipe whose four waves split K (gemm_x3q.h)
extern "C" int rl_launch_gemm_lds(int bt, int la, int lb, const GemmBatch* gb, int total_tiles, int fin_blocks, hipStream_t st) {
    if (total_tiles <= 0) return 0;
    int dir[GEMM_MAX_TASKS], fdir[GEMM_MAX_TASKS];
    for (int q = 0; q < GEMM_MAX_TASKS; ++q) {
        dir[q] = q < gb->ntasks ? gb->t[q].tile_base : 0x7fffffff;
        fdir[q] = (q < gb->ntasks && gb->t[q].splits > 1) ? gb->t[q].fin_base : 0x7fffffff;
    }
    int rc = bt == 33 ? launch_x3q(la, lb, dim3(total_tiles), st, *gb, dir)
           : bt == 257 ? launch_x3w(la, lb, total_tiles, st, *gb, dir) : bt == 65 ? launch_x3s(la, lb, dim3(total_tiles), st, *gb, dir) : bt == 129 ? launch_x3(la, lb, dim3(total_tiles), st, *gb, dir)
           : bt == 128 ? launch_bt<128>(la, lb, dim3(total_tiles), st, *gb, dir) : launch_bt<64>(la, lb, dim3(total_tiles), st, *gb, dir);
    if (rc != 0) return rc;
    if (fin_blocks > 0) {
        hipLaunchKernelGGL(gemm_lds_fin_kernel, dim3(fin_blocks), dim3(256), 0, st, GL_DIR_ARGS(fdir), *gb);
        ++g_rl_launches;              // split-K: the stage is two kernels
        rc = (int)hipGetLastError();
    }
    return rc;
}

// Which sides of a task can use 16-byte accesses?  (dimensions only; pointers are checked by rl_gemm_lds_ptr_flags)
extern "C" int rl_gemm_lds_dim_flags(const GemmTask* t, int la, int lb) {
    int f = 0;
    if ((t->lda & 3) || (la == LD_ROW ? (t->K & 3) : (t->R & 3))) f |= FLAG_SCALAR_A;
    if ((t->ldb & 3) || (lb == LD_ROW ? (t->K & 3) : (t->Cn & 3))) f |= FLAG_SCALAR_B;
    if ((t->Cn & 3) || (t->ldc & 3)) f |= FLAG_SCALAR_C;
    if (t->epi == EPI_DX && t->act != ACT_NONE && (t->ldaux & 3)) f |= FLAG_SCALAR_C;
    if (t->epi == EPI_FWD && t->act == ACT_SIN && (t->ldout2 & 3)) f |= FLAG_SCALAR_C;
    return f;
}
extern "C" int rl_gemm_lds_align_ok(const GemmTask* t, int la, int lb) {
    if (t->epi != EPI_FWD && t->epi != EPI_DX && t->epi != EPI_DW) return 0;
    (void)la; (void)lb;
    return 1;
}
// Is the task large enough to be worth leaving the latency-tuned 16-row engine?  Judged by dimensions alone, so that the
// dry sizing pass and the real pass of the program builder agree.
extern "C" int rl_gemm_lds_dims_ok(const GemmTask* t, int la, int lb) {
    if (!rl_gemm_lds_align_ok(t, la, lb) || t->K < 64) return 0;
    return 2.0 * t->R * t->Cn * t->K >= 2.0e8;
}
static bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
// pointer alignment adds to the scalar flags (never a reason to leave the engine)
extern "C" int rl_gemm_lds_ptr_flags(const GemmTask* t) {
    int f = 0;
    if (!al16(t->A)) f |= FLAG_SCALAR_A;
    if (!al16(t->B)) f |= FLAG_SCALAR_B;
    if (!al16(t->C)) f |= FLAG_SCALAR_C;
    if (t->epi == EPI_FWD && t->bias && !al16(t->bias)) f |= FLAG_SCALAR_C;
    if (t->epi == EPI_FWD && t->act == ACT_SIN && !al16(t->out2)) f |= FLAG_SCALAR_C;
    if (t->epi == EPI_DX && t->act != ACT_NONE && !al16(t->aux)) f |= FLAG_SCALAR_C;
    if (t->epi == EPI_DX && t->r1u && !al16(t->r1v)) f |= FLAG_SCALAR_C;
    return f;
}
extern "C" int rl_gemm_lds_ptrs_ok(const GemmTask* t) { (void)t; return 1; }
// tile edge and split count for a task (dimensions only).  128-wide tiles run two workgroups per CU, so a grid that
// already has >= 256 of them is left alone and a smaller one is split along K towards 512 workgroups (each split keeps
// at least eight 32-deep slices); outputs too small for that use 64-wide tiles under the same rule.
extern "C" void rl_gemm_lds_plan(const GemmTask* t, int* bt, int* splits, int* kchunk) {
    auto tiles = [&](int b) { return (long long)((t->R + b - 1) / b) * ((t->Cn + b - 1) / b); };
    auto spl = [&](int b) {
        const long long n = tiles(b);
        if (n >= 256) return 1;
        long long s = (512 + n - 1) / n;
        const long long mx = t->K / 256 > 1 ? t->K / 256 : 1;
        if (s > mx) s = mx;
        if (s > 32) s = 32;
        return (int)(s < 1 ? 1 : s);
    };
    int b = 128, s = spl(128);
    if (tiles(128) * s < 256) { b = 64; s = spl(64); }
    int kc = ((t->K + s - 1) / s + GL_BK - 1) / GL_BK * GL_BK;
    s = (t->K + kc - 1) / kc;
    *bt = b; *splits = s; *kchunk = kc;
}

// The program builder's routing decision for one task, from dimensions alone (pointer alignment may add scalar flags
// later, which also rules out bf16x3): returns 0 = stays on the 16-row engine, 64 / 128 = fp32-MFMA tile, 129 = the 128-wide
// tile on the bf16 pipe, 257 = the 256 x 128 persistent tile on the bf16 pipe, 65 = the 64-wide one.  Products of >= 2 GFLOP with a row-major A (forward, dX: diffsrsac's 202-GFLOP nabla-mu head) take
// bf16x3: 159 / 137 TF against 110 on the fp32 pipe; the k-major/k-major weight-gradient form stays on fp32 MFMA (107 vs 116).
extern "C" int rl_gemm_lds_route(const GemmTask* t, int la, int lb, int extra_flags, int* splits, int* kchunk, int* flags) {
    const bool combo = (la == LD_ROW && lb == LD_ROW) || (la == LD_ROW && lb == LD_COL) || (la == LD_COL && lb == LD_COL);
    if (!combo || rl_off("gemm_lds") || !rl_gemm_lds_dims_ok(t, la, lb)) return 0;
    int bt = 0;
    rl_gemm_lds_plan(t, &bt, splits, kchunk);
    *flags = rl_gemm_lds_dim_flags(t, la, lb) | extra_flags;
    // (the k-major / k-major weight-gradient form takes bf16x3 too since round 4: gemm_x3t_kernel; RLREP_DISABLE=x3_dw keeps it on the fp32 tile)
    const bool x3 = bt == 128 && (la == LD_ROW || (lb == LD_COL && !rl_off("x3_dw"))) && 2.0 * t->R * t->Cn * t->K >= 2e9 && !rl_off("x3") &&
                    !(*flags & (FLAG_SCALAR_A | FLAG_SCALAR_B));
    if (x3) {
        // the 256 x 128 persistent tile (gemm_x3w.h) where its tiles fill the chip's 256 CUs evenly: >= 85 % of the last round of workgroups busy
        // (RLREP_DISABLE=x3w keeps the 128 x 128 tile)
        const long long wt = (long long)((t->R + 255) / 256) * ((t->Cn + 127) / 128) * *splits;
        const bool fills = wt >= 256 && (double)wt / (double)(((wt + 255) / 256) * 256) >= 0.85;
        if (fills && x3w_epilogue_ok(t) && !rl_off("x3w")) return 257;
        return 129;
    }
    // 64-wide tiles: on the bf16 pipe too when both operands allow 16-byte staging (gemm_x3s_kernel); RLREP_DISABLE=x3s keeps the fp32 tile
    // (the program builder keeps a STAGE on one engine: a stage whose tasks would be split between this tile and the fp32 one becomes two dependent
    // launches, which costs more than the faster tile returns -- spedersac: 69 -> 81 launches on the feature chain, 961 -> 903 train()/s)
    // (operands that are not 16-byte regular -- rows of 119 floats, a block that starts inside another buffer -- take the any-alignment loaders of the
    // same tile (x3s_load_*<2>); RLREP_DISABLE=x3s_unaligned keeps such stages on the fp32 tile as before.  A 4-byte-staging instantiation was built and
    // measured first -- spedersac 968 with it against 1 002 with those stages on fp32, ctrlsac F = 2048 865 against 904)
    // few rows, long inner dimension (ctrlsac's M = 256 layers at main.py's dimensions): where the 64-wide tile would cut K into slabs and a
    // 32 x 32 tiling has enough workgroups without them, the tile whose four waves split K among themselves (gemm_x3q.h): no slab, no finishing
    // launch.  RLREP_DISABLE=x3q keeps the 64-wide tile.
    if (bt == 64 && *splits > 1 && la == LD_ROW && !rl_off("x3") && !rl_off("x3q") && !(*flags & (FLAG_SCALAR_A | FLAG_SCALAR_B)) &&
        (t->K & 15) == 0 && t->K >= 512 && (lb == LD_ROW || (t->Cn & 7) == 0) && t->Cn >= 32 && t->R <= 256 &&
        (long long)((t->R + 31) / 32) * ((t->Cn + 31) / 32) >= 192) {          // (R <= 256: spedersac's M = 1024 critic layers measured 0.8 % slower on it)
        *splits = 1; *kchunk = t->K;
        return 33;
    }
    if (bt == 64 && !rl_off("x3") && !rl_off("x3s") &&
        (!(*flags & (FLAG_SCALAR_A | FLAG_SCALAR_B)) || (x3s_unaligned_ok(t) && !rl_off("x3s_unaligned")))) return 65;
    return bt;
}


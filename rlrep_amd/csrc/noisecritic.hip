// vlsac noise-averaged critic, first layer (reference agent/vlsac/vlsac_agent.py:44-63), gfx950.
//
//   x[(b,n), :] = mean[b,:] + exp(log_std[b,:]) * noise[n,:]           (n < N = 20, never materialised)
//   Hm[b, j]    = (1/N) * sum_n elu( W[j,:] . x[(b,n),:] + bias[j] )
//
// Ten [B*20 x 256] x [256 x 256] products per train() = 63 % of a vlsac train()'s FLOPs, in three forms:
//   nc_fwd_kernel : Hm (and the elu outputs U for the backward passes)
//   nc_dx_kernel  : dL/d(mean, log_std)  (actor step; both heads summed)
//   nc_dw_kernel  : dL/dW, dL/dbias      (critic step)
// All three run on v_mfma_f32_16x16x4_f32 (exact fp32).  The [B*20, F] input and the [B*20, H] gradient
// of the pre-activation are generated on the fly inside the operand fragments; HBM/L2 only see W, U and
// the [B, .] tensors.
//
// Row-mapping trick (fwd, dx): the 16 rows of MFMA A-fragment f are assigned (b' = row>>2, n = 4*f + (row&3)).
// With the 16x16x4 C/D map (row = 4*(lane>>4) + reg) every lane then owns ALL 20 noise rows of batch row
// b' = lane>>4 across its 5 accumulator fragments: the reductions over the noise axis (mean of elu; dmean,
// dlog_std) are in-register sums of 20 values -- no shuffles, no LDS, no atomics.
//
// Occupancy: f32 MFMA issues at 32 cycles/SIMD, so the kernels are MFMA-bound only if every SIMD always has
// an MFMA ready.  Each kernel is shaped to put >= 2 waves on every SIMD of all 256 CUs and prefetches the
// next step's operands into registers before issuing the current step's MFMAs.
#include <type_traits>
#include "common.h"
extern long long g_rl_launches;
#include "kparams.h"
#include "x3.h"

#define NC_NF 5          // N / 4 accumulator fragments per batch-row group (N = 20)

extern __shared__ __attribute__((aligned(16))) float nc_smem[];

#ifdef RL_TIMING_NC
// Instrumented build (tools/exp/nc_timeline.py): thread 0 of every workgroup stamps the shader clock at entry, after the table
// staging barrier, after the MFMA loop and at exit, plus the 100 MHz wall clock at entry / exit.
__device__ unsigned long long g_nc_tim[6 * 4096];
extern "C" int rl_nc_timing_fetch(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_nc_tim), sizeof(unsigned long long) * n);
}
#define NCT(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_nc_tim[6 * blockIdx.x + (k)] = (k) < 4 ? clock64() : wall_clock64(); } while (0)
#else
#define NCT(k) do {} while (0)
#endif

__device__ __forceinline__ void ld4(const float* p, bool vec, int valid, float (&v)[4]) {
    // valid = number of in-range elements (0..4)
    if (vec && valid == 4) { f32x4 x = *reinterpret_cast<const f32x4*>(p); v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3]; }
    else {
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = (s < valid) ? p[s] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// forward.  workgroup = 4 waves = (4*G2 batch rows) x 20 noise rows x 64 hidden units; K loop over F.
// ------------------------------------------------------------------------------------------------
// NW waves per workgroup = 16 * NW hidden units: the tables depend on the batch rows only, so a wider workgroup stages them
// once for more columns (NW = 8: half the table traffic and staging time of NW = 4 at the same waves per SIMD)
template <int G2, int NW>
__global__ __launch_bounds__(64 * NW) void nc_fwd_kernel(NcFwdBatch nb) {
    const int bid = blockIdx.x;
    NCT(0); NCT(4);
    int ti = 0;
#pragma unroll
    for (int q = 1; q < NC_MAX_TASKS; ++q) if (q < nb.ntasks && bid >= nb.t[q].tile_base) ti = q;
    const NcFwdTask& t = nb.t[ti];
    const int local = bid - t.tile_base;
    const int tb = local / t.tiles_h, th = local - tb * t.tiles_h;
    const int RB = 4 * G2;
    const int b0 = tb * RB, n0 = th * (16 * NW);
    const int F = t.F, H = t.H, N = t.N;
    const int Fp = (F + 15) & ~15;
    const int LDS_LD = Fp + 16;
    float* mu_s = nc_smem;                     // [RB][LDS_LD]
    float* sg_s = mu_s + RB * LDS_LD;          // [RB][LDS_LD]
    float* nz_s = sg_s + RB * LDS_LD;          // [N][LDS_LD]

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m16 = lane & 15, kq = lane >> 4;
    const int bp = m16 >> 2, nn = m16 & 3;
    const int col = n0 + 16 * w + m16;
    const bool colok = col < H;
    const bool vecW = ((F & 3) == 0) && ((((uintptr_t)t.W) & 15) == 0);
    const float* wrow = t.W + (size_t)(colok ? col : 0) * F;

    // first W fragment is in flight while the tables are staged
    float wv[4], wn[4];
    {
        const int k0 = 4 * kq;
        const int valid = colok ? max(0, min(4, F - k0)) : 0;
        ld4(wrow + k0, vecW, valid, wv);
    }
    // Table staging: one wave-wide 16-byte load moves a whole 256-float row; every wave issues ALL of its
    // row loads back to back (fixed trip count, fully unrolled) so their L2 latencies overlap instead of
    // serialising (the element-wise loop this replaces cost 7.7 us of a 40 us launch).
    {
        constexpr int NROWS = 2 * RB + 4 * NC_NF;          // mean rows, sigma rows, noise rows (N = 20)
        constexpr int SLOTS = (NROWS + NW - 1) / NW;
        for (int cb = 0; cb < Fp; cb += 256) {
            const int k = cb + 4 * lane;
            f32x4 v[SLOTS];
#pragma unroll
            for (int q = 0; q < SLOTS; ++q) {
                const int row = NW * q + w;
                v[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (row < NROWS && k < F) {
                    if (row < 2 * RB) {
                        const int rr = row < RB ? row : row - RB;
                        const float* src = (row < RB ? t.mean : t.lstd) + (size_t)(b0 + rr) * t.ld_ml + k;
                        if (b0 + rr < t.B) v[q] = *reinterpret_cast<const f32x4*>(src);
                    } else {
                        v[q] = *reinterpret_cast<const f32x4*>(t.noise + (size_t)(row - 2 * RB) * F + k);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < SLOTS; ++q) {
                const int row = NW * q + w;
                if (row >= NROWS || k >= Fp) continue;
                f32x4 x = v[q];
                if (row >= RB && row < 2 * RB) {
                    const int rr = row - RB;
                    const bool ok = (b0 + rr < t.B) && (k < F);
#pragma unroll
                    for (int s = 0; s < 4; ++s) x[s] = ok ? expf(clamp_lstd(x[s])) : 0.f;
                    if (t.sigma_out && th == 0 && ok) *reinterpret_cast<f32x4*>(t.sigma_out + (size_t)(b0 + rr) * F + k) = x;
                }
                float* dst = (row < RB ? mu_s + row * LDS_LD : row < 2 * RB ? sg_s + (row - RB) * LDS_LD : nz_s + (row - 2 * RB) * LDS_LD) + k;
                *reinterpret_cast<f32x4*>(dst) = x;
            }
        }
    }
    __syncthreads();
    NCT(1);

    f32x4 acc[G2][NC_NF];
#pragma unroll
    for (int g = 0; g < G2; ++g)
#pragma unroll
        for (int f = 0; f < NC_NF; ++f) acc[g][f] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // (A variant with two named register sets, which issued the LDS table reads and the W fragment of chunk c+1 before the MFMAs
    // of chunk c, measured the same 47.8k cycles per workgroup for this loop -- tools/exp/nc_timeline.py: staging 4.9k, loop
    // 48.2k, epilogue 3.6k cycles at 2.1 GHz.  The loop runs at 85 % of 32 cycles per MFMA, the rate tools/exp/mfma_peak.hip
    // measures for four waves per SIMD: 126 of 157 TF.)
    for (int kb = 0; kb < Fp; kb += 16) {
        const int k0 = kb + 4 * kq;
        {   // prefetch the next W fragment
            const int k1 = k0 + 16;
            const int valid = colok ? max(0, min(4, F - k1)) : 0;
            if (kb + 16 < Fp) ld4(wrow + k1, vecW, valid, wn);
        }
        f32x4 mu4[G2], sg4[G2], nz4[NC_NF];
#pragma unroll
        for (int g = 0; g < G2; ++g) {
            mu4[g] = *reinterpret_cast<const f32x4*>(&mu_s[(4 * g + bp) * LDS_LD + k0]);
            sg4[g] = *reinterpret_cast<const f32x4*>(&sg_s[(4 * g + bp) * LDS_LD + k0]);
        }
#pragma unroll
        for (int f = 0; f < NC_NF; ++f) nz4[f] = *reinterpret_cast<const f32x4*>(&nz_s[(4 * f + nn) * LDS_LD + k0]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int g = 0; g < G2; ++g)
#pragma unroll
                for (int f = 0; f < NC_NF; ++f)
                    acc[g][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(fmaf(sg4[g][s], nz4[f][s], mu4[g][s]), wv[s], acc[g][f], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) wv[s] = wn[s];
    }

    NCT(2);
    if (!colok) return;
    const float bj = t.bias[col];
    const float invN = 1.0f / (float)N;
#pragma unroll
    for (int g = 0; g < G2; ++g) {
        const int b = b0 + 4 * g + (lane >> 4);
        if (b >= t.B) continue;
        float sum = 0.f;
#pragma unroll
        for (int f = 0; f < NC_NF; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y = elu_fast(acc[g][f][r] + bj);
                sum += y;
#ifndef RL_NC_NOU
                if (t.U) t.U[((size_t)b * N + 4 * f + r) * H + col] = y;
#endif
            }
        t.Hm[(size_t)b * H + col] = sum * invN;
    }
    NCT(3); NCT(5);
}

// ------------------------------------------------------------------------------------------------
// forward on the BF16 matrix pipe at fp32 accuracy (bf16x3, x3.h): v_mfma_f32_16x16x32_bf16 has the C/D map of the fp32
// 16x16x4 instruction, so the row-mapping trick and the epilogue above carry over unchanged; six bf16 MFMAs of 16 cycles
// replace eight fp32 MFMAs of 32 per 16 x 16 x 32 block (0.375x the matrix cycles).
//
// workgroup = 8 batch rows x 20 noise rows x 64 or 128 hidden units, K loop in steps of 32.
// The operand x = mean + sigma * noise is generated AND split once per workgroup, not per wave (the split is ~5.5 VALU ops per
// element and the matrix pipe only leaves two VALU issue slots per MFMA): the three bf16 images [160 rows][32 k] of a step
// (80-byte rows, fragment order: rows (5g + f) 16 .. + 15 are fragment f of batch group g) live in one of two LDS buffers; the
// consuming waves read their A fragments with ds_read_b128 and split their own 16 x 32 W fragments in registers; one barrier
// per step.
// ------------------------------------------------------------------------------------------------
#define NX_RSB 80
#define NX_ROWS 160
#define NX_IMGB (NX_ROWS * NX_RSB)
#define NX_BUFB (3 * NX_IMGB)

// LDS fragment reads as inline asm: hipcc otherwise keeps ONE register set per operand image and puts s_waitcnt lgkmcnt(0) between
// every ds_read and the MFMA that uses it (it re-sinks explicit prefetches, and sched_group_barrier pipelines came out scrambled).
// The reads of fragment j + 1 are issued, then `s_waitcnt lgkmcnt(3)` claims fragment j (LDS returns in order: at most the three
// newest reads are still in flight); compiler-issued LDS traffic in between only makes the wait more conservative.
template <int OFF> __device__ __forceinline__ void nx_read(u32x4& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int J> __device__ __forceinline__ void nx_fload(u32x4 (&d)[3], unsigned addr) {
    nx_read<J * 16 * NX_RSB>(d[0], addr);
    nx_read<J * 16 * NX_RSB + NX_IMGB>(d[1], addr);
    nx_read<J * 16 * NX_RSB + 2 * NX_IMGB>(d[2], addr);
}
template <int PENDING> __device__ __forceinline__ void nx_claim(u32x4 (&d)[3]) {
    if (PENDING) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]));
}
// claim a fragment while NEWER reads (a multiple of three, at most 12) are still in flight
template <int NEWER> __device__ __forceinline__ void nx_claim_n(u32x4 (&d)[3]) {
    static_assert(NEWER >= 0 && NEWER <= 12 && NEWER % 3 == 0, "reads in flight behind the claimed fragment");
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]) : "n"(NEWER));
}

template <int KV> struct NxVec;
template <> struct NxVec<4> { typedef f32x4 T; };
template <> struct NxVec<2> { typedef f32x2v T; };

// CG = 16-column groups per consumer wave (1: 64 hidden units per workgroup, 2: 128).  8 waves in two roles, one of each on every SIMD:
//   waves 0-3 CONSUME: split their CG W fragments, read the ten A fragments (two register sets, one fragment ahead), 60 CG MFMAs per
//              step, ELU / mean-over-noise epilogue;
//   waves 4-7 PRODUCE: thread (k chunk of 4, batch row, noise group of 5) builds and splits its 20 elements of the NEXT step and writes
//              the three images; the tables of the step after that are already in flight (two named register sets).
// As one role per wave the step cost the SUM of the two instruction streams (the bf16 MFMA holds the vector issue port for 8 of its
// 16 cycles, so the split -- ~5.5 VALU per element -- does not hide behind a wave's own MFMAs): 3.3k cycles per step at two waves
// per SIMD, against 1.9k of matrix-pipe time.
template <int CG>
__global__ __launch_bounds__(512) void nc_fwd_x3_kernel(NcFwdBatch nb) {
    constexpr int G2 = 2, KV = 4, NFR = G2 * NC_NF;
    unsigned char* const L = reinterpret_cast<unsigned char*>(nc_smem);
    const int bid = blockIdx.x;
    NCT(0); NCT(4);
    int ti = 0;
#pragma unroll
    for (int q = 1; q < NC_MAX_TASKS; ++q) if (q < nb.ntasks && bid >= nb.t[q].tile_base) ti = q;
    const NcFwdTask& t = nb.t[ti];
    const int local = bid - t.tile_base;
    const int tb = local / t.tiles_h, th = local - tb * t.tiles_h;
    const int b0 = tb * 8, n0 = th * (64 * CG);
    const int F = t.F, H = t.H, N = t.N;
    const int S = F >> 5;                                   // K steps (F % 32 == 0, checked by the launcher)
    const int w8 = threadIdx.x >> 6;

    if (w8 >= 4) {
        // ================= producer =================
        const int tid = threadIdx.x - 256;
        const int kc = tid & 7, pb = (tid >> 3) & 7, ng = tid >> 6;
        const bool okb = b0 + pb < t.B;
        const int bsrc = min(b0 + pb, t.B - 1);
        const float* const pmu = t.mean + (size_t)bsrc * t.ld_ml + kc * KV;
        const float* const pls = t.lstd + (size_t)bsrc * t.ld_ml + kc * KV;
        const float* const pnz = t.noise + (size_t)(5 * ng) * F + kc * KV;
        int wofs[5];                                        // byte offset of this thread's chunk in an image, per noise row
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int n = 5 * ng + i;
            wofs[i] = (((pb >> 2) * 5 + (n >> 2)) * 16 + (pb & 3) * 4 + (n & 3)) * NX_RSB + kc * KV * 2;
        }
        struct PReg { f32x4 mu, ls, nz[5]; };
        auto gload = [&](int s, PReg& r) {
            const int k = 32 * min(s, S - 1);               // past the end: re-read the last step (no branch), never produced
            r.mu = *reinterpret_cast<const f32x4*>(pmu + k);
            r.ls = *reinterpret_cast<const f32x4*>(pls + k);
#pragma unroll
            for (int i = 0; i < 5; ++i) r.nz[i] = *reinterpret_cast<const f32x4*>(pnz + (size_t)i * F + k);
        };
        auto produce = [&](const PReg& r, unsigned char* buf) {
            float sg[KV];
#pragma unroll
            for (int q = 0; q < KV; ++q) sg[q] = okb ? __expf(clamp_lstd(r.ls[q])) : 0.f;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                u32x2 h, m, l;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float x0 = okb ? fmaf(sg[2 * q], r.nz[i][2 * q], r.mu[2 * q]) : 0.f;
                    const float x1 = okb ? fmaf(sg[2 * q + 1], r.nz[i][2 * q + 1], r.mu[2 * q + 1]) : 0.f;
                    unsigned hh, mm, ll;
                    x3_split2(x0, x1, hh, mm, ll);
                    h[q] = hh; m[q] = mm; l[q] = ll;
                }
                unsigned char* p = buf + wofs[i];
                *reinterpret_cast<u32x2*>(p) = h;
                *reinterpret_cast<u32x2*>(p + NX_IMGB) = m;
                *reinterpret_cast<u32x2*>(p + 2 * NX_IMGB) = l;
            }
        };
        PReg ra, rb;
        gload(0, ra); gload(1, rb);
        produce(ra, L);
        gload(2, ra);
        __syncthreads();
        for (int s = 0; s < S; s += 2) {
            if (s + 1 < S) produce(rb, L + NX_BUFB);        // images of step s + 1
            gload(s + 3, rb);
            __syncthreads();
            if (s + 1 < S) {
                if (s + 2 < S) produce(ra, L);              // images of step s + 2
                gload(s + 4, ra);
                __syncthreads();
            }
        }
        // exp(clamp(log_std)) for the backward passes: the producer threads of column tile 0, noise group 0, after their last image
        if (t.sigma_out && th == 0 && ng == 0 && okb) {
            float* const sig_dst = t.sigma_out + (size_t)(b0 + pb) * F + kc * KV;
            for (int s = 0; s < S; ++s) {
                const f32x4 ls = *reinterpret_cast<const f32x4*>(pls + 32 * s);
                f32x4 o;
#pragma unroll
                for (int q = 0; q < KV; ++q) o[q] = expf(clamp_lstd(ls[q]));
                *reinterpret_cast<f32x4*>(sig_dst + 32 * s) = o;
            }
        }
        return;
    }

    // ================= consumer =================
    const int lane = threadIdx.x & 63, w = w8;
    const int m16 = lane & 15, kq = lane >> 4;
    int col[CG]; bool colok[CG]; const float* wrow[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) {
        col[c] = n0 + 16 * (CG * w + c) + m16;
        colok[c] = col[c] < H;
        wrow[c] = t.W + (size_t)min(col[c], H - 1) * F + 8 * kq;
    }
    const int aofs = m16 * NX_RSB + kq * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)L + (unsigned)aofs;
    // with bf16x3 images of W (NcFwdTask::W3): v[c][0..2] are the hi / mid / lo fragments themselves, 16 bytes each at
    // ((3 s + image) H + col) 64 + kq 16; without: v[c][0..1] are eight fp32 values that cstep splits
    struct WReg { f32x4 v[CG][3]; };
    const bool w3 = t.W3 != nullptr;
    const unsigned char* w3row[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) w3row[c] = w3 ? t.W3 + (size_t)min(col[c], H - 1) * 64 + kq * 16 : nullptr;
    auto wload = [&](int s, WReg& r, auto w3_tag) {
        const int sc = min(s, S - 1);
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            if (decltype(w3_tag)::value) {
#pragma unroll
                for (int img = 0; img < 3; ++img) r.v[c][img] = *reinterpret_cast<const f32x4*>(w3row[c] + ((size_t)(3 * sc + img) * H) * 64);
            } else {
                r.v[c][0] = *reinterpret_cast<const f32x4*>(wrow[c] + 32 * sc);
                r.v[c][1] = *reinterpret_cast<const f32x4*>(wrow[c] + 32 * sc + 4);
            }
        }
    };
    f32x4 acc[CG][NFR];
#pragma unroll
    for (int c = 0; c < CG; ++c)
#pragma unroll
        for (int f = 0; f < NFR; ++f) acc[c][f] = (f32x4){0.f, 0.f, 0.f, 0.f};

    WReg wa, wb;                                            // W fragments of steps s and s + 1; refilled two steps ahead
    if (w3) { wload(0, wa, std::true_type{}); wload(1, wb, std::true_type{}); } else { wload(0, wa, std::false_type{}); wload(1, wb, std::false_type{}); }
    __syncthreads();
    NCT(1);
    auto cstep = [&](int s, WReg& rw, auto w3_tag) {
        constexpr bool W3 = decltype(w3_tag)::value;
        const unsigned aaddr = lds0 + (unsigned)((s & 1) * NX_BUFB);
        u32x4 fa[2][3];
        nx_fload<0>(fa[0], aaddr);
        bf16x8 Bh[CG], Bm[CG], Bl[CG];
        if (W3) {
#pragma unroll
            for (int c = 0; c < CG; ++c) { Bh[c] = __builtin_bit_cast(bf16x8, rw.v[c][0]); Bm[c] = __builtin_bit_cast(bf16x8, rw.v[c][1]); Bl[c] = __builtin_bit_cast(bf16x8, rw.v[c][2]); }
        } else
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            u32x4 bh, bm, bl;
            unsigned h, m, l;
            x3_split2(rw.v[c][0][0], rw.v[c][0][1], h, m, l); bh[0] = h; bm[0] = m; bl[0] = l;
            x3_split2(rw.v[c][0][2], rw.v[c][0][3], h, m, l); bh[1] = h; bm[1] = m; bl[1] = l;
            x3_split2(rw.v[c][1][0], rw.v[c][1][1], h, m, l); bh[2] = h; bm[2] = m; bl[2] = l;
            x3_split2(rw.v[c][1][2], rw.v[c][1][3], h, m, l); bh[3] = h; bm[3] = m; bl[3] = l;
            Bh[c] = __builtin_bit_cast(bf16x8, bh); Bm[c] = __builtin_bit_cast(bf16x8, bm); Bl[c] = __builtin_bit_cast(bf16x8, bl);
        }
        wload(s + 2, rw, w3_tag);
#define NX_FRAG(J)                                                                                                     \
        {                                                                                                              \
            if ((J) + 1 < NFR) nx_fload<((J) + 1 < NFR ? (J) + 1 : 0)>(fa[((J) + 1) & 1], aaddr);                      \
            nx_claim<((J) + 1 < NFR)>(fa[(J) & 1]);                                                                    \
            const bf16x8 Ah = __builtin_bit_cast(bf16x8, fa[(J) & 1][0]), Am = __builtin_bit_cast(bf16x8, fa[(J) & 1][1]), \
                         Al = __builtin_bit_cast(bf16x8, fa[(J) & 1][2]);                                              \
            _Pragma("unroll") for (int c = 0; c < CG; ++c) {                                                           \
                f32x4 d = acc[c][(J)];                                                                                 \
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bh[c], d, 0, 0, 0);                                    \
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bl[c], d, 0, 0, 0);                                    \
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bm[c], d, 0, 0, 0);                                    \
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bh[c], d, 0, 0, 0);                                    \
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bm[c], d, 0, 0, 0);                                    \
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bh[c], d, 0, 0, 0);                                    \
                acc[c][(J)] = d;                                                                                       \
            }                                                                                                          \
        }
        NX_FRAG(0) NX_FRAG(1) NX_FRAG(2) NX_FRAG(3) NX_FRAG(4) NX_FRAG(5) NX_FRAG(6) NX_FRAG(7) NX_FRAG(8) NX_FRAG(9)
#undef NX_FRAG
        __syncthreads();
    };
    if (w3) {
        for (int s = 0; s < S; s += 2) {
            cstep(s, wa, std::true_type{});
            if (s + 1 < S) cstep(s + 1, wb, std::true_type{});
        }
    } else {
        for (int s = 0; s < S; s += 2) {
            cstep(s, wa, std::false_type{});
            if (s + 1 < S) cstep(s + 1, wb, std::false_type{});
        }
    }
    NCT(2);

    const float invN = 1.0f / (float)N;
#pragma unroll
    for (int c = 0; c < CG; ++c) {
        if (!colok[c]) continue;
        const float bj = t.bias[col[c]];
#pragma unroll
        for (int g = 0; g < G2; ++g) {
            const int b = b0 + 4 * g + (lane >> 4);
            if (b >= t.B) continue;
            float sum = 0.f;
#pragma unroll
            for (int f = 0; f < NC_NF; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float y = elu_fast(acc[c][g * NC_NF + f][r] + bj);
                    sum += y;
                    if (t.U) t.U[((size_t)b * N + 4 * f + r) * H + col[c]] = y;
                }
            t.Hm[(size_t)b * H + col[c]] = sum * invN;
        }
    }
    NCT(3); NCT(5);
}

// The one-role form of the same step (every wave produces its share of the images AND consumes): NW = 8 waves = 128 hidden units.
// Kept for the four-head critic launch: 156 VGPRs x 2 waves per SIMD leave room for any launch of the feature chain beside it,
// which the two-role kernel at two column groups per wave (174) does not.
template <int NW>
__global__ __launch_bounds__(64 * NW) void nc_fwd_x3w_kernel(NcFwdBatch nb) {
    constexpr int NT = 64 * NW, KV = 1024 / NT, NKC = 32 / KV, G2 = 2;
    typedef typename NxVec<KV>::T vec_t;
    unsigned char* const L = reinterpret_cast<unsigned char*>(nc_smem);
    const int bid = blockIdx.x;
    NCT(0); NCT(4);
    int ti = 0;
#pragma unroll
    for (int q = 1; q < NC_MAX_TASKS; ++q) if (q < nb.ntasks && bid >= nb.t[q].tile_base) ti = q;
    const NcFwdTask& t = nb.t[ti];
    const int local = bid - t.tile_base;
    const int tb = local / t.tiles_h, th = local - tb * t.tiles_h;
    const int b0 = tb * 8, n0 = th * (16 * NW);
    const int F = t.F, H = t.H, N = t.N;
    const int S = F >> 5;                                   // K steps (F % 32 == 0, checked by the launcher)

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m16 = lane & 15, kq = lane >> 4;

    // ---- producer role ----
    const int kc = tid % NKC, pb = (tid / NKC) & 7, ng = tid / (NKC * 8);
    const bool okb = b0 + pb < t.B;
    const int bsrc = min(b0 + pb, t.B - 1);
    const float* const pmu = t.mean + (size_t)bsrc * t.ld_ml + kc * KV;
    const float* const pls = t.lstd + (size_t)bsrc * t.ld_ml + kc * KV;
    const float* const pnz = t.noise + (size_t)(5 * ng) * F + kc * KV;
    int wofs[5];                                            // byte offset of this thread's chunk in an image, per noise row
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int n = 5 * ng + i;
        wofs[i] = (((pb >> 2) * 5 + (n >> 2)) * 16 + (pb & 3) * 4 + (n & 3)) * NX_RSB + kc * KV * 2;
    }
    float* const sig_dst = (t.sigma_out && th == 0 && ng == 0 && okb) ? t.sigma_out + (size_t)(b0 + pb) * F + kc * KV : nullptr;

    // ---- consumer role ----
    const int col = n0 + 16 * w + m16;
    const bool colok = col < H;
    const float* const wrow = t.W + (size_t)min(col, H - 1) * F + 8 * kq;
    const int aofs = m16 * NX_RSB + kq * 16;

    // (a second register set for these, loaded at the START of step s for step s + 2, measured slower: 30.9k vs 26.7k cycles for
    // the eight steps -- the step is bound by vector-instruction issue, not by the latency of these loads)
    vec_t rmu, rls, rnz[5];
    f32x4 rw0, rw1;
    auto gload = [&](int s) {
        const int k = 32 * s;
        rmu = *reinterpret_cast<const vec_t*>(pmu + k);
        rls = *reinterpret_cast<const vec_t*>(pls + k);
#pragma unroll
        for (int i = 0; i < 5; ++i) rnz[i] = *reinterpret_cast<const vec_t*>(pnz + (size_t)i * F + k);
    };
    auto wload = [&](int s) {
        rw0 = *reinterpret_cast<const f32x4*>(wrow + 32 * s);
        rw1 = *reinterpret_cast<const f32x4*>(wrow + 32 * s + 4);
    };
    float sg[KV];
    auto produce_begin = [&]() {
#pragma unroll
        for (int q = 0; q < KV; ++q) sg[q] = okb ? __expf(clamp_lstd(rls[q])) : 0.f;
    };
    auto produce_row = [&](int i, unsigned char* buf) {
        unsigned h[KV / 2], m[KV / 2], l[KV / 2];
#pragma unroll
        for (int q = 0; q < KV / 2; ++q) {
            const float x0 = okb ? fmaf(sg[2 * q], rnz[i][2 * q], rmu[2 * q]) : 0.f;
            const float x1 = okb ? fmaf(sg[2 * q + 1], rnz[i][2 * q + 1], rmu[2 * q + 1]) : 0.f;
            x3_split2(x0, x1, h[q], m[q], l[q]);
        }
        unsigned char* p = buf + wofs[i];
        if (KV == 4) {
            *reinterpret_cast<u32x2*>(p) = (u32x2){h[0], h[KV / 2 - 1]};
            *reinterpret_cast<u32x2*>(p + NX_IMGB) = (u32x2){m[0], m[KV / 2 - 1]};
            *reinterpret_cast<u32x2*>(p + 2 * NX_IMGB) = (u32x2){l[0], l[KV / 2 - 1]};
        } else {
            *reinterpret_cast<unsigned*>(p) = h[0];
            *reinterpret_cast<unsigned*>(p + NX_IMGB) = m[0];
            *reinterpret_cast<unsigned*>(p + 2 * NX_IMGB) = l[0];
        }
    };

    f32x4 acc[G2][NC_NF];
#pragma unroll
    for (int g = 0; g < G2; ++g)
#pragma unroll
        for (int f = 0; f < NC_NF; ++f) acc[g][f] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)L + (unsigned)aofs;

    // One K step.  MORE = not the last step: the producer half (images of step s + 1, one noise row per two fragments; global
    // operands of step s + 2) is interleaved with the fragments in program order.
    auto step = [&](int s, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        const unsigned aaddr = lds0 + (unsigned)((s & 1) * NX_BUFB);
        unsigned char* const nxt = L + ((s + 1) & 1) * NX_BUFB;
        u32x4 fa[2][3];
        nx_fload<0>(fa[0], aaddr);
        u32x4 bh, bm, bl;
        {
            unsigned h, m, l;
            x3_split2(rw0[0], rw0[1], h, m, l); bh[0] = h; bm[0] = m; bl[0] = l;
            x3_split2(rw0[2], rw0[3], h, m, l); bh[1] = h; bm[1] = m; bl[1] = l;
            x3_split2(rw1[0], rw1[1], h, m, l); bh[2] = h; bm[2] = m; bl[2] = l;
            x3_split2(rw1[2], rw1[3], h, m, l); bh[3] = h; bm[3] = m; bl[3] = l;
        }
        const bf16x8 Bh = __builtin_bit_cast(bf16x8, bh), Bm = __builtin_bit_cast(bf16x8, bm), Bl = __builtin_bit_cast(bf16x8, bl);
        if (MORE) { wload(s + 1); produce_begin(); }
#define NX_FRAG(J)                                                                                                     \
        {                                                                                                              \
            if ((J) + 1 < G2 * NC_NF) nx_fload<((J) + 1 < G2 * NC_NF ? (J) + 1 : 0)>(fa[((J) + 1) & 1], aaddr);        \
            if (MORE && ((J) & 1) == 0) produce_row((J) >> 1, nxt);                                                    \
            nx_claim<((J) + 1 < G2 * NC_NF)>(fa[(J) & 1]);                                                             \
            const bf16x8 Ah = __builtin_bit_cast(bf16x8, fa[(J) & 1][0]), Am = __builtin_bit_cast(bf16x8, fa[(J) & 1][1]), \
                         Al = __builtin_bit_cast(bf16x8, fa[(J) & 1][2]);                                              \
            f32x4 c = acc[(J) / NC_NF][(J) % NC_NF];                                                                   \
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bh, c, 0, 0, 0);                                           \
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bl, c, 0, 0, 0);                                           \
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bm, c, 0, 0, 0);                                           \
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bh, c, 0, 0, 0);                                           \
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bm, c, 0, 0, 0);                                           \
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bh, c, 0, 0, 0);                                           \
            acc[(J) / NC_NF][(J) % NC_NF] = c;                                                                         \
        }
        NX_FRAG(0) NX_FRAG(1) NX_FRAG(2) NX_FRAG(3) NX_FRAG(4) NX_FRAG(5) NX_FRAG(6) NX_FRAG(7) NX_FRAG(8) NX_FRAG(9)
#undef NX_FRAG
        if (MORE) gload(min(s + 2, S - 1));                 // (the last one re-reads step S - 1: no branch)
        __syncthreads();
    };

    gload(0); wload(0);
    produce_begin();
#pragma unroll
    for (int i = 0; i < 5; ++i) produce_row(i, L);
    gload(S > 1 ? 1 : 0);
    __syncthreads();
    NCT(1);
    for (int s = 0; s + 1 < S; ++s) step(s, std::true_type{});
    step(S - 1, std::false_type{});
    NCT(2);

    if (sig_dst) {          // exp(clamp(log_std)) for the backward passes: one thread group of column tile 0 per batch row
        for (int s = 0; s < S; ++s) {
            const vec_t ls = *reinterpret_cast<const vec_t*>(pls + 32 * s);
            vec_t o;
#pragma unroll
            for (int q = 0; q < KV; ++q) o[q] = expf(clamp_lstd(ls[q]));
            *reinterpret_cast<vec_t*>(sig_dst + 32 * s) = o;
        }
    }

    if (!colok) return;
    const float bj = t.bias[col];
    const float invN = 1.0f / (float)N;
#pragma unroll
    for (int g = 0; g < G2; ++g) {
        const int b = b0 + 4 * g + (lane >> 4);
        if (b >= t.B) continue;
        float sum = 0.f;
#pragma unroll
        for (int f = 0; f < NC_NF; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y = elu_fast(acc[g][f][r] + bj);
                sum += y;
#ifndef RL_NC_NOU
                if (t.U) t.U[((size_t)b * N + 4 * f + r) * H + col] = y;
#endif
            }
        t.Hm[(size_t)b * H + col] = sum * invN;
    }
    NCT(3); NCT(5);
}

// The same step on v_mfma_f32_32x32x16_bf16 (one-role, 4 waves x 32 hidden units = 128 per workgroup, ONE wave per SIMD): a 32x32x16
// MFMA holds the vector issue port for 8 of its 32 cycles instead of 8 of 16, so the producer's VALU work fits in the shadow of the
// wave's own MFMAs; a 16-byte fragment read feeds 2x the flops; 322 VGPRs per SIMD stay free for the feature chain.
// Row mapping: the C/D map of the 32x32 tile gives lane half h = lane >> 5 the rows 8 q + 4 h + i (q, i < 4) of every tile, 16 of its
// 32 rows; over the five tiles of a step's 160 rows that is 80 slots sigma = 16 t + 4 q + i, which are dealt to (batch row 4 h + sigma / 20,
// noise row sigma % 20): every lane again owns ALL 20 noise rows of four batch rows for its column, in compile-time register positions.
typedef float nq_f32x16 __attribute__((ext_vector_type(16)));
template <int T5, int KB> __device__ __forceinline__ void nq_fload(u32x4 (&d)[3], unsigned addr) {
    nx_read<T5 * 32 * NX_RSB + KB * 32>(d[0], addr);
    nx_read<T5 * 32 * NX_RSB + KB * 32 + NX_IMGB>(d[1], addr);
    nx_read<T5 * 32 * NX_RSB + KB * 32 + 2 * NX_IMGB>(d[2], addr);
}

__global__ __launch_bounds__(256) void nc_fwd_x3q_kernel(NcFwdBatch nb) {
    constexpr int KV = 4;
    unsigned char* const L = reinterpret_cast<unsigned char*>(nc_smem);
    const int bid = blockIdx.x;
    NCT(0); NCT(4);
    int ti = 0;
#pragma unroll
    for (int q = 1; q < NC_MAX_TASKS; ++q) if (q < nb.ntasks && bid >= nb.t[q].tile_base) ti = q;
    const NcFwdTask& t = nb.t[ti];
    const int local = bid - t.tile_base;
    const int tb = local / t.tiles_h, th = local - tb * t.tiles_h;
    const int b0 = tb * 8, n0 = th * 128;
    const int F = t.F, H = t.H, N = t.N;
    const int S = F >> 5;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;

    // ---- producer role: thread = (k chunk of 4, batch row, noise group of 5) ----
    const int kc = tid & 7, pb = (tid >> 3) & 7, ng = tid >> 6;
    const bool okb = b0 + pb < t.B;
    const int bsrc = min(b0 + pb, t.B - 1);
    const float* const pmu = t.mean + (size_t)bsrc * t.ld_ml + kc * KV;
    const float* const pls = t.lstd + (size_t)bsrc * t.ld_ml + kc * KV;
    const float* const pnz = t.noise + (size_t)(5 * ng) * F + kc * KV;
    int wofs[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int sig = (pb & 3) * 20 + 5 * ng + i, rem = sig & 15;
        const int lrow = 32 * (sig >> 4) + 8 * (rem >> 2) + 4 * (pb >> 2) + (rem & 3);
        wofs[i] = lrow * NX_RSB + kc * KV * 2;
    }
    float* const sig_dst = (t.sigma_out && th == 0 && ng == 0 && okb) ? t.sigma_out + (size_t)(b0 + pb) * F + kc * KV : nullptr;

    // ---- consumer role: 32 columns per wave ----
    const int c32 = lane & 31, half = lane >> 5;
    const int col = n0 + 32 * w + c32;
    const bool colok = col < H;
    const float* const wrow = t.W + (size_t)min(col, H - 1) * F + 8 * half;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)L + (unsigned)(c32 * NX_RSB + half * 16);

    f32x4 rmu, rls, rnz[5];
    f32x4 rw[2][2];
    // with bf16x3 images of W (NcFwdTask::W3, kept by the optimizer launch): the three B fragments of a k block as they are multiplied,
    // 16 bytes each at ((3 s + image) H + col) 64 + (2 kb + half) 16 -- no split in this kernel (88 of its ~260 VALU instructions per step)
    u32x4 rb[2][3];
    const unsigned char* const w3p = t.W3 ? t.W3 + (size_t)min(col, H - 1) * 64 + half * 16 : nullptr;
    auto gload = [&](int s) {
        const int k = 32 * s;
        rmu = *reinterpret_cast<const f32x4*>(pmu + k);
        rls = *reinterpret_cast<const f32x4*>(pls + k);
#pragma unroll
        for (int i = 0; i < 5; ++i) rnz[i] = *reinterpret_cast<const f32x4*>(pnz + (size_t)i * F + k);
    };
    auto wload = [&](int s, auto w3_tag) {
        if (decltype(w3_tag)::value) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int img = 0; img < 3; ++img) rb[kb][img] = *reinterpret_cast<const u32x4*>(w3p + ((size_t)(3 * s + img) * H) * 64 + kb * 32);
        } else {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                rw[kb][0] = *reinterpret_cast<const f32x4*>(wrow + 32 * s + 16 * kb);
                rw[kb][1] = *reinterpret_cast<const f32x4*>(wrow + 32 * s + 16 * kb + 4);
            }
        }
    };
    float sg[KV];
    auto produce_begin = [&]() {
#pragma unroll
        for (int q = 0; q < KV; ++q) sg[q] = okb ? __expf(clamp_lstd(rls[q])) : 0.f;
    };
    auto produce_row = [&](int i, unsigned char* buf) {
        u32x2 h, m, l;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float x0 = okb ? fmaf(sg[2 * q], rnz[i][2 * q], rmu[2 * q]) : 0.f;
            const float x1 = okb ? fmaf(sg[2 * q + 1], rnz[i][2 * q + 1], rmu[2 * q + 1]) : 0.f;
            unsigned hh, mm, ll;
            x3_split2(x0, x1, hh, mm, ll);
            h[q] = hh; m[q] = mm; l[q] = ll;
        }
        unsigned char* p = buf + wofs[i];
        *reinterpret_cast<u32x2*>(p) = h;
        *reinterpret_cast<u32x2*>(p + NX_IMGB) = m;
        *reinterpret_cast<u32x2*>(p + 2 * NX_IMGB) = l;
    };

    nq_f32x16 acc[5];
#pragma unroll
    for (int t5 = 0; t5 < 5; ++t5)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t5][r] = 0.f;

    // (splitting W one step ahead, a quarter per odd fragment in the shadow of the previous step's MFMAs, measured slower: 28.8k vs 25.5k
    // cycles for the eight steps)
    auto step = [&](int s, auto more_tag, auto w3_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        constexpr bool W3 = decltype(w3_tag)::value;
        const unsigned aaddr = lds0 + (unsigned)((s & 1) * NX_BUFB);
        unsigned char* const nxt = L + ((s + 1) & 1) * NX_BUFB;
        u32x4 fa[2][3];
        nq_fload<0, 0>(fa[0], aaddr);
        bf16x8 Bh[2], Bm[2], Bl[2];
        if (W3) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) { Bh[kb] = __builtin_bit_cast(bf16x8, rb[kb][0]); Bm[kb] = __builtin_bit_cast(bf16x8, rb[kb][1]); Bl[kb] = __builtin_bit_cast(bf16x8, rb[kb][2]); }
        } else
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            u32x4 bh, bm, bl;
            unsigned h, m, l;
            x3_split2(rw[kb][0][0], rw[kb][0][1], h, m, l); bh[0] = h; bm[0] = m; bl[0] = l;
            x3_split2(rw[kb][0][2], rw[kb][0][3], h, m, l); bh[1] = h; bm[1] = m; bl[1] = l;
            x3_split2(rw[kb][1][0], rw[kb][1][1], h, m, l); bh[2] = h; bm[2] = m; bl[2] = l;
            x3_split2(rw[kb][1][2], rw[kb][1][3], h, m, l); bh[3] = h; bm[3] = m; bl[3] = l;
            Bh[kb] = __builtin_bit_cast(bf16x8, bh); Bm[kb] = __builtin_bit_cast(bf16x8, bm); Bl[kb] = __builtin_bit_cast(bf16x8, bl);
        }
        if (MORE) { wload(s + 1, w3_tag); produce_begin(); }
        // fragment J = (tile J / 2, k block J % 2); one noise row of the next step's images per two fragments
#define NQ_FRAG(J)                                                                                                     \
        {                                                                                                              \
            if ((J) + 1 < 10) nq_fload<((J) + 1 < 10 ? ((J) + 1) / 2 : 0), ((J) + 1) % 2>(fa[((J) + 1) & 1], aaddr);   \
            if (MORE && ((J) & 1) == 0) produce_row((J) >> 1, nxt);                                                    \
            nx_claim<((J) + 1 < 10)>(fa[(J) & 1]);                                                                     \
            const bf16x8 Ah = __builtin_bit_cast(bf16x8, fa[(J) & 1][0]), Am = __builtin_bit_cast(bf16x8, fa[(J) & 1][1]), \
                         Al = __builtin_bit_cast(bf16x8, fa[(J) & 1][2]);                                              \
            nq_f32x16 c = acc[(J) / 2];                                                                                \
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh[(J) % 2], c, 0, 0, 0);                                  \
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl[(J) % 2], c, 0, 0, 0);                                  \
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm[(J) % 2], c, 0, 0, 0);                                  \
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh[(J) % 2], c, 0, 0, 0);                                  \
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm[(J) % 2], c, 0, 0, 0);                                  \
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh[(J) % 2], c, 0, 0, 0);                                  \
            acc[(J) / 2] = c;                                                                                          \
        }
        NQ_FRAG(0) NQ_FRAG(1) NQ_FRAG(2) NQ_FRAG(3) NQ_FRAG(4) NQ_FRAG(5) NQ_FRAG(6) NQ_FRAG(7) NQ_FRAG(8) NQ_FRAG(9)
#undef NQ_FRAG
        // (a sched_group_barrier pipeline {1 MFMA, 4-6 VALU} x 60 over this block changed nothing: 22.7-24.5 us against 22.7)
        if (MORE) gload(min(s + 2, S - 1));
        __syncthreads();
    };

    gload(0);
    if (w3p) wload(0, std::true_type{}); else wload(0, std::false_type{});
    produce_begin();
#pragma unroll
    for (int i = 0; i < 5; ++i) produce_row(i, L);
    gload(S > 1 ? 1 : 0);
    __syncthreads();
    NCT(1);
    if (w3p) {
        for (int s = 0; s + 1 < S; ++s) step(s, std::true_type{}, std::true_type{});
        step(S - 1, std::false_type{}, std::true_type{});
    } else {
        for (int s = 0; s + 1 < S; ++s) step(s, std::true_type{}, std::false_type{});
        step(S - 1, std::false_type{}, std::false_type{});
    }
    NCT(2);

    if (sig_dst) {
        for (int s = 0; s < S; ++s) {
            const f32x4 ls = *reinterpret_cast<const f32x4*>(pls + 32 * s);
            f32x4 o;
#pragma unroll
            for (int q = 0; q < KV; ++q) o[q] = expf(clamp_lstd(ls[q]));
            *reinterpret_cast<f32x4*>(sig_dst + 32 * s) = o;
        }
    }
    if (!colok) return;
    // slot sigma = 16 t5 + r of this lane's half: batch row b0 + 4 half + sigma / 20, noise row sigma % 20.  Walked per batch row (static
    // register positions; the row / store guards are four uniform-ish branches, not eighty)
    const float bj = t.bias[col];
    const float invN = 1.0f / (float)N;
#pragma unroll
    for (int bl = 0; bl < 4; ++bl) {
        const int b = b0 + 4 * half + bl;
        float y[20];
        float sum = 0.f;
#pragma unroll
        for (int n = 0; n < 20; ++n) {
            const int sig = 20 * bl + n;
            y[n] = elu_fast(acc[sig >> 4][sig & 15] + bj);
            sum += y[n];
        }
        if (b < t.B) {
            t.Hm[(size_t)b * H + col] = sum * invN;
#ifndef RL_NC_NOU            /* (timing-only diagnostic build: no U stores) */
            if (t.U) {
                float* up = t.U + ((size_t)b * N) * H + col;
                if (nb.nt_u) {
                    // (10.5 MB of elu outputs per forward that nothing reads before the backward -- streamed past the L2 lines the
                    //  feature chain's launches live in; RLREP_ENABLE=nc_u_nt, measured in docs/history/r06.md)
#pragma unroll
                    for (int n = 0; n < 20; ++n) __builtin_nontemporal_store(y[n], up + (size_t)n * H);
                } else {
#pragma unroll
                    for (int n = 0; n < 20; ++n) up[(size_t)n * H] = y[n];
                }
            }
#endif
        }
    }
    NCT(3); NCT(5);
}

// ------------------------------------------------------------------------------------------------
// dL/d(mean, log_std), both heads (actor step).
// workgroup = 8 waves: waves 0-3 consume head 0, waves 4-7 head 1; wave (w&3) owns 16 feature columns;
// tile = 4 batch rows x 20 noise rows x 64 feature columns; inner loop over the H hidden units of the head.
//
// The A operand dPre[(b,n), j] = GH[b,j]/N * elu'(U[(b,n), j]) is needed by all four column-group waves of a
// head, so it is produced ONCE per workgroup: all 512 threads load their share of the [80 x 16] U tile of each head
// with one 16-byte load, apply elu' and the GH scale, and park the result in an LDS ring; the consumer waves only
// issue ds_read_b128 + MFMA.  Pipeline per inner step s: global loads at iteration s-3 (registers), LDS write at
// s-2, consumed at s -- two iterations (>= 1280 MFMA cycles per SIMD) cover the L2/MALL latency of U, which the
// previous (register-prefetch, per-wave) version could not (WAIT_ANY 45 % of wave cycles, profiles/r01_pmc_summary).
// ------------------------------------------------------------------------------------------------
#define NCDX_RING 4
#define NCDX_ALD 20          // LDS row stride of the A tile in floats (16 + 4 pad, keeps 16-byte alignment)

template <bool VEC>
__global__ __launch_bounds__(512) void nc_dx_kernel(NcDxTask t) {
    __shared__ __attribute__((aligned(16))) float a_s[NCDX_RING][2][80 * NCDX_ALD];   // 51 KB
    __shared__ float red[4][NC_NF][4][64];                                            // 20 KB: head-1 partial accumulators
    // XCD-aware tile order: workgroups b and b+8 share an XCD (and its L2).  The tiles_k column tiles of one
    // batch tile read the SAME rows of U, so they are given the same (blockIdx % 8) whenever the grid allows.
    int bid = blockIdx.x;
    {
        const int ntb = t.ntiles / t.tiles_k;
        if ((ntb & 7) == 0) {
            const int x = bid & 7, y = bid >> 3;
            bid = ((y / t.tiles_k) * 8 + x) * t.tiles_k + (y % t.tiles_k);
        }
    }
    const int tb = bid / t.tiles_k, tk = bid - tb * t.tiles_k;
    const int b0 = tb * 4, kc0 = tk * 64;
    const int F = t.F, H = t.H, N = t.N;
    const int tid = threadIdx.x, lane = tid & 63, w8 = tid >> 6;
    const int w = w8 & 3, h = (t.nheads > 1) ? (w8 >> 2) : 0;
    const bool consumer = (t.nheads > 1) || (w8 < 4);
    const int m16 = lane & 15, kq = lane >> 4;
    const int bp = m16 >> 2, nn = m16 & 3;
    const int kcol = kc0 + 16 * w + m16;
    const bool colok = kcol < F;
    const float invN = 1.0f / (float)N;
    const int T = (H + 15) / 16;                                  // inner steps per head

    // ---- loader role: element q of the step's 2 x [80 x 16] tile = (head, row, 4-wide j group) ----
    // q in [0, 640): thread tid owns q = tid and (tid < 128) q = 512 + tid.  All loads are BRANCH-FREE: rows / heads /
    // inner indices outside the problem are clamped to a valid address and the value is zeroed when it is written to
    // LDS (conditional loads compile to s_cbranch_execz + s_waitcnt vmcnt(0) per load: 359 branches and 21 full
    // drains in the previous build of this kernel).
    struct LReg { float u[4]; float g[4]; };
    struct LSrc { const float* up; const float* gp; int jq; int lofs; bool ok; };
    auto lsrc = [&](int q) {
        LSrc o;
        const int hh = q / 320, e = q - hh * 320, row = e >> 2;
        o.jq = e & 3;
        const int hc = min(hh, t.nheads - 1);
        const int lb = row / 20, n = row - lb * 20;
        const int b = min(b0 + lb, t.B - 1);
        o.up = (hc ? t.U[1] : t.U[0]) + ((size_t)b * N + n) * H;
        o.gp = (hc ? t.GH[1] : t.GH[0]) + (size_t)b * t.ldgh;
        o.lofs = hh * (80 * NCDX_ALD) + row * NCDX_ALD + 4 * o.jq;
        o.ok = (hh < t.nheads) && (b0 + lb < t.B);
        return o;
    };
    const LSrc sa = lsrc(tid), sb = lsrc(512 + (tid & 127));
    auto lload = [&](const LSrc& src, int step, LReg& r) {
        const int j0 = step * 16 + 4 * src.jq;
        if (VEC) {
            const int jc = min(j0, H - 4);
            const f32x4 u = *reinterpret_cast<const f32x4*>(src.up + jc);
            const f32x4 g = *reinterpret_cast<const f32x4*>(src.gp + jc);
#pragma unroll
            for (int s = 0; s < 4; ++s) { r.u[s] = u[s]; r.g[s] = g[s]; }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) { const int jc = min(j0 + s, H - 1); r.u[s] = src.up[jc]; r.g[s] = src.gp[jc]; }
        }
    };
    auto lwrite = [&](const LSrc& src, int step, const LReg& r) {
        const int j0 = step * 16 + 4 * src.jq;
        // clamped loads returned finite in-matrix values, so out-of-range elements are removed by a zero SCALE (a
        // select on the whole expression is compiled into control flow around it)
        f32x4 v;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float m = (src.ok && j0 + (VEC ? 0 : s) < H) ? invN : 0.f;
            const float u = r.u[s];
            v[s] = (r.g[s] * m) * fminf(u + 1.f, 1.f);          // elu'(out) = out > 0 ? 1 : out + 1 = min(out + 1, 1)
        }
        *reinterpret_cast<f32x4*>(&a_s[step % NCDX_RING][0][src.lofs]) = v;
    };
    const bool two = tid < 128;
    LReg ra0, rb0, ra1, rb1;                                      // two register sets x (first, second element); named, never
                                                                  // runtime-indexed (a runtime index would put them in scratch)
    const float* W = t.W[h] + (colok ? kcol : 0);
    const unsigned Fu = (unsigned)F;
    float wv[4], wn[4];
    auto wload = [&](int step, float (&o)[4]) {
        const int j0 = step * 16 + 4 * kq;
#pragma unroll
        for (int s = 0; s < 4; ++s) o[s] = W[(unsigned)min(j0 + s, H - 1) * Fu];
    };
    auto wmask = [&](int step, float (&o)[4]) {
        const int j0 = step * 16 + 4 * kq;
#pragma unroll
        for (int s = 0; s < 4; ++s) o[s] = (consumer && colok && j0 + s < H) ? o[s] : 0.f;
    };
    // epilogue operands (this lane's noise column and log-std), fetched now as volatile asm loads so that hipcc cannot
    // sink them below the final barrier; they are older than every load of the pipeline below and vmcnt retires in
    // order, so the compiler's counted waits stay valid; claimed by the explicit wait before the epilogue
    float nzv[20], lsv;
    {
        const int kc = colok ? kcol : 0;
#pragma unroll
        for (int q = 0; q < 20; ++q) {
            const float* p = t.noise + (size_t)min(q, N - 1) * F + kc;
            asm volatile("global_load_dword %0, %1, off" : "=v"(nzv[q]) : "v"(p));
        }
        const float* p = t.lstd + (size_t)min(b0 + (lane >> 4), t.B - 1) * t.ld_l + kc;
        asm volatile("global_load_dword %0, %1, off" : "=v"(lsv) : "v"(p));
    }
    // prologue: steps 0 and 1 are fetched together (one round trip), staged, and step 2 is left in flight in set 0
    lload(sa, 0, ra0); if (two) lload(sb, 0, rb0);
    if (1 < T) { lload(sa, 1, ra1); if (two) lload(sb, 1, rb1); }
    wload(0, wv);
    __builtin_amdgcn_sched_barrier(0);
    lwrite(sa, 0, ra0); if (two) lwrite(sb, 0, rb0);
    if (1 < T) { lwrite(sa, 1, ra1); if (two) lwrite(sb, 1, rb1); }
    if (2 < T) { lload(sa, 2, ra0); if (two) lload(sb, 2, rb0); }
    wmask(0, wv);
    __syncthreads();

    f32x4 acc[NC_NF];
#pragma unroll
    for (int f = 0; f < NC_NF; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // one inner step; (an, bn) = register set that receives step+3, (ao, bo) = set holding step+2
    auto iter = [&](int step, LReg& an, LReg& bn, LReg& ao, LReg& bo) {
        const int ls = step + 3;
        // program order matters: vmcnt retires in order, so the W fragment (needed at the END of this iteration) is
        // issued BEFORE the U/GH loads (needed an iteration later); waiting for W then leaves the U loads in flight
        if (step + 1 < T) wload(step + 1, wn);
        if (ls < T) { lload(sa, ls, an); if (two) lload(sb, ls, bn); }
        __builtin_amdgcn_sched_barrier(0);
        if (consumer) {
            const float* as = &a_s[step % NCDX_RING][h][0];
            f32x4 a4[NC_NF];
#pragma unroll
            for (int f = 0; f < NC_NF; ++f) a4[f] = *reinterpret_cast<const f32x4*>(&as[(bp * 20 + 4 * f + nn) * NCDX_ALD + 4 * kq]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int f = 0; f < NC_NF; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[f][s], wv[s], acc[f], 0, 0, 0);
        }
        if (step + 1 < T) {
            wmask(step + 1, wn);
#pragma unroll
            for (int s = 0; s < 4; ++s) wv[s] = wn[s];
        }
        if (step + 2 < T) { lwrite(sa, step + 2, ao); if (two) lwrite(sb, step + 2, bo); }
        __syncthreads();
    };
    for (int step = 0; step < T; step += 2) {
        iter(step, ra1, rb1, ra0, rb0);                            // step even: step+2 sits in set 0, step+3 goes to set 1
        if (step + 1 < T) iter(step + 1, ra0, rb0, ra1, rb1);
    }

    // head-1 partial sums travel through LDS (with one head, waves 4-7 never issued an MFMA: they deposit zeros)
    if (w8 >= 4) {
#pragma unroll
        for (int f = 0; f < NC_NF; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[w][f][r][lane] = acc[f][r];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // claim the epilogue operands fetched in the prologue
#pragma unroll
    for (int q = 0; q < 20; ++q) asm volatile("" : "+v"(nzv[q]));
    asm volatile("" : "+v"(lsv));
    __syncthreads();
    if (w8 >= 4 || !colok) return;
    const int bo = b0 + (lane >> 4);
    if (bo >= t.B) return;
    float dmu = 0.f, dls = 0.f;
#pragma unroll
    for (int f = 0; f < NC_NF; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = acc[f][r] + red[w][f][r][lane];
            dmu += v;
            dls = fmaf(v, nzv[4 * f + r], dls);
        }
    t.G[(size_t)bo * t.ldg + kcol] = dmu;
    t.G[(size_t)bo * t.ldg + F + kcol] = dls * expf(clamp_lstd(lsv)) * lstd_mask(lsv);
}

// ------------------------------------------------------------------------------------------------
// dL/d(mean, log_std) on the BF16 matrix pipe at fp32 accuracy (bf16x3): the same tile as nc_dx_kernel (4 batch rows x 20 noise
// rows x 64 feature columns, one wave per 16 columns) with the machinery of nc_fwd_x3_kernel: the A operand
// dPre[(b,n), j] = GH[b,j]/N * elu'(U[(b,n), j]) is generated and split ONCE per workgroup into three bf16 images [80 rows][32 j]
// in LDS (two buffers, one barrier per 32-deep step), the waves read their fragments with pipelined inline-asm ds_read_b128 and
// split their own 32 x 16 W fragment in registers.  The inner loop runs over BOTH heads (K = nheads * H), so no partial sums
// cross waves: the epilogue is the in-register reduction over the noise rows.  Needs H % 32 == 0 and 8-byte rows.
// ------------------------------------------------------------------------------------------------
#define NDX_ROWS 80
#define NDX_IMGB (NDX_ROWS * NX_RSB)
#define NDX_BUFB (3 * NDX_IMGB)

template <int OFF> __device__ __forceinline__ void ndx_fload_at(u32x4 (&d)[3], unsigned addr) {
    nx_read<OFF>(d[0], addr); nx_read<OFF + NDX_IMGB>(d[1], addr); nx_read<OFF + 2 * NDX_IMGB>(d[2], addr);
}

__global__ __launch_bounds__(512) void nc_dx_x3_kernel(NcDxTask t) {
    // 8 waves, two roles: waves 0-3 CONSUME (split their W fragment, read the A fragments, 30 MFMAs per step, epilogue), waves 4-7
    // PRODUCE (stream U and GH, build and split dPre, write the images of the next step).  With one workgroup per CU that puts one
    // wave of each role on every SIMD: the producer's vector work runs in the shadow of the consumer's MFMAs (as one role per wave,
    // 256 threads, the step was their SUM: 2.0k cycles for 480 of MFMA).
    __shared__ __attribute__((aligned(16))) unsigned char L[2 * NDX_BUFB];      // 38.4 KB
    int bid = blockIdx.x;
    NCT(0); NCT(4);
    {   // XCD-aware tile order (as nc_dx_kernel): the column tiles of one batch tile read the same rows of U
        const int ntb = t.ntiles / t.tiles_k;
        if ((ntb & 7) == 0) {
            const int x = bid & 7, y = bid >> 3;
            bid = ((y / t.tiles_k) * 8 + x) * t.tiles_k + (y % t.tiles_k);
        }
    }
    const int tb = bid / t.tiles_k, tk = bid - tb * t.tiles_k;
    const int b0 = tb * 4, kc0 = tk * 64;
    const int F = t.F, H = t.H, N = t.N;
    const int SH = H >> 5;                                  // 32-deep steps per head
    const int S = SH * t.nheads;
    const int w8 = threadIdx.x >> 6;

    if (w8 >= 4) {
        // ================= producer: thread = (j pair jc, batch row pb, noise group ng of 5 rows) =================
        const int tid = threadIdx.x - 256;
        const int jc = tid & 15, pb = tid >> 6, ng = (tid >> 4) & 3;
        const bool okb = b0 + pb < t.B;
        const int bsrc = min(b0 + pb, t.B - 1);
        const size_t urow = ((size_t)bsrc * N + 5 * ng) * H + 2 * jc;
        const size_t grow = (size_t)bsrc * t.ldgh + 2 * jc;
        const float m = okb ? 1.0f / (float)N : 0.f;
        int wofs[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int n = 5 * ng + i;
            wofs[i] = ((n >> 2) * 16 + pb * 4 + (n & 3)) * NX_RSB + jc * 4;
        }
        struct PReg { f32x2v u[5], g; };
        auto gload = [&](int s, PReg& r) {
            s = min(s, S - 1);                              // past the end: re-read the last step (no branch), never produced
            const int h = s >= SH ? 1 : 0, j0 = 32 * (s - h * SH);
            const float* up = (h ? t.U[1] : t.U[0]) + urow + j0;
            const float* gp = (h ? t.GH[1] : t.GH[0]) + grow + j0;
            r.g = *reinterpret_cast<const f32x2v*>(gp);
#pragma unroll
            for (int i = 0; i < 5; ++i) r.u[i] = *reinterpret_cast<const f32x2v*>(up + (size_t)i * H);
        };
        auto produce = [&](const PReg& r, unsigned char* buf) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const float a0 = (r.g[0] * m) * fminf(r.u[i][0] + 1.f, 1.f);      // elu'(out) = min(out + 1, 1)
                const float a1 = (r.g[1] * m) * fminf(r.u[i][1] + 1.f, 1.f);
                unsigned h, md, l;
                x3_split2(a0, a1, h, md, l);
                unsigned char* p = buf + wofs[i];
                *reinterpret_cast<unsigned*>(p) = h;
                *reinterpret_cast<unsigned*>(p + NDX_IMGB) = md;
                *reinterpret_cast<unsigned*>(p + 2 * NDX_IMGB) = l;
            }
        };
        // two named register sets, loads two steps ahead of their use.  (Four sets / four steps ahead measured the same: the step is
        // bound by vector-instruction ISSUE on the SIMD the two roles share -- consumer 460 cycles of W split + 240 of MFMA issue,
        // producer ~500 -- not by the latency of U; tools/exp/nc_timeline.py.)
        PReg ra, rb;
        gload(0, ra); gload(1, rb);
        produce(ra, L);
        gload(2, ra);
        __syncthreads();
        for (int s = 0; s < S; s += 2) {
            if (s + 1 < S) produce(rb, L + NDX_BUFB);       // images of step s + 1
            gload(s + 3, rb);
            __syncthreads();
            if (s + 1 < S) {
                if (s + 2 < S) produce(ra, L);              // images of step s + 2
                gload(s + 4, ra);
                __syncthreads();
            }
        }
        return;
    }

    // ================= consumer =================
    const int lane = threadIdx.x & 63, w = w8;
    const int m16 = lane & 15, kq = lane >> 4;
    const int kcol = kc0 + 16 * w + m16;
    const bool colok = kcol < F;
    const int kcl = colok ? kcol : 0;
    const int aofs = m16 * NX_RSB + kq * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)L + (unsigned)aofs;
    struct WReg { float v[8]; };
    auto wload = [&](int s, WReg& r) {
        s = min(s, S - 1);
        const int h = s >= SH ? 1 : 0, j0 = 32 * (s - h * SH);
        const float* wp = (h ? t.W[1] : t.W[0]) + (size_t)(j0 + 8 * kq) * F + kcl;
#pragma unroll
        for (int q = 0; q < 8; ++q) r.v[q] = wp[(size_t)q * F];
    };
    // epilogue operands (this lane's noise column and log-std): volatile asm loads, older than every load of the loop
    float nzv[20], lsv;
    {
#pragma unroll
        for (int q = 0; q < 20; ++q) {
            const float* p = t.noise + (size_t)min(q, N - 1) * F + kcl;
            asm volatile("global_load_dword %0, %1, off" : "=v"(nzv[q]) : "v"(p));
        }
        const float* p = t.lstd + (size_t)min(b0 + (lane >> 4), t.B - 1) * t.ld_l + kcl;
        asm volatile("global_load_dword %0, %1, off" : "=v"(lsv) : "v"(p));
    }
    f32x4 acc[NC_NF];
#pragma unroll
    for (int f = 0; f < NC_NF; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    WReg wa, wb;                                            // W fragments of steps s and s + 1; refilled two steps ahead
    wload(0, wa); wload(1, wb);
    __syncthreads();
    NCT(1);
    auto cstep = [&](int s, WReg& rw) {
        const unsigned aaddr = lds0 + (unsigned)((s & 1) * NDX_BUFB);
        // two register sets, one fragment ahead (all five fragments in flight at the step start measured slower: 1.8k vs 1.5k cycles)
        u32x4 fa[2][3];
        ndx_fload_at<0>(fa[0], aaddr);
        u32x4 bh, bm, bl;
        {
            unsigned h, m, l;
#pragma unroll
            for (int q = 0; q < 4; ++q) { x3_split2(rw.v[2 * q], rw.v[2 * q + 1], h, m, l); bh[q] = h; bm[q] = m; bl[q] = l; }
        }
        const bf16x8 Bh = __builtin_bit_cast(bf16x8, bh), Bm = __builtin_bit_cast(bf16x8, bm), Bl = __builtin_bit_cast(bf16x8, bl);
        wload(s + 2, rw);
#define NDX_FRAG(J)                                                                                                    \
        {                                                                                                              \
            if ((J) + 1 < NC_NF) ndx_fload_at<((J) + 1 < NC_NF ? (J) + 1 : 0) * 16 * NX_RSB>(fa[((J) + 1) & 1], aaddr); \
            nx_claim<((J) + 1 < NC_NF)>(fa[(J) & 1]);                                                                  \
            const bf16x8 Ah = __builtin_bit_cast(bf16x8, fa[(J) & 1][0]), Am = __builtin_bit_cast(bf16x8, fa[(J) & 1][1]), \
                         Al = __builtin_bit_cast(bf16x8, fa[(J) & 1][2]);                                              \
            f32x4 c = acc[(J)];                                                                                        \
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bh, c, 0, 0, 0);                                           \
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bl, c, 0, 0, 0);                                           \
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bm, c, 0, 0, 0);                                           \
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bh, c, 0, 0, 0);                                           \
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bm, c, 0, 0, 0);                                           \
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bh, c, 0, 0, 0);                                           \
            acc[(J)] = c;                                                                                              \
        }
        NDX_FRAG(0) NDX_FRAG(1) NDX_FRAG(2) NDX_FRAG(3) NDX_FRAG(4)
#undef NDX_FRAG
        __syncthreads();
    };
    for (int s = 0; s < S; s += 2) {
        cstep(s, wa);
        if (s + 1 < S) cstep(s + 1, wb);
    }

    NCT(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // claim the epilogue operands fetched in the prologue
#pragma unroll
    for (int q = 0; q < 20; ++q) asm volatile("" : "+v"(nzv[q]));
    asm volatile("" : "+v"(lsv));
    if (!colok) return;
    const int bo = b0 + (lane >> 4);
    if (bo >= t.B) return;
    float dmu = 0.f, dls = 0.f;
#pragma unroll
    for (int f = 0; f < NC_NF; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = acc[f][r];
            dmu += v;
            dls = fmaf(v, nzv[4 * f + r], dls);
        }
    t.G[(size_t)bo * t.ldg + kcol] = dmu;
    t.G[(size_t)bo * t.ldg + F + kcol] = dls * expf(clamp_lstd(lsv)) * lstd_mask(lsv);
    NCT(3); NCT(5);
}

// ------------------------------------------------------------------------------------------------
// dL/dW[j,k] = sum_{(b,n)} dPre[(b,n), j] * x[(b,n), k],   dL/dbias[j] = sum dPre[(b,n), j]
//   dPre[(b,n), j] = GH[b,j]/N * elu'(U[(b,n), j])
// workgroup = 8 waves sharing one 16(j) x 32(k) output tile; the 20*B inner rows are dealt to the waves in
// chunks of 16 rows (lane group kq takes 4 consecutive rows = one batch row b, four noise rows);
// partial tiles are summed through LDS in fixed wave order.
//
// Only U is streamed (4 dword loads per lane and chunk).  Everything indexed by the batch row alone -- the tile's 32
// columns of mean and sigma, its 16 columns of GH/N -- is staged ONCE per block of NCDW_BB batch rows in LDS
// (row strides 80 / 16 floats: the two batch rows a chunk can touch fall on disjoint banks).  That takes the loads in
// flight per group of four chunks from 36 to 16, so that THREE groups (48 < the 63 the vmcnt counter can track)
// are in flight behind the group being multiplied, in four named register sets: 3 x 1024 MFMA cycles x 2 waves per
// SIMD cover the L2/MALL latency of U.  (The previous version kept one group ahead, copied it with `cur = nxt` --
// a full vmcnt(0) drain per iteration -- and ran at 32 us for 11 us of MFMA work.)
// ------------------------------------------------------------------------------------------------
#define NCDW_BB 256          // batch rows per staged block
#define NCDW_TLD 80          // [mean 32 | sigma 32 | pad 16]
// LEAN: the variant for the deferred critic / actor chain, which runs BESIDE the latency-bound feature chain of the next train().  A
// 16-row-engine workgroup (one wave per SIMD, 100-196 VGPRs) can only start on a CU whose SIMDs still have that many registers free:
// the full variant (233 VGPRs x 2 waves per SIMD) leaves 46 and so blocked every feature-chain launch for its whole 25 us; capped at
// 128 VGPRs (two register sets of U in flight instead of four) it is slower alone and the pair of chains 7 % faster (2480 -> 2650
// train()/s).
template <bool LEAN>
__global__ __launch_bounds__(512, LEAN ? 4 : 2) void nc_dw_kernel(NcDwBatch nb) {
    __shared__ float red[8][2][4][64];          // 16 KB
    __shared__ float bsum[8][16];
    constexpr int NZLD = 36;                    // 4 rows apart -> 16 banks apart: the four kq lane groups do not collide
    __shared__ float nz_s[32 * NZLD];           // noise[n][32 cols of this tile]
    float* tab_s = nc_smem;                     // [NCDW_BB][NCDW_TLD]
    float* gh_s = nc_smem + NCDW_BB * NCDW_TLD; // [NCDW_BB][16]   (already scaled by 1/N)
    const int bid = blockIdx.x;
    const int ti = (nb.ntasks > 1 && bid >= nb.t[1].tile_base) ? 1 : 0;
    const NcDwTask& t = nb.t[ti];
    int local = bid - t.tile_base;
    {   // XCD-aware order: the tiles_k column tiles of one row tile stream the same 16 columns of U
        const int ntj = t.ntiles / t.tiles_k;
        if ((ntj & 7) == 0 && (t.tile_base & 7) == 0) {
            const int x = local & 7, y = local >> 3;
            local = ((y / t.tiles_k) * 8 + x) * t.tiles_k + (y % t.tiles_k);
        }
    }
    const int tj = local / t.tiles_k, tk = local - tj * t.tiles_k;
    const int j0 = tj * 16, k0 = tk * 32;
    const int F = t.F, H = t.H, B = t.B;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int j = j0 + i;
    const bool jok = j < H;
    const float invN = 1.0f / (float)(4 * NC_NF);
    const bool want_bias = (tk == 0);

    for (int e = tid; e < 4 * NC_NF * 32; e += 512) {
        const int n = e >> 5, c = e & 31;
        nz_s[n * NZLD + c] = t.noise[(size_t)n * F + min(k0 + c, F - 1)];      // columns past F are never stored
    }

    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    float asum = 0.f;
    const float* Uj = t.U + (jok ? j : 0);
    const unsigned Hu = (unsigned)H;
    struct Chunk { float u[4]; };
    // 16-byte table loads when the whole tile is inside the matrices and rows are 4-float aligned (uniform)
    const bool vecT = (k0 + 32 <= F) && (j0 + 16 <= H) && (((F | H | t.ld_ml | t.ldgh) & 3) == 0) &&
                      (((((uintptr_t)t.mean) | ((uintptr_t)t.sigma) | ((uintptr_t)t.GH)) & 15) == 0);

    for (int bb0 = 0; bb0 < B; bb0 += NCDW_BB) {
        const int nb_rows = min(NCDW_BB, B - bb0);
        const int Mb = 4 * NC_NF * nb_rows;                    // inner rows of this block (a multiple of 4)
        const float* Ub = Uj + (size_t)bb0 * (4 * NC_NF) * Hu;
        // BRANCH-FREE fetch: rows past the block are clamped to its last 4-row group and contribute gh = 0
        auto load = [&](int c, Chunk& q) {
            const unsigned r0 = min(16u * (unsigned)c + 4u * (unsigned)kq, (unsigned)Mb - 4u);
            const float* up = Ub + r0 * Hu;
#pragma unroll
            for (int s = 0; s < 4; ++s) q.u[s] = up[s * Hu];
        };
        Chunk S0[4], S1[4], S2[LEAN ? 1 : 4], S3[LEAN ? 1 : 4];
        auto loadg = [&](int g, Chunk (&S)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) load(w + 8 * u + 32 * g, S[u]);
        };
        // the first four (LEAN: two) groups of U go out BEFORE the tables are staged: one combined round trip instead of two
        loadg(0, S0); loadg(1, S1);
        if constexpr (!LEAN) { loadg(2, S2); loadg(3, S3); }
        __builtin_amdgcn_sched_barrier(0);
        if (bb0) __syncthreads();                              // the previous block's tables are still being read
        // ---- stage the block's tables (clamped addresses; whatever lies past F / H / B is multiplied by gh = 0) ----
#pragma unroll
        for (int e = tid; e < NCDW_BB * 16; e += 512) {        // 16 float4 per row: 8 mean + 8 sigma
            const int rb = e >> 4, q = e & 15, c4 = (q & 7) * 4;
            const int bsrc = bb0 + min(rb, nb_rows - 1);
            const float* src = (q < 8) ? t.mean + (size_t)bsrc * t.ld_ml : t.sigma + (size_t)bsrc * F;
            f32x4 v;
            if (vecT) v = *reinterpret_cast<const f32x4*>(src + k0 + c4);
            else {
#pragma unroll
                for (int s = 0; s < 4; ++s) v[s] = src[min(k0 + c4 + s, F - 1)];
            }
            *reinterpret_cast<f32x4*>(&tab_s[rb * NCDW_TLD + (q < 8 ? 0 : 32) + c4]) = v;
        }
#pragma unroll
        for (int e = tid; e < NCDW_BB * 4; e += 512) {         // 4 float4 per row of GH
            const int rb = e >> 2, c4 = (e & 3) * 4;
            const bool rok = rb < nb_rows;
            const float* src = t.GH + (size_t)(bb0 + min(rb, nb_rows - 1)) * t.ldgh;
            f32x4 v;
            if (vecT) {
                v = *reinterpret_cast<const f32x4*>(src + j0 + c4);
#pragma unroll
                for (int s = 0; s < 4; ++s) v[s] *= rok ? invN : 0.f;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) v[s] = src[min(j0 + c4 + s, H - 1)] * ((rok && j0 + c4 + s < H) ? invN : 0.f);
            }
            *reinterpret_cast<f32x4*>(&gh_s[rb * 16 + c4]) = v;
        }
        __syncthreads();

        // One chunk in two stages: prep() turns the loaded U values and the LDS tables into the 4 A values and 8 B
        // values of the chunk's MFMAs; mac() issues them.  compg() runs prep(u+1) in the shadow of mac(u): with the LDS
        // reads issued just in time, every chunk exposed 2-3 LDS latencies per 256 MFMA cycles.
        struct Prep { float a[4], x0[4], x1[4]; };
        auto prep = [&](int c, const Chunk& q, Prep& o) {
            const unsigned r0 = 16u * (unsigned)c + 4u * (unsigned)kq;
            const bool ok = r0 < (unsigned)Mb;
            const unsigned rc = ok ? r0 : (unsigned)Mb - 4u;
            const unsigned lb = rc / 20u, n0 = rc - lb * 20u;
            const float ghr = gh_s[lb * 16 + i];
            const float gh = ok ? ghr : 0.f;
            const float* tb = &tab_s[lb * NCDW_TLD + i];
            const float mu0 = tb[0], mu1 = tb[16], sg0 = tb[32], sg1 = tb[48];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float* nz = &nz_s[(n0 + s) * NZLD + i];
                o.a[s] = gh * fminf(q.u[s] + 1.f, 1.f);          // elu'(out) = min(out + 1, 1)
                o.x0[s] = fmaf(sg0, nz[0], mu0);
                o.x1[s] = fmaf(sg1, nz[16], mu1);
            }
        };
        auto mac = [&](const Prep& o) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[s], o.x0[s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[s], o.x1[s], acc[1], 0, 0, 0);
            }
            if (want_bias) asum += (o.a[0] + o.a[1]) + (o.a[2] + o.a[3]);
        };
        // wave w takes chunks w, w+8, ... in GROUPS of four (32 MFMAs); named sets, never copied
        const int nchunks = (Mb + 15) / 16;
        const int ngroups = (nchunks + 31) / 32;
        auto compg = [&](int g, const Chunk (&S)[4]) {
            const int c = w + 32 * g;
            Prep p0, p1;
            prep(c, S[0], p0);
            __builtin_amdgcn_sched_barrier(0);
            prep(c + 8, S[1], p1); mac(p0);
            __builtin_amdgcn_sched_barrier(0);
            prep(c + 16, S[2], p0); mac(p1);
            __builtin_amdgcn_sched_barrier(0);
            prep(c + 24, S[3], p1); mac(p0);
            __builtin_amdgcn_sched_barrier(0);
            mac(p1);
        };
        // Issue order is the SAME on every path that loads (the s_waitcnt immediates are static: a load that is issued
        // on one path only would force the conservative count, i.e. a full drain, on all of them).  Set k is refilled
        // with group g+4+k right after group g+k has been multiplied; only the last trip issues nothing.
        if constexpr (LEAN) {
            for (int g = 0; g < ngroups; g += 2) {
                __builtin_amdgcn_sched_barrier(0);
                if (g + 2 < ngroups) {
                    compg(g, S0);     loadg(g + 2, S0); __builtin_amdgcn_sched_barrier(0);
                    compg(g + 1, S1); loadg(g + 3, S1);          // (past the end: clamped addresses, never multiplied)
                } else {
                    compg(g, S0);
                    if (g + 1 < ngroups) compg(g + 1, S1);
                }
            }
        } else
        for (int g = 0; g < ngroups; g += 4) {
            __builtin_amdgcn_sched_barrier(0);
            if (g + 4 < ngroups) {
                compg(g, S0);     loadg(g + 4, S0); __builtin_amdgcn_sched_barrier(0);
                compg(g + 1, S1); loadg(g + 5, S1); __builtin_amdgcn_sched_barrier(0);
                compg(g + 2, S2); loadg(g + 6, S2); __builtin_amdgcn_sched_barrier(0);
                compg(g + 3, S3); loadg(g + 7, S3);
            } else {
                compg(g, S0);
                if (g + 1 < ngroups) compg(g + 1, S1);
                if (g + 2 < ngroups) compg(g + 2, S2);
                if (g + 3 < ngroups) compg(g + 3, S3);
            }
        }
    }
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[w][f][r][lane] = acc[f][r];
    if (want_bias) {
        asum += __shfl_xor(asum, 16, 64);
        asum += __shfl_xor(asum, 32, 64);
        if (lane < 16) bsum[w][lane] = asum;
    }
    __syncthreads();
    {   // 512 threads -> 16 x 32 outputs
        const int ol = threadIdx.x & 63, q = threadIdx.x >> 6;     // q: 0..7 -> (frag = q>>2, reg = q&3)
        const int f = q >> 2, r = q & 3;
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) v += red[ww][f][r][ol];
        const int row = j0 + (ol >> 4) * 4 + r, colk = k0 + 16 * f + (ol & 15);
        if (row < H && colk < F) {
            const size_t o = (size_t)row * F + colk;
            t.gW[o] = v;
        }
    }
    if (want_bias && threadIdx.x < 16 && j0 + (int)threadIdx.x < H) {
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) v += bsum[ww][threadIdx.x];
        t.gb[j0 + threadIdx.x] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// dW / db on the BF16 matrix pipe at fp32 accuracy (bf16x3), as a split-K GEMM with generated operands:
//   dW[j, k] = sum_r dPre[r, j] * x[r, k],   r = (b, n) over 20 B rows,   dPre = GH[b, j]/N * elu'(U[r, j]),   x = mean + sigma * noise.
// Output tile 64 (j) x 64 (k) per workgroup, `splits` ranges of batch rows per tile (B = 256, 8 splits: 640 rows = twenty 32-deep
// steps); 8 waves in two roles as in nc_dx_x3_kernel: waves 4-7 build and split BOTH operands of the next step -- thread (column, row
// octet) loads 8 rows of U for its j and the tables for its k, and writes one 16-byte chunk per image: the images are [column][32 r],
// i.e. already transposed for the MFMA (a lane needs 8 consecutive r of one column) -- waves 0-3 own a 32 x 32 quarter of the tile
// (four 16 x 16 accumulators, 24 MFMAs and 12 fragment reads per step).  Against the fp32 kernel above (16 x 32 tiles, every dPre
// element regenerated by 8 workgroups and every x element by 16): 4x / 4x, 100 VGPRs instead of 233, no spills.  Partial tiles and
// bias partials go to slabs; nc_dw_fin_kernel adds them in split order (no float atomics: bit-reproducible).
// ------------------------------------------------------------------------------------------------
#define NDW_IMGB (64 * NX_RSB)                 /* one image: 64 columns x 80-byte rows */
#define NDW_OPB (3 * NDW_IMGB)                 /* one operand: three images */
#define NDW_BUFB (2 * NDW_OPB)                 /* one step: A and B */

template <int OFF> __device__ __forceinline__ void ndw_fload_at(u32x4 (&d)[3], unsigned addr) {
    nx_read<OFF>(d[0], addr); nx_read<OFF + NDW_IMGB>(d[1], addr); nx_read<OFF + 2 * NDW_IMGB>(d[2], addr);
}

__global__ __launch_bounds__(512) void nc_dw_x3_kernel(NcDwBatch nb) {
    __shared__ __attribute__((aligned(16))) unsigned char L[2 * NDW_BUFB];      // 61 440 B
    const int bid = blockIdx.x;
    const int q = (nb.ntasks > 1 && bid >= nb.t[1].tile_base) ? 1 : 0;
    const NcDwTask& t = nb.t[q];
    const int F = t.F, H = t.H, B = t.B;
    constexpr int N = 4 * NC_NF;                                     // 20 noise rows (checked by the launcher): divisions by it compile to mul-shift
    const int TK = (F + 63) >> 6, TJ = (H + 63) >> 6;
    int local = bid - t.tile_base;
    {   // XCD-aware order: the TK column tiles of one (split, row tile) stream the same rows of U -- same blockIdx % 8, same L2
        const int ngrp = TJ * nb.splits;
        if ((ngrp & 7) == 0 && (t.tile_base & 7) == 0) {
            const int x = local & 7, y = local >> 3;
            local = ((y / TK) * 8 + x) * TK + (y % TK);
        }
    }
    const int sp = local / (TJ * TK); local -= sp * (TJ * TK);
    const int tj = local / TK, tk = local - tj * TK;
    const int j0 = tj * 64, k0 = tk * 64;
    const int bs = (B + nb.splits - 1) / nb.splits;                 // batch rows per split
    const int bbeg = min(sp * bs, B), nbr = min(bs, B - bbeg);
    const int nrows = N * nbr;                                       // inner rows of this split
    const int S = (nrows + 31) >> 5;
    const int w8 = threadIdx.x >> 6;
    float* const slab = nb.slab + ((size_t)(q * nb.splits + sp) * H) * F;
    if (nbr <= 0) {                                                   // a split without rows (tiny batches): its partials are zero
        for (int e = threadIdx.x; e < 64 * 64; e += 512) {
            const int jj = j0 + (e >> 6), kk = k0 + (e & 63);
            if (jj < H && kk < F) slab[(size_t)jj * F + kk] = 0.f;
        }
        if (tk == 0 && threadIdx.x < 64 && j0 + (int)threadIdx.x < H) nb.bslab[(size_t)(q * nb.splits + sp) * H + j0 + threadIdx.x] = 0.f;
        return;
    }

    if (w8 >= 4) {
        // ================= producer: thread = (column c of BOTH 64-wide operands, row octet rq) =================
        const int tid = threadIdx.x - 256;
        const int c = tid & 63, rq = tid >> 6;
        const int j = min(j0 + c, H - 1), k = min(k0 + c, F - 1);
        const float invN = 1.0f / (float)N;
        const float* const Uc = t.U + (size_t)bbeg * N * H + j;
        const float* const Gc = t.GH + (size_t)bbeg * t.ldgh + j;
        const float* const Mc = t.mean + (size_t)bbeg * t.ld_ml + k;
        const float* const Sc = t.sigma + (size_t)bbeg * F + k;
        const float* const Zc = t.noise + k;
        const int wofs = c * NX_RSB + rq * 16;
        // The octet r0 .. r0 + 7 is two groups of four rows, and a group never straddles a batch row or the end of the split (N and the
        // row count are multiples of 4): per GROUP one batch row lb[h], one validity flag -- folded into the group's GH / sigma / mean
        // values (zero them and the rows contribute nothing), so the per-element work is add, min, mul (dPre) and one fma (x).
        // Static register indices only (a runtime index would put the set in scratch).
        struct PReg { float u[8], g[2], mu[2], sg[2], nz[8]; };
        auto gload = [&](int s, PReg& r) {
            const int r0 = 32 * min(s, S - 1) + 8 * rq;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rg = min(r0 + 4 * h, nrows - 4);           // clamped group start (still a multiple of 4)
                const int lb = rg / N, n0 = rg - lb * N;
                const float* up = Uc + (size_t)rg * H;
#pragma unroll
                for (int e = 0; e < 4; ++e) { r.u[4 * h + e] = up[(size_t)e * H]; r.nz[4 * h + e] = Zc[(size_t)(n0 + e) * F]; }
                r.g[h] = Gc[(size_t)lb * t.ldgh]; r.mu[h] = Mc[(size_t)lb * t.ld_ml]; r.sg[h] = Sc[(size_t)lb * F];
            }
        };
        float bsum = 0.f;
        auto produce = [&](int s, const PReg& r, unsigned char* buf) {
            const int r0 = 32 * s + 8 * rq;
            float a[8], x[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bool valid = r0 + 4 * h < nrows;
                const float g = valid ? r.g[h] * invN : 0.f, mu = valid ? r.mu[h] : 0.f, sg = valid ? r.sg[h] : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[4 * h + e] = g * fminf(r.u[4 * h + e] + 1.f, 1.f);            // elu'(out) = min(out + 1, 1)
                    x[4 * h + e] = fmaf(sg, r.nz[4 * h + e], mu);
                    bsum += a[4 * h + e];
                }
            }
            u32x4 ah, am, al, xh, xm, xl;
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) {
                unsigned h, m, l;
                x3_split2(a[2 * p2], a[2 * p2 + 1], h, m, l); ah[p2] = h; am[p2] = m; al[p2] = l;
                x3_split2(x[2 * p2], x[2 * p2 + 1], h, m, l); xh[p2] = h; xm[p2] = m; xl[p2] = l;
            }
            unsigned char* pa = buf + wofs;
            *reinterpret_cast<u32x4*>(pa) = ah;
            *reinterpret_cast<u32x4*>(pa + NDW_IMGB) = am;
            *reinterpret_cast<u32x4*>(pa + 2 * NDW_IMGB) = al;
            *reinterpret_cast<u32x4*>(pa + NDW_OPB) = xh;
            *reinterpret_cast<u32x4*>(pa + NDW_OPB + NDW_IMGB) = xm;
            *reinterpret_cast<u32x4*>(pa + NDW_OPB + 2 * NDW_IMGB) = xl;
        };
        PReg ra, rb;
        gload(0, ra); gload(1, rb);
        produce(0, ra, L);
        gload(2, ra);
        __syncthreads();
        for (int s = 0; s < S; s += 2) {
            if (s + 1 < S) produce(s + 1, rb, L + NDW_BUFB);
            gload(s + 3, rb);
            __syncthreads();
            if (s + 1 < S) {
                if (s + 2 < S) produce(s + 2, ra, L);
                gload(s + 4, ra);
                __syncthreads();
            }
        }
        // bias partial of this split: sum over the four row octets of a column (fixed order), column tile 0 only
        float* const red = reinterpret_cast<float*>(L);              // the image buffers are free: every consumer is past its last barrier
        __syncthreads();
        red[rq * 64 + c] = bsum;
        __syncthreads();
        if (tk == 0 && rq == 0 && j0 + c < H)
            nb.bslab[(size_t)(q * nb.splits + sp) * H + j0 + c] = ((red[c] + red[64 + c]) + red[128 + c]) + red[192 + c];
        return;
    }

    // ================= consumer: wave (wj, wk) owns rows j0 + 32 wj .. + 31, columns k0 + 32 wk .. + 31 of the tile =================
    const int lane = threadIdx.x & 63, w = w8;
    const int m16 = lane & 15, kq = lane >> 4;
    const int wj = w >> 1, wk = w & 1;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)L + (unsigned)(m16 * NX_RSB + kq * 16);
    const unsigned aofs = (unsigned)(32 * wj * NX_RSB), bofs = (unsigned)(NDW_OPB + 32 * wk * NX_RSB);
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) acc[a][b2] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int s = 0; s < S; ++s) {
        const unsigned base = lds0 + (unsigned)((s & 1) * NDW_BUFB);
        u32x4 fa[2][3], fb[2][3];
        ndw_fload_at<0>(fa[0], base + aofs);
        ndw_fload_at<0>(fb[0], base + bofs);
        ndw_fload_at<16 * NX_RSB>(fa[1], base + aofs);
        ndw_fload_at<16 * NX_RSB>(fb[1], base + bofs);
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2]));
#define NDW_MM(A, Bq)                                                                                                   \
        {                                                                                                               \
            const bf16x8 Ah = __builtin_bit_cast(bf16x8, fa[A][0]), Am = __builtin_bit_cast(bf16x8, fa[A][1]), Al = __builtin_bit_cast(bf16x8, fa[A][2]); \
            const bf16x8 Bh = __builtin_bit_cast(bf16x8, fb[Bq][0]), Bm = __builtin_bit_cast(bf16x8, fb[Bq][1]), Bl = __builtin_bit_cast(bf16x8, fb[Bq][2]); \
            f32x4 d = acc[A][Bq];                                                                                       \
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bh, d, 0, 0, 0);                                            \
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bl, d, 0, 0, 0);                                            \
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bm, d, 0, 0, 0);                                            \
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bh, d, 0, 0, 0);                                            \
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bm, d, 0, 0, 0);                                            \
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bh, d, 0, 0, 0);                                            \
            acc[A][Bq] = d;                                                                                             \
        }
        NDW_MM(0, 0)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[1][2]));
        NDW_MM(0, 1) NDW_MM(1, 0) NDW_MM(1, 1)
#undef NDW_MM
        __syncthreads();
    }
    __syncthreads();                                                 // (the producers' bias reduction: two more barriers for every wave)
    __syncthreads();
    // partial tile -> slab: C/D map col = lane & 15 (k), row = 4 (lane >> 4) + reg (j)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
            const int kk = k0 + 32 * wk + 16 * b2 + m16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jj = j0 + 32 * wj + 16 * a + 4 * kq + r;
                if (jj < H && kk < F) slab[(size_t)jj * F + kk] = acc[a][b2][r];
            }
        }
}

// sums the split partials in split order: gW = sum_sp slab[sp], gb = sum_sp bslab[sp]
__global__ __launch_bounds__(256) void nc_dw_fin_kernel(NcDwBatch nb) {
    const int q = blockIdx.y;
    const NcDwTask& t = nb.t[q];
    const size_t HF = (size_t)t.H * t.F;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const float* s0 = nb.slab + (size_t)q * nb.splits * HF;
    if (e < HF) {
        float v = s0[e];
        for (int sp = 1; sp < nb.splits; ++sp) v += s0[(size_t)sp * HF + e];
        t.gW[e] = v;
    }
    if (e < (size_t)t.H) {
        const float* b0 = nb.bslab + (size_t)q * nb.splits * t.H;
        float v = b0[e];
        for (int sp = 1; sp < nb.splits; ++sp) v += b0[(size_t)sp * t.H + e];
        t.gb[e] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// hidden units per nc_fwd workgroup (64 or 128); the builder sizes tiles_h with it
extern "C" int rl_nc_fwd_cols() { return 128; }
// Tiling of one nc_fwd launch: engine (0 fp32 MFMA, 1 bf16x3), batch-row groups per workgroup (g2, 4 rows each) and hidden
// units per workgroup.  bf16x3 needs 32-deep K steps and 16-byte rows everywhere; RLREP_DISABLE=x3 keeps fp32 (tests: both engines).
extern "C" void rl_nc_fwd_plan(const NcFwdTask* tasks, int ntasks, int* engine, int* g2, int* cols) {
    const int B = tasks[0].B, F = tasks[0].F, H = tasks[0].H;
    const bool want_x3 = !rl_off("x3");             // read per plan (agent construction), so a test can flip it
    bool x3 = want_x3 && (F % 32) == 0 && F >= 64;
    for (int q = 0; q < ntasks; ++q) {
        const NcFwdTask& t = tasks[q];
        x3 = x3 && t.N == 4 * NC_NF && (t.ld_ml & 3) == 0 && t.F == F && t.H == H && t.B == B &&
             ((((uintptr_t)t.mean) | ((uintptr_t)t.lstd) | ((uintptr_t)t.noise) | ((uintptr_t)t.W) | ((uintptr_t)t.sigma_out)) & 15) == 0;
    }
    if (x3) {
        *engine = 1; *g2 = 2;
        const long long wg128 = (long long)ntasks * ((B + 7) / 8) * ((H + 127) / 128);
        *cols = wg128 >= 256 ? 128 : 64;
        return;
    }
    *engine = 0;
    *cols = rl_nc_fwd_cols();
    *g2 = ((long long)ntasks * ((B + 3) / 4) * ((H + 63) / 64) <= 2048) ? 1 : 2;
}
extern "C" int rl_launch_nc_fwd(const NcFwdBatch* nb, int total_tiles, int g2, hipStream_t st) {
    if (total_tiles <= 0) return 0;
    for (int q = 0; q < nb->ntasks; ++q) if (nb->t[q].N != 4 * NC_NF) return -2;      // the row mapping is built for N = 20
    const int F = nb->t[0].F, N = nb->t[0].N;
    if (nb->engine == 1) {
        if ((F % 32) != 0 || g2 != 2) return -3;
#ifdef RL_EXPERIMENTS
        // (the superseded two-role / 16x16x32 forms of the 128-wide tile: experiments library only)
        const int wide = rl_off("nc_x3_wide") ? 0 : 1;
        const int quad = rl_off("nc_x3_q") ? 0 : 1;
        if (nb->cols == 128 && quad) hipLaunchKernelGGL(nc_fwd_x3q_kernel, dim3(total_tiles), dim3(256), 2 * NX_BUFB, st, *nb);
        else if (nb->cols == 128 && wide) hipLaunchKernelGGL((nc_fwd_x3w_kernel<8>), dim3(total_tiles), dim3(512), 2 * NX_BUFB, st, *nb);     // superseded by x3q
        else if (nb->cols == 128) hipLaunchKernelGGL((nc_fwd_x3_kernel<2>), dim3(total_tiles), dim3(512), 2 * NX_BUFB, st, *nb);             // superseded by x3q
#else
        if (nb->cols == 128) hipLaunchKernelGGL(nc_fwd_x3q_kernel, dim3(total_tiles), dim3(256), 2 * NX_BUFB, st, *nb);
#endif
        else hipLaunchKernelGGL((nc_fwd_x3_kernel<1>), dim3(total_tiles), dim3(512), 2 * NX_BUFB, st, *nb);
        return (int)hipGetLastError();
    }
    const int Fp = (F + 15) & ~15;
    const size_t lds = (size_t)(8 * g2 + N) * (Fp + 16) * sizeof(float);
    const int nw = nb->cols / 16;
    if (nw == 8) {
        if (g2 == 1) hipLaunchKernelGGL((nc_fwd_kernel<1, 8>), dim3(total_tiles), dim3(512), lds, st, *nb);
        else if (g2 == 2) hipLaunchKernelGGL((nc_fwd_kernel<2, 8>), dim3(total_tiles), dim3(512), lds, st, *nb);
        else hipLaunchKernelGGL((nc_fwd_kernel<4, 8>), dim3(total_tiles), dim3(512), lds, st, *nb);
    } else {
        if (g2 == 1) hipLaunchKernelGGL((nc_fwd_kernel<1, 4>), dim3(total_tiles), dim3(256), lds, st, *nb);
        else if (g2 == 2) hipLaunchKernelGGL((nc_fwd_kernel<2, 4>), dim3(total_tiles), dim3(256), lds, st, *nb);
        else hipLaunchKernelGGL((nc_fwd_kernel<4, 4>), dim3(total_tiles), dim3(256), lds, st, *nb);
    }
    return (int)hipGetLastError();
}

// engine of the dX launch: 1 = bf16x3 (H % 32 == 0, even row strides, 8-byte aligned U / GH; RLREP_DISABLE=x3 keeps fp32 MFMA)
extern "C" int rl_nc_dx_engine(const NcDxTask* t) {
    if (rl_off("x3")) return 0;
    if (t->N != 4 * NC_NF || (t->H % 32) != 0 || (t->ldgh & 1) || t->nheads < 1 || t->nheads > 2 || t->tiles_k != (t->F + 63) / 64) return 0;
    for (int h = 0; h < t->nheads; ++h) if (((((uintptr_t)t->U[h]) | ((uintptr_t)t->GH[h])) & 7) != 0) return 0;
    return 1;
}
extern "C" int rl_launch_nc_dx(const NcDxTask* t, hipStream_t st) {
    if (t->ntiles <= 0) return 0;
    if (t->N != 4 * NC_NF) return -2;
    // 16-byte loads of U / GH rows need 4-float-aligned rows in every head
    bool vec = ((t->H & 3) == 0) && ((t->ldgh & 3) == 0);
    for (int h = 0; h < t->nheads; ++h) vec = vec && ((((uintptr_t)t->U[h]) | ((uintptr_t)t->GH[h])) & 15) == 0;
    if (rl_nc_dx_engine(t) == 1) {
        hipLaunchKernelGGL(nc_dx_x3_kernel, dim3(t->ntiles), dim3(512), 0, st, *t);
        return (int)hipGetLastError();
    }
    if (vec) hipLaunchKernelGGL(nc_dx_kernel<true>, dim3(t->ntiles), dim3(512), 0, st, *t);
    else hipLaunchKernelGGL(nc_dx_kernel<false>, dim3(t->ntiles), dim3(512), 0, st, *t);
    return (int)hipGetLastError();
}

// one-time setup (agent creation, never inside a stream capture): nc_dw needs more dynamic LDS than the 64 KB default
extern "C" int rl_nc_init() {
    const size_t lds = (size_t)NCDW_BB * (NCDW_TLD + 16) * sizeof(float);
    hipError_t e = hipSuccess;
#ifdef RL_EXPERIMENTS
    e = hipFuncSetAttribute((const void*)nc_fwd_x3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NX_BUFB);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)nc_fwd_x3w_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NX_BUFB);
#endif
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)nc_fwd_x3q_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NX_BUFB);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)nc_fwd_x3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NX_BUFB);
    if (e != hipSuccess) return (int)e;
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)nc_dw_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    return (int)hipFuncSetAttribute((const void*)nc_dw_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

// engine of the dW launch: 1 = bf16x3 split-K (needs slabs from the caller; RLREP_DISABLE=x3 keeps fp32)
extern "C" int rl_nc_dw_engine() { return rl_off("x3") ? 0 : 1; }
extern "C" int rl_nc_dw_splits(int B, int F, int H, int ntasks) {
    const int tiles = ((H + 63) / 64) * ((F + 63) / 64) * ntasks;
    int sp = (256 + tiles - 1) / tiles;                               // about one workgroup per CU
    sp = sp < 1 ? 1 : sp > 16 ? 16 : sp;
    const int maxsp = (B + 7) / 8;                                    // at least 8 batch rows (five 32-deep steps) per split
    return sp > maxsp ? (maxsp < 1 ? 1 : maxsp) : sp;
}
extern "C" int rl_launch_nc_dw(const NcDwBatch* nb, int total_tiles, hipStream_t st) {
    if (total_tiles <= 0) return 0;
    for (int q = 0; q < nb->ntasks; ++q) if (nb->t[q].N != 4 * NC_NF || nb->t[q].B <= 0) return -2;
    if (nb->engine == 1) {
        if (!nb->slab || !nb->bslab || nb->splits < 1) return -3;
        const NcDwTask& t0 = nb->t[0];
        for (int q = 1; q < nb->ntasks; ++q) if (nb->t[q].H != t0.H || nb->t[q].F != t0.F || nb->t[q].B != t0.B) return -3;
        hipLaunchKernelGGL(nc_dw_x3_kernel, dim3(total_tiles), dim3(512), 0, st, *nb);
        if (nb->fin_in_adam) return (int)hipGetLastError();      // the critic group's optimizer launch adds the slabs (AdamTask::Slab)
        const size_t HF = (size_t)t0.H * t0.F;
        hipLaunchKernelGGL(nc_dw_fin_kernel, dim3((unsigned)((HF + 255) / 256), nb->ntasks), dim3(256), 0, st, *nb);
        ++g_rl_launches;              // this stage is two kernels
        return (int)hipGetLastError();
    }
    const size_t lds = (size_t)NCDW_BB * (NCDW_TLD + 16) * sizeof(float);          // 96 KB of staged per-batch-row tables
    if (nb->lean) hipLaunchKernelGGL(nc_dw_kernel<true>, dim3(total_tiles), dim3(512), lds, st, *nb);
    else hipLaunchKernelGGL(nc_dw_kernel<false>, dim3(total_tiles), dim3(512), lds, st, *nb);
    return (int)hipGetLastError();
}

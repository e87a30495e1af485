// gemm_x3q_kernel: the bf16x3 tile for products with FEW ROWS and a LONG inner dimension -- ctrlsac's M = 256 layers at main.py's dimensions
// (256 x {1024, 2048} x {1024, 2048}: agent/ctrlsac/ctrlsac_agent.py:54-102, 213-251), diffsrsac's at HalfCheetah dims.  (Included by gemm_lds.hip.)
//
// On the 64 x 64 tile (gemm_x3s_kernel) such a product has 64 - 128 output tiles for 256 CUs, so it was cut along K into slabs: 256 - 512
// workgroups of ONE wave per SIMD each, whose slice is the serial split -> barrier -> 12 MFMAs -> barrier of a lone workgroup, plus a finishing launch
// that adds the slabs (19 - 32 us per layer; the finishers alone were 13 % of ctrlsac's GPU time: profiles/r05_ctrlsac_*_kernel_stats.csv;
// VERDICT r05 item 2).  Here the output tile is 32 x 32 -- 256 / 512 workgroups WITHOUT slabs -- and the K split moves INSIDE the workgroup:
//
//   * 4 waves = 4 quarters of K.  A wave is a workgroup of its own for the whole main loop: it stages ITS 32 x 32 slices of A and B (64 lanes x
//     4 16-byte loads per operand), splits them three ways and passes them through a PRIVATE 12 KB patch of LDS into MFMA fragments -- no
//     __syncthreads in the loop, so the four waves of a workgroup (one per SIMD) and the two or three workgroups of a CU drift apart and cover
//     each other's load / split / multiply phases;
//   * LDS is only the transposition buffer: a slice's twelve fragments are read into registers at once, the next slice is split into the same
//     patch while the MFMAs of this one run (same-wave LDS operations execute in order); loads run two slices ahead in two register sets
//     (three sets, loads three slices ahead, were measured: 12.6 us against 10.7 for 256 x 1024 x 1024 -- 216 VGPRs, a ninth slice of zeros per
//     quarter of eight -- and ctrlsac 850 against 873 train()/s: docs/history/r06.md);
//   * two accumulators (k-block 0 / k-block 1 of every slice) halve the dependent-MFMA chain;
//   * at the end the four partial tiles meet in LDS, every wave adds its 8 rows of them IN QUARTER ORDER and runs the epilogue (gl_epilogue4: the
//     forward / dX epilogues of the engine) -- deterministic, no slab, no finishing launch.
//
// A is row-major [R, K]; B is row-major [Cn, K] (forward) or k-major [K, Cn] (dX).  Needs K % 16 == 0, 16-byte-regular operands, and Cn % 8 == 0 for
// the k-major B (rl_gemm_lds_route sends everything else to the 64-wide tile as before).
#pragma once

#define X3Q_IMG (32 * X3_RSB)              /* one bf16 image of a 32-row slice: 32 rows x 64 bytes (x3r_off swizzle) */
#define X3Q_WAVE (6 * X3Q_IMG)             /* per wave: three images of A, three of B */

// 16 consecutive k of one row (lane = (row, k half)): raw loads (PHASE 0) / zero fill past the wave's K range (PHASE 1), as in x3s_load_row
template <int PHASE>
__device__ __forceinline__ void x3q_load_row(const float* __restrict__ p, int k0, int kend, int Klim, int kh, f32x4 (&e)[4]) {
    const int k = k0 + 16 * kh;
    if constexpr (PHASE == 0) {
        const float* q = p + min(k, Klim - 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = *reinterpret_cast<const f32x4*>(q + 4 * j);
    } else {
        const bool ok = k < kend;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = ok ? e[j] : zero;
    }
}
// k-major operand: two consecutive k (lane = (k pair, group of 8 columns)): e[0..1] = row k, e[2..3] = row k + 1
template <int PHASE>
__device__ __forceinline__ void x3q_load_col(const float* __restrict__ p, int ld, int k0, int kend, int Klim, int kp, f32x4 (&e)[4]) {
    const int k = k0 + 2 * kp;
    if constexpr (PHASE == 0) {
        const float* q = p + (size_t)min(k, Klim - 2) * ld;
        e[0] = *reinterpret_cast<const f32x4*>(q); e[1] = *reinterpret_cast<const f32x4*>(q + 4);
        e[2] = *reinterpret_cast<const f32x4*>(q + ld); e[3] = *reinterpret_cast<const f32x4*>(q + ld + 4);
    } else {
        const bool ok = k < kend;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = ok ? e[j] : zero;
    }
}
// split 16 consecutive k of row `row` into the three images at img (chunks 2 kh, 2 kh + 1 of the row: two 16-byte writes per image)
__device__ __forceinline__ void x3q_write_row(unsigned char* __restrict__ img, int row, int kh, const f32x4 (&e)[4]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        u32x4 hi, mid, lo;
        unsigned h, m, l;
        x3_split2(e[2 * j][0], e[2 * j][1], h, m, l); hi[0] = h; mid[0] = m; lo[0] = l;
        x3_split2(e[2 * j][2], e[2 * j][3], h, m, l); hi[1] = h; mid[1] = m; lo[1] = l;
        x3_split2(e[2 * j + 1][0], e[2 * j + 1][1], h, m, l); hi[2] = h; mid[2] = m; lo[2] = l;
        x3_split2(e[2 * j + 1][2], e[2 * j + 1][3], h, m, l); hi[3] = h; mid[3] = m; lo[3] = l;
        unsigned char* p = img + x3r_off(row, 2 * kh + j);
        *reinterpret_cast<u32x4*>(p) = hi;
        *reinterpret_cast<u32x4*>(p + X3Q_IMG) = mid;
        *reinterpret_cast<u32x4*>(p + 2 * X3Q_IMG) = lo;
    }
}
// k-major: (k, k + 1) of eight columns -> one packed pair per column, transposed into the row-major image (image row = column)
__device__ __forceinline__ void x3q_write_col(unsigned char* __restrict__ img, int kp, int cg, const f32x4 (&e)[4]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        unsigned h, m, l;
        x3_split2(e[j >> 2][j & 3], e[2 + (j >> 2)][j & 3], h, m, l);
        unsigned char* p = img + x3r_off(8 * cg + j, kp >> 2) + 4 * (kp & 3);
        *reinterpret_cast<unsigned*>(p) = h;
        *reinterpret_cast<unsigned*>(p + X3Q_IMG) = m;
        *reinterpret_cast<unsigned*>(p + 2 * X3Q_IMG) = l;
    }
}

template <int LB>
__global__ __launch_bounds__(256, 2) void gemm_x3q_kernel(GL_DIR_PARAMS, GemmBatch gb) {
    const int gdir[GEMM_MAX_TASKS] = {d0, d1, d2, d3, d4, d5, d6, d7};
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * X3Q_WAVE];          // 48 KB: a 12 KB patch per wave; the epilogue's [4][32][36] floats afterwards
    const int bid = blockIdx.x;
    int ti = 0;
#pragma unroll
    for (int q = 1; q < GEMM_MAX_TASKS; ++q) if (bid >= gdir[q]) ti = q;
    const GemmTask& t = gb.t[ti];
    const float* const pA = t.A; const float* const pB = t.B;
    const int lda = t.lda, ldb = t.ldb, R = t.R, Cn = t.Cn, K = t.K;
    const int tiles_r = (R + 31) >> 5;
    const int local = gl_xcd_remap(bid - t.tile_base, t.ntiles);
    const int tc = local / tiles_r, tr = local - tc * tiles_r;         // (neighbours share the 32 weight rows / columns of B)
    const int r0 = tr * 32, c0 = tc * 32;

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int kq = (((K + 3) >> 2) + 31) & ~31;                      // this wave's quarter of K, whole 32-deep slices
    const int kbeg = min(w * kq, K), kend = min(K, kbeg + kq);
    const int nk = (kend - kbeg + 31) >> 5;
    unsigned char* const Lw = lds + w * X3Q_WAVE;

    // staging roles
    const int srow = lane >> 1, skh = lane & 1;                      // row-major: (row, k half)
    const int skp = lane >> 2, scg = lane & 3;                       // k-major: (k pair, column group)
    const float* const pa = pA + (size_t)min(r0 + srow, R - 1) * lda;
    const float* const pb = LB == LD_ROW ? pB + (size_t)min(c0 + srow, Cn - 1) * ldb : pB + min(c0 + 8 * scg, Cn - 8);
    // fragment roles (32x32x16: lane = (row | column r32, k half hh))
    const int r32 = lane & 31, hh = lane >> 5;
    const int fo = x3r_off(r32, hh), fsw = x3r_off(r32, 2 + hh) - x3r_off(r32, hh);

    f32x16 acc0, acc1;
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc0[q] = 0.f; acc1[q] = 0.f; }
    f32x4 ea[2][4], eb[2][4];
#define X3Q_LOAD(Z, KS)                                                                                                       \
    {                                                                                                                         \
        const int kz_ = kbeg + 32 * (KS);                                                                                     \
        x3q_load_row<0>(pa, kz_, kend, K, skh, ea[Z]);                                                                        \
        if constexpr (LB == LD_ROW) x3q_load_row<0>(pb, kz_, kend, K, skh, eb[Z]); else x3q_load_col<0>(pb, ldb, kz_, kend, K, skp, eb[Z]); \
    }
#define X3Q_SPLIT(Z, KS)                                                                                                      \
    {                                                                                                                         \
        const int kz_ = kbeg + 32 * (KS);                                                                                     \
        x3q_load_row<1>(pa, kz_, kend, K, skh, ea[Z]);                                                                        \
        if constexpr (LB == LD_ROW) x3q_load_row<1>(pb, kz_, kend, K, skh, eb[Z]); else x3q_load_col<1>(pb, ldb, kz_, kend, K, skp, eb[Z]); \
        x3q_write_row(Lw, srow, skh, ea[Z]);                                                                                  \
        if constexpr (LB == LD_ROW) x3q_write_row(Lw + 3 * X3Q_IMG, srow, skh, eb[Z]); else x3q_write_col(Lw + 3 * X3Q_IMG, skp, scg, eb[Z]); \
    }
    // one slice: its twelve fragments out of the patch, the loads of slice KT + 2 into the set it frees, the MFMAs, and -- under them -- the split of
    // slice KT + 1 into the same patch (the fragment reads above have been issued: same-wave LDS operations execute in order)
#define X3Q_ITER(Z, KT)                                                                                                       \
    {                                                                                                                         \
        bf16x8 a[2][3], b[2][3];                                                                                              \
        _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                                                       \
            _Pragma("unroll") for (int m = 0; m < 3; ++m) {                                                                   \
                a[c][m] = *reinterpret_cast<const bf16x8*>(Lw + m * X3Q_IMG + fo + fsw * c);                                  \
                b[c][m] = *reinterpret_cast<const bf16x8*>(Lw + (3 + m) * X3Q_IMG + fo + fsw * c);                            \
            }                                                                                                                 \
        }                                                                                                                     \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                                                                \
        X3Q_LOAD(Z, (KT) + 2)                                                                                                 \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][0], b[0][2], acc0, 0, 0, 0);                                      \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][0], b[1][2], acc1, 0, 0, 0);                                      \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][2], b[0][0], acc0, 0, 0, 0);                                      \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][2], b[1][0], acc1, 0, 0, 0);                                      \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][1], b[0][1], acc0, 0, 0, 0);                                      \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][1], b[1][1], acc1, 0, 0, 0);                                      \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][0], b[0][1], acc0, 0, 0, 0);                                      \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][0], b[1][1], acc1, 0, 0, 0);                                      \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][1], b[0][0], acc0, 0, 0, 0);                                      \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][1], b[1][0], acc1, 0, 0, 0);                                      \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][0], b[0][0], acc0, 0, 0, 0);                                      \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][0], b[1][0], acc1, 0, 0, 0);                                      \
        X3Q_SPLIT((Z) ^ 1, (KT) + 1)                                                                                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                                                                \
    }
    if (nk > 0) {
        X3Q_LOAD(0, 0) X3Q_LOAD(1, 1)
        X3Q_SPLIT(0, 0)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // (pairs of slices: one basic block per pair, static register sets; a slice past the end of an odd quarter multiplies the zeros of its own fill)
        for (int kt = 0; kt < nk; kt += 2) {
            X3Q_ITER(0, kt)
            X3Q_ITER(1, kt + 1)
        }
    }
#undef X3Q_ITER
#undef X3Q_SPLIT
#undef X3Q_LOAD
    __syncthreads();                                                 // every wave is done with its patch: the partial tiles meet in LDS

    // accumulator (32x32 C/D map: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) -> this wave's [32][36] patch
    float* const E = reinterpret_cast<float*>(lds) + w * (32 * 36);
#pragma unroll
    for (int q = 0; q < 16; ++q) E[((q & 3) + 8 * (q >> 2) + 4 * hh) * 36 + r32] = acc0[q] + acc1[q];
    __syncthreads();
    // wave w finishes rows 8 w .. 8 w + 7 of the tile: the four quarters in order, then the epilogue
    const int rr = 8 * w + (lane >> 3), cc = (lane & 7) * 4;
    const float* const E0 = reinterpret_cast<const float*>(lds) + rr * 36 + cc;
    f32x4 v = *reinterpret_cast<const f32x4*>(E0);
#pragma unroll
    for (int p = 1; p < 4; ++p) v += *reinterpret_cast<const f32x4*>(E0 + p * (32 * 36));
    const int r = r0 + rr, c = c0 + cc;
    if (r < R && c < Cn) gl_epilogue4(t, r, c, v);
}

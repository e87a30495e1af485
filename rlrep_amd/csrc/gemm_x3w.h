// ================================================================================================
// gemm_x3w_kernel: bf16x3 on a 256 x 128 tile, one PERSISTENT workgroup per CU -- the 200-GFLOP products of diffsrsac's nabla-mu head
// (forward 2048 x 96 256 x 512, dX 2048 x 512 x 96 256, dW 96 256 x 512 x 2048), included by gemm_lds.hip after the 128 x 128 kernels.
//
// What the 128 x 128 kernels (gemm_x3_kernel / gemm_x3t_kernel) leave on the table at these sizes: a wave owns 32 x 64 of the tile, so every
// 12 MFMAs cost 9 fragment reads and the split of 8 staged elements per thread; one LDS buffer, two barriers per 32-deep block, so the eight
// waves of a workgroup are all in their VALU phase or all in their matrix phase and only the co-resident workgroup fills the other pipe; the
// global loads of block i + 1 are issued one matrix phase before they are needed (an HBM miss is longer); the prologue and the 64 KB epilogue
// of a tile overlap with nothing of the same workgroup.  143-180 TF of 417.
//
// This kernel: eight waves as 4 (rows) x 2 (columns), a wave owns 64 x 64 = four 32x32 accumulators (6 fragment reads and 6 staged elements
// per 12 MFMAs); TWO LDS stages of 72 KB (three bf16 images of 256 rows of A and of 128 rows of B, 32 deep) and ONE barrier per block: block
// i + 1 is split and written while block i is multiplied, and the two waves of a SIMD (w and w + 4) run the two halves of an iteration in
// OPPOSITE order -- one multiplies while the other splits; the workgroup walks its tiles (index = blockIdx + k * gridDim) as ONE stream of
// (tile, block) items, so the loads of the next tile's first blocks are in flight during the last blocks of this one and the accumulator
// stores of a tile drain under the next tile's MFMAs.  The MFMA takes the B fragment as its first operand: a lane then holds FOUR CONSECUTIVE
// COLUMNS of one output row per register quad, and the tile is stored from the accumulators with 16-byte stores -- no LDS patch (there is no
// LDS left for one: 2 x 72 KB + the 16 KB of the bias-gradient sums = 160 KB).
// Image layouts are those of the 128 x 128 kernels: row-major operands [row][64 B] with swizzled 16-byte chunks (x3r_off, ds_read_b128),
// k-major operands [32 k][128 rows] per 128-row half (x3t_off, ds_read_b64_tr_b16).
// ================================================================================================
#ifndef X3W_ABL
#define X3W_ABL 0          /* timing-only ablations: 1 no global loads, 2 no split / LDS writes, 4 no MFMAs, 8 no fragment reads */
#endif
#define X3W_BM 256
#define X3W_BN 128
#define X3W_AIMG (X3W_BM * 64)                       /* bytes per A image */
#define X3W_BIMG (X3W_BN * 64)                       /* bytes per B image */
#define X3W_STAGE (3 * X3W_AIMG + 3 * X3W_BIMG)      /* 73 728 */
#define X3W_LDSB (2 * X3W_STAGE + 16384)             /* + [256 rows][16 k slots] floats of the bias-gradient sums */

// 512 threads load a [64 NJ rows] x 32 slice: NJ 16-byte loads per thread
template <int LD, int NJ>
__device__ __forceinline__ void x3w_load(const float* __restrict__ P, int ld, int base, int lim, int k0, int kend, f32x4 (&e)[NJ]) {
    if (LD == LD_ROW) {
        const int kc = (int)(threadIdx.x & 7) * 4;
        const int k = min(k0 + kc, kend - 4);
        const bool ok = (k0 + kc) < kend;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int r = min(base + (int)(threadIdx.x >> 3) + 64 * j, lim - 1);
            const f32x4 x = *reinterpret_cast<const f32x4*>(P + (size_t)r * ld + k);
            e[j] = ok ? x : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    } else {
        const int c4 = (int)(threadIdx.x & 31) * 4;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = base + 128 * (j >> 1) + c4, kk = (int)(threadIdx.x >> 5) + 16 * (j & 1);
            const int i = min(col, lim - 4), k = min(k0 + kk, kend - 1);
            const f32x4 x = *reinterpret_cast<const f32x4*>(P + (size_t)k * ld + i);
            e[j] = (col < lim && (k0 + kk) < kend) ? x : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
}
// split and write them: img = LDS byte address of the operand's first image, IMG = bytes per image
template <int LD, int NJ, int IMG>
__device__ __forceinline__ void x3w_write(unsigned img, const f32x4 (&e)[NJ]) {
    typedef __attribute__((address_space(3))) u32x2* lp;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        unsigned p;
        if (LD == LD_ROW) {
            const int row = (int)(threadIdx.x >> 3) + 64 * j, kc = (int)(threadIdx.x & 7) * 4;
            p = img + (unsigned)x3r_off(row, kc >> 3) + 8u * ((kc >> 2) & 1);
        } else {
            const int c4 = (int)(threadIdx.x & 31) * 4, kk = (int)(threadIdx.x >> 5) + 16 * (j & 1);
            p = img + 8192u * (j >> 1) + x3t_off(kk, c4 >> 3) + 8u * ((c4 >> 2) & 1);
        }
        u32x2 hi, mid, lo;
        unsigned h, m, l;
        x3_split2(e[j][0], e[j][1], h, m, l); hi[0] = h; mid[0] = m; lo[0] = l;
        x3_split2(e[j][2], e[j][3], h, m, l); hi[1] = h; mid[1] = m; lo[1] = l;
        *(lp)(uintptr_t)p = hi;
        *(lp)(uintptr_t)(p + IMG) = mid;
        *(lp)(uintptr_t)(p + 2 * IMG) = lo;
    }
}
template <int OFF> __device__ __forceinline__ bf16x8 x3w_rd128(unsigned a) {
    return *(__attribute__((address_space(3))) bf16x8*)(uintptr_t)(a + OFF);
}

struct X3wTile { int ti, r0, c0, kbeg, kend, nk, split, tc; };

template <int LA, int LB>
__global__ __launch_bounds__(512, 2) void gemm_x3w_kernel(GL_DIR_PARAMS, GemmBatch gb, int total_tiles) {
    const int gdir[GEMM_MAX_TASKS] = {d0, d1, d2, d3, d4, d5, d6, d7};
    __shared__ __attribute__((aligned(16))) float lds[X3W_LDSB / 4];
    const unsigned Lb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)reinterpret_cast<unsigned char*>(lds);

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int r32 = lane & 31, hh = lane >> 5, g1 = (lane >> 4) & 1;

    // tile -> task, split, row / column tile.  Each XCD class (index % 8) walks a contiguous run of a task's tiles (gl_xcd_remap); inside the run the
    // SHORTER tile dimension is the fast one, so that the 32 tiles an XCD works on together share the panels of the larger operand in its L2
    auto tile_of = [&](int T) __attribute__((always_inline)) {
        X3wTile x;
        int ti = 0;
#pragma unroll
        for (int q = 1; q < GEMM_MAX_TASKS; ++q) if (T >= gdir[q]) ti = q;
        const GemmTask& t = gb.t[ti];
        const int tiles_c = t.tiles_c, tiles_r = (t.R + X3W_BM - 1) / X3W_BM;
        const int local = gl_xcd_remap(T - t.tile_base, t.ntiles);
        const int per_split = tiles_r * tiles_c;
        const int split = local / per_split, rem = local - split * per_split;
        int tr, tc;
        if (tiles_r >= tiles_c) { tr = rem / tiles_c; tc = rem - tr * tiles_c; }
        else { tc = rem / tiles_r; tr = rem - tc * tiles_r; }
        x.ti = ti; x.r0 = tr * X3W_BM; x.c0 = tc * X3W_BN; x.tc = tc; x.split = split;
        x.kbeg = split * t.kchunk; x.kend = min(t.K, x.kbeg + t.kchunk);
        x.nk = (x.kend - x.kbeg + GL_BK - 1) / GL_BK;
        return x;
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][y][q] = 0.f;
    f32x4 rs[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, rs_done[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // (k-major A: bias-gradient partial sums)

    // fragment addresses inside a stage (the stage offset is added per iteration)
    // row-major: chunk 2 c + hh of this lane's row; the swizzle term (row >> 2) & 3 is the same for rows r32, 32 + r32, ...
    const unsigned faR = (unsigned)x3r_off(wr * 64 + r32, hh), fbR = 3 * X3W_AIMG + (unsigned)x3r_off(wc * 64 + r32, hh);
    const int fsw = x3r_off(r32, 2 + hh) - x3r_off(r32, hh);
    // k-major: transposed-read addresses of k-blocks 8 hh and 8 hh + 4 (x3t_addr), A: 128-row half wr >> 1, rows 64 (wr & 1) + 32 i + 16 g1
    // (one address pair per 32-row block: the chunk index is XOR-swizzled, so "+ 32 rows" is not a constant byte offset)
    unsigned aA[2][2], aB[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            aA[i][h] = x3t_addr(8192u * (wr >> 1), 8 * hh + 4 * h, 8 * (wr & 1) + 4 * i + 2 * g1);
            aB[i][h] = x3t_addr(3 * X3W_AIMG, 8 * hh + 4 * h, 8 * wc + 4 * i + 2 * g1);
        }

    f32x4 ea[4], eb[2];
    bool e_valid = false, e_first = false;

    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
    X3wTile cur = tile_of(tile);
    int ckt = 0;
    // the loader's position in the stream
    int ltile = tile, lkt = 0;
    X3wTile lt = cur;
    auto issue_loads = [&]() __attribute__((always_inline)) {
        e_valid = ltile < total_tiles;
        if (!e_valid) return;
        const GemmTask& t = gb.t[lt.ti];
        const int k0 = lt.kbeg + GL_BK * lkt;
        if (!(X3W_ABL & 1) || (ltile == (int)blockIdx.x && lkt == 0)) {
            x3w_load<LA, 4>(t.A, t.lda, lt.r0, t.R, k0, lt.kend, ea);
            x3w_load<LB, 2>(t.B, t.ldb, lt.c0, t.Cn, k0, lt.kend, eb);
        }
        e_first = lkt == 0;
        if (++lkt == lt.nk) { lkt = 0; ltile += gridDim.x; if (ltile < total_tiles) lt = tile_of(ltile); }
    };
    auto split_write = [&](unsigned stage) __attribute__((always_inline)) {
        if (!e_valid) return;
        if constexpr (LA == LD_COL) {
            if (e_first) { rs_done[0] = rs[0]; rs_done[1] = rs[1]; rs[0] = rs[1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            rs[0] += ea[0] + ea[1]; rs[1] += ea[2] + ea[3];
        }
        if (!(X3W_ABL & 2)) {
            x3w_write<LA, 4, X3W_AIMG>(Lb + stage, ea);
            x3w_write<LB, 2, X3W_BIMG>(Lb + stage + 3 * X3W_AIMG, eb);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(ea[j]));
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" :: "v"(eb[j]));
        }
    };

#define X3W_MMA(C)                                                                                                            \
    {                                                                                                                         \
        bf16x8 a[2][3], b[2][3];                                                                                              \
        if constexpr (LA == LD_ROW) {                                                                                         \
            const unsigned p = Lb + stage + faR + (unsigned)(fsw * (C));                                                      \
            a[0][0] = x3w_rd128<0>(p); a[0][1] = x3w_rd128<X3W_AIMG>(p); a[0][2] = x3w_rd128<2 * X3W_AIMG>(p);                 \
            a[1][0] = x3w_rd128<32 * 64>(p); a[1][1] = x3w_rd128<32 * 64 + X3W_AIMG>(p); a[1][2] = x3w_rd128<32 * 64 + 2 * X3W_AIMG>(p); \
        } else {                                                                                                              \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                   \
                const unsigned p0 = Lb + stage + aA[i][0], p1 = Lb + stage + aA[i][1];                                        \
                a[i][0] = x3t_frag<(C) * 4096>(p0, p1); a[i][1] = x3t_frag<(C) * 4096 + X3W_AIMG>(p0, p1);                     \
                a[i][2] = x3t_frag<(C) * 4096 + 2 * X3W_AIMG>(p0, p1);                                                        \
            }                                                                                                                 \
        }                                                                                                                     \
        if constexpr (LB == LD_ROW) {                                                                                         \
            const unsigned p = Lb + stage + fbR + (unsigned)(fsw * (C));                                                      \
            b[0][0] = x3w_rd128<0>(p); b[0][1] = x3w_rd128<X3W_BIMG>(p); b[0][2] = x3w_rd128<2 * X3W_BIMG>(p);                 \
            b[1][0] = x3w_rd128<32 * 64>(p); b[1][1] = x3w_rd128<32 * 64 + X3W_BIMG>(p); b[1][2] = x3w_rd128<32 * 64 + 2 * X3W_BIMG>(p); \
        } else {                                                                                                              \
            _Pragma("unroll") for (int y = 0; y < 2; ++y) {                                                                   \
                const unsigned p0 = Lb + stage + aB[y][0], p1 = Lb + stage + aB[y][1];                                        \
                b[y][0] = x3t_frag<(C) * 4096>(p0, p1); b[y][1] = x3t_frag<(C) * 4096 + X3W_BIMG>(p0, p1);                     \
                b[y][2] = x3t_frag<(C) * 4096 + 2 * X3W_BIMG>(p0, p1);                                                        \
            }                                                                                                                 \
        }                                                                                                                     \
        if (X3W_ABL & 4) {                                                                                                    \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int m = 0; m < 3; ++m) { asm volatile("" :: "v"(a[i][m])); asm volatile("" :: "v"(b[i][m])); } \
        } else                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                         \
        _Pragma("unroll") for (int y = 0; y < 2; ++y) {                                                                       \
            f32x16 v = acc[i][y];                                                                                             \
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[y][2], a[i][0], v, 0, 0, 0);                                        \
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[y][0], a[i][2], v, 0, 0, 0);                                        \
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[y][1], a[i][1], v, 0, 0, 0);                                        \
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[y][1], a[i][0], v, 0, 0, 0);                                        \
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[y][0], a[i][1], v, 0, 0, 0);                                        \
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[y][0], a[i][0], v, 0, 0, 0);                                        \
            acc[i][y] = v;                                                                                                    \
        }                                                                                                                     \
    }

    // the tile's accumulators -> memory (lane: output row r32 of its row block; register quad g: columns 8 g + 4 hh .. + 3 of its column block)
    auto store_tile = [&](const f32x4 (&rsum)[2]) __attribute__((always_inline)) {
        const GemmTask& t = gb.t[cur.ti];
        const int R = t.R, Cn = t.Cn, splits = t.splits;
        if constexpr (LA == LD_COL) {
            if (t.epi == EPI_DW && (t.flags & FLAG_BIASGRAD) && cur.tc == 0) {
                // bias gradient = row sums of operand A: this thread holds rows 128 s + 4 (tid % 32) .. + 3 of its k slot -> LDS -> fixed-order sum over the 16 slots
                float* part = lds + 2 * X3W_STAGE / 4;                  // [256 rows][16 k slots]
                const int c4 = (int)(threadIdx.x & 31) * 4, ks = (int)(threadIdx.x >> 5);
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int q = 0; q < 4; ++q) part[(128 * s + c4 + q) * 16 + ks] = rsum[s][q];
                __syncthreads();
                if (threadIdx.x < 256) {
                    const float* q = part + threadIdx.x * 16;
                    float s0 = 0.f;
#pragma unroll
                    for (int z = 0; z < 16; ++z) s0 += q[z];
                    const int r = cur.r0 + (int)threadIdx.x;
                    if (r < R) { if (splits > 1) t.bslab[(size_t)cur.split * R + r] = s0; else t.out2[r] = s0; }
                }
                __syncthreads();
            }
        }
        f32x4 bias[2][4];
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int g = 0; g < 4; ++g) bias[y][g] = splits > 1 ? (f32x4){0.f, 0.f, 0.f, 0.f} : gl_bias4(t, cur.c0 + wc * 64 + 32 * y + 8 * g + 4 * hh);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = cur.r0 + wr * 64 + 32 * i + r32;
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = cur.c0 + wc * 64 + 32 * y + 8 * g + 4 * hh;
                    const f32x4 v = {acc[i][y][4 * g], acc[i][y][4 * g + 1], acc[i][y][4 * g + 2], acc[i][y][4 * g + 3]};
                    if (r < R && c < Cn) {
                        if (splits > 1) st4(t.slab + ((size_t)cur.split * R + r) * ((Cn + 3) & ~3) + c, v);
                        else gl_epilogue4(t, r, c, v, &bias[y][g]);
                    }
                }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][y][q] = 0.f;
    };

    // prologue: item 0 into stage 0, item 1 into the staging registers
    issue_loads();
    split_write(0);
    issue_loads();
    unsigned stage = 0;
    for (;;) {
        __syncthreads();
        const bool last = ckt + 1 == cur.nk;
        if (w & 4) {
            const bool handed = e_valid && e_first;          // (the next tile's first block is split now: this tile's sums move to rs_done)
            split_write(X3W_STAGE - stage);
            issue_loads();
            X3W_MMA(0) X3W_MMA(1)
            if (last) { if (handed) store_tile(rs_done); else store_tile(rs); }
        } else {
            X3W_MMA(0) X3W_MMA(1)
            if (last) store_tile(rs);
            split_write(X3W_STAGE - stage);
            issue_loads();
        }
        if (last) {
            tile += gridDim.x;
            if (tile >= total_tiles) break;
            cur = tile_of(tile); ckt = 0;
        } else ++ckt;
        stage = X3W_STAGE - stage;
    }
#undef X3W_MMA
}

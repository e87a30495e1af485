// ================================================================================================
// gemm_x3w_kernel: bf16x3 on a 256 x 128 tile, one PERSISTENT workgroup per CU -- the 200-GFLOP products of diffsrsac's nabla-mu head
// (forward 2048 x 96 256 x 512, dX 2048 x 512 x 96 256, dW 96 256 x 512 x 2048), included by gemm_lds.hip after the 128 x 128 kernels.
//
// What the 128 x 128 kernels (gemm_x3_kernel / gemm_x3t_kernel) leave on the table at these sizes: a wave owns 32 x 64 of the tile, so every
// 12 MFMAs cost 9 fragment reads and the split of 8 staged elements per thread; one LDS buffer, two barriers per 32-deep block, so the eight
// waves of a workgroup are all in their VALU phase or all in their matrix phase and only the co-resident workgroup fills the other pipe; the
// global loads of block i + 1 are issued one matrix phase before they are needed (an HBM miss is longer); the prologue and the 64 KB epilogue
// of a tile overlap with nothing of the same workgroup.  130-190 TF of the 305-340 the matrix pipe sustains (tools/exp/mfma_bf16.hip).
//
// This kernel: eight waves as 4 (rows) x 2 (columns), a wave owns 64 x 64 = four 32x32 accumulators (6 fragment reads and 6 staged elements
// per 12 MFMAs); TWO LDS stages of 72 KB (three bf16 images of 256 rows of A and of 128 rows of B, 32 deep) and ONE barrier per block.  An
// iteration multiplies block i out of one stage while it splits block i + 1 (in the staging registers since the previous iteration) into the
// other and refills those registers with block i + 2 -- written SLOT BY SLOT (48 MFMAs, each with its share of the rest behind it: see
// `iteration` below), because no scheduler directive kept that order for all three operand layouts.  The workgroup walks its tiles
// (index = blockIdx + k * gridDim) as ONE stream of (tile, block) items, so the loads of the next tile's first blocks are in flight during
// the last blocks of this one; the tile is stored from the accumulators, a wave-instruction = two whole 128-byte lines (lane = column) --
// there is no LDS left for a patch: 2 x 72 KB + the 16 KB of the bias-gradient sums = 160 KB.  Epilogues: forward / dX with no activation,
// ReLU or ELU, dW (+ bias gradient), split-K slabs.  207 / 219 / 236 VGPRs (row x row / row x k-major / k-major x k-major), two waves per SIMD.
// Measured (tools/exp/x3w_check.py, x3w_time.py): the nabla-mu head forward / dX / dW 1 029 / 955 / 997 us (196 / 211 / 202 TF) against 1 535 /
// 1 105 / 1 149 on the 128 x 128 tiles; 4096^3 615-646 us (213-223 TF) on random operands, 439 us (313 TF) on zeros: the matrix pipe's power.
// Image layouts are those of the 128 x 128 kernels: row-major operands [row][64 B] with swizzled 16-byte chunks (x3r_off, ds_read_b128),
// k-major operands [32 k][128 rows] per 128-row half (x3t_off, ds_read_b64_tr_b16).
// ================================================================================================
#ifndef X3W_ABL
#define X3W_ABL 0          /* timing-only ablations: 1 no global loads, 2 no split / LDS writes, 4 no MFMAs */
#endif
#include <utility>
#include <type_traits>
#define X3W_BM 256
#define X3W_BN 128
#define X3W_AIMG (X3W_BM * 64)                       /* bytes per A image */
#define X3W_BIMG (X3W_BN * 64)                       /* bytes per B image */
#define X3W_STAGE (3 * X3W_AIMG + 3 * X3W_BIMG)      /* 73 728 */

// LDS accesses go through pointers into the kernel's __shared__ arrays (address space 3), byte offsets as immediates
typedef __attribute__((address_space(3))) unsigned char* x3w_lds;
template <int OFF> __device__ __forceinline__ bf16x8 x3w_rd128(x3w_lds a) {
    return *(__attribute__((address_space(3))) bf16x8*)(a + OFF);
}
// k-major fragment: two transposed reads (k-blocks h = 0, 1 at a0 / a1; x3t_frag with pointers)
template <int OFF> __device__ __forceinline__ bf16x8 x3w_tr(x3w_lds a0, x3w_lds a1) {
    typedef __attribute__((address_space(3))) x3t_s16x4* tp;
    const x3t_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tp)(a0 + OFF)), hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tp)(a1 + OFF));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
// which tasks the tile's epilogue covers (the host routes the others to the 128 x 128 tile)
static inline bool x3w_epilogue_ok(const GemmTask* t) {
    return (t->epi == EPI_FWD || t->epi == EPI_DX || t->epi == EPI_DW) && (t->act == ACT_NONE || t->act == ACT_RELU || t->act == ACT_ELU);
}

template <class F, int... I> __device__ __forceinline__ void x3w_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void x3w_for(F&& f) { x3w_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

struct X3wTile { int ti, r0, c0, kbeg, kend, nk, split, tc; };

template <int LA, int LB>
__global__ __launch_bounds__(512, 2) void gemm_x3w_kernel(GL_DIR_PARAMS, GemmBatch gb, int total_tiles) {
    const int gdir[GEMM_MAX_TASKS] = {d0, d1, d2, d3, d4, d5, d6, d7};
    __shared__ __attribute__((aligned(16))) unsigned char stages[2 * X3W_STAGE];
    __shared__ __attribute__((aligned(16))) float bias_part[4096];          // [256 rows][16 k slots]
    const x3w_lds S0 = (x3w_lds)stages;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
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
    f32x4 rs[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // (k-major A: bias-gradient partial sums of the tile whose blocks are being split)

    // fragment addresses inside stage 0
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
    // staging-write addresses (image 0 of stage 0; slot j by constant offset: 4096 j in either layout)
    const unsigned wA = (LA == LD_ROW ? (unsigned)x3r_off(tid >> 3, (tid & 7) >> 1) + 8u * (tid & 1)
                                           : x3t_off(tid >> 5, (tid & 31) >> 1) + 8u * (tid & 1));
    const unsigned wB = 3 * X3W_AIMG + (LB == LD_ROW ? (unsigned)x3r_off(tid >> 3, (tid & 7) >> 1) + 8u * (tid & 1)
                                                          : x3t_off(tid >> 5, (tid & 31) >> 1) + 8u * (tid & 1));

    // ---- the loader: position (lt, lkt) in the stream of (tile, block) items, two items ahead of the multiplier -------------------------
    // per-thread base pointers of the loader's tile: row-major operand: the thread's rows 64 j + tid / 8 (k offset added per block); k-major: its columns
    // 128 s + 4 (tid % 32) (k row added per block)
    const float* pa[4]; const float* pb[2];
    int l_lda = 0, l_ldb = 0, l_kend = 0;
    int ltile = blockIdx.x, lkt = 0;
    if (ltile >= total_tiles) return;
    X3wTile lt = tile_of(ltile);
    auto loader_setup = [&]() __attribute__((always_inline)) {
        const GemmTask& t = gb.t[lt.ti];
        l_lda = t.lda; l_ldb = t.ldb; l_kend = lt.kend;
        if (LA == LD_ROW) {
#pragma unroll
            for (int j = 0; j < 4; ++j) pa[j] = t.A + (size_t)min(lt.r0 + (tid >> 3) + 64 * j, t.R - 1) * t.lda;
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s) pa[s] = t.A + min(lt.r0 + 128 * s + (tid & 31) * 4, t.R - 4);
            pa[2] = pa[3] = nullptr;
        }
        if (LB == LD_ROW) {
#pragma unroll
            for (int j = 0; j < 2; ++j) pb[j] = t.B + (size_t)min(lt.c0 + (tid >> 3) + 64 * j, t.Cn - 1) * t.ldb;
        } else { pb[0] = t.B + min(lt.c0 + (tid & 31) * 4, t.Cn - 4); pb[1] = nullptr; }
    };
    loader_setup();
    // The loader's item: per-thread element offsets (no bounds selects: the K tail is zeroed when the block is split, and rows / columns beyond the edge
    // are clamped duplicates whose outputs are never stored).  The six 16-byte loads themselves are issued one by one (load_piece), each right after
    // the split of the piece whose registers it refills: ONE set of staging registers, and every load still has a whole iteration to land.
    bool l_tail = false, l_dup = false;        // (the item being loaded ends inside its 32-deep block; the loader has run out of items and repeats the last)
    int l_k0 = 0;
    size_t oA0 = 0, oA1 = 0, oB0 = 0, oB1 = 0;
    auto loader_offsets = [&]() __attribute__((always_inline)) {
        const int k0 = lt.kbeg + GL_BK * lkt;
        l_k0 = k0; l_tail = k0 + GL_BK > l_kend;
        const int kr = min(k0 + (tid & 7) * 4, l_kend - 4), k_lo = min(k0 + (tid >> 5), l_kend - 1), k_hi = min(k0 + (tid >> 5) + 16, l_kend - 1);
        if (LA == LD_ROW) oA0 = (size_t)kr; else { oA0 = (size_t)k_lo * l_lda; oA1 = (size_t)k_hi * l_lda; }
        if (LB == LD_ROW) oB0 = (size_t)kr; else { oB0 = (size_t)k_lo * l_ldb; oB1 = (size_t)k_hi * l_ldb; }
    };
    auto load_piece = [&](int j, f32x4& e) __attribute__((always_inline)) {          // j = 0..3: A slot j; 4, 5: B slot j - 4
        if (X3W_ABL & 1) return;
        if (j < 4) e = LA == LD_ROW ? ld4(pa[j] + oA0) : ld4(pa[j >> 1] + ((j & 1) ? oA1 : oA0));
        else e = LB == LD_ROW ? ld4(pb[j - 4] + oB0) : ld4(pb[0] + ((j & 1) ? oB1 : oB0));
    };
    // step the loader; at the end of the stream it stays on the last item (the duplicates it loads are split into a stage nobody reads)
    auto loader_step = [&]() __attribute__((always_inline)) {
        if (++lkt == lt.nk) {
            if (ltile + (int)gridDim.x < total_tiles) { ltile += gridDim.x; lt = tile_of(ltile); lkt = 0; loader_setup(); }
            else { lkt = lt.nk - 1; l_dup = true; }
        }
    };
    // zero the K tail of a loaded block (rare: only a block that ends inside its 32 k)
    auto zero_tail = [&](f32x4 (&ea)[4], f32x4 (&eb)[2], int k0, int kend) __attribute__((always_inline)) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (LA == LD_ROW) { if (k0 + (tid & 7) * 4 >= kend) { ea[0] = z; ea[1] = z; ea[2] = z; ea[3] = z; } }
        else { if (k0 + (tid >> 5) >= kend) { ea[0] = z; ea[2] = z; } if (k0 + (tid >> 5) + 16 >= kend) { ea[1] = z; ea[3] = z; } }
        if (LB == LD_ROW) { if (k0 + (tid & 7) * 4 >= kend) { eb[0] = z; eb[1] = z; } }
        else { if (k0 + (tid >> 5) >= kend) eb[0] = z; if (k0 + (tid >> 5) + 16 >= kend) eb[1] = z; }
    };

    // ---- the multiplier's position ---------------------------------------------------------------------------------------------------------
    int tile = blockIdx.x, ckt = 0;
    X3wTile cur = lt;

    // the tile's accumulators -> memory (32x32 C/D map: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)): a wave-instruction stores
    // 32 consecutive columns of two rows
    auto store_tile = [&]() __attribute__((always_inline)) {
        const GemmTask& t = gb.t[cur.ti];
        const int R = t.R, Cn = t.Cn, splits = t.splits;
        if constexpr (LA == LD_COL) {
            if (t.epi == EPI_DW && (t.flags & FLAG_BIASGRAD) && cur.tc == 0) {
                // bias gradient = row sums of operand A: this thread holds rows 128 s + 4 (tid % 32) .. + 3 of its k slot -> LDS -> fixed-order sum over the 16 slots
                // (the partial sums were put into bias_part when the first block of the NEXT tile came up for splitting: hand_over below)
                float* part = bias_part;
                __syncthreads();
                if (tid < 256) {
                    const float* q = part + tid * 16;
                    float s0 = 0.f;
#pragma unroll
                    for (int z = 0; z < 16; ++z) s0 += q[z];
                    const int r = cur.r0 + tid;
                    if (r < R) { if (splits > 1) t.bslab[(size_t)cur.split * R + r] = s0; else t.out2[r] = s0; }
                }
                __syncthreads();
            }
        }
        // The case analysis is done ONCE per tile, outside the element loops: inside them every store was followed by uniform branches and, where a value
        // loaded before the loop (the bias) was first used, an `s_waitcnt vmcnt(0)` -- which also waits for every store issued so far: 20 us per tile
        const int epi = t.epi, act = t.act;
        const bool accum = (t.flags & FLAG_ACCUM) != 0;
        const float scale = t.scale;
        const int ldo = splits > 1 ? ((Cn + 3) & ~3) : t.ldc;
        float* const obase = splits > 1 ? t.slab + (size_t)cur.split * R * ldo : t.C;
        auto each = [&](auto fn) __attribute__((always_inline)) {
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const int c = cur.c0 + wc * 64 + 32 * y + r32;
                if (c < Cn) {
                    float* const col = obase + c;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const int r = cur.r0 + wr * 64 + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * hh;
                            if (r < R) fn(y, r, c, col + (size_t)r * ldo, acc[i][y][q]);
                        }
                }
            }
        };
        if (splits > 1) each([&](int, int, int, float* cp, float v) __attribute__((always_inline)) { *cp = v; });
        else if (epi == EPI_FWD) {
            float bias[2] = {0.f, 0.f};
            if (t.bias) {
#pragma unroll
                for (int y = 0; y < 2; ++y) { const int c = cur.c0 + wc * 64 + 32 * y + r32; bias[y] = t.bias[min(c, Cn - 1)]; }
            }
            asm volatile("" :: "v"(bias[0]), "v"(bias[1]));             // (the loads have landed before the first store is issued)
            if (act == ACT_NONE) each([&](int y, int, int, float* cp, float v) __attribute__((always_inline)) { *cp = v * scale + bias[y]; });
            else if (act == ACT_RELU) each([&](int y, int, int, float* cp, float v) __attribute__((always_inline)) { *cp = fmaxf(v * scale + bias[y], 0.f); });
            else each([&](int y, int, int, float* cp, float v) __attribute__((always_inline)) { *cp = elu_f(v * scale + bias[y]); });
        } else if (!accum && !(epi == EPI_DX && (act != ACT_NONE || t.r1u))) {
            each([&](int, int, int, float* cp, float v) __attribute__((always_inline)) { *cp = v * scale; });
        } else {
            const bool dx = epi == EPI_DX;
            float r1v[2] = {0.f, 0.f};
            if (dx && t.r1u) {
#pragma unroll
                for (int y = 0; y < 2; ++y) { const int c = cur.c0 + wc * 64 + 32 * y + r32; r1v[y] = t.r1v[min(c, Cn - 1)]; }
            }
            each([&](int y, int r, int c, float* cp, float v) __attribute__((always_inline)) {
                v *= scale;
                if (dx) {
                    if (t.r1u) v += t.r1u[r] * r1v[y];
                    if (act != ACT_NONE) {
                        const float a = t.aux[(size_t)r * t.ldaux + c];
                        v = act == ACT_RELU ? (a > 0.f ? v : 0.f) : v * elu_grad_from_out(a);
                    }
                }
                if (accum) v += *cp;
                *cp = v;
            });
        }
        // The stores must have left before the loop goes on: a store's source registers count as busy until its vmcnt retires, the compiler cannot tell
        // an iteration that follows a tile's end from one that does not, and so it put an `s_waitcnt vmcnt(0)` in front of the first instruction of
        // EVERY iteration that overwrites such a register -- a wait for the six loads issued in the previous iteration, the youngest 0.4 us earlier:
        // the iteration took 2.5 us instead of 1.7.  Waited for here, once per tile (gfx9 encoding: vmcnt 0, expcnt 7, lgkmcnt 15).
        __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][y][q] = 0.f;
    };

    // ---- one iteration, slot by slot ---------------------------------------------------------------------------------------------------------
    // 48 slots = the 48 MFMAs of a 32-deep block (16-deep half c, row block i, column block y, six products), in this order, each followed by what
    // rides in its shadow, and a sched_barrier(0) so that it stays there:
    //   * slots 0-11: one fragment read of the SECOND half (the first half's twelve are issued before slot 0);
    //   * slots 0-41: one seventh of the split of one 16-byte piece of the next block (piece = slot / 7: hi pair; residuals of elements 0,1; of 2,3; mid
    //     pair + the LDS writes of hi and mid; second residuals x 2; lo pair + its LDS write + the global load that refills the piece's registers).
    // Every piece's load is thus issued 42 slots before its first use and waited for alone (vmcnt(5)); nothing is left to the scheduler, whose
    // sched_group_barrier pipelines held for one operand layout and fell apart for the next, or pulled the first instructions of several pieces --
    // and with them a wait for ALL loads -- to the top of the iteration (docs/history/r05.md).
    bf16x8 fa[2][2][3], fb[2][2][3];                        // [half c][32-row block][image]
    unsigned sh0 = 0, sh1 = 0, sm0 = 0, sm1 = 0;             // (the piece being split: packed hi / mid pairs, residuals)
    float sr0 = 0.f, sr1 = 0.f, sr2 = 0.f, sr3 = 0.f;
    auto frag_read = [&](auto C_, auto IDX_, x3w_lds ST) __attribute__((always_inline)) {
        constexpr int C = decltype(C_)::value, IDX = decltype(IDX_)::value, blk = (IDX % 6) / 3, img = IDX % 3;      // 0..5: A; 6..11: B
        if constexpr (IDX < 6) {
            if constexpr (LA == LD_ROW) fa[C][blk][img] = x3w_rd128<2048 * blk + img * X3W_AIMG>(ST + (faR + (unsigned)(fsw * C)));
            else fa[C][blk][img] = x3w_tr<C * 4096 + img * X3W_AIMG>(ST + aA[blk][0], ST + aA[blk][1]);
        } else {
            if constexpr (LB == LD_ROW) fb[C][blk][img] = x3w_rd128<2048 * blk + img * X3W_BIMG>(ST + (fbR + (unsigned)(fsw * C)));
            else fb[C][blk][img] = x3w_tr<C * 4096 + img * X3W_BIMG>(ST + aB[blk][0], ST + aB[blk][1]);
        }
    };
    auto mma = [&](auto N_) __attribute__((always_inline)) {
        constexpr int n = decltype(N_)::value, c = n / 24, i = ((n % 24) / 6) >> 1, y = ((n % 24) / 6) & 1, pr = n % 6;
        constexpr int IA[6] = {0, 2, 1, 0, 1, 0}, IB[6] = {2, 0, 1, 1, 0, 0};            // smallest partial products first
        if (!(X3W_ABL & 4)) acc[i][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][i][IA[pr]], fb[c][y][IB[pr]], acc[i][y], 0, 0, 0);
        else { asm volatile("" :: "v"(fa[c][i][IA[pr]]), "v"(fb[c][y][IB[pr]])); }
    };
    auto split_step = [&](auto N_, x3w_lds WR, f32x4 (&ea)[4], f32x4 (&eb)[2], bool reload) __attribute__((always_inline)) {
        constexpr int n = decltype(N_)::value, pc = n / 7, st = n % 7;
        if constexpr (n < 42) {
            typedef __attribute__((address_space(3))) u32x2* lp;
            f32x4& E = [&]() -> f32x4& { if constexpr (pc < 4) return ea[pc]; else return eb[pc - 4]; }();
            constexpr int IMG = pc < 4 ? X3W_AIMG : X3W_BIMG, OFF = 4096 * (pc < 4 ? pc : pc - 4);
            const x3w_lds P = WR + (pc < 4 ? wA : wB);
            if (X3W_ABL & 2) { if constexpr (st == 6) { asm volatile("" :: "v"(E)); if (reload) load_piece(pc, E); } return; }
            if constexpr (st == 0) {
                if constexpr (LA == LD_COL && pc < 4) rs[pc >> 1] += E;
                sh0 = x3_pk(E[0], E[1]); sh1 = x3_pk(E[2], E[3]);
            } else if constexpr (st == 1) {
                sr0 = E[0] - __builtin_bit_cast(float, sh0 << 16); sr1 = E[1] - __builtin_bit_cast(float, sh0 & 0xffff0000u);
            } else if constexpr (st == 2) {
                sr2 = E[2] - __builtin_bit_cast(float, sh1 << 16); sr3 = E[3] - __builtin_bit_cast(float, sh1 & 0xffff0000u);
            } else if constexpr (st == 3) {
                sm0 = x3_pk(sr0, sr1); sm1 = x3_pk(sr2, sr3);
                *(lp)(P + OFF) = (u32x2){sh0, sh1};
                *(lp)(P + OFF + IMG) = (u32x2){sm0, sm1};
            } else if constexpr (st == 4) {
                sr0 -= __builtin_bit_cast(float, sm0 << 16); sr1 -= __builtin_bit_cast(float, sm0 & 0xffff0000u);
            } else if constexpr (st == 5) {
                sr2 -= __builtin_bit_cast(float, sm1 << 16); sr3 -= __builtin_bit_cast(float, sm1 & 0xffff0000u);
            } else {
                *(lp)(P + OFF + 2 * IMG) = (u32x2){x3_pk(sr0, sr1), x3_pk(sr2, sr3)};
                if (reload) load_piece(pc, E);
            }
        }
    };
    auto iteration = [&](x3w_lds RD, x3w_lds WR, f32x4 (&ea)[4], f32x4 (&eb)[2]) __attribute__((always_inline)) {
        loader_offsets();
        x3w_for<12>([&](auto I) __attribute__((always_inline)) { frag_read(std::integral_constant<int, 0>{}, I, RD); });
        __builtin_amdgcn_sched_barrier(0);
        x3w_for<48>([&](auto N) __attribute__((always_inline)) {
            mma(N);
            if constexpr (decltype(N)::value < 12) frag_read(std::integral_constant<int, 1>{}, N, RD);
            split_step(N, WR, ea, eb, true);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // (the prologue's split of item 0: all seven steps of all six pieces, no MFMAs)
    auto split_all = [&](x3w_lds WR, f32x4 (&ea)[4], f32x4 (&eb)[2]) __attribute__((always_inline)) {
        x3w_for<42>([&](auto N) __attribute__((always_inline)) { split_step(N, WR, ea, eb, true); });
    };

    // One iteration: multiply the block in the stage RD, split the block in (ea, eb) into the stage WR and refill (ea, eb) with the loader's block.
    // first / tail / k0 / kend of the block in (ea, eb) were noted when its loads were issued (F_*), those of the block being loaded go to (N_*).
    auto hand_over = [&]() __attribute__((always_inline)) {
        // (ea, eb) hold the first block of the next tile: the bias-gradient sums of the tile being finished are complete -> LDS (store_tile reduces them)
        const int c4 = (tid & 31) * 4, ks = tid >> 5;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q) bias_part[(128 * s + c4 + q) * 16 + ks] = rs[s][q];
        rs[0] = rs[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
#define X3W_ITER(RD, WR, F, N)                                                                                                \
    {                                                                                                                         \
        __syncthreads();                                                                                                      \
        if (__builtin_expect(F##_tail, 0)) { asm volatile("" ::: "memory"); zero_tail(ea, eb, F##_k0, F##_kend); }   /* (kept a BRANCH: as selects it waits for every load) */ \
        if constexpr (LA == LD_COL) { if (F##_first) hand_over(); }                                                           \
        iteration(RD, WR, ea, eb);                                                                                            \
        N##_tail = l_tail; N##_k0 = l_k0; N##_kend = l_kend; N##_first = l_dup || lkt == 0;                                   \
        loader_step();                                                                                                        \
        if (ckt + 1 == cur.nk) {                                                                                              \
            store_tile();                                                                                                     \
            tile += gridDim.x;                                                                                                \
            if (tile >= total_tiles) break;                                                                                   \
            cur = tile_of(tile); ckt = 0;                                                                                     \
        } else ++ckt;                                                                                                         \
    }

    // prologue: item 0 -> stage 0, item 1 -> (ea, eb)
    f32x4 ea[4], eb[2];
    bool p_tail = false, p_first = false;
    int p_k0 = 0, p_kend = 0;
    loader_offsets();
#pragma unroll
    for (int j = 0; j < 4; ++j) load_piece(j, ea[j]);
#pragma unroll
    for (int j = 0; j < 2; ++j) load_piece(4 + j, eb[j]);
    if (l_tail) zero_tail(ea, eb, l_k0, l_kend);
    loader_step();
    loader_offsets();
    p_tail = l_tail; p_k0 = l_k0; p_kend = l_kend; p_first = l_dup || lkt == 0;
    split_all(S0, ea, eb);
    loader_step();
    // (ONE copy of the iteration, the stage offsets toggling: with the order of every LDS access fixed by hand the compiler needs no proof that the
    // two stages do not alias; two unrolled copies made it alternate the staging registers between two physical sets in the k-major form, with
    // conservative vmcnt waits at the top of each copy)
    unsigned rd = 0;
    for (;;) {
        X3W_ITER(S0 + rd, S0 + (X3W_STAGE - rd), p, p)
        rd = X3W_STAGE - rd;
    }
#undef X3W_ITER
}

// heads_vae_tile: the two Gaussian heads of the vlsac feature step (encoder, f: [B, K] x [2F, K]^T each) AND vae_mid on ONE 16-row x
// 16-feature-column tile: the workgroup computes the four 16 x 16 products it needs -- encoder mean / log_std, f mean / log_std of those
// columns -- on v_mfma_f32_16x16x4_f32 (exact fp32), wave w taking the 16-deep chunks w, w + 4, ... of the inner dimension; the four partial
// tiles meet in LDS, and thread (row, column) then does what vae_mid_kernel does for its element (vlsac_agent.py:135-150).
// Callers: heads_vae_kernel (elementwise.hip, one tile per workgroup) and xchain_kernel (xchain.hip, COH = true: the trunk outputs were
// written by other workgroups of the same launch and are read with sc1 loads).
#pragma once
#include "common.h"
#include "kparams.h"
#include "gemm16_tile.h"
#include <type_traits>

template <bool COH, class ParT = HeadsVae>
__device__ __forceinline__ void heads_vae_tile(const ParT& p, const int tr, const int tc, const float* __restrict__ eps,
                                               float (&red)[4][4][256], float (&sh)[4]) {
    const int F = p.F, K = p.K, B = p.B;
    const int tile = tr * p.tiles_c + tc;
    const int r0 = tr * 16, c0 = tc * 16;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int arow = min(r0 + i, B - 1), wcol = min(c0 + i, F - 1);
    const float* pa0[2] = {p.Ae, p.Af};
    const size_t aoff = (size_t)arow * p.lda;
    const float* pw[2][2] = {{p.We + (size_t)wcol * K, p.We + (size_t)(F + wcol) * K}, {p.Wf + (size_t)wcol * K, p.Wf + (size_t)(F + wcol) * K}};
    const bool vec = !(K & 3) && !(p.lda & 3) && !((((uintptr_t)p.Ae) | ((uintptr_t)p.Af) | ((uintptr_t)p.We) | ((uintptr_t)p.Wf)) & 15);
    f32x4 acc[2][2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[n][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // The access width is decided ONCE (two copies of the loop), never per load: a branch around a load makes hipcc drain vmcnt at every
    // merge point, i.e. one exposed round trip per load instead of one per group.  Inside a copy every load is unconditional: clamped
    // address + select.  Four 16-deep chunks (K = 256: all of this wave's share) are loaded before the first of them is multiplied.
    auto body = [&](auto vec_tag) {
        constexpr bool VEC = decltype(vec_tag)::value;
        auto ldx = [&](auto coh_tag, const float* base, size_t off, int k, float (&v)[4]) {
            constexpr bool CH = decltype(coh_tag)::value;
            if (VEC) {
                const f32x4 x = rl_ld4<CH>(base, off + min(k, K - 4));
#pragma unroll
                for (int s = 0; s < 4; ++s) v[s] = (k + s < K) ? x[s] : 0.f;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) { const float x = rl_ld<CH>(base, off + min(k + s, K - 1)); v[s] = (k + s < K) ? x : 0.f; }
            }
        };
        for (int kb0 = 16 * w; kb0 < K; kb0 += 256) {
            float a[4][2][4], b[4][2][2][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = kb0 + 64 * c + 4 * kq;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    ldx(std::integral_constant<bool, COH>(), pa0[n], aoff, k, a[c][n]);
                    ldx(std::false_type(), pw[n][0], 0, k, b[c][n][0]);
                    ldx(std::false_type(), pw[n][1], 0, k, b[c][n][1]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (kb0 + 64 * c >= K) break;
                // (the four accumulators in turn: consecutive MFMAs are independent)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int q = 0; q < 2; ++q) acc[n][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][n][s], b[c][n][q][s], acc[n][q], 0, 0, 0);
            }
        }
    };
    if (vec) body(std::true_type()); else body(std::false_type());
    // C/D map: lane (i, kq) holds rows 4 kq + r, column i
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[w][n * 2 + q][(4 * kq + r) * 16 + i] = acc[n][q][r];
    __syncthreads();
    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
    const int bq = r0 + row, j = c0 + col;
    float kl = 0.f;
    if (bq < B && j < F) {
        float h[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q] = ((red[0][q][threadIdx.x] + red[1][q][threadIdx.x]) + red[2][q][threadIdx.x]) + red[3][q][threadIdx.x];
        const float m1 = h[0] + p.be[j], l1r = h[1] + p.be[F + j], m2 = h[2] + p.bf[j], l2r = h[3] + p.bf[F + j];
        if (p.EH) { p.EH[(size_t)bq * 2 * F + j] = m1; p.EH[(size_t)bq * 2 * F + F + j] = l1r; }
        if (p.FH) { p.FH[(size_t)bq * 2 * F + j] = m2; p.FH[(size_t)bq * 2 * F + F + j] = l2r; }
        // ---- vae_mid (vlsac_agent.py:135-150), as vae_mid_kernel ----
        const float l1 = clamp_lstd(l1r), l2 = clamp_lstd(l2r);
        const float es = eps[(size_t)bq * F + j] * expf(l1);
        p.Z[(size_t)bq * F + j] = m1 + es;
        p.EZ[(size_t)bq * F + j] = es * lstd_mask(l1r);
        const float v1 = expf(2.f * l1), iv2 = expf(-2.f * l2), d = m1 - m2;
        kl = l2 - l1 + 0.5f * (v1 + d * d) * iv2 - 0.5f;
        const float sc = p.scale;
        const float dm1 = d * iv2 * sc;
        p.GEH[(size_t)bq * 2 * F + j] = dm1;
        p.GEH[(size_t)bq * 2 * F + F + j] = (v1 * iv2 - 1.f) * sc * lstd_mask(l1r);
        p.GFH[(size_t)bq * 2 * F + j] = -dm1;
        p.GFH[(size_t)bq * 2 * F + F + j] = (1.f - (v1 + d * d) * iv2) * sc * lstd_mask(l2r);
    }
    const float ssum = block_sum_256(kl, sh);
    if (threadIdx.x == 0) {
        p.partial[tile] = ssum;
        if (tile == 0 && p.step) bump_group(p.step);
    }
}

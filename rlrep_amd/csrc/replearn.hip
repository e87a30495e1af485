// Representation-loss kernels of ctrlsac (InfoNCE), spedersac (spectral) and diffsrsac (score matching), gfx950.
// All are O(B*F) or O(B*B) passes around the GEMMs of gemm16.hip: one wave per row with shuffle reductions,
// deterministic per-block partial sums, gradients written in the layout the backward GEMMs consume.
#include "common.h"
#include "kparams.h"

// ------------------------------------------------------------------------------------------------
// ctrlsac InfoNCE (agent/ctrlsac/ctrlsac_agent.py:226-233)
//   L = mean_i( logsumexp_j S_ij - S_ii ) + 0.5*mean_i (rhat_i - r_i)^2
//   dS = (softmax_row(S) - I)/B  (in place),  drhat = (rhat - r)/B
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void infonce_kernel(InfoNce p) {
    __shared__ float shp[4][2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float accm = 0.f, accr = 0.f;
    for (int i = blockIdx.x * 4 + w; i < p.B; i += gridDim.x * 4) {
        float* row = p.S + (size_t)i * p.ldS;
        float mx = -INFINITY;
        const int NC = p.ncols, di = p.diag_off + i;
        for (int j = lane; j < NC; j += 64) mx = fmaxf(mx, row[j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float se = 0.f;
        for (int j = lane; j < NC; j += 64) se += expf(row[j] - mx);
        se = wave_sum(se);
        const float lse = mx + logf(se);
        const float sii = row[di];
        for (int j = lane; j < NC; j += 64) {
            const float sm = expf(row[j] - lse);
            row[j] = (sm - (j == di ? 1.f : 0.f)) * p.inv_batch;
        }
        float rh;
        if (p.Z) {
            const float* z = p.Z + (size_t)i * p.ldZ;
            float s = 0.f;
            for (int f = lane; f < p.F; f += 64) s = fmaf(z[f], p.theta_w[f], s);
            rh = wave_sum(s) + p.theta_b[0];
        } else rh = p.rhat[i];
        const float dr = rh - p.r[i];
        if (lane == 0) p.drhat[i] = dr * p.inv_batch;
        accm += lse - sii;
        accr += dr * dr;
    }
    if (lane == 0) { shp[w][0] = accm; shp[w][1] = accr; }
    __syncthreads();
    if (threadIdx.x == 0) {
        p.partial[2 * blockIdx.x] = ((shp[0][0] + shp[1][0]) + shp[2][0]) + shp[3][0];
        p.partial[2 * blockIdx.x + 1] = ((shp[0][1] + shp[1][1]) + shp[2][1]) + shp[3][1];
        if (blockIdx.x == 0 && p.step) bump_group(p.step);
    }
}

// ------------------------------------------------------------------------------------------------
// K12 (agent/ctrlsac/ctrlsac_agent.py:226-233): the score matrix AND its InfoNCE loss / gradient in ONE launch, for the small
// products (BASELINE config 3: 256 x 256 scores of 256-wide features).  A workgroup owns 16 WHOLE rows of S = Z ZM^T (so the row-wise log-sum-exp
// needs nobody else): its four waves take the 16-column tiles w, w + 4, ... (v_mfma_f32_16x16x4_f32; operand maps as gemm16_tile.h: a lane takes four
// consecutive inner indices per 16-deep step), the row maximum and the sum of exponentials meet across the waves in LDS in fixed order, and dS =
// (softmax(S) - I) / B goes out from the accumulators -- S itself never exists in memory.  rhat = theta . Z_i + b rides on the A fragments.
// 16 workgroups for B = 256: fine for a 33-MFLOP product on a chain whose cost is its launch count (one dependent launch less per feature step);
// the builder keeps the GEMM + infonce_kernel pair for F > 512 or more than 1 024 columns.
// ------------------------------------------------------------------------------------------------
#define SI_MAX_TILES 4                 /* column tiles per wave: ncols <= 4 * 4 * 16 = 256 */
#define SI_DEPTH 4                     /* 16-deep steps whose operands are in flight (F % 64 == 0) */
typedef float si_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void score_infonce_kernel(InfoNce p) {
    __shared__ float smax[4][16], ssum[4][16], sdiag[16], srow[2][16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;                      // operand row / column of the lane, its group of four inner indices
    const int i0 = blockIdx.x * 16;
    const int NC = p.ncols, ntiles = (NC + 15) >> 4;
    const int nt = (ntiles - w + 3) >> 2;                          // tiles w, w + 4, ... of this wave
    const float* __restrict__ za = p.Z + (size_t)min(i0 + li, p.B - 1) * p.ldZ + 4 * lg;
    const float* __restrict__ th = p.theta_w + 4 * lg;
    const float* zb[SI_MAX_TILES];
#pragma unroll
    for (int t = 0; t < SI_MAX_TILES; ++t) zb[t] = p.ZM + (size_t)min(16 * (w + 4 * t) + li, NC - 1) * p.ldZM + 4 * lg;
    si_f32x4 acc[SI_MAX_TILES];
#pragma unroll
    for (int t = 0; t < SI_MAX_TILES; ++t) acc[t] = (si_f32x4){0.f, 0.f, 0.f, 0.f};
    float rh = 0.f;
    if (w == 0 && lane < 16) sdiag[lane] = 0.f;
    // operands SI_DEPTH steps ahead in SI_DEPTH static register sets (an L2 round trip is ~1 us, the 16 MFMAs of a step 0.2 us: fetched at its own step, every
    // step waited for its loads -- 40 us per launch where the GEMM + loss pair takes 11)
    si_f32x4 ra[SI_DEPTH], rt[SI_DEPTH], rb[SI_DEPTH][SI_MAX_TILES];
    const int Fm = p.F - 16;
#define SI_LOAD(D, K)                                                                                     \
    {                                                                                                     \
        const int kk_ = min((K), Fm);                                                                     \
        ra[D] = *reinterpret_cast<const si_f32x4*>(za + kk_);                                             \
        rt[D] = *reinterpret_cast<const si_f32x4*>(th + kk_);                                             \
        _Pragma("unroll") for (int t = 0; t < SI_MAX_TILES; ++t) rb[D][t] = *reinterpret_cast<const si_f32x4*>(zb[t] + kk_); \
    }
#pragma unroll
    for (int d = 0; d < SI_DEPTH; ++d) SI_LOAD(d, 16 * d)
    for (int k0 = 0; k0 < p.F; k0 += 16 * SI_DEPTH) {
#pragma unroll
        for (int d = 0; d < SI_DEPTH; ++d) {
            const si_f32x4 a = ra[d], tv = rt[d];
            si_f32x4 bb[SI_MAX_TILES];
#pragma unroll
            for (int t = 0; t < SI_MAX_TILES; ++t) bb[t] = rb[d][t];
            SI_LOAD(d, k0 + 16 * (d + SI_DEPTH))
            rh = fmaf(a[0], tv[0], fmaf(a[1], tv[1], fmaf(a[2], tv[2], fmaf(a[3], tv[3], rh))));          // rhat: theta . Z_i over this lane's four inner indices
#pragma unroll
            for (int t = 0; t < SI_MAX_TILES; ++t) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s4], bb[t][s4], acc[t], 0, 0, 0);
            }
        }
    }
#undef SI_LOAD
    // C map: acc[t][q] = S[i0 + 4 lg + q][16 (w + 4 t) + li]
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int t = 0; t < SI_MAX_TILES; ++t) {
        if (t < nt && 16 * (w + 4 * t) + li < NC) {
#pragma unroll
            for (int q = 0; q < 4; ++q) mx[q] = fmaxf(mx[q], acc[t][q]);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx[q] = fmaxf(mx[q], __shfl_xor(mx[q], o, 64));          // over the 16 lanes that share the rows
        if (li == 0) smax[w][4 * lg + q] = mx[q];
    }
    __syncthreads();
    float se[4], sii[4] = {0.f, 0.f, 0.f, 0.f};
    bool has[4] = {false, false, false, false};                    // this lane holds S_ii of row 4 lg + q (exactly one lane of the workgroup does)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 4 * lg + q;
        mx[q] = fmaxf(fmaxf(smax[0][r], smax[1][r]), fmaxf(smax[2][r], smax[3][r]));
        se[q] = 0.f;
    }
#pragma unroll
    for (int t = 0; t < SI_MAX_TILES; ++t) {
        const int col = 16 * (w + 4 * t) + li;
        if (t < nt && col < NC) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                se[q] += expf(acc[t][q] - mx[q]);
                if (col == p.diag_off + i0 + 4 * lg + q) { sii[q] = acc[t][q]; has[q] = true; }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) se[q] += __shfl_xor(se[q], o, 64);
        if (li == 0) ssum[w][4 * lg + q] = se[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) if (has[q]) sdiag[4 * lg + q] = sii[q];          // (the one lane of the workgroup that holds S_ii of the row files it)
    float lse[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 4 * lg + q;
        lse[q] = mx[q] + logf(((ssum[0][r] + ssum[1][r]) + ssum[2][r]) + ssum[3][r]);
    }
    // dS out of the accumulators: (softmax - I) / B
#pragma unroll
    for (int t = 0; t < SI_MAX_TILES; ++t) {
        const int col = 16 * (w + 4 * t) + li;
        if (t < nt && col < NC) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + 4 * lg + q;
                if (i < p.B) p.S[(size_t)i * p.ldS + col] = (expf(acc[t][q] - lse[q]) - (col == p.diag_off + i ? 1.f : 0.f)) * p.inv_batch;
            }
        }
    }
    __syncthreads();
    // per-row terms, rows of this workgroup: wave 0 holds rhat (four inner groups to add), every lane group lg holds lse of rows 4 lg + q
    if (w == 0) {
        rh += __shfl_xor(rh, 16, 64);
        rh += __shfl_xor(rh, 32, 64);                               // lanes li, li + 16, li + 32, li + 48 all hold row li's dot product now
        if (li == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) srow[0][4 * lg + q] = lse[q] - sdiag[4 * lg + q];
        }
        if (lane < 16) {
            const int i = i0 + lane;
            const float dr = (rh + p.theta_b[0]) - p.r[min(i, p.B - 1)];
            srow[1][lane] = i < p.B ? dr * dr : 0.f;
            if (i < p.B) p.drhat[i] = dr * p.inv_batch;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float am = 0.f, ar = 0.f;
        for (int r = 0; r < 16; ++r) { if (i0 + r < p.B) am += srow[0][r]; ar += srow[1][r]; }
        p.partial[2 * blockIdx.x] = am;
        p.partial[2 * blockIdx.x + 1] = ar;
        if (blockIdx.x == 0 && p.step) bump_group(p.step);
    }
}

// ------------------------------------------------------------------------------------------------
// weighted column sum: out[f] = sum_k w[k] * X[k, f]   (w == nullptr: plain column sum); fixed order
// ------------------------------------------------------------------------------------------------
// 16 columns per workgroup, 64 row groups (16 waves x 4 row sub-groups): a [1024, 512] operand is spread over 32
// workgroups x 16 waves with eight independent loads in flight per lane (the first version walked 256 rows per lane
// as one dependent chain on 8 workgroups: 64 us of the 2.1 ms spedersac train(), 12 times).
__global__ __launch_bounds__(1024) void colsum_kernel(ColSum q) {
    __shared__ float sh[64][17];
    const bool second = blockIdx.y == 1;
    // (the set this workgroup works on, as the one-set kernel's parameter block)
    struct { const float* X; int ldX; const float* w; float* out; int rows, F; } p = {second ? q.X2 : q.X, second ? q.ldX2 : q.ldX, second ? q.w2 : q.w,
                                                                                      second ? q.out2 : q.out, second ? q.rows2 : q.rows, q.F};
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane & 15, rg = w * 4 + (lane >> 4);
    const int f = blockIdx.x * 16 + c;
    const int fc = min(f, p.F - 1);
    const float* __restrict__ X = p.X + fc;
    float wsum = 0.f;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    for (int k0 = rg; k0 < p.rows; k0 += 64 * 8) {
        float x[8], wt[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 64 * u, kc = min(k, p.rows - 1);
            x[u] = X[(size_t)kc * p.ldX];
            wt[u] = p.w ? p.w[kc] : 1.f;
            if (k >= p.rows) wt[u] = 0.f;
            wsum += wt[u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = fmaf(wt[u], x[u], acc[u]);
    }
    sh[rg][c] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    const bool push = !second && q.dp.world > 1;        // (block-uniform)
    unsigned dpe = 0;
    if (push) dpe = dp_slots_epoch(q.dp);
    if (threadIdx.x < 16 && f < p.F) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 64; ++r) s += sh[r][threadIdx.x];
        p.out[f] = s;
        if (push) dp_slots_put(q.dp, dpe, f, s);          // this rank's partial into every rank's slot area (dp_pull.h)
    }
    if (push) dp_slots_publish(q.dp, dpe);
    // the sum of the second set's weights (fixed order: the row groups of column 0 of workgroup 0, then across them)
    if (second && q.outb2 && blockIdx.x == 0) {
        __syncthreads();
        if (c == 0) sh[rg][0] = wsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.f;
            for (int r = 0; r < 64; ++r) s += sh[r][0];
            q.outb2[0] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// spedersac spectral loss (agent/spedersac/spedersac_agent.py:186-208)
//   L = -(2/B) sum_i phi_i.mu'_i + (1/B^2) sum_k (Phibar.mu_r,k)^2 + 0.5*mean (theta.phi_i + b - r_i)^2
// rows kernel: c_k = Phibar.mu_r,k ; d_i = phi_i.mu'_i ; rhat_i ; partial sums
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void speder_rows_kernel(SpederRows p) {
    __shared__ float shp[4][3];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int i = blockIdx.x * 4 + w; i < p.B; i += gridDim.x * 4) {
        const float* ph = p.phi + (size_t)i * p.F;
        const float* mu = p.mu + (size_t)i * p.F;
        const float* mr = p.mu_r + (size_t)i * p.F;
        float d = 0.f, c = 0.f, rh = 0.f;
        for (int f = lane; f < p.F; f += 64) {
            const float x = ph[f];
            d = fmaf(x, mu[f], d);
            c = fmaf(mr[f], p.phibar[f], c);
            rh = fmaf(x, p.theta_w[f], rh);
        }
        d = wave_sum(d); c = wave_sum(c); rh = wave_sum(rh) + p.theta_b[0];
        const float dr = rh - p.r[i];
        if (lane == 0) { p.c[i] = c; p.drhat[i] = dr * p.inv_batch; }
        a0 += d; a1 += c * c; a2 += dr * dr;
    }
    if (lane == 0) { shp[w][0] = a0; shp[w][1] = a1; shp[w][2] = a2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int q = threadIdx.x;
        p.partial[3 * blockIdx.x + q] = ((shp[0][q] + shp[1][q]) + shp[2][q]) + shp[3][q];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.step) bump_group(p.step);
}

// gradients w.r.t. the four feature matrices, written as one [2B, F] block each for phi and mu
// (rows 0..B-1: batch 1, rows B..2B-1: the "random" batch) so that the backward GEMMs see M = 2B
__global__ __launch_bounds__(256) void speder_grads_kernel(SpederGrads p) {
    const long long n = (long long)p.B * p.F;
    const float k1 = -2.f * p.inv_batch, k2 = 2.f * p.inv_batch * p.inv_batch;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int i = (int)(e / p.F), f = (int)(e - (long long)i * p.F);
        p.Gphi[e] = k1 * p.mu[e] + p.drhat[i] * p.theta_w[f];
        p.Gmu[e] = k1 * p.phi[e];
        p.Gphi[n + e] = k2 * p.v[f];
        p.Gmu[n + e] = k2 * p.c[i] * p.phibar[f];
    }
}

// ------------------------------------------------------------------------------------------------
// diffsrsac denoising score matching (agent/diffsrsac/diffsrsac_agent.py:271-305)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void diffsr_perturb_kernel(DiffsrPerturb p) {
    const long long n = (long long)p.B * (p.S + 1);
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int b = (int)(e / (p.S + 1)), s = (int)(e - (long long)b * (p.S + 1));
        const float ab = p.alphabars[p.idx[b]];
        if (s == p.S) { p.XN[e] = ab; continue; }
        const float sq = sqrtf(ab), x = p.s2[(size_t)b * p.ld_s2 + s];
        const float pert = sq * x + sqrtf(1.0f - ab) * p.eps[(size_t)b * p.S + s];
        p.XN[e] = pert;
        p.TGT[(size_t)b * p.S + s] = -(pert - sq * x);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { if (p.step0) bump_group(p.step0); if (p.step1) bump_group(p.step1); }
}

// one workgroup per sample: score_s = sum_z phi_z U[z,s]; loss; dphi_z = sum_s dscore_s U[z,s]; U <- dU = phi_z dscore_s
__global__ __launch_bounds__(256) void diffsr_score_kernel(DiffsrScore p) {
    extern __shared__ float sm[];               // [4][S] partial scores, then [S] dscore
    __shared__ float shl[4];
    const int b = blockIdx.x, S = p.S, F = p.F;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* U = p.U + (size_t)b * F * S;
    const float* phi = p.PHI + (size_t)b * F;
    float* part = sm;                            // [4][S]
    float* dsc = sm + 4 * S;                     // [S]
    // pass 1: wave w accumulates z = w, w+4, ... ; lanes over s
    for (int s = lane; s < S; s += 64) {
        float a = 0.f;
        for (int z = w; z < F; z += 4) a = fmaf(phi[z], U[(size_t)z * S + s], a);
        part[w * S + s] = a;
    }
    __syncthreads();
    const float ab = p.alphabars[p.idx[b]];
    const float coef = (1.0f - ab) * p.sigma;
    float l = 0.f;
    for (int s = threadIdx.x; s < S; s += 256) {
        const float score = ((part[s] + part[S + s]) + part[2 * S + s]) + part[3 * S + s];
        const float diff = p.TGT[(size_t)b * S + s] - coef * score;
        l += diff * diff;
        dsc[s] = -2.f * coef * diff * p.inv_batch;
    }
    const float lb = block_sum_256(l, shl);
    if (threadIdx.x == 0) p.partial[b] = lb;
    __syncthreads();
    // pass 2
    for (int z = w; z < F; z += 4) {
        const float pz = phi[z];
        float a = 0.f;
        for (int s = lane; s < S; s += 64) {
            const float u = U[(size_t)z * S + s], d = dsc[s];
            a = fmaf(d, u, a);
            U[(size_t)z * S + s] = pz * d;
        }
        a = wave_sum(a);
        if (lane == 0) p.GPHI[(size_t)b * F + z] = a;
    }
}

// Small state dimensions (S <= 32: HalfCheetah 17, Hopper 11, Walker 17, Ant 27): a THREAD per row z of U[b] -- its S values are one contiguous
// run (a wave reads 64 consecutive rows: fully coalesced) and stay in registers between the two phases, so U is read once; score_s is reduced over
// the rows with S wave reductions and a fixed-order sum over the waves, and dphi_z = sum_s dscore_s U[z,s] needs no reduction at all.  (The
// lanes-over-s kernel above keeps 17 of 64 lanes busy and walks the rows as a dependent chain: 31 us per launch at HalfCheetah dims, 4 launches
// per train().)  NR rows per thread: F <= 256 * NR.
template <int NR>
__global__ __launch_bounds__(256) void diffsr_score_small_kernel(DiffsrScore p) {
    constexpr int SM = 32;
    __shared__ float part[4][SM];
    __shared__ float dsc[SM];
    __shared__ float shl[4];
    const int b = blockIdx.x, S = p.S, F = p.F;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* U = p.U + (size_t)b * F * S;
    const float* phi = p.PHI + (size_t)b * F;
    float u[NR][SM], pz[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int z = threadIdx.x + 256 * j, zc = min(z, F - 1);
        pz[j] = z < F ? phi[zc] : 0.f;
#pragma unroll
        for (int s = 0; s < SM; ++s) u[j][s] = U[(size_t)zc * S + min(s, S - 1)];
    }
    // phase 1: score_s = sum_z phi_z U[z, s]
#pragma unroll
    for (int s = 0; s < SM; ++s) {
        if (s < S) {            // (uniform; no `break`: the loop must unroll fully, u[][] are registers)
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < NR; ++j) a = fmaf(pz[j], u[j][s], a);
            a = wave_sum(a);
            if (lane == 0) part[w][s] = a;
        }
    }
    __syncthreads();
    const float ab = p.alphabars[p.idx[b]];
    const float coef = (1.0f - ab) * p.sigma;
    float l = 0.f;
    if ((int)threadIdx.x < S) {
        const int s = threadIdx.x;
        const float score = ((part[0][s] + part[1][s]) + part[2][s]) + part[3][s];
        const float diff = p.TGT[(size_t)b * S + s] - coef * score;
        l = diff * diff;
        dsc[s] = -2.f * coef * diff * p.inv_batch;
    }
    const float lb = block_sum_256(l, shl);
    if (threadIdx.x == 0) p.partial[b] = lb;
    __syncthreads();
    // phase 2: dphi_z = sum_s dscore_s U[z, s];  U <- dU = phi_z dscore_s
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int z = threadIdx.x + 256 * j;
        float a = 0.f;
#pragma unroll
        for (int s = 0; s < SM; ++s) {
            if (s < S && z < F) {
                const float d = dsc[s];
                a = fmaf(d, u[j][s], a);
                U[(size_t)z * S + s] = pz[j] * d;
            }
        }
        if (z < F) p.GPHI[(size_t)b * F + z] = a;
    }
}

// The same with 16-byte accesses (S % 4 == 0, e.g. Humanoid's 376): a wave walks a row of U as 16-byte lanes (two per
// lane up to S = 512), four rows in flight per wave; 583 -> us at Humanoid dims, where this kernel moves 2.4 GB per call.
__global__ __launch_bounds__(256) void diffsr_score_vec_kernel(DiffsrScore p) {
    extern __shared__ float sm[];               // [4][S] partial scores, then [S] dscore
    __shared__ float shl[4];
    const int b = blockIdx.x, S = p.S, F = p.F, S4 = S >> 2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* U = p.U + (size_t)b * F * S;
    const float* phi = p.PHI + (size_t)b * F;
    float* part = sm;                            // [4][S]
    float* dsc = sm + 4 * S;                     // [S]
    const int c0 = lane, c1 = lane + 64;
    const bool ok0 = c0 < S4, ok1 = c1 < S4;
    const int q0 = ok0 ? c0 : 0, q1 = ok1 ? c1 : 0;
    // pass 1: wave w accumulates rows z = w, w+4, ...
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    for (int z = w; z < F; z += 16) {
        f32x4 u0[4], u1[4]; float pz[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int zz = min(z + 4 * k, F - 1);
            const f32x4* row = reinterpret_cast<const f32x4*>(U + (size_t)zz * S);
            u0[k] = row[q0]; u1[k] = row[q1];
            pz[k] = (z + 4 * k < F) ? phi[zz] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { a0 += pz[k] * u0[k]; a1 += pz[k] * u1[k]; }
    }
    if (ok0) *reinterpret_cast<f32x4*>(part + w * S + 4 * c0) = a0;
    if (ok1) *reinterpret_cast<f32x4*>(part + w * S + 4 * c1) = a1;
    __syncthreads();
    const float ab = p.alphabars[p.idx[b]];
    const float coef = (1.0f - ab) * p.sigma;
    float l = 0.f;
    for (int s = threadIdx.x; s < S; s += 256) {
        const float score = ((part[s] + part[S + s]) + part[2 * S + s]) + part[3 * S + s];
        const float diff = p.TGT[(size_t)b * S + s] - coef * score;
        l += diff * diff;
        dsc[s] = -2.f * coef * diff * p.inv_batch;
    }
    const float lb = block_sum_256(l, shl);
    if (threadIdx.x == 0) p.partial[b] = lb;
    __syncthreads();
    // pass 2: dphi_z = sum_s dscore_s U[z,s];  U <- dU = phi_z dscore_s
    const f32x4 d0 = ok0 ? *reinterpret_cast<const f32x4*>(dsc + 4 * c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 d1 = ok1 ? *reinterpret_cast<const f32x4*>(dsc + 4 * c1) : (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int z = w; z < F; z += 16) {
        f32x4 u0[4], u1[4]; float pz[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int zz = min(z + 4 * k, F - 1);
            const f32x4* row = reinterpret_cast<const f32x4*>(U + (size_t)zz * S);
            u0[k] = row[q0]; u1[k] = row[q1];
            pz[k] = phi[zz];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (z + 4 * k >= F) break;
            const f32x4 t0 = d0 * u0[k], t1 = d1 * u1[k];
            float a = ((t0[0] + t0[1]) + (t0[2] + t0[3])) + ((t1[0] + t1[1]) + (t1[2] + t1[3]));
            a = wave_sum(a);
            f32x4* row = reinterpret_cast<f32x4*>(U + (size_t)(z + 4 * k) * S);
            if (ok0) row[c0] = pz[k] * d0;
            if (ok1) row[c1] = pz[k] * d1;
            if (lane == 0) p.GPHI[(size_t)b * F + z + 4 * k] = a;
        }
    }
}

// ONE pass over U (the default where it applies: S % 4 == 0, F <= 256): the two kernels above read every U[b] twice -- 385 KB per sample at
// Humanoid dims, 788 MB per launch, far more than the L2s hold between the passes: 2.36 GB of HBM traffic for 1.58 GB of algorithmic bytes
// (U read once, dU written once).  Here a workgroup takes its sample's block through LDS in column chunks of CH floats: all F rows of the
// chunk are read once ([F][CH] floats of LDS, eight 16-byte loads in flight per lane), the chunk's scores are reduced over the row groups in
// fixed order, and the second phase (dU = phi_z dscore_s out, dphi_z += sum_s dscore_s U[z,s]) reads the chunk back from LDS.  dphi
// accumulates over the chunks in order in registers: no partial buffers, no atomics.  (Issuing the NEXT chunk's loads before the second phase --
// a register-held software pipeline -- was measured: 396 us against 367: the live registers cost the overlap between the two co-resident
// workgroups more than the overlap inside one wins.)
// CH: floats per column chunk (32 / 64 / 128): LDS = F * CH * 4 bytes per workgroup decides how many workgroups share a CU
template <int CH>
__global__ __launch_bounds__(256) void diffsr_score_lds_kernel(DiffsrScore p) {
    constexpr int LPR = CH / 4;                 // lanes per row
    constexpr int NRG = 256 / LPR;              // row groups
    constexpr int MAXK = 256 / NRG;             // rows per thread (F <= 256)
    extern __shared__ float sm[];               // [F][CH] chunk of U | [NRG][CH] partial scores | [CH] dscore
    __shared__ float shl[4];
    const int b = blockIdx.x, S = p.S, F = p.F;
    float* U = p.U + (size_t)b * F * S;
    const float* phi = p.PHI + (size_t)b * F;
    float* Uc = sm;
    float* part = sm + (size_t)F * CH;
    float* dsc = part + NRG * CH;
    const int tid = threadIdx.x, rg = tid / LPR, c4 = (tid % LPR) * 4;
    const float ab = p.alphabars[p.idx[b]];
    const float coef = (1.0f - ab) * p.sigma;
    float l = 0.f;
    float gacc[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) gacc[k] = 0.f;
    for (int cb = 0; cb < S; cb += CH) {
        const int cw = min(CH, S - cb);
        const bool okc = c4 < cw;
        const int cc = okc ? c4 : 0;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        // phase 1: rows rg, rg + NRG, ... of the chunk: global -> LDS, score partial of this row group
        constexpr int GRP = MAXK < 16 ? MAXK : 16;       // 16-byte loads in flight per lane (16 against 8: 380 -> 367 us)
#pragma unroll
        for (int k0 = 0; k0 < MAXK; k0 += GRP) {
            f32x4 u[GRP]; float pz[GRP];
#pragma unroll
            for (int q = 0; q < GRP; ++q) {
                const int r = rg + NRG * (k0 + q), rr = min(r, F - 1);
                u[q] = *reinterpret_cast<const f32x4*>(U + (size_t)rr * S + cb + cc);
                pz[q] = r < F ? phi[rr] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < GRP; ++q) {
                const int r = rg + NRG * (k0 + q);
                if (r < F) *reinterpret_cast<f32x4*>(Uc + (size_t)r * CH + c4) = u[q];
                a += pz[q] * u[q];
            }
        }
        *reinterpret_cast<f32x4*>(part + rg * CH + c4) = okc ? a : (f32x4){0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        if (tid < cw) {
            float score = part[tid];
#pragma unroll 8
            for (int g = 1; g < NRG; ++g) score += part[g * CH + tid];
            const float diff = p.TGT[(size_t)b * S + cb + tid] - coef * score;
            l += diff * diff;
            dsc[tid] = -2.f * coef * diff * p.inv_batch;
        }
        __syncthreads();
        const f32x4 d4 = okc ? *reinterpret_cast<const f32x4*>(dsc + c4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        // phase 2 from LDS: dU out, dphi partial of this chunk (the LPR lanes of a row group hold one row)
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
            const int r = rg + NRG * k;
            if (r < F) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(Uc + (size_t)r * CH + c4);
                const f32x4 t = d4 * u;
                float sdot = (t[0] + t[1]) + (t[2] + t[3]);
#pragma unroll
                for (int o = LPR / 2; o >= 1; o >>= 1) sdot += __shfl_xor(sdot, o, 64);
                gacc[k] += sdot;
                if (okc) *reinterpret_cast<f32x4*>(U + (size_t)r * S + cb + c4) = phi[r] * d4;
            }
        }
        __syncthreads();
    }
    if ((tid % LPR) == 0) {
#pragma unroll
        for (int k = 0; k < MAXK; ++k) { const int r = rg + NRG * k; if (r < F) p.GPHI[(size_t)b * F + r] = gacc[k]; }
    }
    const float lb = block_sum_256(l, shl);
    if (threadIdx.x == 0) p.partial[b] = lb;
}

__global__ __launch_bounds__(256) void copy2_kernel(const float* __restrict__ src, float* __restrict__ d1, float* __restrict__ d2, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = src[i];
        d1[i] = v;
        if (d2) d2[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// diffsrsac ELU-layer regulariser statistics (RegStats).  Blocks [0, 4 nbc): 1024 elements of one Gram matrix each; blocks
// [4 nbc, 4 nbc + 4 nbr): four rows of one x each (one wave per row).  Deterministic: one partial per block, summed by the finaliser.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reg_stats_kernel(RegStats p) {
    __shared__ float sh[4];
    const int bid = blockIdx.x;
    const float n = (float)p.B, d = (float)p.H;
    const float a = 1.0f / ((n - 1.0f) * n);
    float v = 0.f;
    if (bid < 4 * p.nbc) {
        const int k = bid / p.nbc, blk = bid - k * p.nbc;
        const long long tot = (long long)p.H * p.H;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long e = (long long)blk * 1024 + u * 256 + threadIdx.x;
            if (e < tot) { const float c = p.C[k][e]; v += c * c; }
        }
        v *= a;
    } else {
        const int rbid = bid - 4 * p.nbc;
        const int k = rbid / p.nbr, blk = rbid - k * p.nbr;
        const int row = blk * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        float s = 0.f;
        if (row < p.B) for (int c = lane; c < p.H; c += 64) { const float x = p.X[k][(size_t)row * p.H + c]; s += x * x; }
        s = wave_sum(s);                                   // |x_row|^2
        if (lane == 0 && row < p.B) v = -a * s * s - 2.0f * s / (n * d);
        if (bid == 4 * p.nbc && threadIdx.x == 0) v += 4.0f / d;      // part3, once per (net, head)
    }
    const float r = block_sum_256(v, sh);
    if (threadIdx.x == 0) p.partial[bid] = p.lambda * r;
}
extern "C" int rl_launch_reg_stats(const RegStats* p, hipStream_t st) {
    hipLaunchKernelGGL(reg_stats_kernel, dim3(4 * (p->nbc + p->nbr)), dim3(256), 0, st, *p);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
static inline int rows_blocks(int B) { int g = (B + 3) / 4; return g > 128 ? 128 : g; }

extern "C" int rl_launch_infonce(const InfoNce* p, hipStream_t st) {
    if (p->ZM) {          // score matrix + loss + gradient in one launch (score_infonce_kernel): 16 whole rows per workgroup
        if (!p->Z || (p->F & 63) || (p->ldZ & 3) || (p->ldZM & 3) || ((((uintptr_t)p->Z) | ((uintptr_t)p->ZM) | ((uintptr_t)p->theta_w)) & 15) || p->ncols > 64 * SI_MAX_TILES ||
            p->nblk != (p->B + 15) / 16) return -7;
        hipLaunchKernelGGL(score_infonce_kernel, dim3(p->nblk), dim3(256), 0, st, *p);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(infonce_kernel, dim3(p->nblk), dim3(256), 0, st, *p);
    return (int)hipGetLastError();
}
extern "C" int rl_launch_colsum(const ColSum* p, hipStream_t st) {
    ColSum q = *p;
    q.dp.nblocks = (p->F + 15) / 16;                   // the first set's blocks take the ticket (dp_slots_publish)
    if (q.dp.world > 1 && (q.dp.n < p->F || !q.dp.slot[q.dp.rank])) return -7;
    hipLaunchKernelGGL(colsum_kernel, dim3((p->F + 15) / 16, p->X2 ? 2 : 1), dim3(1024), 0, st, q);
    return (int)hipGetLastError();
}
extern "C" int rl_launch_speder_rows(const SpederRows* p, hipStream_t st) {
    hipLaunchKernelGGL(speder_rows_kernel, dim3(p->nblk), dim3(256), 0, st, *p);
    return (int)hipGetLastError();
}
extern "C" int rl_launch_speder_grads(const SpederGrads* p, hipStream_t st) {
    long long n = (long long)p->B * p->F;
    int g = (int)((n + 1023) / 1024); if (g > 2048) g = 2048; if (g < 1) g = 1;
    hipLaunchKernelGGL(speder_grads_kernel, dim3(g), dim3(256), 0, st, *p);
    return (int)hipGetLastError();
}
extern "C" int rl_launch_diffsr_perturb(const DiffsrPerturb* p, hipStream_t st) {
    long long n = (long long)p->B * (p->S + 1);
    int g = (int)((n + 255) / 256); if (g > 2048) g = 2048; if (g < 1) g = 1;
    hipLaunchKernelGGL(diffsr_perturb_kernel, dim3(g), dim3(256), 0, st, *p);
    return (int)hipGetLastError();
}
template <int CH> static size_t ds_lds(int F) { return ((size_t)F * CH + (size_t)(1024 / CH) * CH + CH) * sizeof(float); }
extern "C" int rl_replearn_init() {
    hipError_t e = hipFuncSetAttribute((const void*)diffsr_score_lds_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ds_lds<128>(256));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)diffsr_score_lds_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ds_lds<64>(256));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)diffsr_score_lds_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ds_lds<32>(256));
    return (int)e;
}
extern "C" int rl_launch_diffsr_score(const DiffsrScore* p, hipStream_t st) {
    // one pass through LDS where it applies; column chunks of 64 floats (64 KB of LDS at F = 256: two workgroups per CU, whose load and store
    // phases overlap) measured best at Humanoid dims: 380 us against 393 (32), 514 (128: one workgroup per CU) and 462 for the two-pass kernel.
    // RLREP_ENABLE=score_ch=32 / 64 / 128 selects the chunk, 0 the two-pass kernels.
    const char* e = rl_opt("score_ch");
    const int ch = e ? atoi(e) : 64;
    if (ch && (p->S & 3) == 0 && p->F <= 256 && ((((uintptr_t)p->U) & 15) == 0)) {
        if (ch == 128) hipLaunchKernelGGL(diffsr_score_lds_kernel<128>, dim3(p->B), dim3(256), ds_lds<128>(p->F), st, *p);
        else if (ch == 64) hipLaunchKernelGGL(diffsr_score_lds_kernel<64>, dim3(p->B), dim3(256), ds_lds<64>(p->F), st, *p);
        else hipLaunchKernelGGL(diffsr_score_lds_kernel<32>, dim3(p->B), dim3(256), ds_lds<32>(p->F), st, *p);
    }
    else if (p->S <= 32 && p->F <= 1024 && !rl_off("score_small")) {
        if (p->F <= 256) hipLaunchKernelGGL(diffsr_score_small_kernel<1>, dim3(p->B), dim3(256), 0, st, *p);
        else if (p->F <= 512) hipLaunchKernelGGL(diffsr_score_small_kernel<2>, dim3(p->B), dim3(256), 0, st, *p);
        else hipLaunchKernelGGL(diffsr_score_small_kernel<4>, dim3(p->B), dim3(256), 0, st, *p);
    }
    else if ((p->S & 3) == 0 && p->S <= 512 && ((((uintptr_t)p->U) & 15) == 0))
        hipLaunchKernelGGL(diffsr_score_vec_kernel, dim3(p->B), dim3(256), (size_t)5 * p->S * sizeof(float), st, *p);
    else
        hipLaunchKernelGGL(diffsr_score_kernel, dim3(p->B), dim3(256), (size_t)5 * p->S * sizeof(float), st, *p);
    return (int)hipGetLastError();
}
extern "C" int rl_launch_copy2(const float* src, float* d1, float* d2, long long n, hipStream_t st) {
    int g = (int)((n + 1023) / 1024); if (g > 2048) g = 2048; if (g < 1) g = 1;
    hipLaunchKernelGGL(copy2_kernel, dim3(g), dim3(256), 0, st, src, d1, d2, n);
    return (int)hipGetLastError();
}

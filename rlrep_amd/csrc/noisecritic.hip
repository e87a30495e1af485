// vlsac noise-averaged critic, first layer (reference agent/vlsac/vlsac_agent.py:44-63), gfx950.
//
//   x[(b,n), :] = mean[b,:] + exp(log_std[b,:]) * noise[n,:]           (n < N = 20, never materialised)
//   Hm[b, j]    = (1/N) * sum_n elu( W[j,:] . x[(b,n),:] + bias[j] )
//
// This is 63 % of the FLOPs of a vlsac train() (ten [B*20 x 256] x [256 x 256] products).  The [B*20,F]
// input is generated on the fly inside the A-operand fragment from three small LDS-resident tables
// (8 rows of mean, 8 rows of sigma, the 20 noise rows), so HBM/L2 only see W and the outputs.
//
// Row mapping trick: the 16 rows of an MFMA A-fragment f are assigned (b' = row>>2, n = 4*f + (row&3)).
// With the 16x16x4 C/D map (row = 4*(lane>>4) + reg) every lane then owns ALL 20 noise rows of one batch
// row b' = lane>>4 across its 5 accumulator fragments, so the mean over the noise axis is a purely
// in-register sum of 20 values: no shuffles, no LDS, no atomics.
#include "common.h"
#include "kparams.h"

#define NC_NF 5          // N / 4 accumulator fragments per batch-row group (N = 20)
#define NC_G2 2          // batch-row groups (of 4 rows) per wave -> 8 batch rows per workgroup

extern __shared__ __attribute__((aligned(16))) float nc_smem[];

__global__ __launch_bounds__(256) void nc_fwd_kernel(const NcFwdTask* __restrict__ tasks, int ntasks) {
    const int bid = blockIdx.x;
    int ti = 0;
    for (int q = 1; q < ntasks; ++q) if (bid >= tasks[q].tile_base) ti = q;
    const NcFwdTask& t = tasks[ti];
    const int local = bid - t.tile_base;
    const int tb = local / t.tiles_h, th = local - tb * t.tiles_h;
    const int b0 = tb * 8, n0 = th * 64;
    const int F = t.F, H = t.H, N = t.N;
    const int Fp = (F + 15) & ~15;
    const int LDS_LD = Fp + 16;
    float* mu_s = nc_smem;                    // [8][LDS_LD]
    float* sg_s = mu_s + 8 * LDS_LD;          // [8][LDS_LD]
    float* nz_s = sg_s + 8 * LDS_LD;          // [N][LDS_LD]

    for (int e = threadIdx.x; e < 8 * Fp; e += 256) {
        const int rr = e / Fp, k = e - rr * Fp;
        float m = 0.f, s = 0.f;
        if (b0 + rr < t.B && k < F) {
            m = t.mean[(size_t)(b0 + rr) * t.ld_ml + k];
            s = expf(clamp_lstd(t.lstd[(size_t)(b0 + rr) * t.ld_ml + k]));
        }
        mu_s[rr * LDS_LD + k] = m;
        sg_s[rr * LDS_LD + k] = s;
    }
    for (int e = threadIdx.x; e < N * Fp; e += 256) {
        const int rr = e / Fp, k = e - rr * Fp;
        nz_s[rr * LDS_LD + k] = (k < F) ? t.noise[(size_t)rr * F + k] : 0.f;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m16 = lane & 15, kq = lane >> 4;
    const int bp = m16 >> 2, nn = m16 & 3;
    const int col = n0 + 16 * w + m16;           // B-operand column (hidden unit) of this lane
    const bool colok = col < H;
    const bool vecW = ((F & 3) == 0) && ((((uintptr_t)t.W) & 15) == 0);

    f32x4 acc[NC_G2][NC_NF];
#pragma unroll
    for (int g = 0; g < NC_G2; ++g)
#pragma unroll
        for (int f = 0; f < NC_NF; ++f) acc[g][f] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kb = 0; kb < Fp; kb += 16) {
        const int k0 = kb + 4 * kq;
        float wv[4] = {0.f, 0.f, 0.f, 0.f};
        if (colok && k0 < F) {
            const float* wp = t.W + (size_t)col * F + k0;
            if (vecW) { f32x4 x = *reinterpret_cast<const f32x4*>(wp); wv[0] = x[0]; wv[1] = x[1]; wv[2] = x[2]; wv[3] = x[3]; }
            else {
#pragma unroll
                for (int s = 0; s < 4; ++s) if (k0 + s < F) wv[s] = wp[s];
            }
        }
        f32x4 mu4[NC_G2], sg4[NC_G2], nz4[NC_NF];
#pragma unroll
        for (int g = 0; g < NC_G2; ++g) {
            mu4[g] = *reinterpret_cast<const f32x4*>(&mu_s[(4 * g + bp) * LDS_LD + k0]);
            sg4[g] = *reinterpret_cast<const f32x4*>(&sg_s[(4 * g + bp) * LDS_LD + k0]);
        }
#pragma unroll
        for (int f = 0; f < NC_NF; ++f) nz4[f] = *reinterpret_cast<const f32x4*>(&nz_s[(4 * f + nn) * LDS_LD + k0]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int g = 0; g < NC_G2; ++g)
#pragma unroll
                for (int f = 0; f < NC_NF; ++f)
                    acc[g][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(fmaf(sg4[g][s], nz4[f][s], mu4[g][s]), wv[s], acc[g][f], 0, 0, 0);
    }

    if (!colok) return;
    const float bj = t.bias[col];
    const float invN = 1.0f / (float)N;
#pragma unroll
    for (int g = 0; g < NC_G2; ++g) {
        const int b = b0 + 4 * g + (lane >> 4);
        if (b >= t.B) continue;
        float sum = 0.f;
#pragma unroll
        for (int f = 0; f < NC_NF; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y = elu_f(acc[g][f][r] + bj);
                sum += y;
                if (t.U) t.U[((size_t)b * N + 4 * f + r) * H + col] = y;
            }
        t.Hm[(size_t)b * H + col] = sum * invN;
    }
}

// dL/d(mean, log_std) of the noise critic's first layer, both heads summed (actor step).
__global__ __launch_bounds__(256) void nc_dx_kernel(const NcDxTask* __restrict__ tasks, int ntasks) {
    const int bid = blockIdx.x;
    int ti = 0;
    for (int q = 1; q < ntasks; ++q) if (bid >= tasks[q].tile_base) ti = q;
    const NcDxTask& t = tasks[ti];
    const int local = bid - t.tile_base;
    const int tb = local / t.tiles_k, tk = local - tb * t.tiles_k;
    const int b0 = tb * 8, kc0 = tk * 64;
    const int F = t.F, H = t.H, N = t.N;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m16 = lane & 15, kq = lane >> 4;
    const int bp = m16 >> 2, nn = m16 & 3;
    const int kcol = kc0 + 16 * w + m16;          // feature column of this lane (B operand / output)
    const bool colok = kcol < F;
    const float invN = 1.0f / (float)N;
    const bool vecG = ((H & 3) == 0) && ((t.ldgh & 3) == 0);

    f32x4 acc[NC_G2][NC_NF];
#pragma unroll
    for (int g = 0; g < NC_G2; ++g)
#pragma unroll
        for (int f = 0; f < NC_NF; ++f) acc[g][f] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int h = 0; h < t.nheads; ++h) {
        const float* GH = t.GH[h];
        const float* U = t.U[h];
        const float* W = t.W[h];
        for (int jb = 0; jb < H; jb += 16) {
            const int j0 = jb + 4 * kq;
            float wv[4] = {0.f, 0.f, 0.f, 0.f};
            if (colok) {
#pragma unroll
                for (int s = 0; s < 4; ++s) if (j0 + s < H) wv[s] = W[(size_t)(j0 + s) * F + kcol];
            }
            float av[NC_G2][NC_NF][4];
#pragma unroll
            for (int g = 0; g < NC_G2; ++g) {
                const int b = b0 + 4 * g + bp;
                float gh[4] = {0.f, 0.f, 0.f, 0.f};
                const bool rok = (b < t.B) && (j0 < H);
                if (rok) {
                    const float* gp = GH + (size_t)b * t.ldgh + j0;
                    if (vecG) { f32x4 x = *reinterpret_cast<const f32x4*>(gp); gh[0] = x[0]; gh[1] = x[1]; gh[2] = x[2]; gh[3] = x[3]; }
                    else {
#pragma unroll
                        for (int s = 0; s < 4; ++s) if (j0 + s < H) gh[s] = gp[s];
                    }
                }
#pragma unroll
                for (int f = 0; f < NC_NF; ++f) {
                    float u[4] = {0.f, 0.f, 0.f, 0.f};
                    if (rok) {
                        const float* up = U + ((size_t)b * N + 4 * f + nn) * H + j0;
                        if (vecG) { f32x4 x = *reinterpret_cast<const f32x4*>(up); u[0] = x[0]; u[1] = x[1]; u[2] = x[2]; u[3] = x[3]; }
                        else {
#pragma unroll
                            for (int s = 0; s < 4; ++s) if (j0 + s < H) u[s] = up[s];
                        }
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) av[g][f][s] = gh[s] * invN * elu_grad_from_out(u[s]);
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int g = 0; g < NC_G2; ++g)
#pragma unroll
                    for (int f = 0; f < NC_NF; ++f)
                        acc[g][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][f][s], wv[s], acc[g][f], 0, 0, 0);
        }
    }

    if (!colok) return;
    float nz[NC_NF][4];
#pragma unroll
    for (int f = 0; f < NC_NF; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) nz[f][r] = t.noise[(size_t)(4 * f + r) * F + kcol];
#pragma unroll
    for (int g = 0; g < NC_G2; ++g) {
        const int b = b0 + 4 * g + (lane >> 4);
        if (b >= t.B) continue;
        float dmu = 0.f, dls = 0.f;
#pragma unroll
        for (int f = 0; f < NC_NF; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) { dmu += acc[g][f][r]; dls = fmaf(acc[g][f][r], nz[f][r], dls); }
        const float l = t.lstd[(size_t)b * t.ld_l + kcol];
        t.G[(size_t)b * t.ldg + kcol] = dmu;
        t.G[(size_t)b * t.ldg + F + kcol] = dls * expf(clamp_lstd(l)) * lstd_mask(l);
    }
}

extern "C" int rl_launch_nc_fwd(const NcFwdTask* tasks_dev, int ntasks, int total_tiles, int F, int N, hipStream_t st) {
    if (total_tiles <= 0) return 0;
    const int Fp = (F + 15) & ~15;
    const size_t lds = (size_t)(16 + N) * (Fp + 16) * sizeof(float);
    hipLaunchKernelGGL(nc_fwd_kernel, dim3(total_tiles), dim3(256), lds, st, tasks_dev, ntasks);
    return (int)hipGetLastError();
}

extern "C" int rl_launch_nc_dx(const NcDxTask* tasks_dev, int ntasks, int total_tiles, hipStream_t st) {
    if (total_tiles <= 0) return 0;
    hipLaunchKernelGGL(nc_dx_kernel, dim3(total_tiles), dim3(256), 0, st, tasks_dev, ntasks);
    return (int)hipGetLastError();
}

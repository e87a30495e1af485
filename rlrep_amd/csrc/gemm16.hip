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
// backward, weight/bias gradient (+Adam), mse loss, tanh-Gaussian policy forward/backward).
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
#include "common.h"

// BRANCH-FREE operand fetch.  Out-of-range rows / inner indices are handled by CLAMPING the address into the
// matrix and zeroing the value with a select: no exec-masked branch around any load.  (With `if (in_range) load`
// hipcc wraps every load in s_cbranch_execz + s_waitcnt vmcnt(0): 147 branches and 21 full drains in a kernel
// with 16 MFMAs, i.e. the eight 16-byte loads a wave needs were serialised instead of overlapped.)
template <int LOADER, bool VEC>
__device__ __forceinline__ void load_raw(const float* __restrict__ P, int ld, int base, int lim,
                                         int i, int k0, int K, float (&v)[4]) {
    const int idx = min(base + i, lim - 1);
    if (LOADER == LD_ROW) {
        if (VEC) {                       // K % 4 == 0, 16-byte aligned rows: the 4 indices are valid together
            const f32x4 x = *reinterpret_cast<const f32x4*>(P + (size_t)idx * ld + min(k0, K - 4));
            v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
        } else {
            const float* p = P + (size_t)idx * ld;
#pragma unroll
            for (int s = 0; s < 4; ++s) v[s] = p[min(k0 + s, K - 1)];
        }
    } else {  // LD_COL
        const float* p = P + idx;
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = p[(size_t)min(k0 + s, K - 1) * ld];
    }
}
// zero what the clamped load fetched from outside the matrix
__device__ __forceinline__ void mask_frag(int base, int lim, int i, int k0, int K, float (&v)[4]) {
    const bool rok = (base + i) < lim;
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = (rok && k0 + s < K) ? v[s] : 0.f;
}

template <int LA, int LB, int NF, bool VA, bool VB>
__global__ __launch_bounds__(256) void gemm16_kernel(GemmBatch gb) {
    __shared__ float red[4][NF][4][64];
    __shared__ float bsum[4][16];

    const int bid = blockIdx.x;
    int ti = 0;
#pragma unroll
    for (int q = 1; q < GEMM_MAX_TASKS; ++q) if (q < gb.ntasks && bid >= gb.t[q].tile_base) ti = q;
    const GemmTask& t = gb.t[ti];
    const int local = bid - t.tile_base;
    const int tr = local / t.tiles_c, tc = local - tr * t.tiles_c;
    const int r0 = tr * 16, c0 = tc * 16 * NF;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int K = t.K;

    f32x4 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float asum = 0.f;
    const bool want_bias = (t.epi == EPI_DW) && (t.flags & FLAG_BIASGRAD) && (tc == 0);

    // this thread's output elements are known up front: issue the epilogue's operand loads (bias, saved
    // activation, accumulate-into value) NOW so their latency overlaps the operand stream
    const int ol = threadIdx.x & 63, oreg = threadIdx.x >> 6;
    const int r = r0 + (ol >> 4) * 4 + oreg;
    float e0[NF], cold[NF], cold2[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        e0[f] = cold[f] = cold2[f] = 0.f;
        const int c = c0 + 16 * f + (ol & 15);
        if (r >= t.R || c >= t.Cn) continue;
        float* cp = t.C + (size_t)r * t.ldc + c;
        if (t.epi == EPI_FWD) { if (t.bias) e0[f] = t.bias[c]; }
        else if (t.epi == EPI_DX) {
            if (t.act != ACT_NONE) e0[f] = t.aux[(size_t)r * t.ldaux + c];
            if (t.flags & FLAG_ACCUM) cold[f] = *cp;
            if (t.r1u) cold2[f] = t.r1u[r] * t.r1v[c];
        }
        else if (t.epi == EPI_FWD_MSE) { e0[f] = t.bias[c]; cold[f] = (c < t.n0) ? t.x0[(size_t)r * t.ldx0 + c] : t.x1[r]; }
        else if (t.epi == EPI_FWD_POLICY) { e0[f] = t.bias[c]; if (c < t.n0 && t.x2) cold[f] = t.x2[(size_t)r * t.n0 + c]; }
        else if (t.epi == EPI_DX_POLICYBWD) { e0[f] = t.x0[(size_t)r * 2 * t.n0 + t.n0 + c]; cold[f] = t.x2[(size_t)r * t.n0 + c]; cold2[f] = t.x1[(size_t)r * t.ldx1 + c]; }
        else if (t.epi == EPI_DX_REPARAM) { e0[f] = t.aux3[(size_t)r * t.ldaux3 + c]; cold[f] = *cp; cold2[f] = cp[t.F]; }
        else if (t.flags & FLAG_ACCUM) cold[f] = *cp;
    }

    AdamScal adsc;
    if (t.epi == EPI_DW && t.ad_p) adsc = t.ad_grp->sc;

    // wave w owns the 16-wide inner chunks w, w+4, w+8, ...; four chunks (all of K <= 256) are loaded
    // back to back before the first MFMA so that their L2 latencies overlap
    for (int kb = w * 16; kb < K; kb += 256) {
        float a[4][4], b[4][NF][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            load_raw<LA, VA>(t.A, t.lda, r0, t.R, i, kb + 64 * u + 4 * kq, K, a[u]);
#pragma unroll
            for (int f = 0; f < NF; ++f) load_raw<LB, VB>(t.B, t.ldb, c0 + 16 * f, t.Cn, i, kb + 64 * u + 4 * kq, K, b[u][f]);
        }
        // every load of the group is issued before anything consumes one (hipcc otherwise sinks each load next to
        // its MFMA and drains vmcnt(0) in between: eight serialised L2 round trips instead of one)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            mask_frag(r0, t.R, i, kb + 64 * u + 4 * kq, K, a[u]);
#pragma unroll
            for (int f = 0; f < NF; ++f) mask_frag(c0 + 16 * f, t.Cn, i, kb + 64 * u + 4 * kq, K, b[u][f]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int f = 0; f < NF; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][s], b[u][f][s], acc[f], 0, 0, 0);
            if (want_bias) asum += (a[u][0] + a[u][1]) + (a[u][2] + a[u][3]);
        }
    }

    // cross-wave reduction in fixed order
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[w][f][q][lane] = acc[f][q];
    if (want_bias) {
        asum += __shfl_xor(asum, 16, 64);
        asum += __shfl_xor(asum, 32, 64);
        if (lane < 16) bsum[w][lane] = asum;
    }
    __syncthreads();

    if (want_bias && threadIdx.x < 16 && r0 + (int)threadIdx.x < t.R) {
        const int q = threadIdx.x;
        const float gbv = ((bsum[0][q] + bsum[1][q]) + bsum[2][q]) + bsum[3][q];
        t.out2[r0 + q] = gbv;
        if (t.ad_pb) adam_elem(adsc, gbv, t.ad_pb + r0 + q, t.ad_mb + r0 + q, t.ad_vb + r0 + q, t.ad_tb ? t.ad_tb + r0 + q : nullptr);
    }

    if (t.epi == EPI_FWD_MSE) {
        // decoder heads of the vlsac ELBO (vlsac_agent.py:137-140): the gradient of 0.5*mse replaces the prediction
        float es = 0.f, er = 0.f;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int c = c0 + 16 * f + (ol & 15);
            if (r >= t.R || c >= t.Cn) continue;
            const float v = (((red[0][f][oreg][ol] + red[1][f][oreg][ol]) + red[2][f][oreg][ol]) + red[3][f][oreg][ol]) * t.scale;
            const float d = (v + e0[f]) - cold[f];
            float* cp = t.C + (size_t)r * t.ldc + c;
            if (c < t.n0) { es += d * d; *cp = d * t.s0; } else { er += d * d; *cp = d * t.s1; }
        }
        es = wave_sum(es); er = wave_sum(er);
        __syncthreads();
        if (lane == 0) { red[0][0][0][w] = es; red[0][0][1][w] = er; }
        __syncthreads();
        if (threadIdx.x == 0) {
            t.y0[2 * local] = ((red[0][0][0][0] + red[0][0][0][1]) + red[0][0][0][2]) + red[0][0][0][3];
            t.y0[2 * local + 1] = ((red[0][0][1][0] + red[0][0][1][1]) + red[0][0][1][2]) + red[0][0][1][3];
        }
        return;
    }
    if (t.epi == EPI_FWD_POLICY) {
        // (NF == 1 launches only) the whole [mu | rho] row sits in this one 16-column tile: rho_j is A lanes to the right
        const int A = t.n0;
        const int c = c0 + (ol & 15);
        const bool inb = (r < t.R) && (c < t.Cn);
        float lp = 0.f;
        if (inb) {
            const float v = (((red[0][0][oreg][ol] + red[1][0][oreg][ol]) + red[2][0][oreg][ol]) + red[3][0][oreg][ol]) * t.scale;
            const float x0v = v + e0[0];
            t.C[(size_t)r * t.ldc + c] = x0v;
            if (c < A) {
                const int pl = ol + A;
                const float rho = (((red[0][0][oreg][pl] + red[1][0][oreg][pl]) + red[2][0][oreg][pl]) + red[3][0][oreg][pl]) * t.scale + t.bias[c + A];
                const float tt = tanhf(rho);
                const float l = -5.f + 3.5f * (tt + 1.f);
                const float sg = expf(l);
                const float x = x0v + cold[0] * sg;
                t.y0[(size_t)r * t.ldx0 + c] = tanhf(x);
                lp = -0.5f * cold[0] * cold[0] - l - 0.91893853320467274f - 2.f * (0.69314718055994531f - x - softplus_f(-2.f * x));
            }
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) lp += __shfl_xor(lp, o, 64);
        if (inb && c == 0 && t.y1) t.y1[r] = lp;
        return;
    }

#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int c = c0 + 16 * f + (ol & 15);
        if (r >= t.R || c >= t.Cn) continue;
        const float v = (((red[0][f][oreg][ol] + red[1][f][oreg][ol]) + red[2][f][oreg][ol]) + red[3][f][oreg][ol]) * t.scale;
        float* cp = t.C + (size_t)r * t.ldc + c;
        switch (t.epi) {
        case EPI_FWD: {
            const float x = v + e0[f];
            float y;
            switch (t.act) {
            case ACT_RELU: y = fmaxf(x, 0.f); break;
            case ACT_ELU: y = elu_f(x); break;
            case ACT_SIN: y = sinf(x); t.out2[(size_t)r * t.ldout2 + c] = x; break;
            case ACT_TANH: y = tanhf(x); break;
            default: y = x;
            }
            *cp = y;
        } break;
        case EPI_DX: {
            float g = v + cold2[f];
            switch (t.act) {
            case ACT_RELU: g = e0[f] > 0.f ? g : 0.f; break;
            case ACT_ELU: g *= elu_grad_from_out(e0[f]); break;
            case ACT_SIN: g *= cosf(e0[f]); break;
            case ACT_TANH: g *= (1.f - e0[f] * e0[f]); break;
            default: break;
            }
            *cp = cold[f] + g;
        } break;
        case EPI_DX_POLICYBWD: {
            // v = dL/da_c from the critic path; e0 = rho, cold = eps, cold2 = a = tanh(x)
            const int A = t.n0;
            const float g = (float)exp(t.dptr[0]) * t.s0;            // dL/dlogpi = alpha / B
            const float tt = tanhf(e0[f]);
            const float sg = expf(-5.f + 3.5f * (tt + 1.f));
            const float y = cold2[f], e = cold[f];
            const float h = v * (1.f - y * y);
            t.y0[(size_t)r * 2 * A + c] = g * 2.f * y + h;
            t.y0[(size_t)r * 2 * A + A + c] = (g * (-1.f + 2.f * y * e * sg) + h * e * sg) * 3.5f * (1.f - tt * tt);
        } break;
        case EPI_DX_REPARAM:
            // e0 = eps * exp(log_std) * clamp-mask, written by vae_mid_kernel
            *cp = cold[f] + v;
            cp[t.F] = cold2[f] + v * e0[f];
            break;
        case EPI_DW:
        default: {
            const float g = cold[f] + v;
            *cp = g;
            if (t.ad_p) {
                const size_t o = (size_t)r * t.ldc + c;
                adam_elem(adsc, g, t.ad_p + o, t.ad_m + o, t.ad_v + o, t.ad_t ? t.ad_t + o : nullptr);
            }
        } break;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host launcher
// ------------------------------------------------------------------------------------------------
template <int LA, int LB, bool VA, bool VB>
static void launch_nf(int nf, dim3 g, hipStream_t st, const GemmBatch& gb) {
    if (nf == 1) hipLaunchKernelGGL((gemm16_kernel<LA, LB, 1, VA, VB>), g, dim3(256), 0, st, gb);
    else if (nf == 2) hipLaunchKernelGGL((gemm16_kernel<LA, LB, 2, VA, VB>), g, dim3(256), 0, st, gb);
    else hipLaunchKernelGGL((gemm16_kernel<LA, LB, 4, VA, VB>), g, dim3(256), 0, st, gb);
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

extern "C" int rl_launch_gemm16(int la, int lb, int nf, const GemmBatch* gb, int total_tiles, hipStream_t st) {
    if (total_tiles <= 0) return 0;
    dim3 g(total_tiles);
    if (la == LD_ROW && lb == LD_ROW) {
        if (all_vec(*gb, false) && all_vec(*gb, true)) launch_nf<LD_ROW, LD_ROW, true, true>(nf, g, st, *gb);
        else launch_nf<LD_ROW, LD_ROW, false, false>(nf, g, st, *gb);
    } else if (la == LD_ROW && lb == LD_COL) {
        if (all_vec(*gb, false)) launch_nf<LD_ROW, LD_COL, true, false>(nf, g, st, *gb);
        else launch_nf<LD_ROW, LD_COL, false, false>(nf, g, st, *gb);
    } else if (la == LD_COL && lb == LD_COL) launch_nf<LD_COL, LD_COL, false, false>(nf, g, st, *gb);
    else return -1;
    return (int)hipGetLastError();
}

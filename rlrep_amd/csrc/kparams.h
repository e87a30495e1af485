// Kernel parameter blocks shared by the device code and the host-side program builder.
#pragma once
#include <stdint.h>
#include "dp_pull.h"
struct GroupCfg;

struct SlotFill {
    const float* ring; const int* idx;                       // ring gather (ring != nullptr) ...
    const float *s, *a, *r, *s2, *d;                         // ... or five separate arrays
    int B, S, A;
    float *XE, *XF, *XF2, *XFpi, *R, *D;                    // slot buffers
};

// rlrep_select_action: ONE observation through the actor in one launch (elementwise.hip select_action_kernel)
struct SelectAct {
    const float* obs; float* act;                       // [S] in, [A] out: device pointers (of pinned host buffers, as the launcher maps them)
    const float *W1, *b1, *W2, *b2, *W3, *b3;           // actor.trunk.{0,2,4}: [Ha,S], [Ha,Ha], [2A,Ha]
    int S, Ha, A, explore;
    float lo, hi;
    unsigned long long seed, offset;                    // the draw = rlrep_fill_normal(eps[A], 1, seed, offset)
};
struct PhiloxFill {
    float* dst_f; int* dst_i; long long n;
    int kind;                  // 0: normal*std -> dst_f ; 1: uniform int in [0,hi) -> dst_i
    float std; int hi; const int* hi_dev;
    unsigned long long seed, offset; const int* step_dev; uint32_t stream_id;
    int step_add;              // added to *step_dev (1 inside the train prologue, whose launch increments the counter LAST)
};

// One launch at the top of a graph-replayed train(): the index pool, the noise pool, the gather of the FIRST minibatch
// (its indices are recomputed from the counter-based generator, so the gather does not wait for the pool) and the
// steps += 1 of rlrep_begin_train (done by whichever block finishes last).  Replaces four dependent launches.
struct TrainPrologue {
    PhiloxFill idx, eps;
    SlotFill fill;             // fill.idx == nullptr: row b uses element b of the `idx` stream
    int nb_idx, nb_eps, nb_fill;
    int* counter; int* ticket;          // counter: word 0 of the train() counter block (written); idx.step_dev: word 2 (read); ticket: unused
    // transposed weight shadows regenerated from the parameters as they stand at the head of this train() (32 x 32 tiles, one per block)
    const struct ShadowEnt* sh; int nsh, nb_tr; const float* sh_base;
};

// up to 10 float segments + one int copied by ONE launch (rlrep_defer_snapshot)
#define COPY_MAX_SEGS 10
struct CopySegs { int n; const float* src[COPY_MAX_SEGS]; float* dst[COPY_MAX_SEGS]; long long end[COPY_MAX_SEGS]; const int* isrc; int* idst; };
// The deferred chain's snapshot folded into the LAST feature optimizer launch of a train() (rlrep_defer_arm): the small segments (minibatch
// slices, policy noise, step counter) are copied by extra blocks; the block of parameters / targets the chain reads is written by the
// optimizer's own lanes as they produce the new values ([off, off + n) of the task's range; which = 0: parameters, 1: Polyak target).
struct AdamSnap { CopySegs segs; float* block; long long off, n; int which, on; };

struct PolicyFwd {
    const float* O; const float* eps; int B, A;
    float* act; int ld_act; float* logp;
    int clamp; float lo, hi;
};

struct PolicyBwd {
    const float* O; const float* eps; const float* act; int ld_act;
    const float* dA; int ld_dA;
    const double* alpha_state; float inv_batch;
    float* G; int B, A;
};

struct VaeMid {
    const float* EH; const float* FH; const float* eps;
    float* Z; float* EZ; float* GEH; float* GFH; float* partial;
    int B, F, nblk; float scale; GroupCfg* step;
};

// the Gaussian heads of encoder and f AND vae_mid in one launch (elementwise.hip heads_vae_kernel): tile = 16 rows x 16 feature columns
struct HeadsVae {
    const float* Ae; const float* Af; int lda;            // encoder / f trunk outputs [B, K]
    const float* We; const float* be;                     // encoder heads [2F, K] (mean rows, then log_std rows), bias [2F]
    const float* Wf; const float* bf;
    const float* eps;                                      // [B, F]
    float* Z; float* EZ; float* GEH; float* GFH; float* partial;
    float* EH; float* FH;                                  // the heads themselves [B, 2F] (nullable: nothing downstream reads them)
    int B, F, K, tiles_c; float scale; GroupCfg* step;
};

struct VaeMse {
    const float* DH; const float* s2; int ld_s2; const float* r;
    float* GDH; float* partial; int B, S, nblk; float scale_s, scale_r;
};

struct QHeadCritic {
    const float* Et[2]; const float* Ec[2];
    const float* wt[2]; const float* bt[2]; const float* wc[2]; const float* bc[2];
    const float* logp; const float* R; const float* D;
    const double* alpha_state; float gamma, inv_batch;
    float* dq; float* GE[2]; float* partial;
    int B, H, nblk, train; GroupCfg* step;
    int ldE;                   // row stride of Et/Ec/GE (0 -> H)
};

struct QHeadActor {
    const float* Ec[2]; const float* wc[2]; const float* bc[2];
    const float* logp; const double* alpha_state; float inv_batch, target_entropy;
    float* GE[2]; float* partial_loss; float* partial_c;
    int B, H, nblk; GroupCfg* step;
    int ldE;                   // row stride of Ec/GE (0 -> H)
};

enum FinKind : int { FIN_SUM = 0, FIN_COMBINE = 1, FIN_ALPHA = 2, FIN_COPY = 3, FIN_INC = 4, FIN_HISTORY = 5 };    // FIN_INC: *(int*)out += 1 (launch epoch of the cluster programs)
// FIN_HISTORY (last task of the LAST optimizer launch of a train(), run only while rlrep_history is on): the metric slots in_a[0 .. stride) are
// appended to the ring `out` of `count` records of RL_HIST_REC floats, record = sequence number % count, the number itself (bit pattern) in
// word [RL_HIST_TAG]; the sequence counter is *(int*)partials.  A caller that replays whole-train() graphs reads the metrics of call n from
// record n % count, long after the call, without a snapshot launch per call.
#define RL_HIST_REC 20
#define RL_HIST_TAG 16
#define RL_HIST_N 1024

struct FinTask {
    int kind;
    const float* partials; int count, stride; float scale;
    float* out; float* out2;
    const float* in_a; const float* in_b; float scale_b;
    double* alpha_state; float lr, beta1, beta2, eps; int learn;
};

// Metric finalisation + temperature update, run by ONE wave: the trailing block of the Adam launch, or (optimizer fused
// into the weight-gradient launch) the extra trailing workgroup of that launch.
// torch/optim/adam.py::_single_tensor_adam operation order; SURVEY Appendix A.11/A.12
#ifdef __HIPCC__
// dp (data parallel, dp_pull.h): partial sums that live in the gradient arena (FIN_ALPHA's: the temperature gradient) are read from EVERY rank's
// arena, in rank order -- the caller has passed the READY wait
__device__ inline void finalize_tasks(const FinTask* __restrict__ fin, int nfin, int lane, const DpPull* dp = nullptr) {
    for (int q = 0; q < nfin; ++q) {
        const FinTask f = fin[q];
        if (f.kind == FIN_SUM) {
            float s = 0.f;
            for (int i = lane; i < f.count; i += 64) s += f.partials[(size_t)i * f.stride];
            s = wave_sum(s);
            if (lane == 0) *f.out = s * f.scale;
        } else if (f.kind == FIN_COMBINE) {
            if (lane == 0) *f.out = f.scale * (*f.in_a) + f.scale_b * (*f.in_b);
        } else if (f.kind == FIN_INC) {
            if (lane == 0) *reinterpret_cast<int*>(f.out) += 1;
        } else if (f.kind == FIN_COPY) {
            if (lane == 0) *f.out = *f.in_a;
        } else if (f.kind == FIN_ALPHA) {
            // L_alpha = mean(exp(log_alpha) * c), c detached; d/dlog_alpha = alpha * mean(c); fp64 Adam (quirk Q1)
            float s = 0.f;
            if (dp) {
                const long long toff = (long long)(f.partials - dp->base[dp->rank]);
                for (int i = lane; i < f.count; i += 64) s += dp_sum1(*dp, toff + (long long)i * f.stride);
            } else {
                for (int i = lane; i < f.count; i += 64) s += f.partials[(size_t)i * f.stride];
            }
            s = wave_sum(s);
            if (lane == 0) {
                double* st = f.alpha_state;           // log_alpha, m, v, step
                const float mean_c = s * f.scale;
                const double alpha = exp(st[0]);
                *f.out = (float)alpha * mean_c;       // alpha_loss (fp32 product as in the reference)
                if (f.learn) {
                    const double g = (double)mean_c * alpha;
                    const double b1 = (double)f.beta1, b2 = (double)f.beta2;
                    st[3] += 1.0;
                    st[1] = st[1] + (1.0 - b1) * (g - st[1]);
                    st[2] = st[2] * b2 + (1.0 - b2) * g * g;
                    const double bc1 = 1.0 - pow(b1, st[3]), bc2 = 1.0 - pow(b2, st[3]);
                    const double denom = sqrt(st[2]) / sqrt(bc2) + (double)f.eps;
                    st[0] = st[0] - ((double)f.lr / bc1) * (st[1] / denom);
                }
                *f.out2 = (float)exp(st[0]);          // info['alpha'] is read after the optimizer step
            }
        } else if (f.kind == FIN_HISTORY) {
            // the slots were written by earlier launches and by lane 0 of this wave just above: wait until those stores have left (the vector L1 is
            // write-through; the L2 is this XCD's point of coherence), then read past the L1.  (A __threadfence() here was an agent-scope release + acquire:
            // a write-back and an invalidate of the XCD's L2 by the tail of a launch on the critical chain, once per train().)
            __builtin_amdgcn_s_waitcnt(0x0F70);
            asm volatile("" ::: "memory");
            int* seq = reinterpret_cast<int*>(const_cast<float*>(f.partials));
            const int n = *seq;
            float* rec = f.out + (size_t)(n % f.count) * RL_HIST_REC;
            if (lane < f.stride) rec[lane] = __hip_atomic_load(f.in_a + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == RL_HIST_TAG) reinterpret_cast<int*>(rec)[RL_HIST_TAG] = n;
            if (lane == 0) *seq = n + 1;
        }
    }
}
#endif

// vlsac noise critic (noisecritic.hip)
struct NcFwdTask {
    const float* mean; const float* lstd; int ld_ml;
    const float* noise;              // [N, F]
    const float* W; const float* bias;   // [H, F], [H]
    const unsigned char* W3;         // bf16x3 images of W (ShadowEnt kind 1; nullptr: the kernel splits W itself)
    float* Hm;                       // [B, H]
    float* U;                        // [B*N, H] elu outputs (nullptr: not stored)
    float* sigma_out;                // [B, F] exp(clamp(log_std)) (nullptr: not stored)
    int B, F, H, N;
    int tiles_h, tile_base, ntiles;
};

#define NC_MAX_TASKS 4
struct NcFwdBatch { int ntasks; int engine; int cols; int nt_u; NcFwdTask t[NC_MAX_TASKS]; };     // nt_u: U stored non-temporally (RLREP_ENABLE=nc_u_nt)     // passed by value (kernarg)

struct NcDwTask {
    const float* U; const float* GH; int ldgh;     // [B*N, H], [B, H]
    const float* mean; const float* sigma; int ld_ml;   // mean [B,F] (row stride ld_ml), sigma [B,F] dense
    const float* noise;                             // [N, F]
    float* gW; float* gb;                           // [H, F], [H]
    int B, F, H, N;
    int tiles_k, tile_base, ntiles;
};
// lean: the 128-VGPR build of the fp32 kernel (launch beside the feature chain).  engine 1 = bf16x3 split-K form: `splits` row ranges
// per 64 x 64 output tile, partial tiles in slab [task][split][H][F], bias partials in bslab [task][split][H], summed by a finishing launch
// fin_in_adam: the partials are summed by the critic group's optimizer launch (AdamTask::Slab) instead of nc_dw_fin_kernel
struct NcDwBatch { int ntasks; int lean; int engine; int splits; float* slab; float* bslab; int fin_in_adam; int pad_; NcDwTask t[2]; };

struct NcDxTask {
    const float* GH[2]; int ldgh;    // dL/dHm per head [B, H]
    const float* U[2];               // elu outputs per head [B*N, H]
    const float* W[2];               // [H, F]
    const float* noise;              // [N, F]
    const float* lstd; int ld_l;     // raw log-std [B, F]
    float* G; int ldg;               // out [B, 2F]: (dmean | dlog_std)
    int B, F, H, N, nheads;
    int tiles_k, tile_base, ntiles;
};


#define NC_NF_HOST 5   /* vlsac noise rows / 4 (noisecritic.hip NC_NF) */

// representation losses (replearn.hip)
struct InfoNce {
    float* S; int ldS;                 // [B, ncols] scores in, dS out (in place); ncols = world*B, own rows' diagonal at column diag_off + i
    int ncols, diag_off;
    const float* rhat; const float* r; float* drhat;
    float* partial; int B, nblk; float inv_batch; GroupCfg* step;
    // Z != nullptr: rhat_i = theta . Z_i + theta_b[0] is computed here, by the wave that owns row i (a 2048-long dot product per row), instead of
    // arriving in `rhat` from a 16-row-engine launch of its own beside the score matrix
    const float* Z; int ldZ, F; const float* theta_w; const float* theta_b;
    // ZM != nullptr (score_infonce_kernel, K12): the score matrix S = Z ZM^T is computed HERE as well -- ZM [ncols, F] = mu(s') of every rank's
    // minibatch -- by the workgroup that owns 16 whole rows of it; `nblk` = ceil(B / 16) then
    const float* ZM; int ldZM;
};
// second set (X2 != nullptr; blockIdx.y == 1): another weighted column sum in the same launch, plus the sum of its weights (outb2) -- spedersac's
// theta.l weight / bias gradient (sum_i drhat_i phi_i, sum_i drhat_i) rides with v = sum_k c_k mu_r,k instead of a launch of its own
// dp.world > 1 (attached data-parallel agent; dp_pull.h DpSlots): the FIRST set's column sums are this rank's PARTIAL of a sum over the global
// batch -- they are pushed into every rank's slot area and the launch's last block raises READY; a ONE-block launch behind it (comm.hip
// comm_slots_sum_kernel) waits for all ranks and files the rank-ordered sum where the single-GPU kernels read the vector.  (The consumers
// summing the slots themselves -- zero launches -- was built first: 512 workgroups x 1 024 cache-bypassing loads of the SAME two lines took
// 1.4 ms per train(), docs/history/r06.md.)
struct ColSum { const float* X; int ldX; const float* w; float* out; int rows, F; const float* X2; int ldX2; const float* w2; float* out2; float* outb2; int rows2; DpSlots dp; };
#define RL_SLOTS_MAX_F 65536
struct SpederRows {
    const float* phi; const float* mu; const float* mu_r; const float* phibar;
    const float* theta_w; const float* theta_b; const float* r;
    float* c; float* drhat; float* partial; int B, F, nblk; float inv_batch; GroupCfg* step;
};
struct SpederGrads {
    const float* phi; const float* mu; const float* c; const float* drhat; const float* phibar; const float* v;
    const float* theta_w; float* Gphi; float* Gmu; int B, F; float inv_batch;
};
// diffsrsac critic regulariser (diffsrsac_agent.py:62-75): per head x = l2(elu(l2(sin(l1 z)))) [B, H] and its Gram matrix C = x^T x [H, H]:
//   reg = lambda * ( (sum C^2 - sum_i |x_i|^4) / ((B - 1) B)  -  2 mean_i |x_i|^2 / H  +  1 / H ),  sum C^2 = sum_ij (x_i . x_j)^2
// every block writes its share of the sum over the four (net, head) pairs, already scaled, to partial[block]
struct RegStats { const float* X[4]; const float* C[4]; int B, H, nbc, nbr; float lambda; float* partial; };

struct DiffsrPerturb {
    const float* alphabars; const int* idx; const float* s2; int ld_s2; const float* eps;
    float* XN; float* TGT; int B, S; GroupCfg* step0; GroupCfg* step1;
};
struct DiffsrScore {
    float* U; const float* PHI; const float* TGT; const float* alphabars; const int* idx;
    float* GPHI; float* partial; int B, F, S; float sigma, inv_batch;
};

// ------------------------------------------------------------------------------------------------
// xchain.hip: several DEPENDENT row-local stages of a step program in ONE persistent launch, synchronised per XCD
// ------------------------------------------------------------------------------------------------
// The 8 * mpg workgroups form 8 groups (group = blockIdx.x % 8: one XCD under the round-robin dealing of workgroups, verified at run time
// through HW_REG_XCC_ID); group g owns the 16-row blocks g * rbg .. of every task and walks the phases in order, its members handing their
// tiles to each other through that XCD's L2 (plain stores, one flag per member, sc1 loads).  A phase = what used to be one launch.
enum XcKind : int { XC_GEMM = 0, XC_HEADS_VAE = 1, XC_QHEAD_CRITIC = 2, XC_QHEAD_ACTOR = 3 };
struct XcPhase {
    int kind;
    int task0, ntasks;             // XC_GEMM: tasks[task0 .. task0 + ntasks) of the launch's table; GemmTask::tile_base / ntiles = GROUP-LOCAL tile range
    int la, lb, vecA, vecB, pre;   // operand layouts / access widths of the phase, as for a gemm16 launch
    int tiles;                     // tiles (work items) per group
    int aux;                       // XC_HEADS_VAE / XC_QHEAD_*: index into the launch's table of that kind
    int dyn;                       // XC_HEADS_VAE: which dyn[] slot carries eps
    int rbg, R;                    // XC_GEMM: 16-row blocks per group and row count of the phase's tasks
    int tb[GEMM_MAX_TASKS], tcs[GEMM_MAX_TASKS];   // XC_GEMM: group-local first tile and column tiles of each task (the decode reads no task record)
};
#define XC_GROUPS 8
#define XC_FLAG_STRIDE 64          // flags per group: one flag per member, up to 64 members (two 128-byte lines)
struct XcLaunch {
    const XcPhase* ph; int nph;
    const GemmTask* tasks; const HeadsVae* hv; const QHeadCritic* qc; const QHeadActor* qa;
    int ntasks, nhv, nqc, nqa;
    unsigned* flags;               // [XC_GROUPS][XC_FLAG_STRIDE]: (phase counter << 4) | XCC id of the writer; never reset (monotonic, wrap-safe compare)
    unsigned* err;                 // bit 0: a wait timed out; bit 1: members of one group ran on different XCDs (hand-offs not guaranteed)
    int rbg;                       // 16-row blocks per group (of tasks with the launch's common row count)
    int mpg;                       // members (workgroups) per group; grid = XC_GROUPS * mpg
    const float* dyn[3];           // per-call noise pointers (FLAG_DYN_EPS / _EPS2 / _EPS3)
    int low_prio;
};

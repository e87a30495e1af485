// Row-block programs: a chain of MLP layers executed by ONE workgroup per 16-row block of the minibatch (rowprog.hip).
//
// Forward and dX chains of the update path are row-local: only the weight gradient (and Adam) reduce over the batch.  At batch 256
// a launch per layer costs ~6.3 us per dependent launch (boundary + cold operands + epilogue) for ~0.2 us of arithmetic.
// A row-block program instead keeps the block's activations in LDS and walks the layers back to back; the weights stream
// L2 -> VGPR as MFMA B operands (v_mfma_f32_16x16x4_f32: exact fp32), prefetched ahead of their use because they are known when the
// launch starts.  Independent chains (encoder / f / policy) are separate programs of the same launch (separate workgroups); where one
// chain needs a result of another (the KL term needs both Gaussians) the producer publishes its rows through global memory with an
// agent-scope release and a flag, and the consumer acquires (MI355X guide, "Workgroup dispatch ... visibility", valid forms).
#pragma once
#include <stdint.h>
struct GroupCfg;

#define RP_ROWS 16            // batch rows per workgroup (one MFMA row tile)
#define RP_THREADS 512        // 8 waves: 4 column groups x 2 halves of the inner dimension (two waves per SIMD)
#define RP_MAX_PROGS 8
#define RP_XSLOT 1024            // granules per (cluster, hop, member) slot of the exchange buffer (16 rows x up to 64 columns)
#define RP_MAX_HOPS 12
#define RP_MAX_DYN 4
#define RP_LDS_DYN_MAX (122 * 1024)   // dynamic LDS a program may use for its activation buffers (160 KB per CU minus the kernel's static 37 KB)

enum RpKind : int {
    RP_END = 0,
    RP_LOAD = 1,        // global [rows, K] (row stride ldgin) -> LDS dst, zero padded to N columns
    RP_GEMM = 2,        // dst/gout = epilogue(src[16, K] x W)   (RPF_COL: W used as [K, N] = dX form; else W is [N, K] = forward form)
    RP_VAE_MID = 3,     // vlsac: reparameterised sample + KL and its gradients (vlsac_agent.py:135-150)
    RP_MSE = 4,         // vlsac: 0.5 * mse gradients of the decoder heads, in place (vlsac_agent.py:137-140)
    RP_REPARAM = 5,     // vlsac: (dmean, dlog_std) += (dz, dz * eps * sigma)
    RP_SIGNAL = 6,      // publish this block's global stores to the partner workgroup (flag = 1)
    RP_WAIT = 7,        // wait for the partner's flag, acquire, reset it
    RP_STORE = 8,       // LDS src [16, N] -> global gout
    RP_POLICY = 9,      // tanh-Gaussian sampling + log-prob from the actor head output [mu | rho] (agent/sac/actor.py:76-91)
    // ---- cluster programs: C workgroups per row block, member m owns a column slice of every layer --------------------------------
    // LDS src [16, .] holds the full-width vector being assembled; member m owns columns { m*N + q*ldw + c : q < K pieces, c < N }.
    // PUBLISH writes the own slice to exchange slot (cluster, hop = flag, member) as 8-byte granules {float, tag}; GATHER polls the
    // other members' slots (n0 = 0: own cluster, all other members; 1: peer cluster, ALL members; 2: peer cluster, the same member only)
    // and completes the LDS vector.  XCHG = PUBLISH + GATHER(0).
    RP_PUBLISH = 10, RP_GATHER = 11, RP_XCHG = 12
};

#define RPF_COL 1          // RP_GEMM: B operand is W[k][col] (dX = G W); default W[col][k] (Y = X W^T)
#define RPF_BIAS 2         // add bias[col]
#define RPF_MASK_LDS 4     // multiply by act'(aux) with aux in LDS (src2, lds2)
#define RPF_MASK_GLOBAL 8  // multiply by act'(aux) with aux in global memory (gaux, ldgaux)
#define RPF_BUMP 16        // RP_VAE_MID: row block 0 bumps the optimizer group's step counter
#define RPF_FH_INPLACE 32   // RP_VAE_MID: also overwrite src2 (the f heads) with dKL/d(f heads), for a following RP_PUBLISH

struct RpOp {
    int kind, flags;
    int src, lds;              // LDS operand: float offset, row stride (floats)
    int src2, lds2;            // second LDS operand (mask source / second input)
    int dst, ldd;              // LDS result (dst < 0: none)
    int dst2, ldd2;            // second LDS result (RP_VAE_MID: eps * sigma * clamp-mask)
    int K, N;                  // inner length, output width
    int ldw, act;
    const float* W; const float* bias;
    float* gout; float* gout2;             // global results (nullable), row strides ldg / ldg2
    const float* gaux; const float* gin; const float* gin2;
    int ldg, ldg2, ldgaux, ldgin;
    int n0, dyn, flag, wpad;               // dyn: index of the per-call pointer (RpDyn) this op reads (-1: none); wpad: zeroed LDS width
    float s0, s1;
    float* part; GroupCfg* step;
    // cluster programs: per-member increments (floats / LDS floats) added to W, bias, gout, gout2, gaux, gin and to the LDS offsets dst, src2
    int m_w, m_b, m_g, m_g2, m_gaux, m_gin, m_dst, m_s2, m_src, m_dst2;
    int pad0, pad1;
};

struct RpProg { int op_begin, op_end; int block_base, nblocks; int csize, ctype; };   // one kind of workgroup; nblocks = row blocks x csize members (csize 1: no cluster); ctype: which of the two clusters of a row block

struct RpLaunch {
    const RpOp* ops;            // device table
    int* flags;                 // device flags [nflags][row blocks]
    int nprog, B, low_prio, lds_floats;
    RpProg prog[RP_MAX_PROGS];
    const float* dyn[RP_MAX_DYN];          // per-call pointers (noise), patched at launch time
    unsigned long long* xbuf;              // exchange buffer [row block][2 cluster types][RP_MAX_HOPS][csize][RP_XSLOT] granules
    const int* epoch;                      // device counter that differs between any two launches whose granules / flags could be confused
    unsigned* err;                         // error word (rlrep_chain_status bit 2): a wait of this launch timed out
};

extern "C" int rl_launch_rowprog(const RpLaunch* L, int total_blocks, hipStream_t st);
extern "C" int rl_rowprog_init();

// Host-side engine: tensor layout, workspace, step-program builder (librlrep_hip.so internals).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <map>
#include <functional>
#include <cstring>
#include <cstdio>
#include "../../include/rlrep.h"
#include "common.h"
#include "kparams.h"
#include "rowprog.h"

// kernel launchers (defined next to the kernels)
extern "C" {
int rl_launch_gemm16(int la, int lb, int nf, const GemmBatch* gb, int total_tiles, hipStream_t st);
int rl_launch_gemm_lds(int bt, int la, int lb, const GemmBatch* gb, int total_tiles, int fin_blocks, hipStream_t st);
int rl_gemm_lds_align_ok(const GemmTask* t, int la, int lb);
int rl_gemm_lds_dims_ok(const GemmTask* t, int la, int lb);
int rl_gemm_lds_ptrs_ok(const GemmTask* t);
int rl_gemm_lds_dim_flags(const GemmTask* t, int la, int lb);
int rl_gemm_lds_ptr_flags(const GemmTask* t);
int rl_gemm_lds_route(const GemmTask* t, int la, int lb, int extra_flags, int* splits, int* kchunk, int* flags);
void rl_gemm_lds_plan(const GemmTask* t, int* bt, int* splits, int* kchunk);
int rl_launch_nc_fwd(const NcFwdBatch* nb, int total_tiles, int g2, hipStream_t st);
int rl_launch_nc_dx(const NcDxTask* t, hipStream_t st);
int rl_launch_nc_dw(const NcDwBatch* nb, int total_tiles, hipStream_t st);
int rl_nc_init();
int rl_nc_dw_engine();
int rl_nc_dw_splits(int B, int F, int H, int ntasks);
int rl_nc_fwd_cols();
void rl_nc_fwd_plan(const NcFwdTask* tasks, int ntasks, int* engine, int* g2, int* cols);
int rl_launch_fill_slot(const SlotFill* p, hipStream_t st);
int rl_launch_philox(const PhiloxFill* p, hipStream_t st);
int rl_launch_philox_raw(const uint32_t* ck, uint32_t* out, long long n, hipStream_t st);
int rl_launch_policy_fwd(const PolicyFwd* p, hipStream_t st);
int rl_launch_policy_bwd(const PolicyBwd* p, hipStream_t st);
int rl_launch_vae_mid(const VaeMid* p, hipStream_t st);
int rl_launch_heads_vae(const HeadsVae* p, hipStream_t st);
int rl_launch_xchain(const XcLaunch* L, hipStream_t st);
int rl_launch_vae_mse(const VaeMse* p, hipStream_t st);
int rl_launch_qhead_critic(const QHeadCritic* p, hipStream_t st);
int rl_launch_qhead_actor(const QHeadActor* p, hipStream_t st);
int rl_launch_gemm16_duo(int split, int nf2, const GemmBatch* gb, int total_tiles, hipStream_t st);
extern "C" int rl_replearn_init();
int rl_launch_counter_sync(int* c, int mirror, hipStream_t st);
int rl_launch_adam(const AdamTask* task, int adam_blocks, const FinTask* fin, int nfin, const SlotFill* sf, const SlotFill* sf2, const AdamSnap* snap, const DpPull* dp, hipStream_t st);
int rl_launch_adam_l1(const AdamTask* task, int adam_blocks, const FinTask* fin, int nfin, const SlotFill* sf, const GemmTask* g0, const GemmTask* g1, hipStream_t st);
int rl_launch_train_prologue(TrainPrologue* p, hipStream_t st);
int rl_launch_polyak(const PolyakTask* t, hipStream_t st);
int rl_launch_counter_inc(int* c, int mirror, hipStream_t st);
int rl_launch_copy(const float* src, float* dst, long long n, hipStream_t st);
int rl_launch_copy_segs(const CopySegs* p, hipStream_t st);
int rl_launch_shadow(const ShadowEnt* sh_dev, int nsh, int ntiles, const float* base, int target, hipStream_t st);
}

void rl_set_error(const char* fmt, ...);
// diagnostic switches (engine.hip): token listed in RLREP_DISABLE / value of a token of RLREP_ENABLE ("1" when listed bare; nullptr: not listed),
// as parsed at the last library entry (rl_switches_read)
void rl_switches_read();
bool rl_off(const char* token);
const char* rl_opt(const char* token);

// metric slots (union over the agents; names per agent in rlrep_metric_names)
enum Metric : int {
    M_FEAT_TOTAL = 0,   // vae_loss / total_loss / score_loss
    M_FEAT_A = 1,       // ml_loss / model_loss
    M_KL = 2, M_S_LOSS = 3, M_R_LOSS = 4,
    M_Q1_LOSS = 5,      // q1_loss (sac: q_loss; diffsr: q_loss_reg)
    M_Q2_LOSS = 6,      // q2_loss (diffsr: q_loss_noreg)
    M_Q1 = 7, M_Q2 = 8, M_ACTOR_LOSS = 9, M_ALPHA_LOSS = 10, M_ALPHA = 11,
    M_TMP0 = 12, M_TMP1 = 13, M_TMP2 = 14, M_TMP3 = 15,
    M_COUNT = 16
};

// ------------------------------------------------------------------------------------------------
// layout
// ------------------------------------------------------------------------------------------------
struct LT {
    std::string name; int rows, cols; int arena; int group; int64_t off;
};

struct Layout {
    std::vector<LT> t;
    std::map<std::string, int> index;
    int64_t cur[RLREP_ARENA_COUNT] = {0, 0};
    int64_t group_off[4] = {0, 0, 0, 0}, group_n[4] = {0, 0, 0, 0};

    void align(int arena) { cur[arena] = (cur[arena] + 3) & ~int64_t(3); }
    int64_t add(const std::string& name, int rows, int cols, int arena, int group, bool glue = false) {
        if (!glue) align(arena);
        LT e{name, rows, cols, arena, group, cur[arena]};
        index[name] = (int)t.size();
        t.push_back(e);
        cur[arena] += (int64_t)rows * cols;
        return e.off;
    }
    void lin(const std::string& p, int out_f, int in_f, int arena, int group) {
        add(p + ".weight", out_f, in_f, arena, group);
        add(p + ".bias", out_f, 1, arena, group);
    }
    // two Linear layers sharing the input, stored as one [o1+o2, in] matrix (+ one [o1+o2] bias)
    void lin_pair(const std::string& p1, int o1, const std::string& p2, int o2, int in_f, int arena, int group) {
        add(p1 + ".weight", o1, in_f, arena, group);
        add(p2 + ".weight", o2, in_f, arena, group, true);
        add(p1 + ".bias", o1, 1, arena, group);
        add(p2 + ".bias", o2, 1, arena, group, true);
    }
    void begin_group(int g) { align(RLREP_ARENA_PARAM); group_off[g] = cur[RLREP_ARENA_PARAM]; }
    void end_group(int g) { align(RLREP_ARENA_PARAM); group_n[g] = cur[RLREP_ARENA_PARAM] - group_off[g]; }
    const LT& get(const std::string& n) const {
        auto it = index.find(n);
        if (it == index.end()) { fprintf(stderr, "rlrep: unknown tensor %s\n", n.c_str()); abort(); }
        return t[it->second];
    }
};

bool build_layout(const rlrep_dims& d, Layout& L);

// ------------------------------------------------------------------------------------------------
// workspace bump allocator over caller memory
// ------------------------------------------------------------------------------------------------
struct Workspace {
    char* base = nullptr; size_t cap = 0, used = 0; bool dry = false;
    void* alloc(size_t bytes) {
        used = (used + 255) & ~size_t(255);
        void* p = dry ? nullptr : (void*)(base + used);
        used += bytes;
        return p;
    }
    float* f(size_t n) { return (float*)alloc(n * sizeof(float)); }
    bool ok() const { return dry || used <= cap; }
};

// ------------------------------------------------------------------------------------------------
// programs
// ------------------------------------------------------------------------------------------------
// process-wide count of kernel launches issued by the library (rlrep_launch_counter: bench.py counts the launches a captured train() holds)
extern long long g_rl_launches;
extern long long g_rl_front[4];                 // gemm16.hip: launches per front end (fast, fast4, fastpre, record)
extern "C" void rl_gemm16_read_env();           // gemm16.hip: RLREP_DISABLE=gemm16_fast / gemm16_spec, RLREP_ENABLE=gemm16_trace, read at agent creation
// engine / flops / bytes: what rlrep_stage_info reports (include/rlrep.h RLREP_ENGINE_*: which kernel family the stage launches, the
// ALGORITHMIC flops (2 * MAC) of its products and the bytes of their operands and results, each counted once)
struct Stage { std::function<int(hipStream_t)> run; const char* what; int engine = 0; double flops = 0.0, bytes = 0.0; };

struct Program {
    std::vector<Stage> stages;
    int run(hipStream_t st, size_t first = 0) const {
        for (size_t q = first; q < stages.size(); ++q) {
            const Stage& s = stages[q];
            int rc = s.run(st);
            ++g_rl_launches;
            if (rc != 0) { rl_set_error("stage '%s' failed: hip error %d", s.what, rc); return RLREP_ERR_HIP; }
        }
        return 0;
    }
};

struct Mat { float* p; int rows, cols, ld; };

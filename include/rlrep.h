/*
 * rlrep.h -- C ABI of librlrep_hip.so: the MI355X-native (gfx950) update hot path of rl-rep.
 *
 * The reference (haotiansun14/rl-rep) has NO FFI/operator layer: its boundary for this path is the
 * Python class API of the agents.  Each entry point below therefore names the reference method it
 * replaces (paths relative to the reference repository root):
 *
 *   rlrep_agent_create        <- SACAgent.__init__        agent/sac/sac_agent.py:19-81
 *                                VLSACAgent.__init__      agent/vlsac/vlsac_agent.py:71-123
 *                                CTRLSACAgent.__init__    agent/ctrlsac/ctrlsac_agent.py:127-210
 *                                SPEDERSACAgent.__init__  agent/spedersac/spedersac_agent.py:102-179
 *                                DIFFSRSACAgent.__init__  agent/diffsrsac/diffsrsac_agent.py:95-176
 *   rlrep_feature_step        <- feature_step             vlsac_agent.py:126-162, ctrlsac_agent.py:213-251,
 *                                                         spedersac_agent.py:181-219;
 *                                critic_feeder_feature_step  diffsrsac_agent.py:271-318
 *                                (+ update_feature_target: vlsac :240-242, ctrlsac :253-255, speder :221-223)
 *   rlrep_critic_step         <- critic_step              sac_agent.py:105-135 and per-agent overrides
 *   rlrep_actor_alpha_step    <- update_actor_and_alpha   sac_agent.py:138-166 and per-agent overrides
 *   rlrep_prefetch_policy     <- (no counterpart: reorders the forward half of update_actor_and_alpha, sac_agent.py:141-150,
 *                                into critic_step's launches)
 *   rlrep_update_target       <- update_target            sac_agent.py:99-102
 *   rlrep_train               <- train                    sac_agent.py:169-188, vlsac_agent.py:245-273, ...
 *   rlrep_set_batch           <- Batch / unpack_batch     utils/buffer.py:7-10, utils/util.py:10-11
 *   rlrep_replay_add/_sample  <- ReplayBuffer.add/.sample utils/buffer.py:28-48
 *   rlrep_actor_forward       <- SACAgent.select_action   sac_agent.py:89-96
 *
 * Conventions
 *   - plain C types only; every pointer named *_dev is a DEVICE pointer owned by the caller (e.g. a torch
 *     tensor's data_ptr()); the library never allocates or frees device memory on the hot path
 *     (it uploads small task tables into the caller-provided workspace at create time);
 *   - all tensors are contiguous row-major fp32 unless stated; nn.Linear weights are [out,in];
 *   - every call is ASYNCHRONOUS and stream-ordered on `stream` (a hipStream_t passed as void*);
 *     no call synchronises the device; calls on one agent are not thread-safe;
 *   - return value: 0 on success, negative rlrep_status otherwise; rlrep_last_error() gives text.
 */
#ifndef RLREP_H
#define RLREP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RLREP_ABI_VERSION 4      /* 4: exchange scratch + reduced region in the comm block (two-shot, batch-coupled exchanges inside the launches), rlrep_comm_connect_local, time-bounded waits; dims.world_size, layout_info.exchange_floats */

typedef enum {
    RLREP_OK = 0,
    RLREP_ERR_ARG = -1,        /* bad argument / unsupported dimension */
    RLREP_ERR_HIP = -2,        /* a HIP runtime call failed */
    RLREP_ERR_STATE = -3,      /* call order violated (e.g. step before set_batch) */
    RLREP_ERR_NOMEM = -4       /* caller-provided workspace too small */
} rlrep_status;

typedef enum {
    RLREP_ALG_SAC = 0,
    RLREP_ALG_VLSAC = 1,
    RLREP_ALG_CTRLSAC = 2,
    RLREP_ALG_SPEDERSAC = 3,
    RLREP_ALG_DIFFSRSAC = 4
} rlrep_alg;

/* Network dimensions (mirrors the reference constructors' dimension kwargs). */
typedef struct {
    int32_t alg;               /* rlrep_alg */
    int32_t state_dim;         /* S */
    int32_t action_dim;        /* A */
    int32_t hidden_dim;        /* critic hidden (sac/vlsac/ctrlsac `hidden_dim`, speder `critic_and_actor_hidden_dim`) */
    int32_t actor_hidden_dim;  /* actor trunk hidden (sac/vlsac: hidden_dim; ctrlsac: 256; speder: critic_and_actor_hidden_dim) */
    int32_t feature_dim;       /* F */
    int32_t vae_hidden_dim;    /* vlsac Encoder/Decoder/GaussianFeature hidden (reference default 256) */
    int32_t phi_hidden_dim;    /* ctrlsac hidden_dim / speder phi_hidden_dim / diffsr phi_hidden_dim */
    int32_t phi_hidden_depth;  /* speder, diffsr (ctrlsac: 2 fixed) */
    int32_t mu_hidden_dim;     /* ctrlsac hidden_dim / speder mu_hidden_dim / diffsr nabla_mu_hidden_dim */
    int32_t mu_hidden_depth;   /* speder, diffsr (ctrlsac: 2 fixed) */
    int32_t num_noise;         /* vlsac critic noise rows (20) / diffsr num_noises (1000) */
    int32_t max_batch;         /* largest batch size this agent will be stepped with */
    int32_t rank;              /* data-parallel rank of this replica (0 when world_size == 1) */
    int32_t flags;             /* RLREP_FLAG_* */
    int32_t world_size;        /* data-parallel replicas the WORKSPACE is sized for (0 / 1: one; must equal hyper.world_size when > 1): ctrlsac's score
                                  matrix is [B, world * B], and an attached agent (rlrep_comm_attach) keeps its deferred step programs */
} rlrep_dims;

/* vlsac / ctrlsac / spedersac constructed with use_feature_target=False (vlsac_agent.py:113-114,176-179,214-219,257-258;
 * ctrlsac_agent.py:167-168,185-186,268-273,340-346; spedersac_agent.py:150-151,306-307): no Polyak copy of the feature net; vlsac's critic
 * and actor steps read the LIVE f, ctrlsac's critic step reads frozen_phi.  The target tensors stay in the layout and are never touched. */
#define RLREP_FLAG_NO_FEATURE_TARGET 1

/* Hyper-parameters (mirrors the reference constructors' scalar kwargs). */
typedef struct {
    float lr_feature;          /* feature optimizer(s) */
    float lr_critic;
    float lr_actor;            /* actor and log_alpha optimizers */
    float discount;
    float tau;                 /* critic target Polyak rate */
    float feature_tau;         /* feature target Polyak rate */
    float target_entropy;      /* -action_dim */
    float sigma_scale;         /* diffsr sigma_scale_factor */
    int32_t target_update_period;
    int32_t extra_feature_steps;
    int32_t learn_alpha;       /* auto_entropy_tuning */
    int32_t world_size;        /* data-parallel replicas: local losses are scaled by 1/(B*world_size) */
    float beta1, beta2, adam_eps;
    float critic_reg_lambda;   /* diffsr critic_elu_layer_regularizer_lambda (diffsrsac_agent.py:62-90,215-227): enters q_loss_reg only */
} rlrep_hyper;

/* One named tensor inside an arena (what nn.Parameter views / state_dict() are built from). */
typedef struct {
    char name[72];             /* reference state_dict key, e.g. "encoder.mean_linear.weight" */
    int32_t arena;             /* rlrep_arena */
    int32_t group;             /* optimizer group: 0 feature, 1 critic, 2 actor, 3 feature2 (diffsr nabla-mu), -1 none */
    int64_t offset;            /* in floats from the arena base */
    int32_t rows, cols;        /* bias: rows=n, cols=1 */
} rlrep_tensor_desc;

typedef enum {
    RLREP_ARENA_PARAM = 0,     /* trainable parameters            (fp32) */
    RLREP_ARENA_TARGET = 1,    /* target / frozen copies + vlsac noise + diffsr alphabars (fp32) */
    RLREP_ARENA_COUNT = 2
} rlrep_arena;

/* Sizes the caller must allocate (floats unless stated). grad/exp_avg/exp_avg_sq arenas have the
 * PARAM arena's size and layout (+ RLREP_GRAD_TAIL floats of cross-rank-reducible partial sums at the
 * end of the grad arena). */
typedef struct {
    int64_t param_floats;
    int64_t target_floats;
    int64_t grad_floats;       /* = param_floats + RLREP_GRAD_TAIL */
    int64_t workspace_bytes;   /* activations, batch slots, task tables, metric partials */
    int64_t group_offset[4];   /* start of each optimizer group inside the PARAM arena (floats) */
    int64_t group_floats[4];
    int32_t n_tensors;
    int32_t n_metrics;
    int64_t exchange_floats;   /* exchange scratch (rlrep_comm_create) that moves the agent's batch-coupled feature exchanges into its launches (0: none) */
} rlrep_layout_info;

#define RLREP_GRAD_TAIL 256

/* Device pointers handed over at create time; they must outlive the agent. */
typedef struct {
    float* param_dev;
    float* target_dev;
    float* grad_dev;
    float* exp_avg_dev;
    float* exp_avg_sq_dev;
    void* workspace_dev;
    double* alpha_state_dev;   /* 4 doubles: log_alpha, exp_avg, exp_avg_sq, step   (quirk Q1: float64) */
} rlrep_arenas;

/* A minibatch as five separate device arrays (utils/buffer.py:7-10 field order). */
typedef struct {
    const float* state_dev;       /* [B,S] */
    const float* action_dev;      /* [B,A] */
    const float* reward_dev;      /* [B,1] */
    const float* next_state_dev;  /* [B,S] */
    const float* done_dev;        /* [B,1] */
    int32_t batch;
} rlrep_batch;

typedef struct rlrep_agent rlrep_agent;

/* ---- introspection ------------------------------------------------------------------------ */
int32_t rlrep_abi_version(void);
const char* rlrep_last_error(void);

/* Layout of every tensor of algorithm dims->alg.  `descs` may be NULL to query counts only. */
int32_t rlrep_layout(const rlrep_dims* dims, rlrep_layout_info* info, rlrep_tensor_desc* descs, int32_t cap);

/* Names of the metrics slots written by the step programs (reference dict keys). */
int32_t rlrep_metric_names(int32_t alg, char (*names)[32], int32_t cap);

/* ---- lifetime ----------------------------------------------------------------------------- */
int32_t rlrep_agent_create(const rlrep_dims* dims, const rlrep_hyper* hyper, const rlrep_arenas* arenas,
                           void* stream, rlrep_agent** out);
void rlrep_agent_destroy(rlrep_agent* agent);

/* ---- data: minibatch slots ---------------------------------------------------------------- */
/* slot 0 = the batch used by feature/critic/actor steps; slot 1 = spedersac's second ("random") batch. */
int32_t rlrep_set_batch(rlrep_agent* agent, int32_t slot, const rlrep_batch* batch, void* stream);

/* Device-resident replay ring: rows of [s | a | s' | r | d] (row length 2S+A+2 floats). */
int32_t rlrep_replay_row_floats(const rlrep_dims* dims);
/* ReplayBuffer.add (utils/buffer.py:28-36), batched: `nrows` staged rows (pinned host memory, row_floats floats each) enter the ring at
 * slot `ptr` and wrap around its end (ptr' = (ptr + nrows) % capacity is the caller's to keep, as `self.ptr` is in the reference).
 * Asynchronous on `stream`: at most two copies.  nrows <= capacity. */
int32_t rlrep_replay_add(float* ring_dev, int64_t capacity, int32_t row_floats, int64_t ptr,
                         const float* rows_host, int64_t nrows, void* stream);
/* The same plus the ring's new fill level written to the device scalar the index generator reads (`size_dev`, may be NULL), in ONE launch that
 * reads the staged rows in place: `rows_host` must be pinned (mapped) host memory (RLREP_ERR_ARG otherwise).  What ReplayBuffer.flush() issues in
 * front of every train() of main.py's loop. */
int32_t rlrep_replay_add_sized(float* ring_dev, int64_t capacity, int32_t row_floats, int64_t ptr, const float* rows_host, int64_t nrows,
                               int32_t* size_dev, int32_t new_size, void* stream);
/* gather rows idx_dev[0..batch) of the ring into a batch slot */
int32_t rlrep_replay_sample(rlrep_agent* agent, int32_t slot, const float* ring_dev, const int32_t* idx_dev,
                            int32_t batch, void* stream);
/* Philox4x32-10 fills: uniform indices in [0,hi) and N(0,1) noise (counter-based, replayable). */
int32_t rlrep_fill_indices(int32_t* dst_dev, int64_t n, int32_t hi, uint64_t seed, uint64_t offset, void* stream);
int32_t rlrep_fill_normal(float* dst_dev, int64_t n, float std, uint64_t seed, uint64_t offset, void* stream);
/* The bare Philox4x32-10 bijection the fills are built on, for known-answer tests (Random123 kat_vectors):
 * ctr_key_dev uint32[n][6] = (counter x4, key x2) -> out_dev uint32[n][4].  Not on any hot path. */
int32_t rlrep_philox_raw(const uint32_t* ctr_key_dev, uint32_t* out_dev, int64_t n, void* stream);

/* Graph-replay-safe variants: the Philox offset is `offset + *counter_dev` and the index range is `*hi_dev`,
 * both read on the device at execution time (rlrep_steps_dev() is the agent's train() counter). */
int32_t rlrep_fill_indices_dev(int32_t* dst_dev, int64_t n, const int32_t* hi_dev, uint64_t seed, uint64_t offset,
                               const int32_t* counter_dev, void* stream);
int32_t rlrep_fill_normal_dev(float* dst_dev, int64_t n, float std, uint64_t seed, uint64_t offset,
                              const int32_t* counter_dev, void* stream);
const int32_t* rlrep_steps_dev(rlrep_agent* agent);
/* The four optimizer groups' device records: 22 32-bit words each -- {int32 step; float lr, beta1, beta2, eps, tau; 8 derived floats
 * recomputed on every step; 8 words of running beta^step powers (double) with the (step, betas) they belong to}.  A checkpoint restores `step` only and keeps the constructor's hyper-parameters (rlrep_amd/agent/sac). */
const void* rlrep_group_cfg_dev(rlrep_agent* agent);
#define RLREP_GROUP_CFG_WORDS 22

/* ---- launch-saving forms of the sampling calls (graph-replayed train()) -------------------------
 * rlrep_train_prologue == rlrep_begin_train + rlrep_fill_indices_dev(idx_pool) + rlrep_fill_normal_dev(eps_pool) +
 * rlrep_replay_sample(slot 0, ring, idx_pool[0:batch]) in ONE launch (same generator streams, same results): the
 * gather recomputes its indices from the counter-based generator instead of waiting for the pool.  The following
 * rlrep_replay_sample(agent, 0, ring, idx_pool, batch, ..) is recognised and skipped.
 * rlrep_prefetch_batch arms the gather of the NEXT minibatch (slot 0) to ride in the next optimizer launch
 * (rlrep_*_apply / *_step): by then the current step has finished reading the slot.  The matching
 * rlrep_replay_sample(agent, 0, ring, idx, batch, ..) is then skipped.  Returns 1 if armed, 0 if not (batch size change). */
int32_t rlrep_train_prologue(rlrep_agent* agent, const float* ring_dev, const int32_t* size_dev, int32_t* idx_pool_dev, int64_t n_idx,
                             float* eps_pool_dev, int64_t n_eps, uint64_t seed, uint64_t idx_offset, uint64_t eps_offset,
                             int32_t batch, void* stream);
int32_t rlrep_prefetch_batch(rlrep_agent* agent, const float* ring_dev, const int32_t* idx_dev, int32_t batch);
/* the same for minibatch slot `slot` (1: spedersac's second, "random" minibatch, agent/spedersac/spedersac_agent.py:181-186): both gathers of the
 * next feature step ride in this step's optimizer launch */
int32_t rlrep_prefetch_batch_slot(rlrep_agent* agent, int32_t slot, const float* ring_dev, const int32_t* idx_dev, int32_t batch);

/* ---- step programs ------------------------------------------------------------------------ */
/* eps pointers: caller-provided standard-normal noise (parity runs inject the oracle's tensors).
 *   vlsac feature: eps[B,F];  diffsr feature: noise_idx int32[B] + eps[B,S] (already scaled by sigma);
 *   critic/actor: eps[B,A].                                                                        */
int32_t rlrep_feature_step(rlrep_agent* agent, const float* eps_dev, const int32_t* noise_idx_dev, void* stream);
int32_t rlrep_critic_step(rlrep_agent* agent, const float* eps_dev, void* stream);
int32_t rlrep_actor_alpha_step(rlrep_agent* agent, const float* eps_dev, void* stream);
/* Optional launch saving for the sequence critic step -> actor step ON THE SAME BATCH (what train() does,
 * sac_agent.py:180-187): the forward half of update_actor_and_alpha (policy on s, features of (s, a_pi)) reads
 * nothing critic_step writes, so the library can run it inside the critic step's launches.  Call
 * rlrep_prefetch_policy(agent, eps_actor) right before rlrep_critic_step / _backward with the noise the actor step
 * will be given; the following rlrep_actor_alpha_step / _backward called with the SAME eps pointer then resumes after
 * its forward half.  Returns 1 if armed, 0 if this agent / shape has no such variant (then nothing changes), < 0 on
 * error.  Any new batch or feature step in between disarms it; results are identical either way. */
int32_t rlrep_prefetch_policy(rlrep_agent* agent, const float* eps_actor_dev);
/* One step earlier, for agents whose critic / actor steps reuse the minibatch of the LAST feature step (vlsac: train()
 * samples once per feature iteration and keeps the last batch, vlsac_agent.py:246-262): armed before that feature step,
 * BOTH policy forwards (on s' with eps_critic for the critic step's TD target, on s with eps_actor for the actor step)
 * ride in its first launches, and the critic step called with the same eps_critic pointer starts with the three
 * f_target forwards side by side.  Returns 1 if armed, 0 if the agent has no such variant. */
int32_t rlrep_prefetch_policy_early(rlrep_agent* agent, const float* eps_critic_dev, const float* eps_actor_dev);
/* Polyak critic -> critic_target iff (steps % target_update_period == 0), steps kept on the device. */
int32_t rlrep_update_target(rlrep_agent* agent, void* stream);
/* steps += 1 (device counter; graph-replay safe).  Also opens a train() bracket that rlrep_update_target closes: inside
 * it the Polyak critic -> critic_target (same tau, same steps % period gate) is performed by the critic step's Adam
 * launch and rlrep_update_target only closes the bracket -- nothing in between reads critic_target (train():
 * sac_agent.py:169-190).  Without a preceding rlrep_begin_train every entry point does exactly what its reference
 * method does. */
int32_t rlrep_begin_train(rlrep_agent* agent, void* stream);

/* Split entry points for data-parallel training: backward part writes the group's gradient arena, apply part runs
 * Adam / Polyak / temperature update; rlrep_*_step == backward immediately followed by apply.  With hyper.world_size > 1 the
 * gradient arena is COMPLETE after the backward part: the caller all-reduces grad_dev[group_offset .. +group_floats (+tail)]
 * between the two -- or has attached a comm (rlrep_comm_attach), in which case the apply part sums the ranks' arenas itself and
 * the caller issues nothing.  EXCEPTION, hyper.world_size == 1 only: gradients that arrive as split-K partials (vlsac critic
 * l1 / l4, spedersac phi / mu weight gradients) are summed BY THE APPLY PART's optimizer launch, which files the sums in grad_dev
 * too; between backward and apply those ranges of grad_dev hold the previous step's values.  A single-rank caller that inspects
 * or clips gradients between the two parts sets RLREP_DISABLE=fold_ncdw,fold_dwfin (the finishing launches come back). */
int32_t rlrep_feature_backward(rlrep_agent* agent, const float* eps_dev, const int32_t* noise_idx_dev, void* stream);
int32_t rlrep_feature_apply(rlrep_agent* agent, void* stream);
int32_t rlrep_critic_backward(rlrep_agent* agent, const float* eps_dev, void* stream);
int32_t rlrep_critic_apply(rlrep_agent* agent, void* stream);
int32_t rlrep_actor_backward(rlrep_agent* agent, const float* eps_dev, void* stream);
int32_t rlrep_actor_apply(rlrep_agent* agent, void* stream);

/* Batch-coupled representation losses under data parallelism (ctrlsac's in-batch negatives, spedersac's second
 * moment) need a collective INSIDE the feature backward.  The feature backward is therefore cut into
 * rlrep_feature_exchange_count()+1 parts; after part k the caller performs exchange k on a library buffer:
 *   kind 1: all-gather  -- every rank contributes `count` floats at ptr + local_off (in place, rank-major)
 *   kind 2: all-reduce SUM of `count` floats at ptr.
 *   kind 3: `count` floats at ptr = the gradient arena at float offset local_off are FINAL (diffsrsac: the nabla-mu head's weight + bias
 *           gradient, 99 % of that group's bytes, taken first): their all-reduce SUM may be issued now, asynchronously, and has to be complete --
 *           like that of the rest of the group, which the caller reduces after the last part -- before rlrep_feature_apply.
 * With world_size == 1 there are no exchanges and rlrep_feature_backward runs everything. */
int32_t rlrep_feature_exchange_count(rlrep_agent* agent);
int32_t rlrep_feature_exchange(rlrep_agent* agent, int32_t k, int32_t* kind, float** ptr_dev, int64_t* count, int64_t* local_off);
int32_t rlrep_feature_backward_part(rlrep_agent* agent, int32_t part, const float* eps_dev, const int32_t* noise_idx_dev, void* stream);

/* ---- deferred critic / actor steps (vlsac) ---------------------------------------------------
 * The feature steps of train(t+1) read nothing that the critic and actor steps of train(t) write, and the critic / actor steps
 * read only f_target, the last minibatch, their policy noise and the step counter from the feature side.  rlrep_defer_snapshot
 * copies exactly those (ONE launch) after the last feature step of train(t); rlrep_deferred_critic_actor then runs critic_step,
 * update_actor_and_alpha and the (period-gated) critic-target Polyak of train(t) against the snapshot, so a caller may issue it
 * on a second stream / graph branch next to train(t+1)'s feature steps.  Same arithmetic and the same sequence of updates
 * per parameter as the sequential entry points (tests/test_hip_parity.py::test_deferred_pipeline_is_equivalent); the caller
 * must order: snapshot(t) into set s after feature steps(t) AND after the deferred pair that last used set s; deferred(t) after
 * snapshot(t) and after deferred(t-1); anything that reads the critic / actor (select_action, checkpoints, metrics of those steps)
 * after deferred(t).  There are rlrep_defer_supported()
 * sets (3), so with set = t % 3 a snapshot only waits for the pair of train(t-3): the DEVICE needs two, the third keeps a HOST that waits for
 * that older pair before launching the next feature chain a full call ahead of the device. */
int32_t rlrep_defer_supported(rlrep_agent* agent);          /* number of snapshot sets (3) or 0 */
int32_t rlrep_defer_snapshot(rlrep_agent* agent, int32_t set, const float* eps_critic_dev, const float* eps_actor_dev, void* stream);
/* Folded form: called BEFORE the last feature step of train(t), rlrep_defer_arm makes that step's optimizer launch (rlrep_feature_apply)
 * write snapshot set `set` as well -- the minibatch slices, noise and step counter by extra blocks, the f_target block by the lanes that
 * produce its new values -- and the rlrep_defer_snapshot that follows with the same arguments launches nothing (one dependent launch
 * less on the chain that bounds vlsac's train(): agent/vlsac/vlsac_agent.py:245-273 has no counterpart, it is a scheduling device).
 * The caller's ordering duties move accordingly: the set must be free when that LAST feature step is issued.  Returns 1 if armed, 0 if
 * the agent / configuration has no folded form (then rlrep_defer_snapshot copies as before). */
int32_t rlrep_defer_arm(rlrep_agent* agent, int32_t set, const float* eps_critic_dev, const float* eps_actor_dev);
int32_t rlrep_deferred_critic_actor(rlrep_agent* agent, int32_t set, void* stream);
/* The same in four parts for data-parallel callers (all-reduce of the critic / actor gradient slices after parts 0 and 2):
 * 0 critic backward, 1 critic apply (+ period-gated critic-target Polyak), 2 actor backward, 3 actor + temperature apply. */
int32_t rlrep_deferred_part(rlrep_agent* agent, int32_t set, int32_t part, void* stream);
/* Host-only: closes a rlrep_begin_train / rlrep_train_prologue bracket without launching anything (the critic-target update of a
 * deferred train() runs inside rlrep_deferred_critic_actor). */
int32_t rlrep_end_train(rlrep_agent* agent);

/* ctrlsac: frozen_phi, frozen_phi_target <- phi (ctrlsac_agent.py:344-346). No-op for other agents. */
int32_t rlrep_sync_frozen(rlrep_agent* agent, void* stream);

/* ---- inference ---------------------------------------------------------------------------- */
/* action[n,A] = tanh(mu + eps*std) (eps_dev != NULL) or tanh(mu) (NULL), clamped to [lo,hi]. */
int32_t rlrep_actor_forward(rlrep_agent* agent, const float* obs_dev, int32_t n, const float* eps_dev,
                            float lo, float hi, float* action_dev, void* stream);

/* SACAgent.select_action for ONE observation (sac_agent.py:89-96; main.py:126 calls it once per environment step) in ONE launch: the three
 * actor layers, the tanh-Gaussian head and -- explore != 0 -- the standard-normal draw rlrep_fill_normal(eps[A], 1, seed, offset) would give,
 * action = clamp(tanh(mu + eps * std), lo, hi) (explore == 0: tanh(mu)).  obs[S] / action[A]: device pointers, or (the *_on_host flags)
 * pinned host buffers, which the kernel then reads / writes in place: no copy on either side, the caller synchronises `stream` and reads. */
int32_t rlrep_select_action(rlrep_agent* agent, const float* obs, int32_t obs_on_host, int32_t explore, uint64_t seed, uint64_t offset,
                            float lo, float hi, float* action, int32_t action_on_host, void* stream);

/* ---- metrics ------------------------------------------------------------------------------ */
/* device float array of n_metrics slots, valid after the stream has passed the producing step */
const float* rlrep_metrics_dev(rlrep_agent* agent);

/* Profiling hook: launch stage `stage` of step program `program` once (0 feature_bwd, 1 feature_apply,
 * 2 critic_bwd, 3 critic_apply, 4 actor_bwd, 5 actor_apply, 6 update_target); its inputs are whatever the
 * previous full step left in the workspace.  rlrep_stage_count/_name enumerate the stages. */
/* ---- data-parallel exchanges inside the launches (csrc/comm.hip, csrc/dp_pull.h) ---------------------------
 * The reference is one process: no counterpart.  The gradient exchange belongs between its `loss.backward()` and `optimizer.step()` pairs
 * (agent/vlsac/vlsac_agent.py:153-154, 183-184, 229-230; ctrlsac_agent.py:243-244; spedersac_agent.py:211-212; diffsrsac_agent.py:311-314), the
 * batch-coupled ones inside the feature losses (spedersac_agent.py:197-205: Phibar and v over the global batch; ctrlsac_agent.py:226-231: in-batch
 * negatives over the global batch).
 * A comm owns ONE block of device memory, [arena_floats | scratch_floats | reduced region (world >= 3) | flag words], exported over hipIpc and
 * mapped by every peer: the caller passes rlrep_comm_arena() as rlrep_arenas.grad_dev, so every rank's gradients lie where every peer can read
 * them.  Lifecycle, identical on every rank: create -> handle -> (exchange the handles, world x rlrep_comm_handle_bytes() in rank order, by any
 * means: torch.distributed.all_gather_object) -> connect -> rlrep_agent_create(..., grad_dev = rlrep_comm_arena()) -> attach.  (Several ranks in
 * ONE process -- the loopback form of tools/exp/dp_loopback.py and tests -- skip handle / connect and call rlrep_comm_connect_local.)
 * After rlrep_comm_attach the optimizer launch of every attached group (rlrep_*_apply, the fused *_step entry points, the deferred chain)
 * waits -- bounded -- for the peers' gradients of that step, sums every rank's gradient of its elements IN RANK ORDER (bit-identical on every
 * rank; one-shot pull, or reduce-scatter + all-gather inside the same launch for slices of at least two_shot_floats when world >= 3) and does
 * not end before every peer has read this rank's: ZERO launches per all-reduce, nothing on the host changes between calls (hipGraph-capturable),
 * and the caller issues NO collective for those groups.  Groups above max_floats stay with the caller's all-reduce between backward and apply.
 * With scratch_floats >= rlrep_layout_info.exchange_floats the batch-coupled exchanges move into the step programs too (spedersac: pushed by the
 * column-sum launches, summed by their consumers, no launch; ctrlsac: one pull launch each) and rlrep_feature_exchange_count() drops to 0.
 * rlrep_comm_allreduce / _allgather are the same exchanges as stand-alone launches (probe / tests).  A wait that runs out (rlrep_comm_set_timeout;
 * default 120 s: a watchdog) sets a bit of the error word; the launch applies NOTHING and drains: rlrep_comm_status reads that word from mapped
 * host memory WITHOUT synchronising. */
typedef struct rlrep_comm rlrep_comm;
int32_t rlrep_comm_create(int32_t rank, int32_t world, int64_t arena_floats, int64_t scratch_floats, rlrep_comm** out);
float* rlrep_comm_arena(rlrep_comm* comm);
float* rlrep_comm_scratch(rlrep_comm* comm);
int32_t rlrep_comm_handle_bytes(void);
int32_t rlrep_comm_handle(rlrep_comm* comm, void* out, int32_t cap);
int32_t rlrep_comm_connect(rlrep_comm* comm, const void* handles);
int32_t rlrep_comm_connect_local(rlrep_comm* comm, rlrep_comm* const* peers);
int32_t rlrep_comm_set_timeout(rlrep_comm* comm, int64_t timeout_us);
int32_t rlrep_comm_attach(rlrep_agent* agent, rlrep_comm* comm, int64_t max_floats, int64_t two_shot_floats, int32_t* attached_mask);
int32_t rlrep_comm_allreduce(rlrep_comm* comm, int64_t block_offset_floats, int64_t n, float* out_dev, int32_t mode, int64_t timeout_us, void* stream);
int32_t rlrep_comm_allgather(rlrep_comm* comm, int64_t block_offset_floats, int64_t n_per_rank, void* stream);
int32_t rlrep_comm_probe_fill(rlrep_comm* comm, int64_t block_offset_floats, int64_t n, int32_t round, void* stream);
float rlrep_comm_probe_value(int32_t rank, int32_t round, int64_t i);
int32_t rlrep_comm_probe_slots(rlrep_comm* comm, int64_t n, int32_t round, float* out_dev, int64_t timeout_us, void* stream);
int32_t rlrep_comm_status(rlrep_comm* comm, uint32_t* late_mask, int32_t clear);
int32_t rlrep_comm_fine_grained(rlrep_comm* comm);
/* debug / measurement only: mark every peer as arrived for the next `ahead` epochs of `channel` in this rank's flags (one stream then plays several ranks
 * in turn; 1 << 30: one attached replica runs alone and pays the protocol, not the waiting -- tools/exp/dp_loopback.py) */
int32_t rlrep_comm_debug_preset(rlrep_comm* comm, int32_t channel, int32_t ahead);
void rlrep_comm_destroy(rlrep_comm* comm);

/* vlsac noise-critic weight images (bf16x3 shadows of critic.l1 / l4 and their targets; no reference counterpart: nn.Linear has no such
 * copies).  Default: every critic step regenerates them with one launch.  Between rlrep_images_managed(agent, 1) and (agent, 0) the step
 * entry points skip that launch (inside a rlrep_begin_train .. rlrep_update_target bracket the critic group's optimizer launch keeps live and
 * target images current) and the caller runs rlrep_refresh_images after anything else wrote critic / critic_target.  images_managed returns
 * 1 if the agent keeps images, 0 if there is nothing to manage. */
int32_t rlrep_images_managed(rlrep_agent* agent, int32_t on);
int32_t rlrep_refresh_images(rlrep_agent* agent, void* stream);

/* Chained feature steps (vlsac, one GPU; no reference counterpart: a launch-saving form of `for _ in range(extra_feature_steps + 1):
 * feature_step(...)`, vlsac_agent.py:250-256).  Called after rlrep_prefetch_batch has armed the NEXT step's minibatch and before this step's
 * rlrep_feature_step: this step's weight-gradient launch then applies the optimizer to the two first layers in its epilogues, and its optimizer
 * launch skips them and carries the next step's first launch (encoder.l1 / f.l1, rows read from the ring through the index pool) as leading
 * tiles; the next rlrep_feature_backward starts at its second launch.  Same arithmetic in the same order per parameter.  1: armed, 0: not
 * available (nothing changes).  The next call on the agent's feature path must be that next step (plain, no rlrep_prefetch_policy_early). */
int32_t rlrep_feature_chain_next(rlrep_agent* agent);

int32_t rlrep_stage_count(rlrep_agent* agent, int32_t program);
const char* rlrep_stage_name(rlrep_agent* agent, int32_t program, int32_t stage);
int32_t rlrep_run_stage(rlrep_agent* agent, int32_t program, int32_t stage, void* stream);
/* What a stage launches (bench.py groups the stages of a train() into kernel families with it, and prices them): the kernel family
 * (RLREP_ENGINE_*), the ALGORITHMIC flops of its products (2 * rows * columns * inner length per product; 0 for non-GEMM stages) and the
 * algorithmic bytes (every operand and result of a product once, 4 bytes per element; optimizer stages: 28 bytes per parameter + 12 per
 * Polyak-averaged target element).  No reference counterpart (the reference has no kernels of its own). */
#define RLREP_ENGINE_OTHER 0        /* elementwise / loss / gather / copy kernels */
#define RLREP_ENGINE_GEMM16 1       /* gemm16_kernel / gemm16_duo_kernel: the 16-row fp32-MFMA tile engine */
#define RLREP_ENGINE_HEADS_VAE 2    /* heads_vae_kernel (vlsac: both Gaussian heads + sample / KL on a 16 x 16 tile) */
#define RLREP_ENGINE_LDS64 3        /* gemm_lds_kernel<64,...> (+ its split-K finishing blocks) */
#define RLREP_ENGINE_LDS128 4       /* gemm_lds_kernel<128,...> on fp32 MFMA */
#define RLREP_ENGINE_X3 5           /* gemm_x3_kernel: the 128-wide tile on the bf16 pipe, exact three-way split */
#define RLREP_ENGINE_NOISE_CRITIC 6 /* nc_fwd / nc_dx / nc_dw kernels (vlsac noise critic, bf16x3) */
#define RLREP_ENGINE_OPTIMIZER 7    /* adam_kernel (Adam + Polyak + metric finalisation + riders) */
#define RLREP_ENGINE_SCORE 8        /* diffsr score kernels (HBM-bound pass over [B, F*S]) */
int32_t rlrep_stage_info(rlrep_agent* agent, int32_t program, int32_t stage, int32_t* engine, double* flops, double* bytes);

/* Unit-test hook: ONE product on the path's GEMM engines with caller buffers (tests/test_gemm_engines.py checks both
 * engines against NumPy on every operand layout, epilogue, ragged edge and split-K plan).
 *   C[R,Cn] = epilogue( sum_k opA(r,k) * opB(c,k) )
 *   la / lb: 0 = row-major operand [rows, K] (ld = row stride), 1 = k-major operand [K, rows]
 *   epi: 0 forward: act(acc + bias)        (act: 0 none, 1 relu, 2 elu, 3 sin (+ pre-activation to out2), 4 tanh)
 *        1 dX:      acc * act'(aux), flags & 1: C += ...
 *        3 dW:      acc, flags & 1: C += ..., flags & 2: out2[r] = sum_k opA(r,k) (bias gradient)
 *   engine: 0 = 16-row tile engine (gemm16), 1 = LDS-tiled engine (gemm_lds), 2 = its 128-wide tile on the bf16 pipe
 *   (bf16x3: three-way operand split, six MFMAs, fp32 accuracy); bt (0 auto, 64, 128; with engine 2 also 256 = the persistent
 *   256 x 128 tile, which has no sin / tanh epilogue) and splits
 *   (0 auto) override the LDS engine's plan; workspace holds its split-K slabs (splits*R*(Cn+1) floats), which a finishing launch adds
 *   in split order -- or, with flags & 4 on the 64-wide bf16x3 tile (engine 2, bt 64), the last split workgroup of every output tile
 *   (a ticket word per tile behind the slabs: + ceil(R/64)*ceil(Cn/64) floats of workspace), bit-identically and without a second launch.
 * Returns 0, or RLREP_ERR_ARG when the engine cannot run the shape (alignment rules in gemm_lds.hip). */
int32_t rlrep_gemm(int32_t engine, int32_t la, int32_t lb, const float* a_dev, int32_t lda, const float* b_dev, int32_t ldb,
                   float* c_dev, int32_t ldc, int32_t rows, int32_t cols, int32_t inner, int32_t epi, int32_t act, int32_t flags,
                   const float* bias_dev, const float* aux_dev, int32_t ldaux, float* out2_dev, int32_t bt, int32_t splits,
                   float* workspace_dev, int64_t workspace_floats, void* stream);

/* Host-only (no GPU call): the GEMM engine the program builder picks for a product of these dimensions and layouts --
 * *engine 0 = 16-row tile engine, 1 = LDS-tiled fp32-MFMA, 2 = LDS-tiled bf16x3 -- with its tile edge (64 / 128; 256 = the persistent
 * 256 x 128 tile of engine 2), split-K plan and which sides fall back to 4-byte accesses (bit 0: A, bit 1: B, bit 2: C). */
int32_t rlrep_gemm_plan(int32_t la, int32_t lb, int32_t rows, int32_t cols, int32_t inner, int32_t lda, int32_t ldb, int32_t ldc,
                        int32_t* engine, int32_t* tile, int32_t* splits, int32_t* kchunk, int32_t* scalar_sides);

/* Host-only: the engine the builder picks for the vlsac noise critic's first layer (vlsac_agent.py:44-63) with `heads` heads in one
 * launch -- *engine 0 = fp32 MFMA, 1 = bf16x3 (needs F % 32 == 0 and 16-byte rows; RLREP_NC_X3=0 turns it off) -- and its
 * workgroup tile: *rows batch rows x *cols hidden units. */
int32_t rlrep_nc_fwd_plan(int32_t heads, int32_t batch, int32_t feature_dim, int32_t hidden_dim,
                          int32_t* engine, int32_t* rows, int32_t* cols);

/* Chains (csrc/xchain.hip): consecutive row-local stages of a step program -- the forward and dX launches of the reference's
 * feature_step / critic_step / update_actor_and_alpha (agent/vlsac/vlsac_agent.py:126-237 and siblings) -- run as ONE persistent launch
 * whose workgroups synchronise per XCD.  The launch checks its own assumptions on the device: *status receives the error word
 * (0 = clean; bit 0: a wait inside a chain timed out; bit 1: the workgroups of one group did not share an XCD, i.e. the hand-offs
 * were not guaranteed to be seen).  SYNCHRONISES `stream`.  Returns RLREP_ERR_STATE (and sets the error text) when the word is not 0:
 * results of the affected steps are invalid; RLREP_XCHAIN=0 selects one launch per stage. */
int32_t rlrep_chain_status(rlrep_agent* agent, uint32_t* status, void* stream);

/* bit 0: this library was built with RLREP_BUILD_EXPERIMENTS=1 (the opt-in engines that were measured and not adopted: RLREP_ROWPROG,
 * RLREP_XCHAIN, RLREP_FUSE_L1, superseded noise-critic forward kernels).  0: they are not compiled in and their switches are ignored. */
int32_t rlrep_build_flags(void);

/* Metric history.  While on (rlrep_history(agent, 1)), the LAST optimizer launch of every train() -- the actor's, reference
 * agent/sac/sac_agent.py:157-166 -- also appends the 16 metric slots to a device ring of *records records of *record_floats floats: record
 * n % *records holds the metrics of the n-th such launch since the agent was created and n itself (int32 bit pattern) in word *tag_word; *seq is
 * the device counter n.  The reference returns the metrics of a train() as host floats (a device sync per value); a caller
 * that replays train() as a hipGraph reads record n when -- and if -- somebody looks at the returned dict, instead of paying a snapshot
 * launch per call.  A record is overwritten *records calls later: the tag says whether it still is the one asked for.  The switch is read
 * when a step is LAUNCHED (or captured), not on the device.  Not kept by the fused-optimizer variant (RLREP_FUSE_ADAM). */
int32_t rlrep_history(rlrep_agent* agent, int32_t on);
int32_t rlrep_history_dev(rlrep_agent* agent, const float** ring, const int32_t** seq, int32_t* records, int32_t* record_floats, int32_t* tag_word);

/* Diagnostics: one single-thread launch on `stream` that appends (100 MHz device wall clock << 8 | tag) to a ring of `cap` 64-bit words
 * after a running counter in ring[0] (ring: cap + 1 words of device memory, zeroed by the caller).  Captured between the launches of a
 * train() graph it dates the chains on the DEVICE (tools/exp/chain_stamps.py); it is not part of any step. */
int32_t rlrep_debug_stamp(int64_t* ring_dev, int32_t cap, int32_t tag, void* stream);

/* Number of kernel launches the last step program issued (for the latency model in DESIGN.md). */
int32_t rlrep_last_launch_count(rlrep_agent* agent);
/* process-wide number of kernel launches the library has issued so far (a captured train()'s launch count = the difference around its capture) */
int64_t rlrep_launch_counter(void);
/* process-wide launches of the 16-row tile engine per FRONT END so far: out4[0] = gemm16_fast_kernel, [1] = gemm16_fast4_kernel, [2] =
 * gemm16_fastpre_kernel (operand loads issued from preloaded scalars; a launch qualifies by its shapes and by all its operands lying within
 * 16 GiB of one base -- the caller's arenas should be slices of ONE allocation), [3] = the record front end (gemm16_kernel /
 * gemm16_duo_kernel).  Counted when a launch is issued or captured, like rlrep_launch_counter.  The layers these launches compute:
 * reference networks/vae.py:40-57,83-85,112-117 and the MLPs of agent/<alg>/<alg>_agent.py. */
int32_t rlrep_front_end_counts(int64_t* out4);

#ifdef __cplusplus
}
#endif
#endif /* RLREP_H */

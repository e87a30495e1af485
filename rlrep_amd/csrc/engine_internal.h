// Internal definitions shared by engine.hip and agents2.hip (agent state, program builder).
#pragma once
#include "engine.h"
#include <deque>

struct Slot { float *XE, *XF, *XF2, *XFpi, *R, *D; bool filled = false; };

struct Exchange { int after_stage; int kind; float* ptr; int64_t count; int64_t local_off; };

struct rlrep_agent {
    rlrep_dims d; rlrep_hyper h; rlrep_arenas a; Layout L; Workspace ws;
    int B = 0;
    int* steps = nullptr; GroupCfg* adam_step = nullptr; float* metrics = nullptr; float* obs_in = nullptr; float* act_out = nullptr;
    Slot slot[2];
    // per-call dynamic inputs, read by the by-value parameter blocks at launch time
    const float* cur_eps = nullptr; const int* cur_idx = nullptr;
    Program feat_bwd, feat_apply, critic_bwd, critic_apply, actor_bwd, actor_apply, upd_target, infer, sync_prog;
    // The forward half of the actor step (policy on s, features of (s, a_pi)) depends on nothing the critic step
    // changes, so its GEMMs can ride along in the critic step's launches (critic_bwd_h) and the actor step then
    // resumes at stage actor_resume.  Armed per train() by rlrep_prefetch_policy; see there for the protocol.
    Program critic_bwd_h; int actor_resume = 0;
    // Inside a rlrep_begin_train .. rlrep_update_target bracket the Polyak critic -> critic_target is folded into the
    // critic's Adam launch (critic_apply_f): nothing between the two reads critic_target, and the update_target launch
    // (~2 us + a launch boundary) disappears.  Outside a bracket every entry point does exactly what its name says.
    Program critic_apply_f; bool in_train = false, target_done = false;
    // rlrep_prefetch_batch: slot-0 gather that the next optimizer launch performs; pf_done: the matching
    // rlrep_replay_sample(slot 0, pf_ring, pf_idx, B) is then a no-op
    SlotFill pf_fill; bool pf_armed = false, pf_done = false; const float* pf_ring = nullptr; const int* pf_idx = nullptr;
    // the same for slot 1 (spedersac's second, "random" minibatch: rlrep_prefetch_batch_slot)
    SlotFill pf2_fill; bool pf2_armed = false, pf2_done = false; const float* pf2_ring = nullptr; const int* pf2_idx = nullptr;
    int* ticket = nullptr;
    const float* cur_eps2 = nullptr; const float* hoist_req = nullptr; const float* pi_ready = nullptr;
    // One step further: when the critic / actor steps reuse the LAST feature step's minibatch (vlsac), BOTH policy
    // forwards (on s' for the critic step, on s for the actor step) ride in that feature step's first three launches
    // (feat_bwd_h) and the critic step starts with all three f_target forwards in the same launches (critic_bwd_h2).
    Program feat_bwd_h, critic_bwd_h2;
    // CHAINED feature steps (rlrep_feature_chain_next; vlsac, single GPU): when the next feature step follows directly on the minibatch armed by
    // rlrep_prefetch_batch, this step's weight-gradient launch runs the optimizer of the two FIRST layers in its epilogues (feat_bwd_m: the
    // plain program with FLAG_ADAM on those tasks), and its optimizer launch (feat_apply_m) skips them and carries the next step's first
    // launch -- encoder.l1 / f.l1 on rows read straight from the ring -- as leading tiles; the next step then starts at its second stage.
    Program feat_bwd_m, feat_apply_m; bool chain_next = false, chain_bwd_done = false, l1_done = false;
    const float* cur_eps3 = nullptr; const float* early_crit = nullptr; const float* early_act = nullptr;   // armed request
    const float* early_ready_crit = nullptr; const float* early_ready_act = nullptr;                        // done by feat_bwd_h
    // DEFERRED critic / actor steps (vlsac): the feature steps of train(t+1) read nothing the critic and actor steps of train(t)
    // write (and vice versa, once f_target, the minibatch, the policy noise and the step counter are snapshotted), so a caller may
    // run [critic, actor of t] and [feature steps of t+1] as two concurrent branches.  rlrep_defer_snapshot takes the snapshot
    // (one launch) after the last feature step; rlrep_deferred_critic_actor runs the two steps against it.  Same arithmetic,
    // same order of updates per parameter; only the overlap changes.
    // SEVERAL snapshot sets (round robin): train(t) uses set t % NSETS, so the snapshot of train(t+1) never has to wait for the critic /
    // actor pair of train(t) -- only for that of train(t+1-NSETS).  Two sets are enough for the DEVICE; the third is for the HOST, which waits
    // for that older pair before it may launch the next feature chain: with two sets that wait ends ~one graph-launch latency before the
    // running feature chain does, and the feature queue idles between two train() calls (tools/exp/trace_gaps.py).
    struct DeferSet {
        Slot slot; float* block = nullptr; float* eps = nullptr; int* steps = nullptr; CopySegs segs;
        long long block_off = 0, block_n = 0; int block_which = -1;        // the block as the feature optimizer launch sees it (-1: no folded snapshot)
        Program critic_bwd, critic_apply, actor_bwd; int actor_resume = 0; bool valid = false;
    };
    static constexpr int NSETS = 3;             // rlrep_defer_supported() reports it; a caller rotating over all of them only ever waits for the pair of train(t-3)
    DeferSet dset[NSETS]; int dcur = 0;             // dcur: the set the programs under construction belong to
    // while the deferred programs are built, tensors named ov_prefix* resolve into the snapshot block ov_base (same internal layout
    // as the block that starts at ov_first): vlsac f_target.*, ctrlsac phi.*, spedersac phi.trunk.*
    const float* ov_base = nullptr; std::string ov_prefix, ov_first;
    int infer_n = 0; float infer_lo = -1.f, infer_hi = 1.f; size_t prog_end = 0;
    std::vector<Exchange> feat_cuts;      // collectives inside the feature backward (world_size > 1 only)
    int last_launches = 0;
    size_t ws_static = 0;     // workspace bytes used by batch-independent state
    // Transposed shadows of the weight matrices that row programs read in the forward direction (rowprog.hip): per optimizer group a device
    // table + the shadow storage (workspace, batch independent).  Kept current by the group's Adam launch, regenerated from the
    // parameters at the head of every train() (rlrep_train_prologue / rlrep_begin_train) and before an eager step outside a train().
    const ShadowEnt* sh_dev[4] = {nullptr, nullptr, nullptr, nullptr}; int nsh[4] = {0, 0, 0, 0}, sh_tiles[4] = {0, 0, 0, 0};
    std::map<std::string, float*> shadow_of;
    float* PT(const std::string& n) const { auto it = shadow_of.find(n); return it == shadow_of.end() ? nullptr : it->second; }
    bool has_shadows() const { return nsh[0] + nsh[1] + nsh[2] + nsh[3] > 0; }
    // bf16x3 images of the noise critic's first-layer weights (vlsac; ShadowEnt kind 1): live critic.l1 / l4 are kept current by the critic
    // group's Adam launch (sh_dev[1]); ALL of x3_refresh (live and target, absolute sources) is regenerated by a launch at the head of the
    // critic step, which is what serves the target copies and a caller who wrote parameters behind the library's back
    // folded snapshot (rlrep_defer_arm): the next feature optimizer launch also writes snapshot set snap_set; snap_done = the set it wrote
    bool snap_armed = false; int snap_set = -1, snap_done = -1; const float* snap_ec = nullptr; const float* snap_ea = nullptr;
    const ShadowEnt* x3_refresh = nullptr; int x3_n = 0, x3_tiles = 0;
    bool images_managed = false;          // rlrep_images_managed: the step entry points leave the refresh launch to the caller
    std::map<std::string, const unsigned char*> x3_of;
    const unsigned char* W3(const std::string& n) const { auto it = x3_of.find(n); return it == x3_of.end() ? nullptr : it->second; }
    // cluster row programs (RLREP_ENABLE=rowprog=2): the launch epoch that tags their exchange granules (bumped by the feature Adam launch; by an
    // extra launch before an eager step outside a train())
    int* rp_epoch = nullptr;
    float* hist = nullptr; int* hist_seq = nullptr; bool hist_on = false;      // metric history ring (kparams.h FIN_HISTORY, rlrep_history)
    // xchain.hip: error word of the persistent chain launches (rlrep_chain_status) and the names of their stages (owned here: Stage::what is a
    // plain pointer)
    unsigned* xc_err = nullptr; std::deque<std::string> stage_names;
    // data parallel inside the optimizer launches (rlrep_comm_attach, dp_pull.h): the gradient arena is a block every peer has mapped; the
    // optimizer launch of an attached group sums its gradients over the ranks itself (channel = group)
    // the train() counter's mirror (word 2 of `steps`, read by the NEXT train prologue) is refreshed by the optimizer launches that carry
    // AdamTask::sync_steps (group 0; any group for sac).  A caller that runs rlrep_train_prologue and then no such launch (advisor r05: critic /
    // actor steps only) would draw the same Philox indices again: the next prologue then refreshes the mirror itself, with one extra launch.
    bool mirror_pending = false;
    DpPull dp_proto = DpPull(); bool dp_on[4] = {false, false, false, false}, dp_two[4] = {false, false, false, false};
    // ... and the batch-coupled exchanges of the feature step (spedersac Phibar / v: pushed slots, zero launches; ctrlsac mu(s') / dmu': one
    // pull launch each) when the comm carries exchange scratch (rlrep_layout_info.exchange_floats): xfold = the step programs were rebuilt
    // with them inside, feat_cuts is empty and a data-parallel train() is ONE uninterrupted sequence of launches
    bool xfold = false; float* xscratch[RL_DP_MAX_WORLD] = {nullptr}; long long xscratch_floats = 0, xarena_floats = 0;
    DpSlots slots(int channel, long long off_floats, int n) const {          // slot area [2][world][n] of `channel` at scratch offset off_floats
        DpSlots d; memset(&d, 0, sizeof(d));
        d.world = dp_proto.world; d.rank = dp_proto.rank; d.channel = channel; d.n = n; d.timeout = dp_proto.timeout; d.err = dp_proto.err;
        for (int q = 0; q < d.world; ++q) { d.slot[q] = xscratch[q] + off_floats; d.flags[q] = dp_proto.flags[q]; }
        return d;
    }

    float* overridden(const std::string& n) const {
        if (!ov_base || n.compare(0, ov_prefix.size(), ov_prefix) != 0) return nullptr;
        return const_cast<float*>(ov_base) + (L.get(n).off - L.get(ov_first).off);
    }
    float* P(const std::string& n) const { if (float* o = overridden(n)) return o; return a.param_dev ? a.param_dev + L.get(n).off : nullptr; }
    float* T(const std::string& n) const { if (float* o = overridden(n)) return o; return a.target_dev ? a.target_dev + L.get(n).off : nullptr; }
    float* G(const std::string& n) const { return a.grad_dev ? a.grad_dev + L.get(n).off : nullptr; }
    float* Gtail() const { return a.grad_dev ? a.grad_dev + L.cur[RLREP_ARENA_PARAM] : nullptr; }
    float inv_batch() const { return 1.0f / ((float)B * (float)(h.world_size > 0 ? h.world_size : 1)); }
};

// ------------------------------------------------------------------------------------------------
// builder helpers
// ------------------------------------------------------------------------------------------------
struct Builder {
    rlrep_agent* ag; Workspace& ws; bool dry;
    Builder(rlrep_agent* a) : ag(a), ws(a->ws), dry(a->ws.dry) {}

    template <class Tt> const Tt* upload(const std::vector<Tt>& v) {
        Tt* dev = (Tt*)ws.alloc(v.size() * sizeof(Tt));
        if (!dry && ws.ok()) {
            hipError_t e = hipMemcpy(dev, v.data(), v.size() * sizeof(Tt), hipMemcpyHostToDevice);
            if (e != hipSuccess) rl_set_error("table upload failed: %s", hipGetErrorString(e));
        }
        return dev;
    }
    Mat mat(int rows, int cols) { return Mat{ws.f((size_t)rows * cols), rows, cols, cols}; }

    // what rlrep_stage_info reports for the stage pushed last: kernel family, algorithmic flops and bytes of its products
    static void tag(Program& p, int engine, double flops, double bytes) { Stage& s = p.stages.back(); s.engine = engine; s.flops = flops; s.bytes = bytes; }
    static void tag_gemms(Program& p, int engine, const std::vector<GemmTask>& tasks) {
        double fl = 0.0, by = 0.0;
        for (const GemmTask& t : tasks) {
            fl += 2.0 * (double)t.R * (double)t.Cn * (double)t.K;
            by += 4.0 * ((double)t.R * t.K + (double)t.Cn * t.K + (double)t.R * t.Cn);
            if (t.flags & FLAG_PRE) { fl += 2.0 * (double)t.R * (double)t.K * (double)t.n0; by += 4.0 * ((double)t.R * t.n0 + (double)t.n0 * t.K); }    // the fused short product
        }
        tag(p, engine, fl, by);
    }

    // ---- GEMM task constructors -------------------------------------------------------------
    static GemmTask base() { GemmTask t; memset(&t, 0, sizeof(t)); t.scale = 1.f; return t; }
    // Y[B,N] = act(X[B,K] W[N,K]^T + b)
    static GemmTask fwd(const float* X, int ldx, int Bn, int K, const float* W, int ldw, const float* bias, int N,
                        float* Y, int ldy, int act, float* pre = nullptr, int ldpre = 0) {
        GemmTask t = base();
        t.A = X; t.lda = ldx; t.B = W; t.ldb = ldw; t.C = Y; t.ldc = ldy; t.bias = bias;
        t.R = Bn; t.Cn = N; t.K = K; t.epi = EPI_FWD; t.act = act; t.out2 = pre; t.ldout2 = ldpre;
        return t;
    }
    // dX[B,Kout] (=|+=) (G[B,N] W[N, Kout(+off)]) * act'(aux)
    static GemmTask dx(const float* Gm, int ldg, int Bn, int N, const float* W, int ldw, float* dX, int lddx, int Kout,
                       int act, const float* aux, int ldaux, int flags = 0) {
        GemmTask t = base();
        t.A = Gm; t.lda = ldg; t.B = W; t.ldb = ldw; t.C = dX; t.ldc = lddx; t.aux = aux; t.ldaux = ldaux;
        t.R = Bn; t.Cn = Kout; t.K = N; t.epi = EPI_DX; t.act = act; t.flags = flags;
        return t;
    }
    // gW[N,K] = G[M,N]^T X[M,K];  gb[N] = colsum G
    static GemmTask dw(const float* Gm, int ldg, int N, const float* X, int ldx, int K, int M, float* gW, int ldgw, float* gb) {
        GemmTask t = base();
        t.A = Gm; t.lda = ldg; t.B = X; t.ldb = ldx; t.C = gW; t.ldc = ldgw; t.out2 = gb;
        t.R = N; t.Cn = K; t.K = M; t.epi = EPI_DW; t.flags = gb ? FLAG_BIASGRAD : 0;
        return t;
    }

    // Route every task of a stage to the engine that suits its size: the LDS-tiled kernel (gemm_lds.hip; 128- or 64-wide
    // tiles, split-K for small outputs with a long inner dimension) for the large layers of ctrlsac / spedersac /
    // diffsrsac, the 16-row tile engine for everything else.  The routing and the split-K slabs depend on dimensions
    // only, so the dry sizing pass and the real pass allocate identically; a task whose pointers turn out not to be
    // 16-byte aligned simply stays on gemm16 and leaves its slab unused.
    void gemm(Program& p, int la, int lb, std::vector<GemmTask> tasks, const char* what) {
        std::vector<GemmTask> small, big128, big64, bigx3, bigx3s, bigx3w, bigx3q;
        for (auto& t : tasks) {
            int sp = 1, kc = 0, fl = 0;
            // dimensions decide the engine and the slab reservation (identical in the dry and the real pass) ...
            if (rl_gemm_lds_route(&t, la, lb, 0, &sp, &kc, &fl)) {
                // (a task routed to the slab-free 32 x 32 tile still reserves the slabs of its 64-wide plan: pointer alignment, or a neighbour in its
                //  stage, may send it back there)
                { int b0 = 0, k0 = 0; if (sp == 1) rl_gemm_lds_plan(&t, &b0, &sp, &k0); }
                // (the bias-gradient flag depends on a pointer that is null in the dry pass: reserve for every dW task)
                // (+ one ticket word per 64 x 64 output tile behind the slabs: gemm_x3s_kernel's in-launch finish, FLAG_FIN_INLINE; zeroed once, self-resetting)
                const size_t slab_floats = (size_t)sp * t.R * ((t.Cn + 3) & ~3), ticks = (size_t)((t.R + 63) / 64) * ((t.Cn + 63) / 64);
                float* slab = sp > 1 ? ws.f(slab_floats + ticks) : nullptr;
                if (slab && !dry && ws.ok()) (void)hipMemset(slab + slab_floats, 0, ticks * sizeof(int));
                float* bslab = (sp > 1 && t.epi == EPI_DW) ? ws.f((size_t)sp * t.R) : nullptr;
                // ... pointer alignment can only add scalar-access flags (and take bf16x3 away)
                const int code = rl_gemm_lds_route(&t, la, lb, dry ? 0 : rl_gemm_lds_ptr_flags(&t), &sp, &kc, &fl);
                t.splits = sp; t.kchunk = kc; t.slab = slab; t.bslab = bslab; t.flags |= fl;
                (code == 33 ? bigx3q : code == 257 ? bigx3w : code == 129 ? bigx3 : code == 65 ? bigx3s : code == 128 ? big128 : big64).push_back(t);
                continue;
            }
            small.push_back(t);
        }
        // one launch per tile width: if some 64-wide tasks of the stage cannot take the bf16x3 tile (scalar staging), all of them stay on fp32
        if (!big64.empty() && !bigx3s.empty()) { big64.insert(big64.end(), bigx3s.begin(), bigx3s.end()); bigx3s.clear(); }
        // (a stage stays ONE launch where it can: 32 x 32 tasks beside 64-wide ones of the same stage all take the 64-wide tile -- with their split plan)
        if (!bigx3q.empty() && (!bigx3s.empty() || !big64.empty())) {
            for (auto t : bigx3q) {
                int sp = 1, kc = 0, bt0 = 0;
                rl_gemm_lds_plan(&t, &bt0, &sp, &kc);
                t.splits = sp; t.kchunk = kc;
                (big64.empty() ? bigx3s : big64).push_back(t);
            }
            bigx3q.clear();
        }
        if (!bigx3q.empty() || !bigx3w.empty() || !bigx3.empty() || !big128.empty() || !big64.empty() || !bigx3s.empty()) chain_flush();
        if (!bigx3q.empty()) gemm_lds_stage(p, la, lb, 33, bigx3q, what);
        if (!bigx3w.empty()) gemm_lds_stage(p, la, lb, 257, bigx3w, what);
        if (!bigx3.empty()) gemm_lds_stage(p, la, lb, 129, bigx3, what);
        if (!big128.empty()) gemm_lds_stage(p, la, lb, 128, big128, what);
        if (!bigx3s.empty()) gemm_lds_stage(p, la, lb, 65, bigx3s, what);
        if (!big64.empty()) gemm_lds_stage(p, la, lb, 64, big64, what);
        if (!small.empty()) gemm_small(p, la, lb, small, what);
    }
    // fold_group >= 0 (set by an agent's builder around its weight-gradient stage): the split-K weight-gradient tasks built now leave the sum of
    // their partial slabs to that group's optimizer launch (AdamTask::Slab) -- no finishing blocks for them, and when no task of the stage
    // keeps any, no finishing launch.  Only when nothing reads the gradient between the two (one rank), the tensors start on multiples of four
    // floats inside the group and at most eight slabs ride in one launch.
    int fold_group = -1;
    bool fold_fin(GemmTask& t) {
        if (fold_group < 0 || dry || t.epi != EPI_DW || (t.flags & FLAG_ACCUM) || t.splits > 16 || !t.slab || !ag->a.grad_dev || ag->h.world_size > 1) return false;
        const bool bias = (t.flags & FLAG_BIASGRAD) && t.out2 && t.bslab;
        auto& gs = group_slabs[fold_group];
        if (gs.size() + (bias ? 2 : 1) > 8 || t.ldc != t.Cn) return false;
        const int64_t g0 = ag->L.group_off[fold_group], gn = ag->L.group_n[fold_group];
        if (gn & 3) return false;
        const int64_t offw = (t.C - ag->a.grad_dev) - g0, offb = bias ? (t.out2 - ag->a.grad_dev) - g0 : 0;
        const int64_t nw = (int64_t)t.R * t.Cn;
        if (offw < 0 || offw + nw > gn || (offw & 3) || (nw & 3)) return false;
        if (bias && (offb < 0 || offb + t.R > gn || (offb & 3) || (t.R & 3))) return false;
        const int ldpad = (t.Cn + 3) & ~3;
        AdamTask::Slab sw; memset(&sw, 0, sizeof(sw));
        sw.off = offw; sw.n = nw; sw.per = nw; sw.sstride = (long long)t.R * ldpad; sw.slab = t.slab; sw.splits = t.splits; sw.cols = t.Cn; sw.ldpad = ldpad;
        gs.push_back(sw);
        if (bias) {
            AdamTask::Slab sb; memset(&sb, 0, sizeof(sb));
            sb.off = offb; sb.n = t.R; sb.per = t.R; sb.sstride = t.R; sb.slab = t.bslab; sb.splits = t.splits;
            gs.push_back(sb);
        }
        t.flags |= FLAG_FIN_IN_ADAM;
        return true;
    }
    void gemm_lds_stage(Program& p, int la, int lb, int bt, std::vector<GemmTask> tasks, const char* what) {
        int base = 0, fin = 0;
        const int edge = bt == 129 ? 128 : bt == 65 ? 64 : bt == 33 ? 32 : bt;       // 129 / 65 / 33: the 128- / 64- / 32-wide tile on the bf16 pipe
        const int er = bt == 257 ? 256 : edge, ec = bt == 257 ? 128 : edge;     // 257: 256 rows x 128 columns (gemm_x3w.h)
        for (auto& t : tasks) {
            t.tiles_c = (t.Cn + ec - 1) / ec;
            t.ntiles = ((t.R + er - 1) / er) * t.tiles_c * t.splits; t.tile_base = base; base += t.ntiles;
            if (t.splits > 1 && fold_fin(t)) t.fin_base = 0x7fffffff;       // (no finishing block ever matches it)
            // (OPT-IN, RLREP_ENABLE=fin_inline: the last split workgroup of a tile finishes it inside the launch.  Bit-identical and one launch less per
            //  split stage -- and 1.1x (spedersac) to 3.5x (ctrlsac F = 2048) SLOWER per train(): the slabs have to go through to memory and come back past
            //  the L2s, docs/history/r06.md)
            else if (t.splits > 1 && bt == 65 && rl_opt("fin_inline")) { t.flags |= FLAG_FIN_INLINE; t.fin_base = 0x7fffffff; }
            else if (t.splits > 1) {
                const bool bias = t.epi == EPI_DW && (t.flags & FLAG_BIASGRAD);
                t.fin_base = fin;
                fin += (int)(((long long)t.R * ((t.Cn + 3) / 4) + 255) / 256) + (bias ? (t.R + 255) / 256 : 0);
            }
        }
        GemmBatch gb; memset(&gb, 0, sizeof(gb));
        gb.ntasks = (int)tasks.size();
        for (size_t q = 0; q < tasks.size(); ++q) gb.t[q] = tasks[q];
        p.stages.push_back({[=](hipStream_t st) { return rl_launch_gemm_lds(bt, la, lb, &gb, base, fin, st); }, what});
        tag_gemms(p, (bt == 257 || bt == 129 || bt == 65 || bt == 33) ? RLREP_ENGINE_X3 : bt == 128 ? RLREP_ENGINE_LDS128 : RLREP_ENGINE_LDS64, tasks);
    }
    void gemm_small(Program& p, int la, int lb, std::vector<GemmTask> tasks, const char* what) {
        if (chain_take(p, la, lb, tasks, what)) return;
        gemm_small_launch(p, la, lb, tasks, what);
    }
    void gemm_small_launch(Program& p, int la, int lb, std::vector<GemmTask> tasks, const char* what) {
        // column fragments per workgroup: the widest tile that still leaves >= ~1.5 workgroups per CU; the
        // epilogues that need a whole row / per-tile partials in one fragment force NF = 1
        int nf = 1;
        {
            bool force1 = false;
            for (auto& t : tasks) force1 = force1 || t.epi == EPI_FWD_POLICY || t.epi == EPI_FWD_MSE || t.epi == EPI_DX_REPARAM;
            auto count = [&](int f) { long long n = 0; for (auto& t : tasks) n += (long long)((t.R + 15) / 16) * ((t.Cn + 16 * f - 1) / (16 * f)); return n; };
            if (!force1) { if (count(2) >= 384) nf = 2; if (count(4) >= 384) nf = 4; }
            if (!tasks.empty() && (tasks[0].flags & FLAG_PRE)) nf = 1;
        }
        int base_tile = 0;
        for (auto& t : tasks) {
            t.tiles_c = (t.Cn + 16 * nf - 1) / (16 * nf);
            const int tr = (t.R + 15) / 16;
            t.ntiles = tr * t.tiles_c; t.tile_base = base_tile; base_tile += t.ntiles;
        }
        if (tasks.size() > GEMM_MAX_TASKS) { fprintf(stderr, "rlrep: too many tasks in stage %s\n", what); abort(); }
        GemmBatch gb; memset(&gb, 0, sizeof(gb));
        gb.ntasks = (int)tasks.size();
        gb.low_prio = low_prio ? 1 : 0;
        for (size_t q = 0; q < tasks.size(); ++q) gb.t[q] = tasks[q];
        const int total = base_tile;
        bool dyn = false;
        for (auto& t : tasks) dyn = dyn || (t.flags & (FLAG_DYN_EPS | FLAG_DYN_EPS2 | FLAG_DYN_EPS3));
        rlrep_agent* a = ag;
        if (dyn)
            p.stages.push_back({[=](hipStream_t st) {
                GemmBatch g2 = gb;
                for (int q = 0; q < g2.ntasks; ++q) {
                    if (g2.t[q].flags & FLAG_DYN_EPS) g2.t[q].x2 = a->cur_eps;
                    if (g2.t[q].flags & FLAG_DYN_EPS2) g2.t[q].x2 = a->cur_eps2;
                    if (g2.t[q].flags & FLAG_DYN_EPS3) g2.t[q].x2 = a->cur_eps3;
                }
                return rl_launch_gemm16(la, lb, nf, &g2, total, st);
            }, what});
        else
            p.stages.push_back({[=](hipStream_t st) { return rl_launch_gemm16(la, lb, nf, &gb, total, st); }, what});
        tag_gemms(p, RLREP_ENGINE_GEMM16, tasks);
    }

    // Two INDEPENDENT stages of different tile forms -- d1: row-major x k-major products (the dX form), d2: k-major x k-major ones (the
    // weight-gradient form) -- as ONE launch (gemm16_duo_kernel): a dependent launch less.  Falls back to the two stages (d1 first) when a
    // task is routed to the LDS-tiled engines, carries a fused short product, or the table does not fit.  RLREP_DISABLE=duo: always the two stages.
    bool routes_small(const GemmTask& t, int la, int lb) { int sp = 1, kc = 0, fl = 0; GemmTask c = t; return !rl_gemm_lds_route(&c, la, lb, 0, &sp, &kc, &fl); }
    void gemm_duo(Program& p, std::vector<GemmTask> d1, std::vector<GemmTask> d2, const char* w1, const char* w2, const char* what) {
        bool ok = !rl_off("duo") && !chain_prog && !d1.empty() && !d2.empty() && d1.size() + d2.size() <= GEMM_MAX_TASKS;
        for (auto& t : d1) ok = ok && routes_small(t, LD_ROW, LD_COL) && !(t.flags & (FLAG_PRE | FLAG_DYN_EPS | FLAG_DYN_EPS2 | FLAG_DYN_EPS3));
        for (auto& t : d2) ok = ok && routes_small(t, LD_COL, LD_COL) && !(t.flags & (FLAG_PRE | FLAG_DYN_EPS | FLAG_DYN_EPS2 | FLAG_DYN_EPS3));
        if (!ok) { gemm(p, LD_ROW, LD_COL, d1, w1); gemm(p, LD_COL, LD_COL, d2, w2); return; }
        auto count4 = [&]() { long long n = 0; for (auto& t : d2) n += (long long)((t.R + 15) / 16) * ((t.Cn + 63) / 64); return n; };
        const int nf2 = count4() >= 192 ? 4 : 1;
        int base_tile = 0;
        for (auto& t : d1) { t.tiles_c = (t.Cn + 15) / 16; t.ntiles = ((t.R + 15) / 16) * t.tiles_c; t.tile_base = base_tile; base_tile += t.ntiles; }
        for (auto& t : d2) { t.tiles_c = (t.Cn + 16 * nf2 - 1) / (16 * nf2); t.ntiles = ((t.R + 15) / 16) * t.tiles_c; t.tile_base = base_tile; base_tile += t.ntiles; }
        GemmBatch gb; memset(&gb, 0, sizeof(gb));
        gb.ntasks = (int)(d1.size() + d2.size()); gb.low_prio = low_prio ? 1 : 0;
        for (size_t q = 0; q < d1.size(); ++q) gb.t[q] = d1[q];
        for (size_t q = 0; q < d2.size(); ++q) gb.t[d1.size() + q] = d2[q];
        const int total = base_tile, split = (int)d1.size();
        p.stages.push_back({[=](hipStream_t st) { return rl_launch_gemm16_duo(split, nf2, &gb, total, st); }, what});
        { std::vector<GemmTask> both = d1; both.insert(both.end(), d2.begin(), d2.end()); tag_gemms(p, RLREP_ENGINE_GEMM16, both); }
    }

    // ---- chains (xchain.hip) -----------------------------------------------------------------------
    // Between chain_begin(p) and chain_end() consecutive ROW-LOCAL stages of program p (forward and dX stages of the 16-row tile engine, the
    // fused heads + vae_mid stage) are not emitted as launches but collected as PHASES of one persistent launch, in which the workgroups of
    // an XCD hand their tiles to each other through that XCD's L2 (kparams.h XcLaunch).  Anything else that reaches the program inside the
    // bracket must go through chain_flush() first: gemm() does it for the LDS-tiled engines, the builders do it before a stage they push
    // themselves.  A chain of one phase is emitted as the ordinary launch it would have been.
    struct ChainPhase { XcPhase ph; std::vector<GemmTask> tasks; HeadsVae hv; const char* what; int la, lb; };
    std::vector<ChainPhase> chain; Program* chain_prog = nullptr; int chain_R = -1;
    // OPT-IN while it is slower than the launches it replaces beside the deferred chain: RLREP_ENABLE=xchain
#ifdef RL_EXPERIMENTS
    static bool chain_enabled() { const char* e = rl_opt("xchain"); return e && e[0] == '1'; }
#else
    static bool chain_enabled() { return false; }
#endif
    static int chain_mpg() { const char* e = rl_opt("xchain_mpg"); const int v = e ? atoi(e) : 32; return v == 64 ? 64 : 32; }
    bool chain_fits(int R) const { return chain_enabled() && R <= 512 && (chain_R < 0 || chain_R == R); }
    void chain_begin(Program& p) { chain_flush(); chain_prog = &p; chain_R = -1; }
    void chain_end() { chain_flush(); chain_prog = nullptr; }
    static bool vec_ok(const std::vector<GemmTask>& tasks, bool opB, bool dry_) {
        for (const GemmTask& t : tasks) {
            const float* ptr = opB ? t.B : t.A; const int ld = opB ? t.ldb : t.lda;
            if ((ld & 3) || (t.K & 3) || (!dry_ && (((uintptr_t)ptr) & 15))) return false;
        }
        return true;
    }
    bool chain_take(Program& p, int la, int lb, const std::vector<GemmTask>& tasks, const char* what) {
        if (chain_prog != &p) return false;
        bool ok = la == LD_ROW && !tasks.empty();
        for (const GemmTask& t : tasks) {
            ok = ok && tasks.size() <= GEMM_MAX_TASKS && chain_fits(t.R) && t.R == tasks[0].R && !(t.flags & FLAG_PRE_FWD) && t.epi != EPI_DW && t.Cn <= 2048 && t.K <= 2048;
            ok = ok && (((t.flags & FLAG_PRE) != 0) == ((tasks[0].flags & FLAG_PRE) != 0));
        }
        if (!ok) { chain_flush(); return false; }
        ChainPhase c; memset(&c.ph, 0, sizeof(c.ph)); memset(&c.hv, 0, sizeof(c.hv));
        c.ph.kind = XC_GEMM; c.ph.la = la; c.ph.lb = lb; c.ph.pre = (tasks[0].flags & FLAG_PRE) ? 1 : 0;
        c.ph.vecA = vec_ok(tasks, false, dry) ? 1 : 0; c.ph.vecB = vec_ok(tasks, true, dry) ? 1 : 0;
        c.tasks = tasks; c.what = what; c.la = la; c.lb = lb;
        chain_R = tasks[0].R;
        chain.push_back(c);
        return true;
    }
    // the fused Gaussian heads + vae_mid stage (heads_vae_kernel) as a phase, or as its own launch outside a chain
    void heads_vae_stage(Program& p, const HeadsVae& hv, const char* what) {
        rlrep_agent* a = ag;
        if (chain_prog == &p && chain_fits(hv.B)) {
            ChainPhase c; memset(&c.ph, 0, sizeof(c.ph));
            c.ph.kind = XC_HEADS_VAE; c.ph.dyn = 0; c.hv = hv; c.what = what; c.la = c.lb = 0;
            chain_R = hv.B;
            chain.push_back(c);
            return;
        }
        chain_flush();
        p.stages.push_back({[=](hipStream_t st) { HeadsVae q = hv; q.eps = a->cur_eps; return rl_launch_heads_vae(&q, st); }, what});
        tag_heads_vae(p, hv);
    }
    // two [B, K] x [2F, K]^T head products; operands: both activations, both weight pairs; results: z, eps*sigma, two head gradients
    static void tag_heads_vae(Program& p, const HeadsVae& hv) {
        tag(p, RLREP_ENGINE_HEADS_VAE, 2.0 * 2.0 * (double)hv.B * (2.0 * hv.F) * (double)hv.K,
            4.0 * (2.0 * (double)hv.B * hv.K + 2.0 * (2.0 * hv.F) * hv.K + (double)hv.B * hv.F * 7.0));
    }
    void chain_flush() {
        if (chain.empty()) return;
        std::vector<ChainPhase> c; c.swap(chain);
        Program& p = *chain_prog;
        const int R = chain_R; chain_R = -1;
        rlrep_agent* a = ag;
        if (c.size() == 1) {            // nothing to chain: the launch it would have been
            if (c[0].ph.kind == XC_GEMM) gemm_small_launch(p, c[0].la, c[0].lb, c[0].tasks, c[0].what);
            else { const HeadsVae hv = c[0].hv; p.stages.push_back({[=](hipStream_t st) { HeadsVae q = hv; q.eps = a->cur_eps; return rl_launch_heads_vae(&q, st); }, c[0].what}); tag_heads_vae(p, hv); }
            return;
        }
        const int rbg = (((R + 15) / 16) + XC_GROUPS - 1) / XC_GROUPS;      // 16-row blocks per group
        std::vector<XcPhase> phs; std::vector<GemmTask> tasks; std::vector<HeadsVae> hvs;
        std::string name = "chain:";
        for (ChainPhase& cp : c) {
            XcPhase ph = cp.ph;
            if (ph.kind == XC_GEMM) {
                ph.task0 = (int)tasks.size(); ph.ntasks = (int)cp.tasks.size();
                int base = 0, q = 0;
                ph.rbg = rbg; ph.R = R;
                for (GemmTask t : cp.tasks) {
                    t.tiles_c = (t.Cn + 15) / 16; t.ntiles = rbg * t.tiles_c; t.tile_base = base;
                    ph.tb[q] = base; ph.tcs[q] = t.tiles_c; ++q;
                    rl_gemm16_plan(t);
                    base += t.ntiles;
                    tasks.push_back(t);
                }
                ph.tiles = base;
            } else {
                ph.aux = (int)hvs.size(); ph.tiles = rbg * cp.hv.tiles_c;
                hvs.push_back(cp.hv);
            }
            phs.push_back(ph);
            name += std::string(" [") + cp.what + "]";
        }
        if (hvs.empty()) { HeadsVae z; memset(&z, 0, sizeof(z)); hvs.push_back(z); }      // (identical allocations in the dry and the real pass)
        XcLaunch L; memset(&L, 0, sizeof(L));
        L.ntasks = (int)tasks.size(); L.nhv = (int)hvs.size();
        if (tasks.empty()) tasks.push_back(base());
        L.ph = upload(phs); L.nph = (int)phs.size(); L.tasks = upload(tasks); L.hv = upload(hvs);
        L.flags = (unsigned*)ws.alloc(sizeof(unsigned) * XC_GROUPS * XC_FLAG_STRIDE);
        if (!dry && ws.ok() && L.flags) (void)hipMemset(L.flags, 0, sizeof(unsigned) * XC_GROUPS * XC_FLAG_STRIDE);
        L.err = a->xc_err; L.rbg = rbg; L.mpg = chain_mpg(); L.low_prio = low_prio ? 1 : 0;
        a->stage_names.push_back(name);
        const char* what = a->stage_names.back().c_str();
        p.stages.push_back({[=](hipStream_t st) {
            XcLaunch l = L; l.dyn[0] = a->cur_eps; l.dyn[1] = a->cur_eps2; l.dyn[2] = a->cur_eps3;
            return rl_launch_xchain(&l, st);
        }, what});
    }
    bool low_prio = false;        // set while the deferred critic / actor programs are built: their chain has slack next to the feature chain
    float lr_of(int g) const { return g == 1 ? ag->h.lr_critic : g == 2 ? ag->h.lr_actor : ag->h.lr_feature; }

    void fwd_stage(Program& p, std::vector<GemmTask> t, const char* w) { gemm(p, LD_ROW, LD_ROW, t, w); }
    void dx_stage(Program& p, std::vector<GemmTask> t, const char* w) { gemm(p, LD_ROW, LD_COL, t, w); }
    // Two consecutive dX stages of which the FIRST has a short inner length (<= 32) and a ReLU or ELU mask: one launch in which every tile of
    // the second recomputes its 16 rows of the first (gemm16.hip, FLAG_PRE).  Falls back to the two stages when the pair does not fit.
    void dx_stage12(Program& p, GemmTask d1, GemmTask d2, const char* w1, const char* w2) {
        const bool ok = !rl_off("fuse_dx") && d1.epi == EPI_DX && (d1.act == ACT_RELU || d1.act == ACT_ELU) && !(d1.flags & FLAG_ACCUM) && !d1.r1u && d1.K <= 32 &&
                        d1.scale == 1.f && d2.A == d1.C && d2.lda == d1.ldc && d2.K == d1.Cn && d2.R == d1.R &&
                        (d2.epi == EPI_DX || d2.epi == EPI_DX_REPARAM) && ((d2.R + 15) / 16) * ((d2.Cn + 15) / 16) < 384 * 2;
        if (!ok) { dx_stage(p, {d1}, w1); dx_stage(p, {d2}, w2); return; }
        GemmTask t = d2;
        t.flags |= FLAG_PRE | (d1.act == ACT_ELU ? FLAG_PRE_ELU : 0);
        t.x0 = d1.A; t.ldx0 = d1.lda; t.x1 = d1.B; t.ldx1 = d1.ldb; t.n0 = d1.K; t.x2 = d1.aux; t.ldaux2 = d1.ldaux; t.y0 = d1.C; t.ldout2 = d1.ldc;
        gemm_small(p, LD_ROW, LD_COL, {t}, w2);
    }
    // Forward twin: layer pairs (first, second) where the FIRST has a short inner length (<= 48), ReLU and a transposed weight shadow `wt`
    // ([K1][N1], kept by the optimizer launch): one launch in which every tile of the second layer recomputes its 16 rows of the first
    // (gemm16.hip, FLAG_PRE | FLAG_PRE_FWD) and column tile 0 stores them for the backward pass.  False (nothing emitted) if a pair does not fit.
    struct FwdPair { GemmTask first, second; const float* wt; };
    bool fwd_stage12(Program& p, const std::vector<FwdPair>& pairs, const char* what) {
        if (pairs.empty()) return false;
        std::vector<GemmTask> tasks;
        for (const FwdPair& fp : pairs) {
            const GemmTask& d1 = fp.first; const GemmTask& d2 = fp.second;
            const bool ok = (dry || fp.wt) && d1.epi == EPI_FWD && d1.act == ACT_RELU && d1.K <= 48 && d1.scale == 1.f && d2.epi == EPI_FWD &&
                            d2.A == d1.C && d2.lda == d1.ldc && d2.K == d1.Cn && d2.R == d1.R && !d1.out2 &&
                            ((d2.R + 15) / 16) * ((d2.Cn + 15) / 16) < 384 * 2;
            if (!ok) return false;
            GemmTask t = d2;
            t.flags |= FLAG_PRE | FLAG_PRE_FWD;
            t.x0 = d1.A; t.ldx0 = d1.lda; t.x1 = fp.wt; t.ldx1 = d1.Cn; t.n0 = d1.K; t.x2 = d1.bias; t.ldaux2 = 0; t.y0 = d1.C; t.ldout2 = d1.ldc;
            tasks.push_back(t);
        }
        gemm_small(p, LD_ROW, LD_ROW, tasks, what);
        return true;
    }
    // weight-gradient stage
    void dw_stage(Program& p, std::vector<GemmTask> t, const char* w) { gemm(p, LD_COL, LD_COL, t, w); }

    // split-K partial gradients that the group's optimizer launches finish themselves (AdamTask::Slab; set by the agent's builder)
    std::vector<AdamTask::Slab> group_slabs[4];
    void adam(Program& p, int group, float lr, float* target, int64_t pol_off, int64_t pol_n, float tau,
              std::vector<FinTask> fin, const char* what, const int* pol_steps = nullptr, int pol_period = 1) {
        const auto& L = ag->L;
        AdamTask t; memset(&t, 0, sizeof(t));
        const int64_t off = L.group_off[group];
        t.p = ag->a.param_dev ? ag->a.param_dev + off : nullptr;
        t.g = ag->a.grad_dev ? ag->a.grad_dev + off : nullptr;
        t.m = ag->a.exp_avg_dev ? ag->a.exp_avg_dev + off : nullptr;
        t.v = ag->a.exp_avg_sq_dev ? ag->a.exp_avg_sq_dev + off : nullptr;
        t.n = L.group_n[group];
        t.lr = lr; t.beta1 = ag->h.beta1; t.beta2 = ag->h.beta2; t.eps = ag->h.adam_eps;
        t.grp = ag->adam_step + group;
        t.target = target; t.pol_off = pol_off - off; t.pol_n = pol_n; t.tau = tau; t.pol_steps = pol_steps; t.pol_period = pol_period;
        t.sh = ag->sh_dev[group]; t.nsh = ag->nsh[group];
        t.sync_steps = (group == 0 || ag->d.alg == RLREP_ALG_SAC) ? ag->steps : nullptr;      // (launches in the train prologue's own chain)
        t.nslab = (int)std::min<size_t>(group_slabs[group].size(), 8);
        for (int q = 0; q < t.nslab; ++q) t.slabs[q] = group_slabs[group][q];
        const FinTask* fdev = fin.empty() ? nullptr : upload(fin);
        const int nfin_all = (int)fin.size();
        const bool hist_last = !fin.empty() && fin.back().kind == FIN_HISTORY;        // run only while rlrep_history is on
        const int blocks = (int)((t.n + 1023) / 1024);
        rlrep_agent* a = ag;
        p.stages.push_back({[=](hipStream_t st) {
            // a minibatch armed by rlrep_prefetch_batch is gathered by extra blocks of this launch
            const SlotFill* sf = a->pf_armed ? &a->pf_fill : nullptr;
            if (sf) { a->pf_armed = false; a->pf_done = true; a->slot[0].filled = true; a->pi_ready = nullptr; a->early_ready_crit = a->early_ready_act = nullptr; }
            const SlotFill* sf2 = a->pf2_armed ? &a->pf2_fill : nullptr;
            if (sf2) { a->pf2_armed = false; a->pf2_done = true; a->slot[1].filled = true; }
            // a snapshot armed by rlrep_defer_arm rides in the feature group's launch (never beside a minibatch gather: both touch slot 0)
            AdamSnap sn; memset(&sn, 0, sizeof(sn));
            if (group == 0 && a->snap_armed && !sf) {
                const rlrep_agent::DeferSet& D = a->dset[a->snap_set];
                const CopySegs& full = D.segs;
                long long end = 0; int n = 0;
                for (int q = 0; q < full.n; ++q) {
                    const long long cnt = full.end[q] - (q ? full.end[q - 1] : 0);
                    if (full.dst[q] == D.block) continue;                                   // the block is written by the optimizer's own lanes
                    sn.segs.src[n] = q == full.n - 2 ? a->snap_ec : q == full.n - 1 ? a->snap_ea : full.src[q];
                    sn.segs.dst[n] = full.dst[q]; end += cnt; sn.segs.end[n] = end; ++n;
                }
                sn.segs.n = n; sn.segs.isrc = full.isrc; sn.segs.idst = full.idst;
                sn.block = D.block; sn.off = D.block_off; sn.n = D.block_n; sn.which = D.block_which; sn.on = 1;
                a->snap_armed = false; a->snap_done = a->snap_set;
            }
            const int nfin = (hist_last && !a->hist_on) ? nfin_all - 1 : nfin_all;
            DpPull dp = a->dp_proto;
            dp.channel = group; dp.mode = a->dp_two[group] ? 2 : 1;
            if (t.sync_steps) a->mirror_pending = false;
            return rl_launch_adam(&t, blocks, fdev, nfin, sf, sf2, sn.on ? &sn : nullptr, a->dp_on[group] ? &dp : nullptr, st);
        }, what});
        tag(p, RLREP_ENGINE_OPTIMIZER, 0.0, 28.0 * (double)t.n + 12.0 * (double)(target ? pol_n : 0));     // read p, g, m, v; write p, m, v (+ target: read, read source, write)
    }
    // optimizer launch of `group` that skips [skip0, skip0 + n0) and [skip1, skip1 + n1) (group-relative; their Adam ran in the weight-gradient
    // epilogues) and carries the next step's two first-layer tasks as leading tiles (elementwise.hip adam_l1_kernel)
    void adam_l1(Program& p, int group, float lr, float* target, int64_t pol_off, int64_t pol_n, float tau, std::vector<FinTask> fin,
                 int64_t skip0, int64_t n0, int64_t skip1, int64_t n1, GemmTask g0, GemmTask g1, const char* what) {
        const auto& L = ag->L;
        AdamTask t; memset(&t, 0, sizeof(t));
        const int64_t off = L.group_off[group];
        t.p = ag->a.param_dev ? ag->a.param_dev + off : nullptr;
        t.g = ag->a.grad_dev ? ag->a.grad_dev + off : nullptr;
        t.m = ag->a.exp_avg_dev ? ag->a.exp_avg_dev + off : nullptr;
        t.v = ag->a.exp_avg_sq_dev ? ag->a.exp_avg_sq_dev + off : nullptr;
        t.n = L.group_n[group];
        t.lr = lr; t.beta1 = ag->h.beta1; t.beta2 = ag->h.beta2; t.eps = ag->h.adam_eps;
        t.grp = ag->adam_step + group;
        t.target = target; t.pol_off = pol_off - off; t.pol_n = pol_n; t.tau = tau; t.pol_steps = nullptr; t.pol_period = 1;
        t.nskip = 2; t.skip_off[0] = skip0; t.skip_n[0] = n0; t.skip_off[1] = skip1; t.skip_n[1] = n1;
        t.sync_steps = ag->steps;          // (group 0: in the train prologue's own chain)
        if (!fin.empty() && fin.back().kind == FIN_HISTORY) fin.pop_back();
        const FinTask* fdev = fin.empty() ? nullptr : upload(fin);
        const int nfin = (int)fin.size();
        const int blocks = (int)((t.n + 1023) / 1024);
        rlrep_agent* a = ag;
        p.stages.push_back({[=](hipStream_t st) {
            if (!a->pf_armed) return -8;                     // (rlrep_feature_chain_next refuses to arm without a prefetched minibatch)
            a->mirror_pending = false;
            const SlotFill sf = a->pf_fill;
            a->pf_armed = false; a->pf_done = true; a->slot[0].filled = true; a->pi_ready = nullptr; a->early_ready_crit = a->early_ready_act = nullptr;
            return rl_launch_adam_l1(&t, blocks, fdev, nfin, &sf, &g0, &g1, st);
        }, what});
        tag(p, RLREP_ENGINE_OPTIMIZER, 0.0, 28.0 * (double)(t.n - n0 - n1) + 12.0 * (double)(target ? pol_n : 0));
    }
    // metric finalisation without an optimizer (diffsrsac's critic step: quirk Q11, its optimizer is a no-op)
    void finalize_only(Program& p, std::vector<FinTask> fin, const char* what) {
        if (!fin.empty() && fin.back().kind == FIN_HISTORY) fin.pop_back();        // (the history ring is kept by the optimizer launches only)
        if (fin.empty()) return;
        const FinTask* fdev = upload(fin);
        const int nfin = (int)fin.size();
        p.stages.push_back({[=](hipStream_t st) { return rl_launch_adam(nullptr, 0, fdev, nfin, nullptr, nullptr, nullptr, nullptr, st); }, what});
    }

    static FinTask fin_sum(const float* partials, int count, int stride, float scale, float* out) {
        FinTask f; memset(&f, 0, sizeof(f));
        f.kind = FIN_SUM; f.partials = partials; f.count = count; f.stride = stride; f.scale = scale; f.out = out;
        return f;
    }
    static FinTask fin_combine(const float* a, float sa, const float* b, float sb, float* out) {
        FinTask f; memset(&f, 0, sizeof(f));
        f.kind = FIN_COMBINE; f.in_a = a; f.in_b = b; f.scale = sa; f.scale_b = sb; f.out = out;
        return f;
    }
    static FinTask fin_inc(int* counter) {
        FinTask f; memset(&f, 0, sizeof(f)); f.kind = FIN_INC; f.out = reinterpret_cast<float*>(counter); return f;
    }
    static FinTask fin_history(const rlrep_agent* a) {
        FinTask f; memset(&f, 0, sizeof(f));
        f.kind = FIN_HISTORY; f.in_a = a->metrics; f.stride = RL_HIST_TAG; f.out = a->hist; f.count = RL_HIST_N; f.partials = reinterpret_cast<const float*>(a->hist_seq);
        return f;
    }
    static FinTask fin_copy(const float* a, float* out) {
        FinTask f; memset(&f, 0, sizeof(f)); f.kind = FIN_COPY; f.in_a = a; f.out = out; return f;
    }
};


// ------------------------------------------------------------------------------------------------
// row-block programs (rowprog.hip): host-side assembler
// ------------------------------------------------------------------------------------------------
// (opt-in engines: compiled into the experiments library only, rlrep_amd/csrc/build.sh)
#ifdef RL_EXPERIMENTS
static inline bool rl_rowprog_enabled() { const char* e = rl_opt("rowprog"); return e && (e[0] == '1' || e[0] == '2'); }
static inline bool rl_rowprog_cluster() { const char* e = rl_opt("rowprog"); return e && e[0] == '2'; }
#else
static inline bool rl_rowprog_enabled() { return false; }
static inline bool rl_rowprog_cluster() { return false; }
#endif
struct RpBuf { int off, ld, w; };            // LDS buffer: float offset, row stride, zero-padded width (multiple of 32)

struct RpAsm {
    std::vector<RpOp> ops; std::vector<RpProg> progs;
    int top = 0, peak = 0, cur_begin = 0, blocks = 0;
    static RpOp blank(int kind) { RpOp o; memset(&o, 0, sizeof(o)); o.kind = kind; o.dst = -1; o.dyn = -1; return o; }
    // LDS regions are bump-allocated per program; `at` re-uses the storage of a dead buffer for a new shape
    RpBuf buf(int cols) { RpBuf b; b.w = (cols + 31) & ~31; b.ld = b.w + 4; b.off = top; top += RP_ROWS * b.ld; if (top > peak) peak = top; return b; }
    static RpBuf at(const RpBuf& dead, int cols) { RpBuf b; b.w = (cols + 31) & ~31; b.ld = b.w + 4; b.off = dead.off; if (b.ld > dead.ld) { fprintf(stderr, "rlrep: row-program buffer reuse does not fit\n"); abort(); } return b; }
    void begin() { cur_begin = (int)ops.size(); top = 0; }
    void end(int nblocks, int csize = 1, int ctype = 0) {
        RpProg p; p.op_begin = cur_begin; p.op_end = (int)ops.size(); p.block_base = blocks; p.nblocks = nblocks * csize; p.csize = csize; p.ctype = ctype;
        blocks += p.nblocks; progs.push_back(p);
    }
    // cluster programs: exchange ops on the full-width LDS vector `v`; slice = ws columns x np pieces (piece stride ps) per member
    void xop(int kind, const RpBuf& v, int hop, int ws, int np, int ps, int mode = 0) {
        RpOp o = blank(kind); o.src = v.off; o.lds = v.ld; o.N = ws; o.K = np; o.ldw = ps; o.flag = hop; o.n0 = mode; ops.push_back(o);
    }
    void load(const float* g, int ldg, int K, const RpBuf& d) {
        RpOp o = blank(RP_LOAD); o.gin = g; o.ldgin = ldg; o.K = K; o.dst = d.off; o.ldd = d.ld; o.wpad = d.w; ops.push_back(o);
    }
    // Y = act(X W^T + b): W [N, K] row stride ldw
    void fwd(const RpBuf& x, int K, const float* W, int ldw, const float* bias, int N, int act, const RpBuf* d, float* gout, int ldg) {
        RpOp o = blank(RP_GEMM); o.src = x.off; o.lds = x.ld; o.K = K; o.N = N; o.W = W; o.ldw = ldw; o.bias = bias; o.act = act;
        o.flags = bias ? RPF_BIAS : 0; if (d) { o.dst = d->off; o.ldd = d->ld; o.wpad = d->w; } o.gout = gout; o.ldg = ldg; ops.push_back(o);
    }
    // the same layer from the TRANSPOSED weight WT [K, N] (row stride N): the loader of the dX form, whose fragments are contiguous
    void fwdT(const RpBuf& x, int K, const float* WT, const float* bias, int N, int act, const RpBuf* d, float* gout, int ldg) {
        RpOp o = blank(RP_GEMM); o.src = x.off; o.lds = x.ld; o.K = K; o.N = N; o.W = WT; o.ldw = N; o.bias = bias; o.act = act;
        o.flags = RPF_COL | (bias ? RPF_BIAS : 0); if (d) { o.dst = d->off; o.ldd = d->ld; o.wpad = d->w; } o.gout = gout; o.ldg = ldg; ops.push_back(o);
    }
    // dX = (G W) * act'(aux): W [K, N] row stride ldw (a column window of a wider matrix is W + offset with the full row stride)
    void dx(const RpBuf& g, int K, const float* W, int ldw, int N, int act, const RpBuf* mask_lds, const float* mask_g, int ldmask,
            const RpBuf* d, float* gout, int ldg) {
        RpOp o = blank(RP_GEMM); o.src = g.off; o.lds = g.ld; o.K = K; o.N = N; o.W = W; o.ldw = ldw; o.act = act; o.flags = RPF_COL;
        if (act != ACT_NONE && mask_lds) { o.flags |= RPF_MASK_LDS; o.src2 = mask_lds->off; o.lds2 = mask_lds->ld; }
        else if (act != ACT_NONE && mask_g) { o.flags |= RPF_MASK_GLOBAL; o.gaux = mask_g; o.ldgaux = ldmask; }
        else o.act = ACT_NONE;
        if (d) { o.dst = d->off; o.ldd = d->ld; o.wpad = d->w; } o.gout = gout; o.ldg = ldg; ops.push_back(o);
    }
    void signal(int flag) { RpOp o = blank(RP_SIGNAL); o.flag = flag; ops.push_back(o); }
    void wait(int flag) { RpOp o = blank(RP_WAIT); o.flag = flag; ops.push_back(o); }
    void store(const RpBuf& s, int N, float* g, int ldg) { RpOp o = blank(RP_STORE); o.src = s.off; o.lds = s.ld; o.N = N; o.gout = g; o.ldg = ldg; ops.push_back(o); }
    size_t lds_bytes() const { return (size_t)peak * sizeof(float); }
};

struct ActorBufs { float *A1, *A2, *AO, *logp, *dA, *Ghead, *GA2, *GA1; };

int qhead_blocks(int B);
ActorBufs alloc_actor(Builder& b, int B, int A, int Ha);
GemmTask actor_l(rlrep_agent* ag, int layer, const float* X, int ldx, const ActorBufs& ab);
void policy_fwd_stage(Program& p, rlrep_agent* ag, const ActorBufs& ab, float* act, int ld_act, const char* what);
// actor head layer + tanh-Gaussian sampling/log-prob: ONE launch (policy fused into the head GEMM's epilogue) when the
// [mu|rho] row fits one 16-column tile (2A <= 16), else head launch + policy_fwd_kernel.  `extra` tasks share the launch.
void actor_head_stage(Builder& b, Program& p, rlrep_agent* ag, const ActorBufs& ab, float* act, int ld_act, std::vector<GemmTask> extra, const char* what);
bool policy_fusable(const rlrep_agent* ag);
GemmTask policy_head_task(rlrep_agent* ag, const ActorBufs& ab, float* act, int ld_act, int dyn_flag);
// `action_dx` computes dL/da [B,A] into ab.dA; its epilogue takes over policy_bwd when A <= 16
void actor_backward(Builder& b, Program& p, rlrep_agent* ag, const ActorBufs& ab, const float* X, int ldx, const float* act, int ld_act, GemmTask action_dx);
void actor_apply_program(Builder& b, rlrep_agent* ag, const float* partial_loss, int nblk);
std::vector<FinTask> actor_fins(rlrep_agent* ag, const float* partial_loss, int nblk);
void critic_apply_folded(Builder& b, rlrep_agent* ag, const std::string& first_dst, std::vector<FinTask> fins, Program* into = nullptr, const int* steps = nullptr);
void update_target_program(rlrep_agent* ag, const std::string& first_src, const std::string& first_dst);
// Deferred critic / actor programs (rlrep_agent::dset): defer_begin(set) allocates that set's snapshot (minibatch, policy noise, step
// counter and a copy of the `block_n` floats at `block_src`), redirects slot 0 and the tensors named `prefix`* (block starting at tensor
// `first`) and returns the saved slot; the caller then emits its critic / actor programs into dset[set].critic_bwd / .actor_bwd and
// calls defer_end.
Slot defer_begin(Builder& b, rlrep_agent* ag, int set, const char* prefix, const char* first, const float* block_src, int64_t block_n);
void defer_end(Builder& b, rlrep_agent* ag, int set, const Slot& keep, const std::string& critic_target_first, std::vector<FinTask> cfins);

// agents2.hip
void lay_ctrlsac(const rlrep_dims& d, Layout& L);
void lay_spedersac(const rlrep_dims& d, Layout& L);
void lay_diffsrsac(const rlrep_dims& d, Layout& L);
void lay_actor(Layout& L, int S, int A, int Ha, int arena, int group);
void lay_six(Layout& L, const std::string& m, int in_f, int H, int arena, int group);
void build_ctrlsac(Builder& b, rlrep_agent* ag);
void build_spedersac(Builder& b, rlrep_agent* ag);
void build_diffsrsac(Builder& b, rlrep_agent* ag);

// Step programs of ctrlsac, spedersac and diffsrsac (host side; kernels in gemm16.hip / replearn.hip /
// elementwise.hip).  Same construction as sac/vlsac in engine.hip: a short list of grouped launches.
#include "engine_internal.h"

extern "C" {
int rl_launch_infonce(const InfoNce* p, hipStream_t st);
int rl_launch_colsum(const ColSum* p, hipStream_t st);
int rl_launch_reg_stats(const RegStats* p, hipStream_t st);
int rl_launch_speder_rows(const SpederRows* p, hipStream_t st);
int rl_launch_speder_grads(const SpederGrads* p, hipStream_t st);
int rl_launch_diffsr_perturb(const DiffsrPerturb* p, hipStream_t st);
int rl_launch_diffsr_score(const DiffsrScore* p, hipStream_t st);
int rl_launch_copy2(const float* src, float* d1, float* d2, long long n, hipStream_t st);
// comm.hip: ctrlsac's batch-coupled exchanges as launches of the step program (attached agents)
int rl_launch_xchg_gather(const DpPull* proto, int channel, long long off, long long n, int no_done, hipStream_t st);
int rl_launch_xchg_reduce(const DpPull* proto, int channel, long long off, long long n, float* out, int two_shot, int no_done, hipStream_t st);
int rl_launch_slots_sum(const DpSlots* d, float* out, hipStream_t st);
}

// ================================================================================================
// generic MLP (utils/util.py:85-96: Linear(+ELU) x depth, Linear) on Sequential names <prefix>.{0,2,..}
// ================================================================================================
struct Mlp {
    std::string prefix; bool target; int in_f, hid, out_f, depth;
    int width(int l) const { return l == depth ? out_f : hid; }        // output width of layer l
    int in_w(int l) const { return l == 0 ? in_f : hid; }
    std::string name(int l) const { return prefix + "." + std::to_string(2 * l); }
};

static void lay_mlp(Layout& L, const Mlp& m, int arena, int group) {
    for (int l = 0; l <= m.depth; ++l) L.lin(m.name(l), m.width(l), m.in_w(l), arena, group);
}

struct MlpBufs { std::vector<float*> act; std::vector<float*> g; int rows; };     // act[l]: output of layer l; g[l]: dL/d(pre-act l)

static MlpBufs alloc_mlp(Builder& b, const Mlp& m, int rows, bool need_grad) {
    MlpBufs r; r.rows = rows;
    for (int l = 0; l <= m.depth; ++l) {
        r.act.push_back(b.ws.f((size_t)rows * m.width(l)));
        r.g.push_back(need_grad ? b.ws.f((size_t)rows * m.width(l)) : nullptr);
    }
    return r;
}

static float* W(rlrep_agent* ag, const Mlp& m, int l, const char* what) {
    const std::string n = m.name(l) + what;
    return m.target ? ag->T(n) : ag->P(n);
}

// forward task of layer l (input X for l == 0); out_act: activation of the LAST layer (ACT_NONE for plain mlp)
static GemmTask mlp_fwd(rlrep_agent* ag, const Mlp& m, const MlpBufs& mb, int l, const float* X, int ldx, int out_act = ACT_NONE) {
    const float* in = l == 0 ? X : mb.act[l - 1];
    const int ld = l == 0 ? ldx : m.hid;
    return Builder::fwd(in, ld, mb.rows, m.in_w(l), W(ag, m, l, ".weight"), m.in_w(l), W(ag, m, l, ".bias"), m.width(l),
                        mb.act[l], m.width(l), l == m.depth ? out_act : ACT_ELU);
}
// backward-data task through layer l (l >= 1): g[l-1] = (g[l] W_l) * elu'(act[l-1])
static GemmTask mlp_dx(rlrep_agent* ag, const Mlp& m, const MlpBufs& mb, int l) {
    return Builder::dx(mb.g[l], m.width(l), mb.rows, m.width(l), W(ag, m, l, ".weight"), m.in_w(l), mb.g[l - 1], m.hid, m.hid,
                       ACT_ELU, mb.act[l - 1], m.hid);
}
// backward-data of layer 0 restricted to input columns [c0, c0+n)  (dL/d action)
static GemmTask mlp_dx_input(rlrep_agent* ag, const Mlp& m, const MlpBufs& mb, int c0, int n, float* out, int ldo) {
    float* w0 = W(ag, m, 0, ".weight");
    return Builder::dx(mb.g[0], m.width(0), mb.rows, m.width(0), w0 ? w0 + c0 : nullptr, m.in_f, out, ldo, n, ACT_NONE, nullptr, 0);
}
static GemmTask mlp_dw(rlrep_agent* ag, const Mlp& m, const MlpBufs& mb, int l, const float* X, int ldx) {
    const float* in = l == 0 ? X : mb.act[l - 1];
    const int ld = l == 0 ? ldx : m.hid;
    return Builder::dw(mb.g[l], m.width(l), m.width(l), in, ld, m.in_w(l), mb.rows, ag->G(m.name(l) + ".weight"), m.in_w(l),
                       ag->G(m.name(l) + ".bias"));
}

// ------------------------------------------------------------------------------------------------
// RFF critic (spedersac_agent.py:21-50, diffsrsac_agent.py:40-90): q = l3(elu(l2(sin(l1 z)))) per head
// ------------------------------------------------------------------------------------------------
struct RffBufs { float *S1, *PRE1, *E, *GE, *G1; };
static RffBufs alloc_rff(Builder& b, int B, int H, bool grad) {
    RffBufs r;
    r.S1 = b.ws.f((size_t)B * 2 * H); r.PRE1 = b.ws.f((size_t)B * 2 * H); r.E = b.ws.f((size_t)2 * B * H);
    r.GE = grad ? b.ws.f((size_t)2 * B * H) : nullptr; r.G1 = grad ? b.ws.f((size_t)B * 2 * H) : nullptr;
    return r;
}
static GemmTask rff_l1(rlrep_agent* ag, bool target, const float* Z, int F, int H, const RffBufs& r) {
    const char* m = target ? "critic_target" : "critic";
    auto w = [&](const char* n) { return target ? ag->T(std::string(m) + n) : ag->P(std::string(m) + n); };
    return Builder::fwd(Z, F, ag->B, F, w(".l1.weight"), F, w(".l1.bias"), 2 * H, r.S1, 2 * H, ACT_SIN, r.PRE1, 2 * H);
}
static void rff_l2(rlrep_agent* ag, bool target, int H, const RffBufs& r, std::vector<GemmTask>& out) {
    const char* m = target ? "critic_target" : "critic";
    auto w = [&](const char* n) { return target ? ag->T(std::string(m) + n) : ag->P(std::string(m) + n); };
    const size_t BH = (size_t)ag->B * H;
    out.push_back(Builder::fwd(r.S1, 2 * H, ag->B, H, w(".l2.weight"), H, w(".l2.bias"), H, r.E, H, ACT_ELU));
    out.push_back(Builder::fwd(r.S1 + H, 2 * H, ag->B, H, w(".l5.weight"), H, w(".l5.bias"), H, r.E + BH, H, ACT_ELU));
}

// ================================================================================================
// layouts
// ================================================================================================
static Mlp ctrl_phi(const rlrep_dims& d, const std::string& prefix, bool target) { return Mlp{prefix, target, d.state_dim + d.action_dim, d.phi_hidden_dim, d.feature_dim, 2}; }

static void lay_ctrl_phi(Layout& L, const rlrep_dims& d, const std::string& m, int arena, int group) {
    // agent/ctrlsac/ctrlsac_agent.py:68-70 names l1,l2,l3
    L.lin(m + ".l1", d.phi_hidden_dim, d.state_dim + d.action_dim, arena, group);
    L.lin(m + ".l2", d.phi_hidden_dim, d.phi_hidden_dim, arena, group);
    L.lin(m + ".l3", d.feature_dim, d.phi_hidden_dim, arena, group);
}

void lay_ctrlsac(const rlrep_dims& d, Layout& L) {
    const int S = d.state_dim, A = d.action_dim, H = d.hidden_dim, F = d.feature_dim;
    const int P = RLREP_ARENA_PARAM, T = RLREP_ARENA_TARGET;
    L.begin_group(0);
    lay_ctrl_phi(L, d, "phi", P, 0);
    L.lin("mu.l1", d.mu_hidden_dim, S, P, 0);                       // ctrlsac_agent.py:92-94
    L.lin("mu.l2", d.mu_hidden_dim, d.mu_hidden_dim, P, 0);
    L.lin("mu.l3", F, d.mu_hidden_dim, P, 0);
    L.lin("theta.l", 1, F, P, 0);
    L.end_group(0);
    auto critic = [&](const std::string& m, int arena, int g) {    // ctrlsac_agent.py:32-38: l1,l2 / l4,l5; l1|l4 glued
        L.lin_pair(m + ".l1", H, m + ".l4", H, F, arena, g);
        L.lin(m + ".l2", 1, H, arena, g);
        L.lin(m + ".l5", 1, H, arena, g);
    };
    L.begin_group(1); critic("critic", P, 1); L.end_group(1);
    L.begin_group(2); lay_actor(L, S, A, d.actor_hidden_dim, P, 2); L.end_group(2);
    lay_ctrl_phi(L, d, "phi_target", T, -1);
    critic("critic_target", T, -1);
    lay_ctrl_phi(L, d, "frozen_phi", T, -1);
    lay_ctrl_phi(L, d, "frozen_phi_target", T, -1);
}

void lay_spedersac(const rlrep_dims& d, Layout& L) {
    const int S = d.state_dim, A = d.action_dim, F = d.feature_dim;
    const int P = RLREP_ARENA_PARAM, T = RLREP_ARENA_TARGET;
    Mlp phi{"phi.trunk", false, S + A, d.phi_hidden_dim, F, d.phi_hidden_depth};
    Mlp mu{"mu.trunk", false, S, d.mu_hidden_dim, F, d.mu_hidden_depth};
    Mlp phit{"phi_target.trunk", true, S + A, d.phi_hidden_dim, F, d.phi_hidden_depth};
    L.begin_group(0); lay_mlp(L, phi, P, 0); lay_mlp(L, mu, P, 0); L.lin("theta.l", 1, F, P, 0); L.end_group(0);
    L.begin_group(1); lay_six(L, "critic", F, d.hidden_dim, P, 1); L.end_group(1);
    L.begin_group(2); lay_actor(L, S, A, d.actor_hidden_dim, P, 2); L.end_group(2);
    lay_mlp(L, phit, T, -1);
    lay_six(L, "critic_target", F, d.hidden_dim, T, -1);
}

void lay_diffsrsac(const rlrep_dims& d, Layout& L) {
    const int S = d.state_dim, A = d.action_dim, F = d.feature_dim;
    const int P = RLREP_ARENA_PARAM, T = RLREP_ARENA_TARGET;
    Mlp phi{"critic_feed_feature.z_vector", false, S + A, d.phi_hidden_dim, F, d.phi_hidden_depth};
    Mlp nm{"nablamu_net.Mu_z_by_s_layer", false, S + 1, d.mu_hidden_dim, F * S, d.mu_hidden_depth};
    L.begin_group(0); lay_mlp(L, phi, P, 0); L.end_group(0);
    L.begin_group(3); lay_mlp(L, nm, P, 3); L.end_group(3);
    // quirk Q11: the RFF critic is never trained; it lives in the parameter arena (group 1) but no Adam runs on it
    L.begin_group(1); lay_six(L, "critic", F, d.hidden_dim, P, 1); L.end_group(1);
    L.begin_group(2); lay_actor(L, S, A, d.actor_hidden_dim, P, 2); L.end_group(2);
    lay_six(L, "critic_target", F, d.hidden_dim, T, -1);
    L.add("noise_alphabars", d.num_noise, 1, T, -1);
}

// ================================================================================================
// shared: qhead stages
// ================================================================================================
static void qhead_critic_stage(Program& p, rlrep_agent* ag, const float* Et0, const float* Et1, const float* Ec0, const float* Ec1, int ldE,
                               const float* wt0, const float* bt0, const float* wt1, const float* bt1,
                               const float* wc0, const float* bc0, const float* wc1, const float* bc1,
                               const float* logp, float* dq, float* GE0, float* GE1, float* part_q, int H, int nblk, int train) {
    QHeadCritic q; memset(&q, 0, sizeof(q));
    q.Et[0] = Et0; q.Et[1] = Et1; q.Ec[0] = Ec0; q.Ec[1] = Ec1; q.ldE = ldE;
    q.wt[0] = wt0; q.wt[1] = wt1; q.bt[0] = bt0; q.bt[1] = bt1; q.wc[0] = wc0; q.wc[1] = wc1; q.bc[0] = bc0; q.bc[1] = bc1;
    q.logp = logp; q.R = ag->slot[0].R; q.D = ag->slot[0].D; q.alpha_state = ag->a.alpha_state_dev; q.gamma = ag->h.discount;
    q.inv_batch = ag->inv_batch(); q.dq = dq; q.GE[0] = GE0; q.GE[1] = GE1; q.partial = part_q;
    q.B = ag->B; q.H = H; q.nblk = nblk; q.train = train; q.step = train ? ag->adam_step + 1 : nullptr;
    p.stages.push_back({[=](hipStream_t st) { return rl_launch_qhead_critic(&q, st); }, "qhead critic"});
}
static void qhead_actor_stage(Program& p, rlrep_agent* ag, const float* Ec0, const float* Ec1, int ldE, const float* wc0, const float* bc0,
                              const float* wc1, const float* bc1, const float* logp, float* GE0, float* GE1, float* part_l, int H, int nblk) {
    QHeadActor q; memset(&q, 0, sizeof(q));
    q.Ec[0] = Ec0; q.Ec[1] = Ec1; q.ldE = ldE; q.wc[0] = wc0; q.wc[1] = wc1; q.bc[0] = bc0; q.bc[1] = bc1;
    q.logp = logp; q.alpha_state = ag->a.alpha_state_dev; q.inv_batch = ag->inv_batch(); q.target_entropy = ag->h.target_entropy;
    q.GE[0] = GE0; q.GE[1] = GE1; q.partial_loss = part_l; q.partial_c = ag->Gtail();
    q.B = ag->B; q.H = H; q.nblk = nblk; q.step = ag->adam_step + 2;
    p.stages.push_back({[=](hipStream_t st) { return rl_launch_qhead_actor(&q, st); }, "qhead actor"});
}
static std::vector<FinTask> critic_fins(rlrep_agent* ag, const float* part_q, int nblk) {
    const float ib = 1.0f / (float)ag->B;
    float* m = ag->metrics;
    return {Builder::fin_sum(part_q + 0, nblk, 4, ib, m + M_Q1_LOSS), Builder::fin_sum(part_q + 1, nblk, 4, ib, m + M_Q2_LOSS),
            Builder::fin_sum(part_q + 2, nblk, 4, ib, m + M_Q1), Builder::fin_sum(part_q + 3, nblk, 4, ib, m + M_Q2)};
}

// ================================================================================================
// CTRLSAC  (agent/ctrlsac/ctrlsac_agent.py:213-361)
// ================================================================================================
void build_ctrlsac(Builder& b, rlrep_agent* ag) {
    const rlrep_dims& d = ag->d;
    const int S = d.state_dim, A = d.action_dim, H = d.hidden_dim, Ha = d.actor_hidden_dim, F = d.feature_dim, B = ag->B;
    const int Hp = d.phi_hidden_dim, Hm = d.mu_hidden_dim, SA = S + A, KE = 2 * S + A;
    Slot& s0 = ag->slot[0];
    Workspace& ws = b.ws;
    auto Pw = [&](const char* n) { return ag->P(n); };
    auto Tw = [&](const char* n) { return ag->T(n); };
    auto Gw = [&](const char* n) { return ag->G(n); };
    const float* s2 = s0.XE ? s0.XE + SA : nullptr;                 // next_state columns of the [s,a,s'] matrix

    // phi as a 3-layer network with the reference's l1/l2/l3 names
    struct Phi3 { float *P1, *P2, *Z, *G2, *G1, *GZ; };
    auto alloc_phi = [&](bool grad) {
        Phi3 r; r.P1 = ws.f((size_t)B * Hp); r.P2 = ws.f((size_t)B * Hp); r.Z = ws.f((size_t)B * F);
        r.G2 = grad ? ws.f((size_t)B * Hp) : nullptr; r.G1 = grad ? ws.f((size_t)B * Hp) : nullptr; r.GZ = grad ? ws.f((size_t)B * F) : nullptr;
        return r;
    };
    auto phi_fwd = [&](int l, const float* X, const Phi3& r) {
        if (l == 0) return Builder::fwd(X, SA, B, SA, Pw("phi.l1.weight"), SA, Pw("phi.l1.bias"), Hp, r.P1, Hp, ACT_ELU);
        if (l == 1) return Builder::fwd(r.P1, Hp, B, Hp, Pw("phi.l2.weight"), Hp, Pw("phi.l2.bias"), Hp, r.P2, Hp, ACT_ELU);
        return Builder::fwd(r.P2, Hp, B, Hp, Pw("phi.l3.weight"), Hp, Pw("phi.l3.bias"), F, r.Z, F, ACT_NONE);
    };

    // ---- feature step ----
    Phi3 pf = alloc_phi(true);
    // data parallel: the in-batch negatives span ALL ranks' minibatches.  mu(s') of every rank is
    // all-gathered into ZMall [W*B, F] (this rank's rows sit at rank*B), the score matrix is [B, W*B], and the
    // partial dL/dmu'_all [W*B, F] is all-reduced before this rank back-propagates its own B rows.
    const int Wd = ag->h.world_size > 1 ? ag->h.world_size : 1, rank = Wd > 1 ? d.rank : 0, WB = Wd * B;
    // attached (rlrep_comm_attach with exchange scratch, dp_pull.h): mu(s') of all ranks and the partial dmu'_all live in the shared block, and
    // the two exchanges are ONE pull launch each inside the step program (comm.hip comm_gather_kernel / comm_pull_kernel) -- no cut, no host
    const bool xf = ag->xfold && Wd > 1 && (((long long)B * F) & 3) == 0 && 2ll * WB * F <= ag->xscratch_floats;
    float* M1 = ws.f((size_t)B * Hm); float* M2 = ws.f((size_t)B * Hm);
    float* ZMall = xf ? ag->xscratch[rank] : ws.f((size_t)WB * F);
    float* ZM = ZMall ? ZMall + (size_t)rank * B * F : nullptr;
    float* GM2 = ws.f((size_t)B * Hm); float* GM1 = ws.f((size_t)B * Hm);
    float* GZMall = xf ? ag->xscratch[rank] + (size_t)WB * F : ws.f((size_t)WB * F);
    float* GZM = GZMall ? GZMall + (size_t)rank * B * F : nullptr;
    rlrep_agent* const agp = ag;
    float* Sx = ws.f((size_t)B * WB); float* RH = ws.f(B); float* DRH = ws.f(B);
    const int nblk_f = qhead_blocks(B);
    float* part_f = ws.f((size_t)2 * nblk_f);
    {
        Program& p = ag->feat_bwd;
        b.fwd_stage(p, {phi_fwd(0, s0.XF, pf), Builder::fwd(s2, KE, B, S, Pw("mu.l1.weight"), S, Pw("mu.l1.bias"), Hm, M1, Hm, ACT_ELU)}, "phi.l1 mu.l1");
        b.fwd_stage(p, {phi_fwd(1, nullptr, pf), Builder::fwd(M1, Hm, B, Hm, Pw("mu.l2.weight"), Hm, Pw("mu.l2.bias"), Hm, M2, Hm, ACT_ELU)}, "phi.l2 mu.l2");
        b.fwd_stage(p, {phi_fwd(2, nullptr, pf), Builder::fwd(M2, Hm, B, Hm, Pw("mu.l3.weight"), Hm, Pw("mu.l3.bias"), F, ZM, F, ACT_TANH)}, "phi.l3 mu.l3(tanh)");
        // quirk Q6: the score matrix is the GEMM phi mu'^T, not the [B,B,F] broadcast
        if (xf) {
            const long long off = ag->xarena_floats, nseg = (long long)B * F;
            // (no DONE round trip in either exchange launch: this rank's mu(s') rows are next written a whole feature step later, behind the reduce-scatter
            //  below, whose READY a peer sends only after its gather has completed; and its dmu'_all behind the NEXT step's gather, likewise)
            p.stages.push_back({[=](hipStream_t st) { return rl_launch_xchg_gather(&agp->dp_proto, 4, off, nseg, 1, st); }, "all-gather mu(s') (pull)"});
        } else
        if (Wd > 1) ag->feat_cuts.push_back({(int)p.stages.size() - 1, 1, ZMall, (int64_t)B * F, (int64_t)rank * B * F});
        // rhat = theta . phi + b (a [B, 1] product) is computed by the InfoNCE launch itself: beside a score matrix that routes to the LDS-tiled engine it was a
        // 16-row-engine launch of its own on the dependent chain (8 us per feature step at F = 2048); RLREP_DISABLE=fold_theta keeps it
        const bool theta_in_loss = !rl_off("fold_theta");
        // K12: for the small products (BASELINE config 3: F = 256, 256 columns) the score matrix is computed BY the InfoNCE launch, 16 whole rows per
        // workgroup (replearn.hip score_infonce_kernel): one dependent launch less per feature step, S never in memory.  RLREP_DISABLE=fuse_infonce: the pair.
        // (At most 256 columns: four column tiles per wave in static registers -- the data-parallel forms, whose matrix is [B, W B], keep the pair.)
        // OPT-IN (RLREP_ENABLE=fuse_infonce): measured SLOWER -- 3 546 against 4 150 train()/s at config 3: the 64 waves of 16 workgroups run the 256 MFMAs
        // per wave that 1 024 waves of the GEMM launch share, ~21 us against 11 for the pair (docs/history/r06.md).  The arithmetic K12 was declined with, confirmed.
        const bool fuse_score = theta_in_loss && (F & 63) == 0 && F <= 512 && WB <= 256 && rl_opt("fuse_infonce") != nullptr;
        const int nblk_loss = fuse_score ? (B + 15) / 16 : nblk_f;
        if (fuse_score) { /* no score-matrix stage */ }
        else if (theta_in_loss) b.fwd_stage(p, {Builder::fwd(pf.Z, F, B, F, ZMall, F, nullptr, WB, Sx, WB, ACT_NONE)}, "score matrix");
        else
        b.fwd_stage(p, {Builder::fwd(pf.Z, F, B, F, ZMall, F, nullptr, WB, Sx, WB, ACT_NONE),
                        Builder::fwd(pf.Z, F, B, F, Pw("theta.l.weight"), F, Pw("theta.l.bias"), 1, RH, 1, ACT_NONE)}, "score matrix + theta");
        InfoNce nc; memset(&nc, 0, sizeof(nc));
        if (theta_in_loss) { nc.Z = pf.Z; nc.ldZ = F; nc.F = F; nc.theta_w = Pw("theta.l.weight"); nc.theta_b = Pw("theta.l.bias"); }
        nc.S = Sx; nc.ldS = WB; nc.ncols = WB; nc.diag_off = rank * B; nc.rhat = RH; nc.r = s0.R; nc.drhat = DRH; nc.partial = part_f; nc.B = B; nc.nblk = nblk_loss;
        nc.inv_batch = ag->inv_batch(); nc.step = ag->adam_step + 0;
        if (fuse_score) { nc.ZM = ZMall; nc.ldZM = F; }
        p.stages.push_back({[=](hipStream_t st) { return rl_launch_infonce(&nc, st); }, fuse_score ? "score matrix + infonce" : "infonce"});
        if (fuse_score) Builder::tag(p, RLREP_ENGINE_GEMM16, 2.0 * (double)B * (double)WB * (double)F, 4.0 * ((double)B * F + (double)WB * F + (double)B * WB));
        {
            GemmTask t = Builder::dx(Sx, WB, B, WB, ZMall, F, pf.GZ, F, F, ACT_NONE, nullptr, 0);
            t.r1u = DRH; t.r1v = Pw("theta.l.weight");
            GemmTask u = Builder::base();     // dmu'[j,f] = (sum_i dS[i,j] phi[i,f]) * (1 - mu'^2)
            u.A = Sx; u.lda = WB; u.B = pf.Z; u.ldb = F; u.C = GZMall; u.ldc = F; u.R = WB; u.Cn = F; u.K = B;
            u.epi = EPI_DX; u.act = ACT_TANH; u.aux = ZMall; u.ldaux = F;
            if (Wd > 1 && !xf) {          // data parallel over torch.distributed: the all-reduce of dmu'_all sits right behind its own stage
                b.dx_stage(p, {t}, "dphi = dS mu' + drhat theta");
                b.gemm(p, LD_COL, LD_COL, {u}, "dmu'_all = dS^T phi");
                ag->feat_cuts.push_back({(int)p.stages.size() - 1, 2, GZMall, (int64_t)WB * F, 0});
            } else {
                // both products consume the InfoNCE gradient dS and nothing of each other: ONE launch of two tile forms (gemm16_duo_kernel)
                b.gemm_duo(p, {t}, {u}, "dphi = dS mu' + drhat theta", "dmu'_all = dS^T phi", "dphi = dS mu' + drhat theta | dmu'_all = dS^T phi");
                if (xf) {
                    // this rank needs only ITS rows of the sum: a reduce-scatter, in place (peers read their own rows of this block)
                    const long long off = ag->xarena_floats + (long long)WB * F + (long long)rank * B * F, nseg = (long long)B * F;
                    float* out = GZM;
                    p.stages.push_back({[=](hipStream_t st) { return rl_launch_xchg_reduce(&agp->dp_proto, 5, off, nseg, out, 0, 1, st); }, "reduce-scatter dmu' (pull)"});
                }
            }
        }
        b.dx_stage(p, {Builder::dx(pf.GZ, F, B, F, Pw("phi.l3.weight"), Hp, pf.G2, Hp, Hp, ACT_ELU, pf.P2, Hp),
                       Builder::dx(GZM, F, B, F, Pw("mu.l3.weight"), Hm, GM2, Hm, Hm, ACT_ELU, M2, Hm)}, "l3 dx");
        b.dx_stage(p, {Builder::dx(pf.G2, Hp, B, Hp, Pw("phi.l2.weight"), Hp, pf.G1, Hp, Hp, ACT_ELU, pf.P1, Hp),
                       Builder::dx(GM2, Hm, B, Hm, Pw("mu.l2.weight"), Hm, GM1, Hm, Hm, ACT_ELU, M1, Hm)}, "l2 dx");
        b.dw_stage(p, {Builder::dw(pf.GZ, F, F, pf.P2, Hp, Hp, B, Gw("phi.l3.weight"), Hp, Gw("phi.l3.bias")),
                       Builder::dw(GZM, F, F, M2, Hm, Hm, B, Gw("mu.l3.weight"), Hm, Gw("mu.l3.bias")),
                       Builder::dw(pf.G2, Hp, Hp, pf.P1, Hp, Hp, B, Gw("phi.l2.weight"), Hp, Gw("phi.l2.bias")),
                       Builder::dw(GM2, Hm, Hm, M1, Hm, Hm, B, Gw("mu.l2.weight"), Hm, Gw("mu.l2.bias")),
                       Builder::dw(pf.G1, Hp, Hp, s0.XF, SA, SA, B, Gw("phi.l1.weight"), SA, Gw("phi.l1.bias")),
                       Builder::dw(GM1, Hm, Hm, s2, KE, S, B, Gw("mu.l1.weight"), S, Gw("mu.l1.bias")),
                       Builder::dw(DRH, 1, 1, pf.Z, F, F, B, Gw("theta.l.weight"), F, Gw("theta.l.bias"))}, "feature dW");
        const LT& p0 = ag->L.get("phi.l1.weight");
        const LT& pl = ag->L.get("phi.l3.bias");
        float* m = ag->metrics;
        // use_feature_target=False (ctrlsac_agent.py:340-346): no Polyak into phi_target, frozen_phi_target is not written
        const bool nft = (ag->d.flags & RLREP_FLAG_NO_FEATURE_TARGET) != 0;
        b.adam(ag->feat_apply, 0, ag->h.lr_feature, nft ? nullptr : Tw("phi_target.l1.weight"), p0.off, pl.off + pl.rows - p0.off, ag->h.feature_tau,
               {Builder::fin_sum(part_f + 0, nblk_loss, 2, 1.0f / (float)B, m + M_FEAT_A),
                Builder::fin_sum(part_f + 1, nblk_loss, 2, 0.5f / (float)B, m + M_R_LOSS),
                Builder::fin_combine(m + M_FEAT_A, 1.f, m + M_R_LOSS, 1.f, m + M_FEAT_TOTAL)}, "adam feature + polyak phi");
        // ctrlsac_agent.py:344-346: frozen_phi, frozen_phi_target <- phi (quirk Q8)
        const long long pn = pl.off + pl.rows - p0.off;
        float* src = Pw("phi.l1.weight"); float* d1 = Tw("frozen_phi.l1.weight"); float* d2 = nft ? nullptr : Tw("frozen_phi_target.l1.weight");
        ag->sync_prog.stages.push_back({[=](hipStream_t st) { return rl_launch_copy2(src, d1, d2, pn, st); }, "frozen_phi* <- phi"});
    }

    // ---- critic / actor ----
    ActorBufs ab = alloc_actor(b, B, A, Ha);
    Phi3 pa = alloc_phi(true), pb = alloc_phi(false);
    float* Et = ws.f((size_t)B * 2 * H); float* Ec = ws.f((size_t)B * 2 * H); float* GE = ws.f((size_t)B * 2 * H);
    float* dq = ws.f((size_t)2 * B);
    const int nblk = qhead_blocks(B);
    float* part_q = ws.f((size_t)4 * nblk); float* part_l = ws.f(nblk);
    auto critic_program = [&](Program& p) {
        // quirk Q8: frozen_phi_target == frozen_phi == phi at this point of train(); the programs read phi directly
        b.fwd_stage(p, {actor_l(ag, 0, s0.XF2, SA, ab), phi_fwd(0, s0.XF, pa)}, "actor.l1(s') phi.l1(s,a)");
        b.fwd_stage(p, {actor_l(ag, 1, nullptr, 0, ab), phi_fwd(1, nullptr, pa)}, "actor.l2 phi.l2");
        actor_head_stage(b, p, ag, ab, s0.XF2 + S, SA, {phi_fwd(2, nullptr, pa)}, "actor.head phi.l3 + policy");
        b.fwd_stage(p, {phi_fwd(0, s0.XF2, pb)}, "phi.l1(s',a')");
        b.fwd_stage(p, {phi_fwd(1, nullptr, pb)}, "phi.l2");
        b.fwd_stage(p, {phi_fwd(2, nullptr, pb)}, "phi.l3");
        b.fwd_stage(p, {Builder::fwd(pb.Z, F, B, F, Tw("critic_target.l1.weight"), F, Tw("critic_target.l1.bias"), 2 * H, Et, 2 * H, ACT_ELU),
                        Builder::fwd(pa.Z, F, B, F, Pw("critic.l1.weight"), F, Pw("critic.l1.bias"), 2 * H, Ec, 2 * H, ACT_ELU)}, "critic l1|l4");
        qhead_critic_stage(p, ag, Et, Et + H, Ec, Ec + H, 2 * H, Tw("critic_target.l2.weight"), Tw("critic_target.l2.bias"),
                           Tw("critic_target.l5.weight"), Tw("critic_target.l5.bias"), Pw("critic.l2.weight"), Pw("critic.l2.bias"),
                           Pw("critic.l5.weight"), Pw("critic.l5.bias"), ab.logp, dq, GE, GE + H, part_q, H, nblk, 1);
        b.dw_stage(p, {Builder::dw(dq, 1, 1, Ec, 2 * H, H, B, Gw("critic.l2.weight"), H, Gw("critic.l2.bias")),
                       Builder::dw(dq + B, 1, 1, Ec + H, 2 * H, H, B, Gw("critic.l5.weight"), H, Gw("critic.l5.bias")),
                       Builder::dw(GE, 2 * H, 2 * H, pa.Z, F, F, B, Gw("critic.l1.weight"), F, Gw("critic.l1.bias"))}, "critic dW");
    };
    auto actor_program = [&](Program& p) {
        b.fwd_stage(p, {actor_l(ag, 0, s0.XFpi, SA, ab)}, "actor.l1(s)");
        b.fwd_stage(p, {actor_l(ag, 1, nullptr, 0, ab)}, "actor.l2");
        actor_head_stage(b, p, ag, ab, s0.XFpi + S, SA, {}, "actor.head + policy");
        b.fwd_stage(p, {phi_fwd(0, s0.XFpi, pa)}, "phi.l1(s,a_pi)");
        b.fwd_stage(p, {phi_fwd(1, nullptr, pa)}, "phi.l2");
        b.fwd_stage(p, {phi_fwd(2, nullptr, pa)}, "phi.l3");
        b.fwd_stage(p, {Builder::fwd(pa.Z, F, B, F, Pw("critic.l1.weight"), F, Pw("critic.l1.bias"), 2 * H, Ec, 2 * H, ACT_ELU)}, "critic l1|l4");
        qhead_actor_stage(p, ag, Ec, Ec + H, 2 * H, Pw("critic.l2.weight"), Pw("critic.l2.bias"), Pw("critic.l5.weight"), Pw("critic.l5.bias"),
                          ab.logp, GE, GE + H, part_l, H, nblk);
        b.dx_stage(p, {Builder::dx(GE, 2 * H, B, 2 * H, Pw("critic.l1.weight"), F, pa.GZ, F, F, ACT_NONE, nullptr, 0)}, "critic l1|l4 dx");
        b.dx_stage(p, {Builder::dx(pa.GZ, F, B, F, Pw("phi.l3.weight"), Hp, pa.G2, Hp, Hp, ACT_ELU, pa.P2, Hp)}, "phi.l3 dx");
        b.dx_stage(p, {Builder::dx(pa.G2, Hp, B, Hp, Pw("phi.l2.weight"), Hp, pa.G1, Hp, Hp, ACT_ELU, pa.P1, Hp)}, "phi.l2 dx");
        actor_backward(b, p, ag, ab, s0.XFpi, SA, s0.XFpi + S, SA, Builder::dx(pa.G1, Hp, B, Hp, Pw("phi.l1.weight") ? Pw("phi.l1.weight") + S : nullptr, SA, ab.dA, A, A, ACT_NONE, nullptr, 0));
    };
    critic_program(ag->critic_bwd);
    b.adam(ag->critic_apply, 1, ag->h.lr_critic, nullptr, 0, 0, 0.f, critic_fins(ag, part_q, nblk), "adam critic");
    critic_apply_folded(b, ag, "critic_target.l1.weight", critic_fins(ag, part_q, nblk));
    actor_program(ag->actor_bwd);
    actor_apply_program(b, ag, part_l, nblk);
    update_target_program(ag, "critic.l1.weight", "critic_target.l1.weight");
    // deferred variants: the critic / actor programs against a snapshot set (phi copy + minibatch copy); the snapshot launch also runs
    // sync_prog (frozen_phi* <- phi), which belongs to the end of the feature steps
    if (ag->h.world_size <= 1 || xf) {
        const LT& q0 = ag->L.get("phi.l1.weight");
        const LT& ql = ag->L.get("phi.l3.bias");
        for (int set = 0; set < rlrep_agent::NSETS; ++set) {
            const Slot keep = defer_begin(b, ag, set, "phi.", "phi.l1.weight", Pw("phi.l1.weight"), ql.off + ql.rows - q0.off);
            critic_program(ag->dset[set].critic_bwd);
            actor_program(ag->dset[set].actor_bwd);
            ag->dset[set].actor_resume = 0;
            defer_end(b, ag, set, keep, "critic_target.l1.weight", critic_fins(ag, part_q, nblk));
        }
    }
}

// ================================================================================================
// RFF-critic agents share the critic / actor construction (feature map = an Mlp on [s,a])
// ================================================================================================
static void build_rff_critic_actor(Builder& b, rlrep_agent* ag, const Mlp& phi, bool train_critic) {
    const rlrep_dims& d = ag->d;
    const int S = d.state_dim, A = d.action_dim, H = d.hidden_dim, Ha = d.actor_hidden_dim, F = d.feature_dim, B = ag->B;
    const int SA = S + A;
    Slot& s0 = ag->slot[0];
    Workspace& ws = b.ws;
    auto Pw = [&](const char* n) { return ag->P(n); };
    auto Tw = [&](const char* n) { return ag->T(n); };
    auto Gw = [&](const char* n) { return ag->G(n); };
    ActorBufs ab = alloc_actor(b, B, A, Ha);
    MlpBufs pa = alloc_mlp(b, phi, B, true), pb = alloc_mlp(b, phi, B, false);
    RffBufs rt = alloc_rff(b, B, H, false), rc = alloc_rff(b, B, H, true);
    float* dq = ws.f((size_t)2 * B); float* GZ = ws.f((size_t)B * F);
    const int nblk = qhead_blocks(B);
    float* part_q = ws.f((size_t)4 * nblk); float* part_l = ws.f(nblk);
    const size_t BH = (size_t)B * H;
    const int D = phi.depth;
    float* Zc = pa.act[D]; float* Zn = pb.act[D];
    auto critic_program = [&](Program& p, Program& papply, bool emit_apply) {
        // actor(s') next to phi(s,a): layer l of both in one launch while both have a layer l
        for (int l = 0; l < 3 || l <= D; ++l) {
            std::vector<GemmTask> t;
            if (l < 2) t.push_back(actor_l(ag, l, s0.XF2, SA, ab));
            if (l <= D) t.push_back(mlp_fwd(ag, phi, pa, l, s0.XF, SA));
            if (l == 2) actor_head_stage(b, p, ag, ab, s0.XF2 + S, SA, t, "actor.head(s') + policy / phi(s,a) layer");
            else b.fwd_stage(p, t, "actor(s') / phi(s,a) layer");
        }
        for (int l = 0; l <= D; ++l) b.fwd_stage(p, {mlp_fwd(ag, phi, pb, l, s0.XF2, SA)}, "phi(s',a') layer");
        b.fwd_stage(p, {rff_l1(ag, true, Zn, F, H, rt), rff_l1(ag, false, Zc, F, H, rc)}, "critic l1|l4 (sin)");
        {
            std::vector<GemmTask> t; rff_l2(ag, true, H, rt, t); rff_l2(ag, false, H, rc, t);
            b.fwd_stage(p, t, "critic l2/l5");
        }
        qhead_critic_stage(p, ag, rt.E, rt.E + BH, rc.E, rc.E + BH, H, Tw("critic_target.l3.weight"), Tw("critic_target.l3.bias"),
                           Tw("critic_target.l6.weight"), Tw("critic_target.l6.bias"), Pw("critic.l3.weight"), Pw("critic.l3.bias"),
                           Pw("critic.l6.weight"), Pw("critic.l6.bias"), ab.logp, dq, rc.GE, rc.GE + BH, part_q, H, nblk, train_critic ? 1 : 0);
        if (train_critic) {
            b.dx_stage(p, {Builder::dx(rc.GE, H, B, H, Pw("critic.l2.weight"), H, rc.G1, 2 * H, H, ACT_SIN, rc.PRE1, 2 * H),
                           Builder::dx(rc.GE + BH, H, B, H, Pw("critic.l5.weight"), H, rc.G1 + H, 2 * H, H, ACT_SIN, rc.PRE1 + H, 2 * H)}, "critic l2/l5 dx");
            b.dw_stage(p, {Builder::dw(dq, 1, 1, rc.E, H, H, B, Gw("critic.l3.weight"), H, Gw("critic.l3.bias")),
                           Builder::dw(dq + B, 1, 1, rc.E + BH, H, H, B, Gw("critic.l6.weight"), H, Gw("critic.l6.bias")),
                           Builder::dw(rc.GE, H, H, rc.S1, 2 * H, H, B, Gw("critic.l2.weight"), H, Gw("critic.l2.bias")),
                           Builder::dw(rc.GE + BH, H, H, rc.S1 + H, 2 * H, H, B, Gw("critic.l5.weight"), H, Gw("critic.l5.bias")),
                           Builder::dw(rc.G1, 2 * H, 2 * H, Zc, F, F, B, Gw("critic.l1.weight"), F, Gw("critic.l1.bias"))}, "critic dW");
            if (emit_apply) {
                b.adam(papply, 1, ag->h.lr_critic, nullptr, 0, 0, 0.f, critic_fins(ag, part_q, nblk), "adam critic");
                critic_apply_folded(b, ag, "critic_target.l1.weight", critic_fins(ag, part_q, nblk));
            }
        } else if (emit_apply) {
            // diffsrsac (quirk Q11): metrics only; q2 := q1 (Q13).  q_loss_noreg = mse1 + mse2; q_loss_reg adds the ELU-layer regulariser
            // of BOTH nets' two heads (diffsrsac_agent.py:62-90, 215-227), which is lambda * (...) = 0 at the default lambda (Q12)
            const float ib = 1.0f / (float)B;
            float* m = ag->metrics;
            std::vector<FinTask> fins = {Builder::fin_sum(part_q + 0, nblk, 4, ib, m + M_TMP0), Builder::fin_sum(part_q + 1, nblk, 4, ib, m + M_TMP1),
                                         Builder::fin_combine(m + M_TMP0, 1.f, m + M_TMP1, 1.f, m + M_Q2_LOSS),
                                         Builder::fin_sum(part_q + 2, nblk, 4, ib, m + M_Q1), Builder::fin_copy(m + M_Q1, m + M_Q2)};
            // (the regulariser's buffers are reserved whatever lambda is: rlrep_layout sizes the workspace from the dimensions alone)
            float* XR = ws.f((size_t)4 * BH); float* CR = ws.f((size_t)4 * H * H);
            RegStats rs; memset(&rs, 0, sizeof(rs));
            rs.B = B; rs.H = H; rs.lambda = ag->h.critic_reg_lambda;
            rs.nbc = (int)(((long long)H * H + 1023) / 1024); rs.nbr = (B + 3) / 4;
            float* part_r = ws.f((size_t)4 * (rs.nbc + rs.nbr));
            rs.partial = part_r;
            {
                // x = l2(E) once more on the ELU outputs (no activation), C = x^T x through the weight-gradient form, then the statistics.
                // With lambda = 0 the stages are built into a scratch program and dropped: the builder's workspace reservations (split-K
                // slabs) then stay a function of the dimensions alone
                Program scratch;
                const bool reg_on = ag->h.critic_reg_lambda != 0.f;
                Program& pr = reg_on ? p : scratch;
                std::vector<GemmTask> fw, gr;
                for (int k = 0; k < 4; ++k) {
                    const bool tgt = k < 2; const int head = k & 1;
                    const RffBufs& r = tgt ? rt : rc;
                    const std::string mname = tgt ? "critic_target" : "critic";
                    auto w = [&](const char* nm) { return tgt ? ag->T(mname + nm) : ag->P(mname + nm); };
                    float* xk = XR ? XR + (size_t)k * BH : nullptr; float* ck = CR ? CR + (size_t)k * H * H : nullptr;
                    fw.push_back(Builder::fwd(r.E ? r.E + (size_t)head * BH : nullptr, H, B, H, w(head ? ".l5.weight" : ".l2.weight"), H, w(head ? ".l5.bias" : ".l2.bias"), H, xk, H, ACT_NONE));
                    gr.push_back(Builder::dw(xk, H, H, xk, H, H, B, ck, H, nullptr));
                    rs.X[k] = xk; rs.C[k] = ck;
                }
                b.fwd_stage(pr, fw, "critic regulariser: l2/l5 on the ELU outputs");
                b.gemm(pr, LD_COL, LD_COL, gr, "critic regulariser: Gram matrices");
                pr.stages.push_back({[=](hipStream_t st) { return rl_launch_reg_stats(&rs, st); }, "critic regulariser statistics"});
                if (reg_on) {
                    fins.push_back(Builder::fin_sum(part_r, 4 * (rs.nbc + rs.nbr), 1, 1.0f, m + M_TMP2));
                    fins.push_back(Builder::fin_combine(m + M_Q2_LOSS, 1.f, m + M_TMP2, 1.f, m + M_Q1_LOSS));
                } else {
                    fins.push_back(Builder::fin_copy(m + M_Q2_LOSS, m + M_Q1_LOSS));
                    fins.push_back(Builder::fin_copy(m + M_Q2_LOSS, m + M_Q1_LOSS));       // (same table size with and without the regulariser)
                }
            }
            b.finalize_only(papply, fins, "critic metrics");
        }
    };
    auto actor_program = [&](Program& p) {
        b.fwd_stage(p, {actor_l(ag, 0, s0.XFpi, SA, ab)}, "actor.l1(s)");
        b.fwd_stage(p, {actor_l(ag, 1, nullptr, 0, ab)}, "actor.l2");
        actor_head_stage(b, p, ag, ab, s0.XFpi + S, SA, {}, "actor.head + policy");
        for (int l = 0; l <= D; ++l) b.fwd_stage(p, {mlp_fwd(ag, phi, pa, l, s0.XFpi, SA)}, "phi(s,a_pi) layer");
        b.fwd_stage(p, {rff_l1(ag, false, Zc, F, H, rc)}, "critic l1|l4 (sin)");
        {
            std::vector<GemmTask> t; rff_l2(ag, false, H, rc, t);
            b.fwd_stage(p, t, "critic l2/l5");
        }
        qhead_actor_stage(p, ag, rc.E, rc.E + BH, H, Pw("critic.l3.weight"), Pw("critic.l3.bias"), Pw("critic.l6.weight"), Pw("critic.l6.bias"),
                          ab.logp, rc.GE, rc.GE + BH, part_l, H, nblk);
        b.dx_stage(p, {Builder::dx(rc.GE, H, B, H, Pw("critic.l2.weight"), H, rc.G1, 2 * H, H, ACT_SIN, rc.PRE1, 2 * H),
                       Builder::dx(rc.GE + BH, H, B, H, Pw("critic.l5.weight"), H, rc.G1 + H, 2 * H, H, ACT_SIN, rc.PRE1 + H, 2 * H)}, "critic l2/l5 dx");
        // dL/dz = G1 [W1;W4]; when phi has hidden layers the elu' of its last hidden activation is NOT applied here
        b.dx_stage(p, {Builder::dx(rc.G1, 2 * H, B, 2 * H, Pw("critic.l1.weight"), F, pa.g[D], F, F, ACT_NONE, nullptr, 0)}, "critic l1|l4 dx");
        for (int l = D; l >= 1; --l) b.dx_stage(p, {mlp_dx(ag, phi, pa, l)}, "phi dx");
        actor_backward(b, p, ag, ab, s0.XFpi, SA, s0.XFpi + S, SA, mlp_dx_input(ag, phi, pa, S, A, ab.dA, A));
    };
    critic_program(ag->critic_bwd, ag->critic_apply, true);
    actor_program(ag->actor_bwd);
    actor_apply_program(b, ag, part_l, nblk);
    (void)GZ;
    update_target_program(ag, "critic.l1.weight", "critic_target.l1.weight");
    // deferred variants (spedersac: critic and actor read the LIVE phi, so the snapshot carries a copy of it)
    if (train_critic && (ag->h.world_size <= 1 || ag->xfold)) {
        const LT& q0 = ag->L.get(phi.name(0) + ".weight");
        const LT& ql = ag->L.get(phi.name(phi.depth) + ".bias");
        const std::string pre = phi.prefix + ".", first = phi.name(0) + ".weight";
        for (int set = 0; set < rlrep_agent::NSETS; ++set) {
            const Slot keep = defer_begin(b, ag, set, pre.c_str(), first.c_str(), ag->P(first), ql.off + ql.rows - q0.off);
            Program unused;
            critic_program(ag->dset[set].critic_bwd, unused, false);
            actor_program(ag->dset[set].actor_bwd);
            ag->dset[set].actor_resume = 0;
            defer_end(b, ag, set, keep, "critic_target.l1.weight", critic_fins(ag, part_q, nblk));
        }
    }
}

// ================================================================================================
// SPEDERSAC  (agent/spedersac/spedersac_agent.py:181-322)
// ================================================================================================
void build_spedersac(Builder& b, rlrep_agent* ag) {
    const rlrep_dims& d = ag->d;
    const int S = d.state_dim, A = d.action_dim, F = d.feature_dim, B = ag->B;
    const int SA = S + A, KE = 2 * S + A;
    Slot& s0 = ag->slot[0];
    Workspace& ws = b.ws;
    Mlp phi{"phi.trunk", false, SA, d.phi_hidden_dim, F, d.phi_hidden_depth};
    Mlp mu{"mu.trunk", false, S, d.mu_hidden_dim, F, d.mu_hidden_depth};
    // both minibatches as one 2B-row problem (slot buffers are contiguous, engine.hip build_programs)
    MlpBufs pf = alloc_mlp(b, phi, 2 * B, true), mf = alloc_mlp(b, mu, 2 * B, true);
    const float* X2 = s0.XF;                                   // [2B, S+A]
    const float* S2 = s0.XE ? s0.XE + SA : nullptr;            // next_state columns, row stride 2S+A, 2B rows
    float* PHI = pf.act[phi.depth]; float* MU = mf.act[mu.depth];
    float* PHIBAR = ws.f(F); float* V = ws.f(F); float* C = ws.f(B); float* DRH = ws.f(B);
    // attached (rlrep_comm_attach with exchange scratch): Phibar and v are sums over the GLOBAL random batch -- the colsum launches push this rank's
    // partial into every rank's slot area (dp_pull.h DpSlots), a one-block launch behind each waits for all ranks and files the rank-ordered sum
    const int Wd = ag->h.world_size > 1 ? ag->h.world_size : 1;
    const int Fp = (F + 63) & ~63;
    const bool xf = ag->xfold && Wd > 1 && F <= RL_SLOTS_MAX_F && 4ll * Wd * Fp <= ag->xscratch_floats;
    const int nblk_f = qhead_blocks(B);
    float* part_f = ws.f((size_t)3 * nblk_f);
    const size_t BF = (size_t)B * F;
    {
        Program& p = ag->feat_bwd;
        for (int l = 0; l <= phi.depth || l <= mu.depth; ++l) {
            std::vector<GemmTask> t;
            if (l <= phi.depth) t.push_back(mlp_fwd(ag, phi, pf, l, X2, SA));
            if (l <= mu.depth) t.push_back(mlp_fwd(ag, mu, mf, l, S2, KE));
            b.fwd_stage(p, t, "phi / mu layer (both batches)");
        }
        ColSum c1; memset(&c1, 0, sizeof(c1));
        c1.X = PHI ? PHI + BF : nullptr; c1.ldX = F; c1.w = nullptr; c1.out = PHIBAR; c1.rows = B; c1.F = F;
        if (xf) c1.dp = ag->slots(4, 0, Fp);
        p.stages.push_back({[=](hipStream_t st) { return rl_launch_colsum(&c1, st); }, "Phibar = colsum phi_r"});
        // data parallel: Phibar and v are sums over the GLOBAL "random" batch: two F-float all-reduces
        if (ag->h.world_size > 1 && !xf) ag->feat_cuts.push_back({(int)p.stages.size() - 1, 2, PHIBAR, (int64_t)F, 0});
        if (xf) {
            const DpSlots sl = ag->slots(4, 0, Fp);
            p.stages.push_back({[=](hipStream_t st) { return rl_launch_slots_sum(&sl, PHIBAR, st); }, "Phibar over all ranks (pushed slots)"});
        }
        SpederRows sr; memset(&sr, 0, sizeof(sr));
        sr.phi = PHI; sr.mu = MU; sr.mu_r = MU ? MU + BF : nullptr; sr.phibar = PHIBAR; sr.theta_w = ag->P("theta.l.weight"); sr.theta_b = ag->P("theta.l.bias");
        sr.r = s0.R; sr.c = C; sr.drhat = DRH; sr.partial = part_f; sr.B = B; sr.F = F; sr.nblk = nblk_f; sr.inv_batch = ag->inv_batch(); sr.step = ag->adam_step + 0;
        p.stages.push_back({[=](hipStream_t st) { return rl_launch_speder_rows(&sr, st); }, "speder rows (c, d, rhat)"});
        ColSum c2; memset(&c2, 0, sizeof(c2));
        c2.X = MU ? MU + BF : nullptr; c2.ldX = F; c2.w = C; c2.out = V; c2.rows = B; c2.F = F;
        if (xf) c2.dp = ag->slots(5, 2ll * Wd * Fp, Fp);
        // theta.l's gradient (sum_i drhat_i phi_i and sum_i drhat_i over the first batch) is a weighted column sum too: it rides here instead of
        // being a 16-row-engine launch of its own behind the weight-gradient launch (7 us per feature step)
        const bool theta_here = !rl_off("fold_theta");
        if (theta_here) { c2.X2 = PHI; c2.ldX2 = F; c2.w2 = DRH; c2.out2 = ag->G("theta.l.weight"); c2.outb2 = ag->G("theta.l.bias"); c2.rows2 = B; }
        p.stages.push_back({[=](hipStream_t st) { return rl_launch_colsum(&c2, st); }, "v = sum_k c_k mu_r,k"});
        if (ag->h.world_size > 1 && !xf) ag->feat_cuts.push_back({(int)p.stages.size() - 1, 2, V, (int64_t)F, 0});
        if (xf) {
            const DpSlots sl = ag->slots(5, 2ll * Wd * Fp, Fp);
            p.stages.push_back({[=](hipStream_t st) { return rl_launch_slots_sum(&sl, V, st); }, "v over all ranks (pushed slots)"});
        }
        SpederGrads sg; memset(&sg, 0, sizeof(sg));
        sg.phi = PHI; sg.mu = MU; sg.c = C; sg.drhat = DRH; sg.phibar = PHIBAR; sg.v = V; sg.theta_w = ag->P("theta.l.weight");
        sg.Gphi = pf.g[phi.depth]; sg.Gmu = mf.g[mu.depth]; sg.B = B; sg.F = F; sg.inv_batch = ag->inv_batch();
        p.stages.push_back({[=](hipStream_t st) { return rl_launch_speder_grads(&sg, st); }, "speder grads"});
        for (int l = std::max(phi.depth, mu.depth); l >= 1; --l) {
            std::vector<GemmTask> t;
            if (l <= phi.depth) t.push_back(mlp_dx(ag, phi, pf, l));
            if (l <= mu.depth) t.push_back(mlp_dx(ag, mu, mf, l));
            b.dx_stage(p, t, "phi / mu dx");
        }
        {
            std::vector<GemmTask> t;
            for (int l = 0; l <= phi.depth; ++l) t.push_back(mlp_dw(ag, phi, pf, l, X2, SA));
            for (int l = 0; l <= mu.depth; ++l) t.push_back(mlp_dw(ag, mu, mf, l, S2, KE));
            if (!theta_here) t.push_back(Builder::dw(DRH, 1, 1, PHI, F, F, B, ag->G("theta.l.weight"), F, ag->G("theta.l.bias")));
            // the split-K partials of these gradients (K = 2B rows) are summed by the optimizer launch below: no finishing launch in the step
            if (!rl_off("fold_dwfin")) b.fold_group = 0;
            b.dw_stage(p, t, "feature dW");
            b.fold_group = -1;
        }
        const LT& p0 = ag->L.get(phi.name(0) + ".weight");
        const LT& pl = ag->L.get(phi.name(phi.depth) + ".bias");
        float* m = ag->metrics;
        const float ib = 1.0f / (float)B;
        // use_feature_target=False (spedersac_agent.py:306-307): no Polyak into the (never read) phi_target
        b.adam(ag->feat_apply, 0, ag->h.lr_feature, (ag->d.flags & RLREP_FLAG_NO_FEATURE_TARGET) ? nullptr : ag->T("phi_target.trunk.0.weight"), p0.off, pl.off + pl.rows - p0.off, ag->h.feature_tau,
               {Builder::fin_sum(part_f + 0, nblk_f, 3, -2.0f * ib, m + M_TMP0), Builder::fin_sum(part_f + 1, nblk_f, 3, ib * ib, m + M_TMP1),
                Builder::fin_combine(m + M_TMP0, 1.f, m + M_TMP1, 1.f, m + M_FEAT_A),
                Builder::fin_sum(part_f + 2, nblk_f, 3, 0.5f * ib, m + M_R_LOSS),
                Builder::fin_combine(m + M_FEAT_A, 1.f, m + M_R_LOSS, 1.f, m + M_FEAT_TOTAL)}, "adam feature + polyak phi");
    }
    Mlp phi_b{"phi.trunk", false, SA, d.phi_hidden_dim, F, d.phi_hidden_depth};
    build_rff_critic_actor(b, ag, phi_b, true);
}

// ================================================================================================
// DIFFSRSAC  (agent/diffsrsac/diffsrsac_agent.py:205-343)
// ================================================================================================
void build_diffsrsac(Builder& b, rlrep_agent* ag) {
    const rlrep_dims& d = ag->d;
    const int S = d.state_dim, A = d.action_dim, F = d.feature_dim, B = ag->B;
    const int SA = S + A, KE = 2 * S + A;
    Slot& s0 = ag->slot[0];
    Workspace& ws = b.ws;
    Mlp phi{"critic_feed_feature.z_vector", false, SA, d.phi_hidden_dim, F, d.phi_hidden_depth};
    Mlp nm{"nablamu_net.Mu_z_by_s_layer", false, S + 1, d.mu_hidden_dim, F * S, d.mu_hidden_depth};
    MlpBufs pf = alloc_mlp(b, phi, B, true);
    // nabla-mu: the last layer's output U [B, F*S] doubles as its gradient buffer (dU overwrites U in diffsr_score_kernel)
    MlpBufs nf; nf.rows = B;
    for (int l = 0; l <= nm.depth; ++l) {
        nf.act.push_back(ws.f((size_t)B * nm.width(l)));
        nf.g.push_back(l == nm.depth ? nf.act[l] : ws.f((size_t)B * nm.width(l)));
    }
    float* XN = ws.f((size_t)B * (S + 1)); float* TGT = ws.f((size_t)B * S);
    float* part_f = ws.f(B);
    const float* alphabars = ag->T("noise_alphabars");
    {
        Program& p = ag->feat_bwd;
        DiffsrPerturb dp; memset(&dp, 0, sizeof(dp));
        dp.alphabars = alphabars; dp.s2 = s0.XE ? s0.XE + SA : nullptr; dp.ld_s2 = KE; dp.XN = XN; dp.TGT = TGT; dp.B = B; dp.S = S;
        dp.step0 = ag->adam_step + 0; dp.step1 = ag->adam_step + 3;
        p.stages.push_back({[=](hipStream_t st) { DiffsrPerturb q = dp; q.idx = ag->cur_idx; q.eps = ag->cur_eps; return rl_launch_diffsr_perturb(&q, st); }, "perturb s'"});
        for (int l = 0; l <= phi.depth || l <= nm.depth; ++l) {
            std::vector<GemmTask> t;
            if (l <= phi.depth) t.push_back(mlp_fwd(ag, phi, pf, l, s0.XF, SA));
            if (l <= nm.depth) t.push_back(mlp_fwd(ag, nm, nf, l, XN, S + 1));
            b.fwd_stage(p, t, "phi / nabla-mu layer");
        }
        DiffsrScore ds; memset(&ds, 0, sizeof(ds));
        ds.U = nf.act[nm.depth]; ds.PHI = pf.act[phi.depth]; ds.TGT = TGT; ds.alphabars = alphabars; ds.GPHI = pf.g[phi.depth]; ds.partial = part_f;
        ds.B = B; ds.F = F; ds.S = S; ds.sigma = ag->h.sigma_scale; ds.inv_batch = ag->inv_batch();
        p.stages.push_back({[=](hipStream_t st) { DiffsrScore q = ds; q.idx = ag->cur_idx; return rl_launch_diffsr_score(&q, st); }, "score matching loss"});
        Builder::tag(p, RLREP_ENGINE_SCORE, 4.0 * (double)B * (double)F * (double)S, 8.0 * (double)B * (double)F * (double)S);      // U once in, dU once out
        // Data parallel: the nabla-mu HEAD holds 99 % of the feature gradients (F*S x H: 197 MB at Humanoid dims) and its weight gradient needs
        // only dU (just written) and the last hidden activation -- so it is taken FIRST, and exchange 3 tells the caller that the slice
        // [head.weight .. end of group 3] of the gradient arena is complete: its all-reduce can travel while the head's dX (202 GFLOP), the
        // rest of the backward and the small weight gradients run ("198 MB => RCCL, overlapped with backward").  The caller
        // reduces the remainder of the group after the backward.  RLREP_DISABLE=bucket_dp: one all-reduce after the whole backward, as before.
        const bool head_first = (ag->h.world_size > 1 || getenv("RLREP_FORCE_DP")) && nm.depth >= 1 && !rl_off("bucket_dp");      // (RLREP_FORCE_DP: the one-rank RCCL rehearsal)
        if (head_first) {
            b.dw_stage(p, {mlp_dw(ag, nm, nf, nm.depth, XN, S + 1)}, "nabla-mu head dW");
            const LT& hw = ag->L.get(nm.name(nm.depth) + ".weight");
            const int64_t end = ag->L.group_off[3] + ag->L.group_n[3];
            ag->feat_cuts.push_back({(int)p.stages.size() - 1, 3, ag->a.grad_dev ? ag->a.grad_dev + hw.off : nullptr, end - hw.off, hw.off});
        }
        for (int l = std::max(phi.depth, nm.depth); l >= 1; --l) {
            std::vector<GemmTask> t;
            if (l <= phi.depth) t.push_back(mlp_dx(ag, phi, pf, l));
            if (l <= nm.depth) t.push_back(mlp_dx(ag, nm, nf, l));
            b.dx_stage(p, t, "phi / nabla-mu dx");
        }
        {
            std::vector<GemmTask> t;
            for (int l = 0; l <= phi.depth; ++l) t.push_back(mlp_dw(ag, phi, pf, l, s0.XF, SA));
            for (int l = 0; l <= nm.depth; ++l) if (!(head_first && l == nm.depth)) t.push_back(mlp_dw(ag, nm, nf, l, XN, S + 1));
            b.dw_stage(p, t, "feature dW");
        }
        // diffsrsac_agent.py:313-314: nablamu_net_optimizer.step(); phi_optimizer.step()
        b.adam(ag->feat_apply, 3, ag->h.lr_feature, nullptr, 0, 0, 0.f, {}, "adam nabla-mu");
        b.adam(ag->feat_apply, 0, ag->h.lr_feature, nullptr, 0, 0, 0.f,
               {Builder::fin_sum(part_f, B, 1, 1.0f / (float)B, ag->metrics + M_FEAT_TOTAL)}, "adam phi");
    }
    build_rff_critic_actor(b, ag, phi, false);
}

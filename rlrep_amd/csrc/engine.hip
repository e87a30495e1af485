// librlrep_hip.so host side: layout, step-program construction for each agent, C ABI (include/rlrep.h).
//
// A step program is a short, fixed list of kernel launches (stages).  Each stage executes a TABLE of
// independent tasks (grouped GEMMs, see gemm16.hip), so the number of dependent launches equals the depth
// of the agent's computation graph.  Tables live in the caller-provided workspace and are uploaded when
// the batch size changes; the hot path performs no allocation, no host<->device copy and no sync.
#include "engine_internal.h"
#include <cstdarg>
#include <memory>

static thread_local char g_err[512] = "";
long long g_rl_launches = 0;
void rl_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}

// ---- diagnostic switches (engine.h: rl_off / rl_opt) ---------------------------------------------------------------------------------
// TWO environment variables, comma-separated tokens, parsed when the library is entered through rlrep_layout / rlrep_agent_create / rlrep_gemm*
// (never on a launch path): RLREP_DISABLE lists default mechanisms to switch off (every one of them has an equivalence test that compares the two
// forms), RLREP_ENABLE lists opt-in ones, optionally with a value (token=value).  INTEGRATION.md has the table.
static std::map<std::string, std::string> g_sw_off, g_sw_on;
static void sw_parse(const char* env, std::map<std::string, std::string>& m) {
    m.clear();
    const char* e = getenv(env);
    if (!e) return;
    std::string s(e), tok;
    s.push_back(',');
    for (char ch : s) {
        if (ch == ',' || ch == ' ' || ch == ';') {
            if (!tok.empty()) {
                const size_t eq = tok.find('=');
                if (eq == std::string::npos) m[tok] = "1"; else m[tok.substr(0, eq)] = tok.substr(eq + 1);
                tok.clear();
            }
        } else tok.push_back(ch);
    }
}
void rl_switches_read() { sw_parse("RLREP_DISABLE", g_sw_off); sw_parse("RLREP_ENABLE", g_sw_on); rl_gemm16_read_env(); }
bool rl_off(const char* token) { return g_sw_off.count(token) != 0; }
const char* rl_opt(const char* token) { auto it = g_sw_on.find(token); return it == g_sw_on.end() ? nullptr : it->second.c_str(); }

// ================================================================================================
// layout (names = reference state_dict keys; see oracle/shapes.py for the reference order)
// ================================================================================================
void lay_actor(Layout& L, int S, int A, int Ha, int arena, int group) {
    L.lin("actor.trunk.0", Ha, S, arena, group);          // agent/sac/actor.py:63-74
    L.lin("actor.trunk.2", Ha, Ha, arena, group);
    L.lin("actor.trunk.4", 2 * A, Ha, arena, group);
}
static void lay_doubleq(Layout& L, const std::string& m, int SA, int H, int arena, int group) {
    // agent/sac/critic.py:15-36; the two heads' first layers are stored as one [2H, S+A] matrix
    L.lin_pair(m + ".Q1.0", H, m + ".Q2.0", H, SA, arena, group);
    L.lin(m + ".Q1.2", H, H, arena, group);
    L.lin(m + ".Q2.2", H, H, arena, group);
    L.lin(m + ".Q1.4", 1, H, arena, group);
    L.lin(m + ".Q2.4", 1, H, arena, group);
}
void lay_six(Layout& L, const std::string& m, int in_f, int H, int arena, int group) {
    // l1..l6 critics (vlsac_agent.py:33-41, spedersac_agent.py:26-34, diffsrsac_agent.py:51-59); l1|l4 glued
    L.lin_pair(m + ".l1", H, m + ".l4", H, in_f, arena, group);
    L.lin(m + ".l2", H, H, arena, group);
    L.lin(m + ".l5", H, H, arena, group);
    L.lin(m + ".l3", 1, H, arena, group);
    L.lin(m + ".l6", 1, H, arena, group);
}
static void lay_gauss(Layout& L, const std::string& m, int in_f, int Hv, int F, int arena, int group) {
    // networks/vae.py:28-35 / 104-109; mean|log_std heads glued into one [2F, Hv] matrix
    L.lin(m + ".l1", Hv, in_f, arena, group);
    L.lin(m + ".l2", Hv, Hv, arena, group);
    L.lin_pair(m + ".mean_linear", F, m + ".log_std_linear", F, Hv, arena, group);
}

bool build_layout(const rlrep_dims& d, Layout& L) {
    const int S = d.state_dim, A = d.action_dim, H = d.hidden_dim, Ha = d.actor_hidden_dim, F = d.feature_dim;
    const int P = RLREP_ARENA_PARAM, T = RLREP_ARENA_TARGET;
    switch (d.alg) {
    case RLREP_ALG_SAC:
        L.begin_group(1); lay_doubleq(L, "critic", S + A, H, P, 1); L.end_group(1);
        L.begin_group(2); lay_actor(L, S, A, Ha, P, 2); L.end_group(2);
        lay_doubleq(L, "critic_target", S + A, H, T, -1);
        return true;
    case RLREP_ALG_VLSAC: {
        const int Hv = d.vae_hidden_dim;
        L.begin_group(0);
        lay_gauss(L, "encoder", 2 * S + A, Hv, F, P, 0);
        L.lin("decoder.l1", Hv, F, P, 0);                                   // networks/vae.py:74-77
        L.lin_pair("decoder.state_linear", S, "decoder.reward_linear", 1, Hv, P, 0);
        lay_gauss(L, "f", S + A, Hv, F, P, 0);
        L.end_group(0);
        L.begin_group(1); lay_six(L, "critic", F, H, P, 1); L.end_group(1);
        L.begin_group(2); lay_actor(L, S, A, Ha, P, 2); L.end_group(2);
        lay_gauss(L, "f_target", S + A, Hv, F, T, -1);
        lay_six(L, "critic_target", F, H, T, -1);
        L.add("critic.noise", d.num_noise, F, T, -1);                      // quirk Q3: plain attribute
        return true;
    }
    case RLREP_ALG_CTRLSAC: lay_ctrlsac(d, L); return true;
    case RLREP_ALG_SPEDERSAC: lay_spedersac(d, L); return true;
    case RLREP_ALG_DIFFSRSAC: lay_diffsrsac(d, L); return true;
    default:
        rl_set_error("unknown algorithm %d", d.alg);
        return false;
    }
}

// ================================================================================================
// agent
// ================================================================================================
// rows processed per qhead block loop: grid <= 128 blocks of 4 waves
int qhead_blocks(int B) { int g = (B + 3) / 4; return g > 128 ? 128 : g; }

// ------------------------------------------------------------------------------------------------
// shared fragments: actor forward / backward, actor+alpha apply
// ------------------------------------------------------------------------------------------------

ActorBufs alloc_actor(Builder& b, int B, int A, int Ha) {
    ActorBufs r;
    r.A1 = b.ws.f((size_t)B * Ha); r.A2 = b.ws.f((size_t)B * Ha); r.AO = b.ws.f((size_t)B * 2 * A);
    r.logp = b.ws.f(B); r.dA = b.ws.f((size_t)B * A); r.Ghead = b.ws.f((size_t)B * 2 * A);
    r.GA2 = b.ws.f((size_t)B * Ha); r.GA1 = b.ws.f((size_t)B * Ha);
    return r;
}

// GEMM tasks of the three actor trunk layers on input X[B, S] (row stride ldx)
GemmTask actor_l(rlrep_agent* ag, int layer, const float* X, int ldx, const ActorBufs& ab) {
    const int S = ag->d.state_dim, A = ag->d.action_dim, Ha = ag->d.actor_hidden_dim, B = ag->B;
    if (layer == 0) return Builder::fwd(X, ldx, B, S, ag->P("actor.trunk.0.weight"), S, ag->P("actor.trunk.0.bias"), Ha, ab.A1, Ha, ACT_ELU);
    if (layer == 1) return Builder::fwd(ab.A1, Ha, B, Ha, ag->P("actor.trunk.2.weight"), Ha, ag->P("actor.trunk.2.bias"), Ha, ab.A2, Ha, ACT_ELU);
    return Builder::fwd(ab.A2, Ha, B, Ha, ag->P("actor.trunk.4.weight"), Ha, ag->P("actor.trunk.4.bias"), 2 * A, ab.AO, 2 * A, ACT_NONE);
}

void policy_fwd_stage(Program& p, rlrep_agent* ag, const ActorBufs& ab, float* act, int ld_act, const char* what) {
    PolicyFwd pf; memset(&pf, 0, sizeof(pf));
    pf.O = ab.AO; pf.B = ag->B; pf.A = ag->d.action_dim; pf.act = act; pf.ld_act = ld_act; pf.logp = ab.logp;
    p.stages.push_back({[=](hipStream_t st) { PolicyFwd q = pf; q.eps = ag->cur_eps; return rl_launch_policy_fwd(&q, st); }, what});
}

// the tanh-Gaussian sampling runs in the head GEMM's epilogue when [mu | rho] fits one 16-column tile
bool policy_fusable(const rlrep_agent* ag) { return 2 * ag->d.action_dim <= 16 && !rl_off("fuse_policy"); }
GemmTask policy_head_task(rlrep_agent* ag, const ActorBufs& ab, float* act, int ld_act, int dyn_flag) {
    GemmTask head = actor_l(ag, 2, nullptr, 0, ab);
    head.epi = EPI_FWD_POLICY; head.n0 = ag->d.action_dim; head.y0 = act; head.ldx0 = ld_act; head.y1 = ab.logp; head.flags |= dyn_flag;
    return head;
}

void actor_head_stage(Builder& b, Program& p, rlrep_agent* ag, const ActorBufs& ab, float* act, int ld_act, std::vector<GemmTask> extra, const char* what) {
    GemmTask head = actor_l(ag, 2, nullptr, 0, ab);
    if (policy_fusable(ag)) {
        extra.insert(extra.begin(), policy_head_task(ag, ab, act, ld_act, FLAG_DYN_EPS));
        b.fwd_stage(p, extra, what);
    } else {
        extra.insert(extra.begin(), head);
        b.fwd_stage(p, extra, what);
        policy_fwd_stage(p, ag, ab, act, ld_act, "policy");
    }
}

// policy head backward + trunk backward + weight gradients (X = actor input with row stride ldx)
void actor_backward(Builder& b, Program& p, rlrep_agent* ag, const ActorBufs& ab, const float* X, int ldx,
                           const float* act, int ld_act, GemmTask action_dx) {
    const int S = ag->d.state_dim, A = ag->d.action_dim, Ha = ag->d.actor_hidden_dim, B = ag->B;
    if (A <= 16 && !rl_off("fuse_policy")) {
        action_dx.epi = EPI_DX_POLICYBWD; action_dx.n0 = A; action_dx.x0 = ab.AO; action_dx.x1 = act; action_dx.ldx1 = ld_act;
        action_dx.y0 = ab.Ghead; action_dx.dptr = ag->a.alpha_state_dev; action_dx.s0 = ag->inv_batch(); action_dx.flags |= FLAG_DYN_EPS;
        b.dx_stage(p, {action_dx}, "dx(action) -> dL/d[mu|rho]");
    } else {
        b.dx_stage(p, {action_dx}, "dx(action)");
        PolicyBwd pb; memset(&pb, 0, sizeof(pb));
        pb.O = ab.AO; pb.act = act; pb.ld_act = ld_act; pb.dA = ab.dA; pb.ld_dA = A;
        pb.alpha_state = ag->a.alpha_state_dev; pb.inv_batch = ag->inv_batch(); pb.G = ab.Ghead; pb.B = B; pb.A = A;
        p.stages.push_back({[=](hipStream_t st) { PolicyBwd q = pb; q.eps = ag->cur_eps; return rl_launch_policy_bwd(&q, st); }, "policy_bwd"});
    }
    // the head's dX (inner length 2A <= 32) is recomputed by every tile of the second layer's dX launch (FLAG_PRE | FLAG_PRE_ELU): one launch less
    b.dx_stage12(p, Builder::dx(ab.Ghead, 2 * A, B, 2 * A, ag->P("actor.trunk.4.weight"), Ha, ab.GA2, Ha, Ha, ACT_ELU, ab.A2, Ha),
                 Builder::dx(ab.GA2, Ha, B, Ha, ag->P("actor.trunk.2.weight"), Ha, ab.GA1, Ha, Ha, ACT_ELU, ab.A1, Ha), "actor.head dx", "actor.head dx + actor.l2 dx");
    b.dw_stage(p, {Builder::dw(ab.Ghead, 2 * A, 2 * A, ab.A2, Ha, Ha, B, ag->G("actor.trunk.4.weight"), Ha, ag->G("actor.trunk.4.bias")),
                   Builder::dw(ab.GA2, Ha, Ha, ab.A1, Ha, Ha, B, ag->G("actor.trunk.2.weight"), Ha, ag->G("actor.trunk.2.bias")),
                   Builder::dw(ab.GA1, Ha, Ha, X, ldx, S, B, ag->G("actor.trunk.0.weight"), S, ag->G("actor.trunk.0.bias"))},
               "actor dW");
}

std::vector<FinTask> actor_fins(rlrep_agent* ag, const float* partial_loss, int nblk) {
    FinTask fa; memset(&fa, 0, sizeof(fa));
    fa.kind = FIN_ALPHA; fa.partials = ag->Gtail(); fa.count = nblk; fa.stride = 1; fa.scale = ag->inv_batch();
    fa.out = ag->metrics + M_ALPHA_LOSS; fa.out2 = ag->metrics + M_ALPHA; fa.alpha_state = ag->a.alpha_state_dev;
    fa.lr = ag->h.lr_actor; fa.beta1 = ag->h.beta1; fa.beta2 = ag->h.beta2; fa.eps = ag->h.adam_eps; fa.learn = ag->h.learn_alpha;
    // the actor's optimizer launch is the last one of a train(): it also files the call's metrics in the history ring (rlrep_history)
    return {Builder::fin_sum(partial_loss, nblk, 1, 1.0f / (float)ag->B, ag->metrics + M_ACTOR_LOSS), fa, Builder::fin_history(ag)};
}
void actor_apply_program(Builder& b, rlrep_agent* ag, const float* partial_loss, int nblk) {
    b.adam(ag->actor_apply, 2, ag->h.lr_actor, nullptr, 0, 0, 0.f, actor_fins(ag, partial_loss, nblk), "adam actor + alpha");
}

void update_target_program(rlrep_agent* ag, const std::string& first_src, const std::string& first_dst) {
    // critic -> critic_target over the whole critic group (identical internal layouts)
    PolyakTask t; memset(&t, 0, sizeof(t));
    t.src = ag->a.param_dev ? ag->a.param_dev + ag->L.group_off[1] : nullptr;
    t.dst = ag->a.target_dev ? ag->a.target_dev + ag->L.get(first_dst).off : nullptr;
    (void)first_src;
    t.n = ag->L.group_n[1]; t.tau = ag->h.tau; t.steps = ag->steps; t.period = ag->h.target_update_period;
    ag->upd_target.stages.push_back({[=](hipStream_t st) { return rl_launch_polyak(&t, st); }, "polyak critic"});
}
// critic Adam with the target update folded in (same Polyak, same period gate, run by the Adam launch's own lanes)
void critic_apply_folded(Builder& b, rlrep_agent* ag, const std::string& first_dst, std::vector<FinTask> fins, Program* into, const int* steps) {
    if (!into && rl_off("fold_target")) return;
    float* dst = ag->a.target_dev ? ag->a.target_dev + ag->L.get(first_dst).off : nullptr;
    b.adam(into ? *into : ag->critic_apply_f, 1, ag->h.lr_critic, dst, ag->L.group_off[1], ag->L.group_n[1], ag->h.tau, fins, "adam critic + polyak critic",
           steps ? steps : ag->steps, ag->h.target_update_period);
}

// ================================================================================================
// SAC   (agent/sac/sac_agent.py:105-166)
// ================================================================================================
static void build_sac(Builder& b, rlrep_agent* ag) {
    const int S = ag->d.state_dim, A = ag->d.action_dim, H = ag->d.hidden_dim, Ha = ag->d.actor_hidden_dim, B = ag->B;
    const int SA = S + A;
    Slot& s0 = ag->slot[0];
    ActorBufs ab = alloc_actor(b, B, A, Ha);               // policy on s' (critic step)
    ActorBufs ab_pi = alloc_actor(b, B, A, Ha);            // policy on s  (actor step)
    float* E1t = b.ws.f((size_t)B * 2 * H);       // target first-layer activations [B, 2H] (Q1|Q2)
    float* E1c = b.ws.f((size_t)B * 2 * H);
    float* Et = b.ws.f((size_t)2 * B * H);        // second-layer activations, heads stacked [2][B,H]
    float* Ec = b.ws.f((size_t)2 * B * H);
    float* GE = b.ws.f((size_t)2 * B * H);
    float* G1 = b.ws.f((size_t)B * 2 * H);
    float* dq = b.ws.f((size_t)2 * B);
    const int nblk = qhead_blocks(B);
    float* part_q = b.ws.f((size_t)4 * nblk);
    float* part_l = b.ws.f(nblk);
    auto Pw = [&](const char* n) { return ag->P(n); };
    auto Tw = [&](const char* n) { return ag->T(n); };

    // ---- critic step ----
    // hoist: the variant that also carries the forward half of the FOLLOWING actor step (policy on s): its three layers read nothing the
    // critic update writes, and as extra tasks of launches that exist anyway they take three launches off train() (rlrep_prefetch_policy)
    const bool can_hoist = policy_fusable(ag) && !rl_off("hoist");
    auto critic_program = [&](Program& p, bool hoist, bool emit_apply) {
        if (hoist) {
            b.fwd_stage(p, {actor_l(ag, 0, s0.XF2, SA, ab), actor_l(ag, 0, s0.XFpi, SA, ab_pi)}, "actor.l1(s') actor.l1(s)");
            b.fwd_stage(p, {actor_l(ag, 1, nullptr, 0, ab), actor_l(ag, 1, nullptr, 0, ab_pi)}, "actor.l2 x2");
            b.fwd_stage(p, {policy_head_task(ag, ab, s0.XF2 + S, SA, FLAG_DYN_EPS), policy_head_task(ag, ab_pi, s0.XFpi + S, SA, FLAG_DYN_EPS2)},
                        "actor.head x2 + policy");
        } else {
            b.fwd_stage(p, {actor_l(ag, 0, s0.XF2, SA, ab)}, "actor.l1(s')");
            b.fwd_stage(p, {actor_l(ag, 1, nullptr, 0, ab)}, "actor.l2");
            actor_head_stage(b, p, ag, ab, s0.XF2 + S, SA, {}, "actor.head + policy");
        }
        b.fwd_stage(p, {Builder::fwd(s0.XF2, SA, B, SA, Tw("critic_target.Q1.0.weight"), SA, Tw("critic_target.Q1.0.bias"), 2 * H, E1t, 2 * H, ACT_ELU),
                        Builder::fwd(s0.XF, SA, B, SA, Pw("critic.Q1.0.weight"), SA, Pw("critic.Q1.0.bias"), 2 * H, E1c, 2 * H, ACT_ELU)}, "Q l1");
        b.fwd_stage(p, {Builder::fwd(E1t, 2 * H, B, H, Tw("critic_target.Q1.2.weight"), H, Tw("critic_target.Q1.2.bias"), H, Et, H, ACT_ELU),
                        Builder::fwd(E1t + H, 2 * H, B, H, Tw("critic_target.Q2.2.weight"), H, Tw("critic_target.Q2.2.bias"), H, Et + (size_t)B * H, H, ACT_ELU),
                        Builder::fwd(E1c, 2 * H, B, H, Pw("critic.Q1.2.weight"), H, Pw("critic.Q1.2.bias"), H, Ec, H, ACT_ELU),
                        Builder::fwd(E1c + H, 2 * H, B, H, Pw("critic.Q2.2.weight"), H, Pw("critic.Q2.2.bias"), H, Ec + (size_t)B * H, H, ACT_ELU)}, "Q l2");
        QHeadCritic q; memset(&q, 0, sizeof(q));
        q.Et[0] = Et; q.Et[1] = Et + (size_t)B * H; q.Ec[0] = Ec; q.Ec[1] = Ec + (size_t)B * H;
        q.wt[0] = Tw("critic_target.Q1.4.weight"); q.wt[1] = Tw("critic_target.Q2.4.weight");
        q.bt[0] = Tw("critic_target.Q1.4.bias"); q.bt[1] = Tw("critic_target.Q2.4.bias");
        q.wc[0] = Pw("critic.Q1.4.weight"); q.wc[1] = Pw("critic.Q2.4.weight");
        q.bc[0] = Pw("critic.Q1.4.bias"); q.bc[1] = Pw("critic.Q2.4.bias");
        q.logp = ab.logp; q.R = s0.R; q.D = s0.D; q.alpha_state = ag->a.alpha_state_dev; q.gamma = ag->h.discount;
        q.inv_batch = ag->inv_batch(); q.dq = dq; q.GE[0] = GE; q.GE[1] = GE + (size_t)B * H; q.partial = part_q;
        q.B = B; q.H = H; q.nblk = nblk; q.train = 1; q.step = ag->adam_step + 1;
        p.stages.push_back({[=](hipStream_t st) { return rl_launch_qhead_critic(&q, st); }, "qhead critic"});
        b.dx_stage(p, {Builder::dx(GE, H, B, H, Pw("critic.Q1.2.weight"), H, G1, 2 * H, H, ACT_ELU, E1c, 2 * H),
                       Builder::dx(GE + (size_t)B * H, H, B, H, Pw("critic.Q2.2.weight"), H, G1 + H, 2 * H, H, ACT_ELU, E1c + H, 2 * H)}, "Q l2 dx");
        b.dw_stage(p, {Builder::dw(dq, 1, 1, Ec, H, H, B, ag->G("critic.Q1.4.weight"), H, ag->G("critic.Q1.4.bias")),
                       Builder::dw(dq + B, 1, 1, Ec + (size_t)B * H, H, H, B, ag->G("critic.Q2.4.weight"), H, ag->G("critic.Q2.4.bias")),
                       Builder::dw(GE, H, H, E1c, 2 * H, H, B, ag->G("critic.Q1.2.weight"), H, ag->G("critic.Q1.2.bias")),
                       Builder::dw(GE + (size_t)B * H, H, H, E1c + H, 2 * H, H, B, ag->G("critic.Q2.2.weight"), H, ag->G("critic.Q2.2.bias")),
                       Builder::dw(G1, 2 * H, 2 * H, s0.XF, SA, SA, B, ag->G("critic.Q1.0.weight"), SA, ag->G("critic.Q1.0.bias"))}, "Q dW");
        // sac reports q_loss = mse1+mse2 and q2 := q1 (quirk Q13)
        const float ib = 1.0f / (float)B;
        const std::vector<FinTask> cfins = {Builder::fin_sum(part_q + 0, nblk, 4, ib, ag->metrics + M_TMP0),
                Builder::fin_sum(part_q + 1, nblk, 4, ib, ag->metrics + M_TMP1),
                Builder::fin_combine(ag->metrics + M_TMP0, 1.f, ag->metrics + M_TMP1, 1.f, ag->metrics + M_Q1_LOSS),
                Builder::fin_sum(part_q + 2, nblk, 4, ib, ag->metrics + M_Q1),
                Builder::fin_copy(ag->metrics + M_Q1, ag->metrics + M_Q2)};
        if (emit_apply) {
            b.adam(ag->critic_apply, 1, ag->h.lr_critic, nullptr, 0, 0, 0.f, cfins, "adam critic");
            critic_apply_folded(b, ag, "critic_target.Q1.0.weight", cfins);
        }
    };
    critic_program(ag->critic_bwd, false, true);
    if (can_hoist) critic_program(ag->critic_bwd_h, true, false);
    // ---- actor step ----
    {
        Program& p = ag->actor_bwd;
        b.fwd_stage(p, {actor_l(ag, 0, s0.XFpi, SA, ab_pi)}, "actor.l1(s)");
        b.fwd_stage(p, {actor_l(ag, 1, nullptr, 0, ab_pi)}, "actor.l2");
        actor_head_stage(b, p, ag, ab_pi, s0.XFpi + S, SA, {}, "actor.head + policy");
        ag->actor_resume = (int)p.stages.size();              // everything above is what critic_bwd_h already did
        b.fwd_stage(p, {Builder::fwd(s0.XFpi, SA, B, SA, Pw("critic.Q1.0.weight"), SA, Pw("critic.Q1.0.bias"), 2 * H, E1c, 2 * H, ACT_ELU)}, "Q l1");
        b.fwd_stage(p, {Builder::fwd(E1c, 2 * H, B, H, Pw("critic.Q1.2.weight"), H, Pw("critic.Q1.2.bias"), H, Ec, H, ACT_ELU),
                        Builder::fwd(E1c + H, 2 * H, B, H, Pw("critic.Q2.2.weight"), H, Pw("critic.Q2.2.bias"), H, Ec + (size_t)B * H, H, ACT_ELU)}, "Q l2");
        QHeadActor q; memset(&q, 0, sizeof(q));
        q.Ec[0] = Ec; q.Ec[1] = Ec + (size_t)B * H;
        q.wc[0] = Pw("critic.Q1.4.weight"); q.wc[1] = Pw("critic.Q2.4.weight");
        q.bc[0] = Pw("critic.Q1.4.bias"); q.bc[1] = Pw("critic.Q2.4.bias");
        q.logp = ab_pi.logp; q.alpha_state = ag->a.alpha_state_dev; q.inv_batch = ag->inv_batch(); q.target_entropy = ag->h.target_entropy;
        q.GE[0] = GE; q.GE[1] = GE + (size_t)B * H; q.partial_loss = part_l; q.partial_c = ag->Gtail();
        q.B = B; q.H = H; q.nblk = nblk; q.step = ag->adam_step + 2;
        p.stages.push_back({[=](hipStream_t st) { return rl_launch_qhead_actor(&q, st); }, "qhead actor"});
        b.dx_stage(p, {Builder::dx(GE, H, B, H, Pw("critic.Q1.2.weight"), H, G1, 2 * H, H, ACT_ELU, E1c, 2 * H),
                       Builder::dx(GE + (size_t)B * H, H, B, H, Pw("critic.Q2.2.weight"), H, G1 + H, 2 * H, H, ACT_ELU, E1c + H, 2 * H)}, "Q l2 dx");
        // both heads at once: [G1_1 | G1_2] [B,2H] x [W_Q1.0 ; W_Q2.0][:, S:S+A]
        actor_backward(b, p, ag, ab_pi, s0.XFpi, SA, s0.XFpi + S, SA, Builder::dx(G1, 2 * H, B, 2 * H, Pw("critic.Q1.0.weight") ? Pw("critic.Q1.0.weight") + S : nullptr, SA, ab_pi.dA, A, A, ACT_NONE, nullptr, 0));
        actor_apply_program(b, ag, part_l, nblk);
    }
    update_target_program(ag, "critic.Q1.0.weight", "critic_target.Q1.0.weight");
}

// ================================================================================================
// VLSAC   (agent/vlsac/vlsac_agent.py:126-273)
// ================================================================================================
struct GaussBufs { float *H1, *H2, *HH; };

static void gauss_tasks(rlrep_agent* ag, bool target, const std::string& m, const float* X, int ldx, int K, const GaussBufs& g,
                        GemmTask (&out)[3]) {
    const int Hv = ag->d.vae_hidden_dim, F = ag->d.feature_dim, B = ag->B;
    auto W = [&](const std::string& n) { return target ? ag->T(m + n) : ag->P(m + n); };
    out[0] = Builder::fwd(X, ldx, B, K, W(".l1.weight"), K, W(".l1.bias"), Hv, g.H1, Hv, ACT_RELU);
    out[1] = Builder::fwd(g.H1, Hv, B, Hv, W(".l2.weight"), Hv, W(".l2.bias"), Hv, g.H2, Hv, ACT_RELU);
    out[2] = Builder::fwd(g.H2, Hv, B, Hv, W(".mean_linear.weight"), Hv, W(".mean_linear.bias"), 2 * F, g.HH, 2 * F, ACT_NONE);
}

static void build_vlsac(Builder& b, rlrep_agent* ag) {
    const int S = ag->d.state_dim, A = ag->d.action_dim, H = ag->d.hidden_dim, Ha = ag->d.actor_hidden_dim, B = ag->B;
    const int F = ag->d.feature_dim, Hv = ag->d.vae_hidden_dim, N = ag->d.num_noise;
    const int SA = S + A, KE = 2 * S + A;
    Slot& s0 = ag->slot[0];
    Workspace& ws = b.ws;
    // use_feature_target=False (vlsac_agent.py:176-179, 214-219, 257-258): critic and actor steps read the LIVE f, no Polyak into f_target
    const bool nft = (ag->d.flags & RLREP_FLAG_NO_FEATURE_TARGET) != 0;
    const std::string fnet = nft ? "f" : "f_target";
    auto FT = [&](const char* n) { return nft ? ag->P(std::string("f.") + n) : ag->T(std::string("f_target.") + n); };
    auto Pw = [&](const char* n) { return ag->P(n); };
    auto Tw = [&](const char* n) { return ag->T(n); };
    auto Gw = [&](const char* n) { return ag->G(n); };

    // ---- feature step buffers ----
    GaussBufs ge{ws.f((size_t)B * Hv), ws.f((size_t)B * Hv), ws.f((size_t)B * 2 * F)};
    GaussBufs gf{ws.f((size_t)B * Hv), ws.f((size_t)B * Hv), ws.f((size_t)B * 2 * F)};
    float* Z = ws.f((size_t)B * F); float* EZ = ws.f((size_t)B * F);
    float* D1 = ws.f((size_t)B * Hv);
    float* GDH = ws.f((size_t)B * (S + 1)); float* GD1 = ws.f((size_t)B * Hv);
    float* GEH = ws.f((size_t)B * 2 * F); float* GFH = ws.f((size_t)B * 2 * F);
    float* GH2e = ws.f((size_t)B * Hv); float* GH1e = ws.f((size_t)B * Hv);
    float* GH2f = ws.f((size_t)B * Hv); float* GH1f = ws.f((size_t)B * Hv);
    // the Gaussian heads of encoder and f and vae_mid as ONE launch (heads_vae_kernel: a 16 x 16 tile per workgroup, one KL partial each;
    // +3 % against the heads launch + an elementwise vae_mid launch, which is gone)
    // dec.heads + mse inside the dec.l1 dX launch (FLAG_PRE_MSE): needs the 16-byte-aligned rows its first phase loads, K1 = S + 1 <= 32
    const bool fold_mse = !rl_rowprog_enabled() && !Builder::chain_enabled() && (Hv & 3) == 0 && S + 1 <= 32 && !rl_off("fold_mse") && !rl_off("fuse_dx") &&
                          ((B + 15) / 16) * ((F + 15) / 16) < 384 * 2;
    const int tiles_vm = ((B + 15) / 16) * ((F + 15) / 16);
    const int nblk_kl = tiles_vm, nblk_mse_tiles = ((B + 15) / 16) * ((S + 1 + 15) / 16);   // mse partials: one pair per dec.heads tile
    const int nblk_mse = fold_mse ? (B + 15) / 16 : nblk_mse_tiles;                        // (folded: one pair per 16-row tile)
    float* part_kl = ws.f(nblk_kl); float* part_mse = ws.f((size_t)2 * nblk_mse_tiles);
    // actor buffers are needed by the feature program variant that carries the policy forwards
    ActorBufs ab = alloc_actor(b, B, A, Ha);                                                 // policy on s' (critic step)
    ActorBufs ab_pi = alloc_actor(b, B, A, Ha);                                              // policy on s  (actor step)
    // ---- the forward + dX chains of a feature step as ONE launch of row-block programs (rowprog.hip) -------------------------------
    // Two workgroups per 16-row block: program E runs encoder -> sample / KL -> decoder -> decoder and encoder backward, program F runs
    // f forward, publishes its heads to E (the KL term needs both Gaussians), receives dKL/d(f heads) and runs f's backward beside E's
    // decoder passes.  `early`: two more programs run both policy forwards (what the first three launches of feat_bwd_h carry).
    const int nrb = (B + RP_ROWS - 1) / RP_ROWS;
    const int nblk_rp = nrb;
    float* part_kl_rp = ws.f((size_t)nblk_rp * 8); float* part_mse_rp = ws.f((size_t)2 * nblk_rp);       // (KL: one partial per workgroup, up to 8 members per row block)
    int* rp_flags = (int*)ws.alloc(sizeof(int) * 2 * nrb);
    if (!b.dry && rp_flags && b.ws.ok()) (void)hipMemset(rp_flags, 0, sizeof(int) * 2 * nrb);
    auto rp_feature = [&](RpAsm& A_, bool early) {
        const bool pair = !rl_off("rowprog_pair");
        auto Wp = [&](const char* n) { return ag->P(n); };
        const bool useT = ag->nsh[0] > 0;
        // forward layer `name`: from the transposed shadow when the agent keeps one (wide layers), else from W itself
        auto FW = [&](const RpBuf& x, int K, const std::string& name, const std::string& bias_name, int N, int act, const RpBuf* d, float* gout, int ldg) {
            if (useT && N >= 64 && ag->shadow_of.count(name + ".weight")) A_.fwdT(x, K, ag->PT(name + ".weight"), ag->P(bias_name + ".bias"), N, act, d, gout, ldg);
            else A_.fwd(x, K, ag->P(name + ".weight"), K, ag->P(bias_name + ".bias"), N, act, d, gout, ldg);
        };
        auto f_forward = [&](RpBuf& x, RpBuf& h1, RpBuf& h2) {
            x = A_.buf(SA); h1 = A_.buf(Hv); h2 = A_.buf(Hv);
            A_.load(s0.XF, SA, SA, x);
            FW(x, SA, "f.l1", "f.l1", Hv, ACT_RELU, &h1, gf.H1, Hv);
            FW(h1, Hv, "f.l2", "f.l2", Hv, ACT_RELU, &h2, gf.H2, Hv);
        };
        auto f_backward = [&](const RpBuf& g, const RpBuf* m2, const RpBuf* m1, const RpBuf& g2) {
            A_.load(GFH, 2 * F, 2 * F, g);
            A_.dx(g, 2 * F, Wp("f.mean_linear.weight"), Hv, Hv, ACT_RELU, m2, gf.H2, Hv, &g2, GH2f, Hv);
            A_.dx(g2, Hv, Wp("f.l2.weight"), Hv, Hv, ACT_RELU, m1, gf.H1, Hv, nullptr, GH1f, Hv);
        };
        // ---- program E (and, unpaired, the whole step) ----
        A_.begin();
        RpBuf r0b = A_.buf(std::max(KE, S + 1)), r1 = A_.buf(std::max(Hv, F)), r2 = A_.buf(std::max(Hv, F)), rhh = A_.buf(2 * F),
              rfh = A_.buf(std::max(2 * F, Hv)), r5 = A_.buf(std::max(F, Hv));
        if (!pair) {
            // f forward first, on the regions the encoder is about to use; its heads land where the KL op expects them
            RpBuf x = RpAsm::at(r0b, SA), h1 = RpAsm::at(r1, Hv), h2 = RpAsm::at(r2, Hv), fh = RpAsm::at(rfh, 2 * F);
            A_.load(s0.XF, SA, SA, x);
            FW(x, SA, "f.l1", "f.l1", Hv, ACT_RELU, &h1, gf.H1, Hv);
            FW(h1, Hv, "f.l2", "f.l2", Hv, ACT_RELU, &h2, gf.H2, Hv);
            FW(h2, Hv, "f.mean_linear", "f.mean_linear", 2 * F, ACT_NONE, &fh, gf.HH, 2 * F);
        }
        {
            RpBuf x = RpAsm::at(r0b, KE), h1 = RpAsm::at(r1, Hv), h2 = RpAsm::at(r2, Hv);
            A_.load(s0.XE, KE, KE, x);
            FW(x, KE, "encoder.l1", "encoder.l1", Hv, ACT_RELU, &h1, ge.H1, Hv);
            FW(h1, Hv, "encoder.l2", "encoder.l2", Hv, ACT_RELU, &h2, ge.H2, Hv);
            FW(h2, Hv, "encoder.mean_linear", "encoder.mean_linear", 2 * F, ACT_NONE, &rhh, ge.HH, 2 * F);
        }
        const RpBuf fh = RpAsm::at(rfh, 2 * F), z = RpAsm::at(r1, F), ez = RpAsm::at(r2, F);
        if (pair) { A_.wait(0); A_.load(gf.HH, 2 * F, 2 * F, fh); }
        {
            RpOp o = RpAsm::blank(RP_VAE_MID);
            o.src = rhh.off; o.lds = rhh.ld; o.src2 = fh.off; o.lds2 = fh.ld; o.N = F; o.dyn = 0; o.s0 = ag->inv_batch() / (float)F;
            o.dst = z.off; o.ldd = z.ld; o.wpad = z.w; o.dst2 = ez.off; o.ldd2 = ez.ld;
            o.gout = Z; o.ldg = F; o.gout2 = GFH; o.ldg2 = 2 * F; o.part = part_kl_rp; o.step = ag->adam_step + 0; o.flags = RPF_BUMP;
            A_.ops.push_back(o);
        }
        if (pair) A_.signal(1);
        // the f heads are dead: their region takes the decoder's hidden layer, then dz; z's region takes the encoder's dL/dh2
        RpBuf d1 = RpAsm::at(rfh, Hv), dh = RpAsm::at(r0b, S + 1), gd1 = RpAsm::at(r5, Hv), dz = RpAsm::at(rfh, F), g2 = RpAsm::at(r1, Hv);
        FW(z, F, "decoder.l1", "decoder.l1", Hv, ACT_RELU, &d1, D1, Hv);
        A_.fwd(d1, Hv, Wp("decoder.state_linear.weight"), Hv, Wp("decoder.state_linear.bias"), S + 1, ACT_NONE, &dh, nullptr, 0);
        {
            RpOp o = RpAsm::blank(RP_MSE);
            o.src = dh.off; o.lds = dh.ld; o.n0 = S; o.gin = s0.XE ? s0.XE + SA : nullptr; o.ldgin = KE; o.gin2 = s0.R;
            o.s0 = ag->inv_batch() / (float)S; o.s1 = ag->inv_batch(); o.gout = GDH; o.ldg = S + 1; o.part = part_mse_rp;
            A_.ops.push_back(o);
        }
        A_.dx(dh, S + 1, Wp("decoder.state_linear.weight"), Hv, Hv, ACT_RELU, &d1, nullptr, 0, &gd1, GD1, Hv);
        A_.dx(gd1, Hv, Wp("decoder.l1.weight"), F, F, ACT_NONE, nullptr, nullptr, 0, &dz, nullptr, 0);
        {
            RpOp o = RpAsm::blank(RP_REPARAM);
            o.src = dz.off; o.lds = dz.ld; o.src2 = ez.off; o.lds2 = ez.ld; o.dst = rhh.off; o.ldd = rhh.ld; o.N = F; o.gout = GEH; o.ldg = 2 * F;
            A_.ops.push_back(o);
        }
        A_.dx(rhh, 2 * F, Wp("encoder.mean_linear.weight"), Hv, Hv, ACT_RELU, nullptr, ge.H2, Hv, &g2, GH2e, Hv);
        A_.dx(g2, Hv, Wp("encoder.l2.weight"), Hv, Hv, ACT_RELU, nullptr, ge.H1, Hv, nullptr, GH1e, Hv);
        if (!pair) {
            RpBuf g = RpAsm::at(rfh, 2 * F), gg2 = RpAsm::at(r2, Hv);
            f_backward(g, nullptr, nullptr, gg2);
        }
        A_.end(nrb);
        if (pair) {
            A_.begin();
            RpBuf x, h1, h2;
            f_forward(x, h1, h2);
            FW(h2, Hv, "f.mean_linear", "f.mean_linear", 2 * F, ACT_NONE, nullptr, gf.HH, 2 * F);
            A_.signal(0);
            RpBuf g = A_.buf(2 * F), gg2 = A_.buf(Hv);
            A_.wait(1);
            f_backward(g, &h2, &h1, gg2);
            A_.end(nrb);
        }
        if (early) {
            for (int which = 0; which < 2; ++which) {
                const ActorBufs& abx = which == 0 ? ab : ab_pi;
                const float* X = which == 0 ? s0.XF2 : s0.XFpi;
                A_.begin();
                RpBuf x = A_.buf(S), a1 = A_.buf(Ha), a2 = A_.buf(Ha), ao = A_.buf(2 * A);
                A_.load(X, SA, S, x);
                A_.fwd(x, S, Wp("actor.trunk.0.weight"), S, Wp("actor.trunk.0.bias"), Ha, ACT_ELU, &a1, abx.A1, Ha);
                A_.fwd(a1, Ha, Wp("actor.trunk.2.weight"), Ha, Wp("actor.trunk.2.bias"), Ha, ACT_ELU, &a2, abx.A2, Ha);
                A_.fwd(a2, Ha, Wp("actor.trunk.4.weight"), Ha, Wp("actor.trunk.4.bias"), 2 * A, ACT_NONE, &ao, abx.AO, 2 * A);
                RpOp o = RpAsm::blank(RP_POLICY);
                o.src = ao.off; o.lds = ao.ld; o.n0 = A; o.dyn = which == 0 ? 1 : 2;
                o.gout = const_cast<float*>(X) ? const_cast<float*>(X) + S : nullptr; o.ldg = SA; o.gout2 = abx.logp;
                A_.ops.push_back(o);
                A_.end(nrb);
            }
        }
    };
    // OPT-IN (RLREP_ENABLE=rowprog): measured on MI355X at the headline dimensions the fused launch takes 91-95 us against 44 us (stages timed
    // alone) / ~57 us (in the dependent chain) for the nine launches it replaces -- one CU per 16-row block ingests every weight matrix
    // (256 KB per 256 x 256 layer at ~50-65 GB/s per CU) and runs fp32 MFMA at 41-53 cycles per instruction: 5-6 us per layer and row
    // block, i.e. a dependent launch.  DESIGN.md section 5.2 has the per-op timeline.

    // ---- cluster form (RLREP_ENABLE=rowprog=2): C workgroups per row block and chain, member m owns a column slice of every layer and the
    // members complete each other's vectors through tagged 8-byte granules in global memory (RP_XCHG; tools/exp/cluster_hop.hip: 1.9-2.4 us
    // per hop with 256 workgroups exchanging at once).  Masks come from the activations in global memory (each member wrote its own slice).
    int CS = 0;                                   // cluster size: 0 = not applicable
    for (int c : {8, 4, 2}) if (!CS && Hv % c == 0 && F % c == 0 && (long long)nrb * c * 2 <= 256 && 16 * (2 * F / c) <= RP_XSLOT && 16 * (Hv / c) <= RP_XSLOT) CS = c;
    unsigned long long* xbuf = (unsigned long long*)ws.alloc(rl_rowprog_cluster() && CS ? (size_t)nrb * 2 * RP_MAX_HOPS * CS * RP_XSLOT * 8 : 8);
    auto rp_feature_cluster = [&](RpAsm& A_) {
        const int wh = Hv / CS, wf = F / CS;
        auto Wp = [&](const std::string& n) { return ag->P(n); };
        auto last = [&]() -> RpOp& { return A_.ops.back(); };
        // forward slice of layer `name` (N_total outputs; this member: columns col_base + m*ws .. + ws) from the transposed shadow
        auto fwdS = [&](const RpBuf& x, int K, const std::string& name, const std::string& bias_name, int Ntot, int col_base, int wsl, int act,
                        const RpBuf& dfull, float* gout, int ldg) {
            RpBuf d = dfull; d.off += col_base; d.w = wsl;
            const float* WT = ag->PT(name + ".weight");
            const float* bias = Wp(bias_name + ".bias");
            A_.fwdT(x, K, WT ? WT + col_base : nullptr, bias ? bias + col_base : nullptr, wsl, act, &d, gout ? gout + col_base : nullptr, ldg);
            RpOp& o = last(); o.ldw = Ntot; o.wpad = wsl; o.m_w = wsl; o.m_b = wsl; o.m_dst = wsl; o.m_g = wsl;
        };
        // dX slice: columns m*ws .. of (G W) masked by the activation in global memory
        auto dxS = [&](const RpBuf& g, int K, const float* W, int ldw, int wsl, int act, const float* mask_g, int ldmask, const RpBuf* dfull, bool slice_dst,
                       float* gout, int ldg) {
            RpBuf d; if (dfull) { d = *dfull; d.w = wsl; }
            A_.dx(g, K, W, ldw, wsl, act, nullptr, mask_g, ldmask, dfull ? &d : nullptr, gout, ldg);
            RpOp& o = last(); o.wpad = wsl; o.m_w = wsl; o.m_g = wsl; o.m_gaux = wsl; o.m_dst = slice_dst ? 0 : wsl;
        };
        // ---------------- program E ----------------
        A_.begin();
        {
            RpBuf bx = A_.buf(std::max(KE, S + 1)), bA = A_.buf(2 * F > Hv ? 2 * F : Hv), bB = A_.buf(2 * F > Hv ? 2 * F : Hv), bC = A_.buf(std::max(F, Hv));
            RpBuf ezs = A_.buf(wf), dzs = A_.buf(wf);
            A_.load(s0.XE, KE, KE, bx);
            fwdS(bx, KE, "encoder.l1", "encoder.l1", Hv, 0, wh, ACT_RELU, bA, ge.H1, Hv);
            A_.xop(RP_XCHG, bA, 0, wh, 1, 0);
            fwdS(bA, Hv, "encoder.l2", "encoder.l2", Hv, 0, wh, ACT_RELU, bB, ge.H2, Hv);
            A_.xop(RP_XCHG, bB, 1, wh, 1, 0);
            fwdS(bB, Hv, "encoder.mean_linear", "encoder.mean_linear", 2 * F, 0, wf, ACT_NONE, bA, ge.HH, 2 * F);
            fwdS(bB, Hv, "encoder.mean_linear", "encoder.mean_linear", 2 * F, F, wf, ACT_NONE, bA, ge.HH, 2 * F);
            A_.xop(RP_GATHER, bB, 2, wf, 2, F, 2);                      // my slice of the f heads, from member m of the F cluster (its hop 2)
            {
                RpOp o = RpAsm::blank(RP_VAE_MID);
                o.src = bA.off; o.lds = bA.ld; o.src2 = bB.off; o.lds2 = bB.ld; o.N = wf; o.K = F; o.wpad = wf; o.dyn = 0; o.s0 = ag->inv_batch() / (float)F;
                o.dst = bC.off; o.ldd = bC.ld; o.dst2 = ezs.off; o.ldd2 = ezs.ld;
                o.gout = Z; o.ldg = F; o.gout2 = GFH; o.ldg2 = 2 * F; o.part = part_kl_rp; o.step = ag->adam_step + 0; o.flags = RPF_BUMP | RPF_FH_INPLACE;
                o.m_src = wf; o.m_s2 = wf; o.m_dst = wf; o.m_g = wf; o.m_g2 = wf; o.m_gin = wf;
                A_.ops.push_back(o);
            }
            A_.xop(RP_PUBLISH, bB, 7, wf, 2, F);                        // dKL/d(f heads) for the F cluster
            A_.xop(RP_XCHG, bC, 2, wf, 1, 0);                           // z
            fwdS(bC, F, "decoder.l1", "decoder.l1", Hv, 0, wh, ACT_RELU, bB, D1, Hv);
            A_.xop(RP_XCHG, bB, 3, wh, 1, 0);
            RpBuf dh = RpAsm::at(bx, S + 1);
            A_.fwd(bB, Hv, Wp("decoder.state_linear.weight"), Hv, Wp("decoder.state_linear.bias"), S + 1, ACT_NONE, &dh, nullptr, 0);   // every member, whole
            {
                RpOp o = RpAsm::blank(RP_MSE);
                o.src = dh.off; o.lds = dh.ld; o.n0 = S; o.gin = s0.XE ? s0.XE + SA : nullptr; o.ldgin = KE; o.gin2 = s0.R;
                o.s0 = ag->inv_batch() / (float)S; o.s1 = ag->inv_batch(); o.gout = GDH; o.ldg = S + 1; o.part = part_mse_rp;
                A_.ops.push_back(o);
            }
            dxS(dh, S + 1, Wp("decoder.state_linear.weight"), Hv, wh, ACT_RELU, D1, Hv, &bC, false, GD1, Hv);
            A_.xop(RP_XCHG, bC, 4, wh, 1, 0);
            dxS(bC, Hv, Wp("decoder.l1.weight"), F, wf, ACT_NONE, nullptr, 0, &dzs, true, nullptr, 0);
            {
                RpOp o = RpAsm::blank(RP_REPARAM);
                o.src = dzs.off; o.lds = dzs.ld; o.src2 = ezs.off; o.lds2 = ezs.ld; o.dst = bA.off; o.ldd = bA.ld; o.N = wf; o.K = F; o.gout = GEH; o.ldg = 2 * F;
                o.m_dst = wf; o.m_g = wf;
                A_.ops.push_back(o);
            }
            A_.xop(RP_XCHG, bA, 5, wf, 2, F);                           // dL/d(encoder heads), both halves
            dxS(bA, 2 * F, Wp("encoder.mean_linear.weight"), Hv, wh, ACT_RELU, ge.H2, Hv, &bB, false, GH2e, Hv);
            A_.xop(RP_XCHG, bB, 6, wh, 1, 0);
            dxS(bB, Hv, Wp("encoder.l2.weight"), Hv, wh, ACT_RELU, ge.H1, Hv, nullptr, false, GH1e, Hv);
        }
        A_.end(nrb, CS, 0);
        // ---------------- program F ----------------
        A_.begin();
        {
            RpBuf bx = A_.buf(SA), bA = A_.buf(Hv), bB = A_.buf(Hv), bH = A_.buf(2 * F);
            A_.load(s0.XF, SA, SA, bx);
            fwdS(bx, SA, "f.l1", "f.l1", Hv, 0, wh, ACT_RELU, bA, gf.H1, Hv);
            A_.xop(RP_XCHG, bA, 0, wh, 1, 0);
            fwdS(bA, Hv, "f.l2", "f.l2", Hv, 0, wh, ACT_RELU, bB, gf.H2, Hv);
            A_.xop(RP_XCHG, bB, 1, wh, 1, 0);
            fwdS(bB, Hv, "f.mean_linear", "f.mean_linear", 2 * F, 0, wf, ACT_NONE, bH, gf.HH, 2 * F);
            fwdS(bB, Hv, "f.mean_linear", "f.mean_linear", 2 * F, F, wf, ACT_NONE, bH, gf.HH, 2 * F);
            A_.xop(RP_PUBLISH, bH, 2, wf, 2, F);                        // my slice of the heads, for member m of the E cluster
            A_.xop(RP_GATHER, bH, 7, wf, 2, F, 1);                      // dKL/d(f heads): every E member's slice
            dxS(bH, 2 * F, Wp("f.mean_linear.weight"), Hv, wh, ACT_RELU, gf.H2, Hv, &bA, false, GH2f, Hv);
            A_.xop(RP_XCHG, bA, 3, wh, 1, 0);
            dxS(bA, Hv, Wp("f.l2.weight"), Hv, wh, ACT_RELU, gf.H1, Hv, nullptr, false, GH1f, Hv);
        }
        A_.end(nrb, CS, 1);
    };
    bool use_rp = rl_rowprog_enabled();
    const bool use_cluster = rl_rowprog_cluster() && CS > 0 && ag->nsh[0] > 0;
    {
        RpAsm probe; if (use_cluster) rp_feature_cluster(probe); else rp_feature(probe, true);
        if (probe.lds_bytes() > RP_LDS_DYN_MAX || probe.ops.size() > 4096) use_rp = false;
        // the paired programs wait for each other inside the launch: every workgroup of it must be resident at once (256 CUs, as many
        // workgroups per CU as their LDS allows, at most 2 of these 512-thread ones) -- otherwise waiters could hold the chip while their
        // partners are never scheduled
        const int per_cu = std::max(1, std::min(2, (int)((160u * 1024u) / std::max<size_t>(probe.lds_bytes(), 1))));
        if (probe.blocks > 256 * per_cu) use_rp = false;
        for (auto& pr : probe.progs) if (pr.op_end - pr.op_begin > 40) use_rp = false;
    }
    auto with_adam = [&](GemmTask t, bool on, float* tw, float* tb) {
        if (!on || !ag->a.grad_dev || !t.C) return t;
        const int64_t ow = t.C - ag->a.grad_dev, ob = t.out2 - ag->a.grad_dev;
        t.flags |= FLAG_ADAM;
        t.ad_p = ag->a.param_dev + ow; t.ad_m = ag->a.exp_avg_dev + ow; t.ad_v = ag->a.exp_avg_sq_dev + ow; t.ad_t = tw;
        t.ad_pb = ag->a.param_dev + ob; t.ad_mb = ag->a.exp_avg_dev + ob; t.ad_vb = ag->a.exp_avg_sq_dev + ob; t.ad_tb = tb;
        t.ad_grp = ag->adam_step + 0;
        return t;
    };
    // fuse_l1: the weight-gradient tasks of encoder.l1 / f.l1 run their optimizer (and f.l1's Polyak into f_target.l1) in the epilogue
    // (FLAG_ADAM): the variant whose optimizer launch carries the next step's first layers (rlrep_feature_chain_next)
    auto feature_program = [&](Program& p, bool early, bool fuse_l1 = false) {
        GemmTask te[3], tf[3];
        gauss_tasks(ag, false, "encoder", s0.XE, KE, KE, ge, te);
        gauss_tasks(ag, false, "f", s0.XF, SA, SA, gf, tf);
        if (use_rp) {
            RpAsm A_; if (use_cluster && !early) rp_feature_cluster(A_); else rp_feature(A_, early);
            RpLaunch L; memset(&L, 0, sizeof(L));
            L.ops = b.upload(A_.ops); L.flags = rp_flags; L.nprog = (int)A_.progs.size(); L.B = B; L.low_prio = 0;
            L.lds_floats = A_.peak; L.xbuf = xbuf; L.epoch = ag->rp_epoch; L.err = ag->xc_err;
            for (size_t q = 0; q < A_.progs.size(); ++q) L.prog[q] = A_.progs[q];
            const int total = A_.blocks;
            rlrep_agent* a = ag;
            p.stages.push_back({[=](hipStream_t st) {
                RpLaunch l2 = L; l2.dyn[0] = a->cur_eps; l2.dyn[1] = a->cur_eps3; l2.dyn[2] = a->cur_eps2;
                return rl_launch_rowprog(&l2, total, st);
            }, early ? "row programs: encoder | f | policy(s') | policy(s): forward + dX" : "row programs: encoder | f: forward + dX"});
        } else
        b.chain_begin(p);              // everything up to the weight gradients is row-local: ONE persistent launch (xchain.hip)
        if (use_rp) {
        } else if (early) {
            b.fwd_stage(p, {te[0], tf[0], actor_l(ag, 0, s0.XF2, SA, ab), actor_l(ag, 0, s0.XFpi, SA, ab_pi)}, "enc.l1 f.l1 actor.l1(s') actor.l1(s)");
            b.fwd_stage(p, {te[1], tf[1], actor_l(ag, 1, nullptr, 0, ab), actor_l(ag, 1, nullptr, 0, ab_pi)}, "enc.l2 f.l2 actor.l2 x2");
        } else {
            // the first layers ride in the second layers' launch when their transposed shadows exist (one launch less per feature step)
            const bool have_t = ag->shadow_of.count("encoder.l1.weight") && ag->shadow_of.count("f.l1.weight");
            if (!have_t || !b.fwd_stage12(p, {{te[0], te[1], ag->PT("encoder.l1.weight")}, {tf[0], tf[1], ag->PT("f.l1.weight")}}, "enc.l1+l2 f.l1+l2")) {
                b.fwd_stage(p, {te[0], tf[0]}, "enc.l1 f.l1");
                b.fwd_stage(p, {te[1], tf[1]}, "enc.l2 f.l2");
            }
        }
        if (!use_rp) {
        {
            HeadsVae hv; memset(&hv, 0, sizeof(hv));
            hv.Ae = ge.H2; hv.Af = gf.H2; hv.lda = Hv; hv.We = Pw("encoder.mean_linear.weight"); hv.be = Pw("encoder.mean_linear.bias");
            hv.Wf = Pw("f.mean_linear.weight"); hv.bf = Pw("f.mean_linear.bias");
            hv.Z = Z; hv.EZ = EZ; hv.GEH = GEH; hv.GFH = GFH; hv.partial = part_kl; hv.EH = nullptr; hv.FH = nullptr;       // (nothing downstream of vae_mid reads the heads themselves)
            hv.B = B; hv.F = F; hv.K = Hv; hv.tiles_c = (F + 15) / 16; hv.scale = ag->inv_batch() / (float)F; hv.step = ag->adam_step + 0;
            b.heads_vae_stage(p, hv, "enc.heads f.heads + vae_mid");
        }
        if (early)                  // the two policy heads of the early variant ride with the next forward launch instead of the heads launch
            b.fwd_stage(p, {Builder::fwd(Z, F, B, F, Pw("decoder.l1.weight"), F, Pw("decoder.l1.bias"), Hv, D1, Hv, ACT_RELU),
                            policy_head_task(ag, ab, s0.XF2 + S, SA, FLAG_DYN_EPS3), policy_head_task(ag, ab_pi, s0.XFpi + S, SA, FLAG_DYN_EPS2)},
                        "dec.l1 actor.head x2 + policy");
        else
        b.fwd_stage(p, {Builder::fwd(Z, F, B, F, Pw("decoder.l1.weight"), F, Pw("decoder.l1.bias"), Hv, D1, Hv, ACT_RELU)}, "dec.l1");
        if (!fold_mse) {
            // decoder heads with the 0.5*mse loss fused into the epilogue: the launch writes d loss / d[s_hat | r_hat]
            // (GDH) directly and per-tile partial sums of the squared errors (vlsac_agent.py:137-140)
            GemmTask t = Builder::fwd(D1, Hv, B, Hv, Pw("decoder.state_linear.weight"), Hv, Pw("decoder.state_linear.bias"), S + 1, GDH, S + 1, ACT_NONE);
            t.epi = EPI_FWD_MSE; t.n0 = S; t.x0 = s0.XE ? s0.XE + SA : nullptr; t.ldx0 = KE; t.x1 = s0.R;
            t.s0 = ag->inv_batch() / (float)S; t.s1 = ag->inv_batch(); t.y0 = part_mse;
            b.fwd_stage(p, {t}, "dec.heads + mse");
        }
        {
            GemmTask t = Builder::dx(GD1, Hv, B, Hv, Pw("decoder.l1.weight"), F, GEH, 2 * F, F, ACT_NONE, nullptr, 0);
            t.epi = EPI_DX_REPARAM; t.aux3 = EZ; t.ldaux3 = F; t.F = F;
            // the K = 18 product dL/d(dec.l1 output) = d[s_hat|r_hat] W_heads rides in the dec.l1 dX launch (one launch less per feature step)
            if (fold_mse) {
                // ... and so does the heads' forward + mse (FLAG_PRE_MSE): every tile of the launch computes d[s_hat|r_hat] of its 16 rows itself
                t.flags |= FLAG_PRE | FLAG_PRE_MSE;
                t.x0 = GDH; t.ldx0 = S + 1; t.x1 = Pw("decoder.state_linear.weight"); t.ldx1 = Hv; t.n0 = S + 1; t.x2 = D1; t.ldaux2 = Hv; t.y0 = GD1; t.ldout2 = Hv;
                t.bias = Pw("decoder.state_linear.bias"); t.tgs = s0.XE ? s0.XE + SA : nullptr; t.ldtgs = KE; t.tgr = s0.R; t.pad_mse = S;
                t.s0 = ag->inv_batch() / (float)S; t.s1 = ag->inv_batch(); t.mse_part = part_mse;
                b.gemm_small(p, LD_ROW, LD_COL, {t}, "dec.heads + mse | dec.heads dx | dec.l1 dx -> (dmean, dlog_std)");
            } else
            b.dx_stage12(p, Builder::dx(GDH, S + 1, B, S + 1, Pw("decoder.state_linear.weight"), Hv, GD1, Hv, Hv, ACT_RELU, D1, Hv), t,
                         "dec.heads dx", "dec.l1 dx -> (dmean, dlog_std)");
        }
        b.dx_stage(p, {Builder::dx(GEH, 2 * F, B, 2 * F, Pw("encoder.mean_linear.weight"), Hv, GH2e, Hv, Hv, ACT_RELU, ge.H2, Hv),
                       Builder::dx(GFH, 2 * F, B, 2 * F, Pw("f.mean_linear.weight"), Hv, GH2f, Hv, Hv, ACT_RELU, gf.H2, Hv)}, "heads dx");
        b.dx_stage(p, {Builder::dx(GH2e, Hv, B, Hv, Pw("encoder.l2.weight"), Hv, GH1e, Hv, Hv, ACT_RELU, ge.H1, Hv),
                       Builder::dx(GH2f, Hv, B, Hv, Pw("f.l2.weight"), Hv, GH1f, Hv, Hv, ACT_RELU, gf.H1, Hv)}, "l2 dx");
        }
        b.chain_end();
        b.dw_stage(p, {Builder::dw(GDH, S + 1, S + 1, D1, Hv, Hv, B, Gw("decoder.state_linear.weight"), Hv, Gw("decoder.state_linear.bias")),
                       Builder::dw(GD1, Hv, Hv, Z, F, F, B, Gw("decoder.l1.weight"), F, Gw("decoder.l1.bias")),
                       Builder::dw(GEH, 2 * F, 2 * F, ge.H2, Hv, Hv, B, Gw("encoder.mean_linear.weight"), Hv, Gw("encoder.mean_linear.bias")),
                       Builder::dw(GFH, 2 * F, 2 * F, gf.H2, Hv, Hv, B, Gw("f.mean_linear.weight"), Hv, Gw("f.mean_linear.bias")),
                       Builder::dw(GH2e, Hv, Hv, ge.H1, Hv, Hv, B, Gw("encoder.l2.weight"), Hv, Gw("encoder.l2.bias")),
                       Builder::dw(GH2f, Hv, Hv, gf.H1, Hv, Hv, B, Gw("f.l2.weight"), Hv, Gw("f.l2.bias")),
                       with_adam(Builder::dw(GH1e, Hv, Hv, s0.XE, KE, KE, B, Gw("encoder.l1.weight"), KE, Gw("encoder.l1.bias")), fuse_l1, nullptr, nullptr),
                       with_adam(Builder::dw(GH1f, Hv, Hv, s0.XF, SA, SA, B, Gw("f.l1.weight"), SA, Gw("f.l1.bias")), fuse_l1,
                                 nft ? nullptr : Tw("f_target.l1.weight"), nft ? nullptr : Tw("f_target.l1.bias"))}, fuse_l1 ? "feature dW (+ adam l1)" : "feature dW");
    };
    feature_program(ag->feat_bwd, false);
    const bool can_hoist = policy_fusable(ag) && !rl_off("hoist");
    if (can_hoist && !rl_off("early_policy")) feature_program(ag->feat_bwd_h, true);
    {
        // apply: Adam over (encoder, decoder, f) + Polyak f -> f_target (vlsac_agent.py:152-154, 240-242)
        const std::vector<FinTask> feat_fins = {
            Builder::fin_sum(use_rp ? part_kl_rp : part_kl, use_rp ? nblk_rp * (use_cluster ? CS : 1) : nblk_kl, 1, 1.0f / ((float)B * F), ag->metrics + M_KL),
            Builder::fin_sum((use_rp ? part_mse_rp : part_mse) + 0, use_rp ? nblk_rp : nblk_mse, 2, 0.5f / ((float)B * S), ag->metrics + M_S_LOSS),
            Builder::fin_sum((use_rp ? part_mse_rp : part_mse) + 1, use_rp ? nblk_rp : nblk_mse, 2, 0.5f / (float)B, ag->metrics + M_R_LOSS),
            Builder::fin_combine(ag->metrics + M_R_LOSS, 1.f, ag->metrics + M_S_LOSS, 1.f, ag->metrics + M_FEAT_A),
            Builder::fin_combine(ag->metrics + M_FEAT_A, 1.f, ag->metrics + M_KL, 1.f, ag->metrics + M_FEAT_TOTAL),
            // the next row-program launch gets a fresh epoch for its exchange granules (harmless when none is used)
            Builder::fin_inc(ag->rp_epoch)};
        const LT& f0 = ag->L.get("f.l1.weight");
        const LT& flast = ag->L.get("f.log_std_linear.bias");
        const int64_t fn = flast.off + flast.rows - f0.off;
        if (nft) b.adam(ag->feat_apply, 0, ag->h.lr_feature, nullptr, 0, 0, 0.f, feat_fins, "adam feature");
        else b.adam(ag->feat_apply, 0, ag->h.lr_feature, Tw("f_target.l1.weight"), f0.off, fn, ag->h.feature_tau, feat_fins, "adam feature + polyak f");
        // chained form (rlrep_feature_chain_next): first layers' optimizer in the weight-gradient epilogues, the rest + the NEXT step's first layers in one launch
        if (!use_rp && !Builder::chain_enabled() && ag->h.world_size <= 1 && ag->nsh[0] == 0 && !rl_off("chain_next")) {
            feature_program(ag->feat_bwd_m, false, true);
            GemmTask te[3], tf[3];
            gauss_tasks(ag, false, "encoder", s0.XE, KE, KE, ge, te);
            gauss_tasks(ag, false, "f", s0.XF, SA, SA, gf, tf);
            const LT& ew = ag->L.get("encoder.l1.weight"); const LT& eb = ag->L.get("encoder.l1.bias");
            const LT& fw = ag->L.get("f.l1.weight"); const LT& fb = ag->L.get("f.l1.bias");
            const int64_t g0 = ag->L.group_off[0];
            b.adam_l1(ag->feat_apply_m, 0, ag->h.lr_feature, nft ? nullptr : Tw("f_target.l1.weight"), f0.off, nft ? 0 : fn, nft ? 0.f : ag->h.feature_tau, feat_fins,
                      ew.off - g0, eb.off + eb.rows - ew.off, fw.off - g0, fb.off + fb.rows - fw.off, te[0], tf[0],
                      "adam feature (- l1) + polyak f | next enc.l1 f.l1");
        }
    }

    // ---- critic / actor shared buffers ----
    GaussBufs gt{ws.f((size_t)B * Hv), ws.f((size_t)B * Hv), ws.f((size_t)B * 2 * F)};      // f_target(s, a)
    GaussBufs gn{ws.f((size_t)B * Hv), ws.f((size_t)B * Hv), ws.f((size_t)B * 2 * F)};      // f_target(s', a')
    GaussBufs gp{ws.f((size_t)B * Hv), ws.f((size_t)B * Hv), ws.f((size_t)B * 2 * F)};      // f_target(s, a_pi)
    float* HmT = ws.f((size_t)2 * B * H); float* HmC = ws.f((size_t)2 * B * H);
    float* U = ws.f((size_t)2 * B * N * H);
    float* Et = ws.f((size_t)2 * B * H); float* Ec = ws.f((size_t)2 * B * H);
    float* GE = ws.f((size_t)2 * B * H); float* GHm = ws.f((size_t)2 * B * H);
    float* dq = ws.f((size_t)2 * B);
    float* SIG = ws.f((size_t)B * F);
    float* GTH = ws.f((size_t)B * 2 * F); float* GT2 = ws.f((size_t)B * Hv); float* GT1 = ws.f((size_t)B * Hv);
    const int nblk = qhead_blocks(B);
    float* part_q = ws.f((size_t)4 * nblk); float* part_l = ws.f(nblk);
    const size_t BH = (size_t)B * H, BNH = (size_t)B * N * H;
    const float* noise = Tw("critic.noise");

    auto nc_task = [&](const float* HH, const float* W, const float* bias, float* Hm, float* Ubuf, const unsigned char* W3 = nullptr) {
        NcFwdTask t; memset(&t, 0, sizeof(t));
        t.mean = HH; t.lstd = HH ? HH + F : nullptr; t.ld_ml = 2 * F; t.noise = noise; t.W = W; t.W3 = W3; t.bias = bias; t.Hm = Hm; t.U = Ubuf;
        t.B = B; t.F = F; t.H = H; t.N = N;
        return t;
    };
    auto nc_stage = [&](Program& p, std::vector<NcFwdTask> tasks, const char* what) {
        // batch rows per workgroup = 4*g2: the largest tile that still gives every CU a workgroup (more
        // accumulators per wave amortise the LDS-table staging and the epilogue over more MFMAs)
        // measured on MI355X (B=256, F=H=256): 4 batch rows per workgroup (4 waves per SIMD) beats 8 and 16 rows
        // (39.8 / 41.6 / 52.2 us for the 4-head launch): one wave per SIMD cannot keep the f32 MFMA pipe busy
        // behind the VALU that builds its operands.
        // (the bf16x3 engine, when the shapes allow it, takes 8 rows: rl_nc_fwd_plan)
        int g2 = 1, engine = 0, cols = 128;
        rl_nc_fwd_plan(tasks.data(), (int)tasks.size(), &engine, &g2, &cols);
        NcFwdBatch nb; memset(&nb, 0, sizeof(nb));
        int base_tile = 0;
        nb.ntasks = (int)tasks.size(); nb.engine = engine; nb.cols = cols; nb.nt_u = rl_opt("nc_u_nt") ? 1 : 0;
        for (size_t q = 0; q < tasks.size(); ++q) {
            NcFwdTask& t = tasks[q];
            t.tiles_h = (H + cols - 1) / cols; t.ntiles = ((B + 4 * g2 - 1) / (4 * g2)) * t.tiles_h; t.tile_base = base_tile; base_tile += t.ntiles;
            nb.t[q] = t;
        }
        const int total = base_tile;
        p.stages.push_back({[=](hipStream_t st) { return rl_launch_nc_fwd(&nb, total, g2, st); }, what});
        {   // per head: [B*N, F] x [F, H]; reads mean / log_std / W, writes the noise-row mean (and U where the head keeps it)
            double by = 0.0;
            for (auto& t : tasks) by += 4.0 * (2.0 * (double)B * F + (double)F * H + (double)B * H + (t.U ? (double)B * N * H : 0.0));
            Builder::tag(p, RLREP_ENGINE_NOISE_CRITIC, (double)tasks.size() * 2.0 * (double)B * N * F * H, by);
        }
    };

    // ---- critic step (vlsac_agent.py:201-237) ----
    // hoist = true builds the variant that also carries the forward half of the FOLLOWING actor step (policy on s,
    // f_target on (s, a_pi)): those GEMMs read nothing the critic update writes, and as extra tasks of launches that
    // exist anyway they take six launches (~5 us each at B = 256) off the critical path of train().
    const std::vector<FinTask> cfins = {
        Builder::fin_sum(part_q + 0, nblk, 4, 1.0f / (float)B, ag->metrics + M_Q1_LOSS), Builder::fin_sum(part_q + 1, nblk, 4, 1.0f / (float)B, ag->metrics + M_Q2_LOSS),
        Builder::fin_sum(part_q + 2, nblk, 4, 1.0f / (float)B, ag->metrics + M_Q1), Builder::fin_sum(part_q + 3, nblk, 4, 1.0f / (float)B, ag->metrics + M_Q2)};
    // split-K slabs of the noise critic's weight gradient (bf16x3 form): ONE set for every variant of the critic program (no two of them are
    // ever in flight together), summed by the critic group's optimizer launch when there is no all-reduce between the two (AdamTask::Slab)
    const int ncdw_splits = rl_nc_dw_splits(B, F, H, 2);
    float* const ncdw_slab = ws.f((size_t)2 * ncdw_splits * H * F);
    float* const ncdw_bslab = ws.f((size_t)2 * ncdw_splits * H);
    const bool ncdw_in_adam = rl_nc_dw_engine() == 1 && ag->h.world_size <= 1 && ((H * F) & 3) == 0 && (H & 3) == 0 && ncdw_splits <= 16 && !rl_off("fold_ncdw");
    if (ncdw_in_adam) {
        const LT& w1 = ag->L.get("critic.l1.weight"); const LT& b1 = ag->L.get("critic.l1.bias");
        AdamTask::Slab sw; memset(&sw, 0, sizeof(sw));
        sw.off = w1.off - ag->L.group_off[1]; sw.n = 2ll * H * F; sw.per = (long long)H * F; sw.slab = ncdw_slab; sw.splits = ncdw_splits;
        AdamTask::Slab sb = sw;
        sb.off = b1.off - ag->L.group_off[1]; sb.n = 2ll * H; sb.per = H; sb.slab = ncdw_bslab;
        b.group_slabs[1] = {sw, sb};
    }
    auto critic_program = [&](Program& p, int hoist) {      // 0: plain, 1: carries the actor step's forward half, 2: both policies ran already
        GemmTask tt[3], tn[3], tp[3];
        gauss_tasks(ag, !nft, fnet, s0.XF, SA, SA, gt, tt);
        gauss_tasks(ag, !nft, fnet, s0.XF2, SA, SA, gn, tn);
        gauss_tasks(ag, !nft, fnet, s0.XFpi, SA, SA, gp, tp);
        if (hoist == 2) {
            b.fwd_stage(p, {tt[0], tn[0], tp[0]}, "ft.l1(s,a) ft.l1(s',a') ft.l1(s,a_pi)");
            b.fwd_stage(p, {tt[1], tn[1], tp[1]}, "ft.l2 x3");
            b.fwd_stage(p, {tt[2], tn[2], tp[2]}, "ft.heads x3");
        } else if (hoist == 1) {
            b.fwd_stage(p, {actor_l(ag, 0, s0.XF2, SA, ab), actor_l(ag, 0, s0.XFpi, SA, ab_pi), tt[0]}, "actor.l1(s') actor.l1(s) ft.l1(s,a)");
            b.fwd_stage(p, {actor_l(ag, 1, nullptr, 0, ab), actor_l(ag, 1, nullptr, 0, ab_pi), tt[1]}, "actor.l2 x2 ft.l2");
            b.fwd_stage(p, {policy_head_task(ag, ab, s0.XF2 + S, SA, FLAG_DYN_EPS), policy_head_task(ag, ab_pi, s0.XFpi + S, SA, FLAG_DYN_EPS2), tt[2]},
                        "actor.head x2 ft.heads + policy");
            b.fwd_stage(p, {tn[0], tp[0]}, "ft.l1(s',a') ft.l1(s,a_pi)");
            b.fwd_stage(p, {tn[1], tp[1]}, "ft.l2 x2");
            b.fwd_stage(p, {tn[2], tp[2]}, "ft.heads x2");
        } else {
            b.fwd_stage(p, {actor_l(ag, 0, s0.XF2, SA, ab), tt[0]}, "actor.l1(s') ft.l1(s,a)");
            b.fwd_stage(p, {actor_l(ag, 1, nullptr, 0, ab), tt[1]}, "actor.l2 ft.l2");
            actor_head_stage(b, p, ag, ab, s0.XF2 + S, SA, {tt[2]}, "actor.head ft.heads + policy");
            b.fwd_stage(p, {tn[0]}, "ft.l1(s',a')");
            b.fwd_stage(p, {tn[1]}, "ft.l2");
            b.fwd_stage(p, {tn[2]}, "ft.heads");
        }
        {
            NcFwdTask live1 = nc_task(gt.HH, Pw("critic.l1.weight"), Pw("critic.l1.bias"), HmC, U, ag->W3("critic.l1.weight"));
            live1.sigma_out = SIG;
            nc_stage(p, {nc_task(gn.HH, Tw("critic_target.l1.weight"), Tw("critic_target.l1.bias"), HmT, nullptr, ag->W3("critic_target.l1.weight")),
                         nc_task(gn.HH, Tw("critic_target.l4.weight"), Tw("critic_target.l4.bias"), HmT + BH, nullptr, ag->W3("critic_target.l4.weight")),
                         live1,
                         nc_task(gt.HH, Pw("critic.l4.weight"), Pw("critic.l4.bias"), HmC + BH, U + BNH, ag->W3("critic.l4.weight"))}, "noise critic l1/l4 (target+live)");
        }
        b.fwd_stage(p, {Builder::fwd(HmT, H, B, H, Tw("critic_target.l2.weight"), H, Tw("critic_target.l2.bias"), H, Et, H, ACT_ELU),
                        Builder::fwd(HmT + BH, H, B, H, Tw("critic_target.l5.weight"), H, Tw("critic_target.l5.bias"), H, Et + BH, H, ACT_ELU),
                        Builder::fwd(HmC, H, B, H, Pw("critic.l2.weight"), H, Pw("critic.l2.bias"), H, Ec, H, ACT_ELU),
                        Builder::fwd(HmC + BH, H, B, H, Pw("critic.l5.weight"), H, Pw("critic.l5.bias"), H, Ec + BH, H, ACT_ELU)}, "critic l2/l5");
        QHeadCritic q; memset(&q, 0, sizeof(q));
        q.Et[0] = Et; q.Et[1] = Et + BH; q.Ec[0] = Ec; q.Ec[1] = Ec + BH;
        // quirk Q2: BOTH heads end in l3 (l6 is dead)
        q.wt[0] = q.wt[1] = Tw("critic_target.l3.weight"); q.bt[0] = q.bt[1] = Tw("critic_target.l3.bias");
        q.wc[0] = q.wc[1] = Pw("critic.l3.weight"); q.bc[0] = q.bc[1] = Pw("critic.l3.bias");
        q.logp = ab.logp; q.R = s0.R; q.D = s0.D; q.alpha_state = ag->a.alpha_state_dev; q.gamma = ag->h.discount;
        q.inv_batch = ag->inv_batch(); q.dq = dq; q.GE[0] = GE; q.GE[1] = GE + BH; q.partial = part_q;
        q.B = B; q.H = H; q.nblk = nblk; q.train = 1; q.step = ag->adam_step + 1;
        p.stages.push_back({[=](hipStream_t st) { return rl_launch_qhead_critic(&q, st); }, "qhead critic"});
        b.dx_stage(p, {Builder::dx(GE, H, B, H, Pw("critic.l2.weight"), H, GHm, H, H, ACT_NONE, nullptr, 0),
                       Builder::dx(GE + BH, H, B, H, Pw("critic.l5.weight"), H, GHm + BH, H, H, ACT_NONE, nullptr, 0)}, "critic l2/l5 dx");
        b.dw_stage(p, {Builder::dw(dq, 1, 1, Ec, H, H, 2 * B, Gw("critic.l3.weight"), H, Gw("critic.l3.bias")),     // shared l3: heads stacked
                       Builder::dw(GE, H, H, HmC, H, H, B, Gw("critic.l2.weight"), H, Gw("critic.l2.bias")),
                       Builder::dw(GE + BH, H, H, HmC + BH, H, H, B, Gw("critic.l5.weight"), H, Gw("critic.l5.bias"))}, "critic dW l3 l2 l5");
        {
            NcDwBatch nb; memset(&nb, 0, sizeof(nb));
            nb.ntasks = 2;
            nb.lean = b.low_prio ? 1 : 0;     // deferred chain (fp32 kernel only): leave registers for the feature chain's launches
            int base_tile = 0;
            auto ncdw = [&](int q, float* Ubuf, float* GH, float* gW, float* gb) {
                NcDwTask& t = nb.t[q];
                t.U = Ubuf; t.GH = GH; t.ldgh = H; t.mean = gt.HH; t.sigma = SIG; t.ld_ml = 2 * F;
                t.noise = noise; t.gW = gW; t.gb = gb; t.B = B; t.F = F; t.H = H; t.N = N;
                t.tiles_k = (F + 31) / 32; t.ntiles = ((H + 15) / 16) * t.tiles_k; t.tile_base = base_tile; base_tile += t.ntiles;
            };
            ncdw(0, U, GHm, Gw("critic.l1.weight"), Gw("critic.l1.bias"));
            ncdw(1, U + BNH, GHm + BH, Gw("critic.l4.weight"), Gw("critic.l4.bias"));
            // bf16x3 split-K form: 64 x 64 tiles x splits, partial tiles in workspace slabs (reserved in the dry pass as well)
            nb.splits = ncdw_splits; nb.slab = ncdw_slab; nb.bslab = ncdw_bslab;
            nb.engine = rl_nc_dw_engine();
            nb.fin_in_adam = ncdw_in_adam ? 1 : 0;
            if (nb.engine == 1) {
                base_tile = 0;
                for (int q = 0; q < 2; ++q) {
                    nb.t[q].ntiles = ((H + 63) / 64) * ((F + 63) / 64) * nb.splits; nb.t[q].tile_base = base_tile; base_tile += nb.t[q].ntiles;
                }
            }
            const int total = base_tile;
            p.stages.push_back({[=](hipStream_t st) { return rl_launch_nc_dw(&nb, total, st); }, "noise critic dW l1/l4"});
            Builder::tag(p, RLREP_ENGINE_NOISE_CRITIC, 2.0 * 2.0 * (double)B * N * F * H, 2.0 * 4.0 * ((double)B * N * H + (double)B * H + 2.0 * (double)B * F + (double)F * H));
        }
    };
    critic_program(ag->critic_bwd, 0);
    if (can_hoist) critic_program(ag->critic_bwd_h, 1);
    if (!ag->feat_bwd_h.stages.empty()) critic_program(ag->critic_bwd_h2, 2);
    b.adam(ag->critic_apply, 1, ag->h.lr_critic, nullptr, 0, 0, 0.f, cfins, "adam critic");
    critic_apply_folded(b, ag, "critic_target.l1.weight", cfins);

    // ---- actor + temperature step (vlsac_agent.py:165-198) ----
    auto actor_program = [&](Program& p, int& resume) {
        GemmTask tt[3];
        gauss_tasks(ag, !nft, fnet, s0.XFpi, SA, SA, gp, tt);
        b.fwd_stage(p, {actor_l(ag, 0, s0.XFpi, SA, ab_pi)}, "actor.l1(s)");
        b.fwd_stage(p, {actor_l(ag, 1, nullptr, 0, ab_pi)}, "actor.l2");
        actor_head_stage(b, p, ag, ab_pi, s0.XFpi + S, SA, {}, "actor.head + policy");
        b.fwd_stage(p, {tt[0]}, "ft.l1(s,a_pi)");
        b.fwd_stage(p, {tt[1]}, "ft.l2");
        b.fwd_stage(p, {tt[2]}, "ft.heads");
        resume = (int)p.stages.size();                // everything above is what critic_bwd_h already did
        nc_stage(p, {nc_task(gp.HH, Pw("critic.l1.weight"), Pw("critic.l1.bias"), HmC, U, ag->W3("critic.l1.weight")),
                     nc_task(gp.HH, Pw("critic.l4.weight"), Pw("critic.l4.bias"), HmC + BH, U + BNH, ag->W3("critic.l4.weight"))}, "noise critic l1/l4");
        b.fwd_stage(p, {Builder::fwd(HmC, H, B, H, Pw("critic.l2.weight"), H, Pw("critic.l2.bias"), H, Ec, H, ACT_ELU),
                        Builder::fwd(HmC + BH, H, B, H, Pw("critic.l5.weight"), H, Pw("critic.l5.bias"), H, Ec + BH, H, ACT_ELU)}, "critic l2/l5");
        QHeadActor q; memset(&q, 0, sizeof(q));
        q.Ec[0] = Ec; q.Ec[1] = Ec + BH; q.wc[0] = q.wc[1] = Pw("critic.l3.weight"); q.bc[0] = q.bc[1] = Pw("critic.l3.bias");
        q.logp = ab_pi.logp; q.alpha_state = ag->a.alpha_state_dev; q.inv_batch = ag->inv_batch(); q.target_entropy = ag->h.target_entropy;
        q.GE[0] = GE; q.GE[1] = GE + BH; q.partial_loss = part_l; q.partial_c = ag->Gtail();
        q.B = B; q.H = H; q.nblk = nblk; q.step = ag->adam_step + 2;
        p.stages.push_back({[=](hipStream_t st) { return rl_launch_qhead_actor(&q, st); }, "qhead actor"});
        b.dx_stage(p, {Builder::dx(GE, H, B, H, Pw("critic.l2.weight"), H, GHm, H, H, ACT_NONE, nullptr, 0),
                       Builder::dx(GE + BH, H, B, H, Pw("critic.l5.weight"), H, GHm + BH, H, H, ACT_NONE, nullptr, 0)}, "critic l2/l5 dx");
        {
            NcDxTask t; memset(&t, 0, sizeof(t));
            t.GH[0] = GHm; t.GH[1] = GHm + BH; t.ldgh = H; t.U[0] = U; t.U[1] = U + BNH;
            t.W[0] = Pw("critic.l1.weight"); t.W[1] = Pw("critic.l4.weight"); t.noise = noise;
            t.lstd = gp.HH ? gp.HH + F : nullptr; t.ld_l = 2 * F; t.G = GTH; t.ldg = 2 * F;
            t.B = B; t.F = F; t.H = H; t.N = N; t.nheads = 2;
            t.tiles_k = (F + 63) / 64; t.ntiles = ((B + 3) / 4) * t.tiles_k; t.tile_base = 0;
            p.stages.push_back({[=](hipStream_t st) { return rl_launch_nc_dx(&t, st); }, "noise critic dX -> (dmean, dlog_std)"});
            Builder::tag(p, RLREP_ENGINE_NOISE_CRITIC, 2.0 * 2.0 * (double)B * N * F * H, 4.0 * (2.0 * ((double)B * N * H + (double)B * H + (double)F * H) + 3.0 * (double)B * F));
        }
        b.dx_stage(p, {Builder::dx(GTH, 2 * F, B, 2 * F, FT("mean_linear.weight"), Hv, GT2, Hv, Hv, ACT_RELU, gp.H2, Hv)}, "ft.heads dx");
        b.dx_stage(p, {Builder::dx(GT2, Hv, B, Hv, FT("l2.weight"), Hv, GT1, Hv, Hv, ACT_RELU, gp.H1, Hv)}, "ft.l2 dx");
        actor_backward(b, p, ag, ab_pi, s0.XFpi, SA, s0.XFpi + S, SA, Builder::dx(GT1, Hv, B, Hv, FT("l1.weight") ? FT("l1.weight") + S : nullptr, SA, ab_pi.dA, A, A, ACT_NONE, nullptr, 0));
    };
    actor_program(ag->actor_bwd, ag->actor_resume);
    actor_apply_program(b, ag, part_l, nblk);
    update_target_program(ag, "critic.l1.weight", "critic_target.l1.weight");

    // ---- deferred variants: the same critic / actor programs against a snapshot set (see rlrep_agent::dset) ----
    for (int set = 0; set < rlrep_agent::NSETS; ++set) {
        const LT& f0 = ag->L.get(fnet + ".l1.weight");
        const LT& fl = ag->L.get(fnet + ".log_std_linear.bias");
        float* fbase = nft ? ag->a.param_dev : ag->a.target_dev;          // the snapshot is of whichever copy the two steps read
        const Slot keep = defer_begin(b, ag, set, nft ? "f." : "f_target.", nft ? "f.l1.weight" : "f_target.l1.weight", fbase ? fbase + f0.off : nullptr,
                                      fl.off + fl.rows - f0.off);
        {
            // the same block as the feature group's optimizer launch produces it (rlrep_defer_arm): the live f's range of the group, or the
            // range its Polyak writes into f_target (both start at f.l1.weight's offset in the group)
            rlrep_agent::DeferSet& D = ag->dset[set];
            const LT& p0 = ag->L.get("f.l1.weight");
            D.block_off = p0.off - ag->L.group_off[0]; D.block_n = fl.off + fl.rows - f0.off; D.block_which = nft ? 0 : 1;
            // (with N > 1 ranks too: the optimizer launch that writes the snapshot is the one that has summed the ranks' gradients)
            if (rl_off("fold_snapshot")) D.block_which = -1;
        }
        critic_program(ag->dset[set].critic_bwd, can_hoist ? 1 : 0);
        actor_program(ag->dset[set].actor_bwd, ag->dset[set].actor_resume);
        if (!can_hoist) ag->dset[set].actor_resume = 0;
        defer_end(b, ag, set, keep, "critic_target.l1.weight", cfins);
    }
}

Slot defer_begin(Builder& b, rlrep_agent* ag, int set, const char* prefix, const char* first, const float* block_src, int64_t block_n) {
    Workspace& ws = b.ws;
    const int B = ag->B, S = ag->d.state_dim, A = ag->d.action_dim, SA = S + A;
    Slot& s0 = ag->slot[0];
    rlrep_agent::DeferSet& D = ag->dset[set];
    Slot d; d.XE = nullptr; d.XF = ws.f((size_t)B * SA); d.XF2 = ws.f((size_t)B * SA); d.XFpi = ws.f((size_t)B * SA); d.R = ws.f(B); d.D = ws.f(B);
    D.slot = d;
    D.block = block_n > 0 ? ws.f((size_t)block_n) : nullptr;
    D.eps = ws.f((size_t)2 * B * A); D.steps = (int*)ws.alloc(sizeof(int) * 4);
    CopySegs& cs = D.segs; memset(&cs, 0, sizeof(cs));
    long long end = 0; int n = 0;
    auto seg = [&](const float* src, float* dst, long long cnt) { cs.src[n] = src; cs.dst[n] = dst; end += cnt; cs.end[n] = end; ++n; };
    seg(s0.XF, d.XF, (long long)B * SA); seg(s0.XF2, d.XF2, (long long)B * SA); seg(s0.XFpi, d.XFpi, (long long)B * SA);
    seg(s0.R, d.R, B); seg(s0.D, d.D, B);
    if (block_n > 0) seg(block_src, D.block, block_n);
    seg(nullptr, D.eps, (long long)B * A);                             // critic-step policy noise (patched per call)
    seg(nullptr, D.eps ? D.eps + (size_t)B * A : nullptr, (long long)B * A);   // actor-step policy noise
    cs.n = n; cs.isrc = ag->steps; cs.idst = D.steps;
    const Slot keep = s0;
    s0.XF = d.XF; s0.XF2 = d.XF2; s0.XFpi = d.XFpi; s0.R = d.R; s0.D = d.D;
    ag->ov_base = b.dry ? nullptr : D.block; ag->ov_prefix = prefix; ag->ov_first = first;      // the dry pass only sizes the workspace
    ag->dcur = set;
    b.low_prio = true;
    return keep;
}
void defer_end(Builder& b, rlrep_agent* ag, int set, const Slot& keep, const std::string& critic_target_first, std::vector<FinTask> cfins) {
    ag->ov_base = nullptr;
    ag->slot[0] = keep;
    b.low_prio = false;
    critic_apply_folded(b, ag, critic_target_first, cfins, &ag->dset[set].critic_apply, ag->dset[set].steps);
    ag->dset[set].valid = false;
}

// ================================================================================================
// (re)build for a batch size
// ================================================================================================
static int build_programs(rlrep_agent* ag, int B) {
    ag->B = B;
    ag->ws.used = ag->ws_static;
    for (Program* p : {&ag->feat_bwd, &ag->feat_apply, &ag->critic_bwd, &ag->critic_apply, &ag->actor_bwd, &ag->actor_apply, &ag->upd_target, &ag->infer, &ag->sync_prog, &ag->critic_bwd_h, &ag->critic_apply_f, &ag->feat_bwd_h, &ag->critic_bwd_h2, &ag->feat_bwd_m, &ag->feat_apply_m})
        p->stages.clear();
    for (auto& D : ag->dset) for (Program* p : {&D.critic_bwd, &D.critic_apply, &D.actor_bwd}) p->stages.clear();
    ag->infer_n = 0; ag->actor_resume = 0; ag->pi_ready = ag->hoist_req = nullptr; ag->in_train = ag->target_done = false;
    ag->pf_armed = ag->pf_done = false; ag->pf2_armed = ag->pf2_done = false;
    ag->chain_next = ag->chain_bwd_done = ag->l1_done = false;
    ag->early_crit = ag->early_act = ag->early_ready_crit = ag->early_ready_act = nullptr;
    ag->feat_cuts.clear();
    Builder b(ag);
    const int S = ag->d.state_dim, A = ag->d.action_dim;
    {
        // spedersac steps on two minibatches; their [s,a,s'] / [s,a] matrices are allocated back to back so
        // that phi/mu run as ONE 2B-row GEMM per layer
        const int ns = (ag->d.alg == RLREP_ALG_SPEDERSAC) ? 2 : 1;
        float* XE = b.ws.f((size_t)ns * B * (2 * S + A));
        float* XF = b.ws.f((size_t)ns * B * (S + A));
        for (int i = 0; i < ns; ++i) {
            Slot& s = ag->slot[i];
            s.XE = XE ? XE + (size_t)i * B * (2 * S + A) : nullptr;
            s.XF = XF ? XF + (size_t)i * B * (S + A) : nullptr;
            s.XF2 = b.ws.f((size_t)B * (S + A)); s.XFpi = b.ws.f((size_t)B * (S + A)); s.R = b.ws.f(B); s.D = b.ws.f(B); s.filled = false;
        }
    }
    switch (ag->d.alg) {
    case RLREP_ALG_SAC: build_sac(b, ag); break;
    case RLREP_ALG_VLSAC: build_vlsac(b, ag); break;
    case RLREP_ALG_CTRLSAC: build_ctrlsac(b, ag); break;
    case RLREP_ALG_SPEDERSAC: build_spedersac(b, ag); break;
    case RLREP_ALG_DIFFSRSAC: build_diffsrsac(b, ag); break;
    default: rl_set_error("unknown algorithm %d", ag->d.alg); return RLREP_ERR_ARG;
    }
    ag->prog_end = ag->ws.used;
    // room for the B<=max_batch inference program (rlrep_actor_forward)
    const size_t infer_bytes = (size_t)ag->d.max_batch * (2 * ag->d.actor_hidden_dim + 2 * ag->d.action_dim) * sizeof(float) + 8192;
    if (!ag->ws.dry && ag->ws.used + infer_bytes > ag->ws.cap) { rl_set_error("workspace too small: need %zu bytes, have %zu", ag->ws.used + infer_bytes, ag->ws.cap); return RLREP_ERR_NOMEM; }
    if (ag->ws.dry) ag->ws.used += infer_bytes;
    return 0;
}

static bool check_dims(const rlrep_dims* d) {
    if (!d || d->state_dim <= 0 || d->action_dim <= 0 || d->hidden_dim <= 0 || d->actor_hidden_dim <= 0 || d->max_batch <= 0) {
        rl_set_error("bad dimensions"); return false;
    }
    if (d->alg == RLREP_ALG_VLSAC) {
        if (d->num_noise != 4 * NC_NF_HOST) { rl_set_error("vlsac: num_noise must be %d", 4 * NC_NF_HOST); return false; }
        if (d->feature_dim <= 0 || d->vae_hidden_dim <= 0 || (d->feature_dim & 3)) { rl_set_error("vlsac: feature_dim must be a positive multiple of 4"); return false; }
        if ((size_t)(16 + d->num_noise) * (((d->feature_dim + 15) & ~15) + 16) * 4 > 64 * 1024) { rl_set_error("vlsac: feature_dim too large for the LDS-resident noise tables"); return false; }
    }
    if (d->alg == RLREP_ALG_CTRLSAC || d->alg == RLREP_ALG_SPEDERSAC || d->alg == RLREP_ALG_DIFFSRSAC) {
        if (d->feature_dim <= 0 || d->phi_hidden_dim <= 0 || d->mu_hidden_dim <= 0 || d->phi_hidden_depth < 0 || d->mu_hidden_depth < 0 ||
            d->phi_hidden_depth > 3 || d->mu_hidden_depth > 3) { rl_set_error("bad representation-network dimensions"); return false; }
        if (d->alg == RLREP_ALG_DIFFSRSAC && d->num_noise <= 0) { rl_set_error("diffsrsac: num_noise must be positive"); return false; }
    }
    return true;
}

static void static_state(rlrep_agent* ag) {
    Workspace& ws = ag->ws;
    ws.used = 0;
    ag->steps = (int*)ws.alloc(sizeof(int) * 8);
    ag->adam_step = (GroupCfg*)ws.alloc(sizeof(GroupCfg) * 4);     // [4] optimizer groups
    ag->metrics = ws.f(M_COUNT);
    const int S = ag->d.state_dim, A = ag->d.action_dim;
    ag->obs_in = ws.f((size_t)ag->d.max_batch * S);
    ag->act_out = ws.f((size_t)ag->d.max_batch * A);
    ag->rp_epoch = (int*)ws.alloc(sizeof(int) * 4);
    ag->hist = ws.f((size_t)RL_HIST_N * RL_HIST_REC);
    ag->hist_seq = (int*)ws.alloc(sizeof(int) * 4);
    if (!ws.dry && ws.ok()) { (void)hipMemset(ag->hist, 0xff, sizeof(float) * RL_HIST_N * RL_HIST_REC); (void)hipMemset(ag->hist_seq, 0, sizeof(int) * 4); }
    ag->xc_err = (unsigned*)ws.alloc(256);
    if (!ws.dry && ws.ok()) (void)hipMemset(ag->xc_err, 0, 256);
    if (!ws.dry && ws.ok()) { const int one[4] = {1, 0, 0, 0}; (void)hipMemcpy(ag->rp_epoch, one, sizeof(one), hipMemcpyHostToDevice); }
    // transposed weight shadows (see rlrep_agent::sh_dev): vlsac's feature group, read by the feature step's row programs
    ag->shadow_of.clear();
    for (int g = 0; g < 4; ++g) { ag->sh_dev[g] = nullptr; ag->nsh[g] = ag->sh_tiles[g] = 0; }
    // ... and, with RLREP_ENABLE=fuse_l1, the two FIRST layers of its feature nets (K = 2S+A / S+A) for the variant in which they ride in the
    // second layers' launch (Builder::fwd_stage12 reads W1 transposed).  OPT-IN: measured 11.2 us for the fused launch against 3.7 + 4.9 us
    // for the pair (every one of the 512 tiles re-reads W1^T and X: twice the L2 traffic, three times the load instructions): 2 763 vs
    // 2 865 train()/s.
    const char* fl1 = rl_opt("fuse_l1");
#ifdef RL_EXPERIMENTS
    const bool sh_all = rl_rowprog_enabled(), sh_l1 = fl1 && fl1[0] == '1';
#else
    const bool sh_all = false, sh_l1 = false; (void)fl1;       // (the fused-first-layers launch is an experiments-build kernel)
#endif
    if (ag->d.alg == RLREP_ALG_VLSAC && (sh_all || sh_l1) && !rl_off("shadows")) {
        std::vector<ShadowEnt> tab; std::vector<std::string> first;
        const auto& T = ag->L.t;
        for (size_t q = 0; q < T.size(); ++q) {
            const LT& e = T[q];
            if (e.arena != RLREP_ARENA_PARAM || e.group != 0 || e.cols <= 1 || e.name.find(".weight") == std::string::npos) continue;
            if (!sh_all && e.name != "encoder.l1.weight" && e.name != "f.l1.weight") continue;
            // a glued pair (mean | log_std heads, state | reward heads: Layout::lin_pair) is two consecutive blocks [o1, in], [o2, in] = ONE
            // [o1 + o2, in] matrix for the kernels: one shadow, reachable under the first tensor's name
            if (!tab.empty() && q > 0 && T[q - 1].name.find(".weight") != std::string::npos && tab.back().off + tab.back().n == e.off - ag->L.group_off[0] && tab.back().cols == e.cols) {
                tab.back().rows += e.rows; tab.back().n += (long long)e.rows * e.cols;
                continue;
            }
            ShadowEnt se; memset(&se, 0, sizeof(se));
            // offsets RELATIVE to the group's start: the group's Adam launch indexes its arena slice from 0, and the refresh launches are
            // given param_dev + group_off[0] as their base (one table serves both, wherever the group sits in the arena)
            se.off = e.off - ag->L.group_off[0]; se.n = (long long)e.rows * e.cols; se.rows = e.rows; se.cols = e.cols;
            tab.push_back(se); first.push_back(e.name);
        }
        int tiles = 0;
        for (size_t k = 0; k < tab.size(); ++k) {
            tab[k].sp = ws.f((size_t)tab[k].n);
            ag->shadow_of[first[k]] = tab[k].sp;
            tiles += ((tab[k].rows + 31) / 32) * ((tab[k].cols + 31) / 32);
        }
        ShadowEnt* dev = (ShadowEnt*)ws.alloc(tab.size() * sizeof(ShadowEnt));
        if (!ws.dry && ws.ok() && !tab.empty()) (void)hipMemcpy(dev, tab.data(), tab.size() * sizeof(ShadowEnt), hipMemcpyHostToDevice);
        ag->sh_dev[0] = dev; ag->nsh[0] = (int)tab.size(); ag->sh_tiles[0] = tiles;
    }
    // bf16x3 images of the noise critic's first layers (vlsac): see rlrep_agent::x3_refresh.  (Where the shape rules them out -- F not a
    // multiple of 32 -- the kernels split W themselves, as in round 1.)
    ag->x3_refresh = nullptr; ag->x3_n = ag->x3_tiles = 0; ag->x3_of.clear();
    {
        const int F = ag->d.feature_dim, H = ag->d.hidden_dim;
        if (ag->d.alg == RLREP_ALG_VLSAC && !rl_off("x3") && F > 0 && (F % 32) == 0 && ag->L.index.count("critic.l1.weight") && ag->L.index.count("critic_target.l1.weight")) {
            std::vector<ShadowEnt> all, live;
            const char* names[4] = {"critic.l1.weight", "critic.l4.weight", "critic_target.l1.weight", "critic_target.l4.weight"};
            for (int q = 0; q < 4; ++q) {
                const LT& t = ag->L.get(names[q]);
                ShadowEnt se; memset(&se, 0, sizeof(se));
                se.off = t.off; se.n = (long long)t.rows * t.cols; se.rows = t.rows; se.cols = t.cols; se.kind = 1;
                unsigned char* img = (unsigned char*)ws.alloc((size_t)3 * t.rows * t.cols * 2);
                se.sp = reinterpret_cast<float*>(img);
                se.src = q < 2 ? (ag->a.param_dev ? ag->a.param_dev + t.off : nullptr) : (ag->a.target_dev ? ag->a.target_dev + t.off : nullptr);
                ag->x3_of[names[q]] = img;
                all.push_back(se);
                ag->x3_tiles += ((t.rows + 31) / 32) * ((t.cols + 31) / 32);
                if (q < 2) { ShadowEnt lv = se; lv.src = nullptr; lv.off = t.off - ag->L.group_off[1]; live.push_back(lv); }     // (the Adam launch indexes from the group's start)
                (void)H;
            }
            // the critic's Adam launch carries the critic -> critic_target Polyak inside a train(): its lanes then keep the TARGET images current too
            for (size_t q = 0; q < live.size(); ++q) live[q].st = all[q + 2].sp;
            ShadowEnt* dev = (ShadowEnt*)ws.alloc(all.size() * sizeof(ShadowEnt));
            ShadowEnt* dev_live = (ShadowEnt*)ws.alloc(live.size() * sizeof(ShadowEnt));
            if (!ws.dry && ws.ok()) {
                (void)hipMemcpy(dev, all.data(), all.size() * sizeof(ShadowEnt), hipMemcpyHostToDevice);
                (void)hipMemcpy(dev_live, live.data(), live.size() * sizeof(ShadowEnt), hipMemcpyHostToDevice);
            }
            ag->x3_refresh = dev; ag->x3_n = (int)all.size();
            // the critic group's Adam launch keeps the live images current (the actor step reads them right after the critic update)
            ag->sh_dev[1] = dev_live; ag->nsh[1] = (int)live.size(); ag->sh_tiles[1] = 0;      // (no tiles: group 1 is refreshed through x3_refresh)
        }
    }
    ag->ws_static = ws.used;
}

// regenerate the shadows of group g from the parameters as they stand (a launch of its own: the eager entry points)
static int refresh_shadows(rlrep_agent* ag, void* stream) {
    for (int g = 0; g < 4; ++g) {
        if (!ag->nsh[g] || !ag->sh_tiles[g]) continue;
        const int rc = rl_launch_shadow(ag->sh_dev[g], ag->nsh[g], ag->sh_tiles[g], ag->a.param_dev + ag->L.group_off[g], 0, (hipStream_t)stream); ++g_rl_launches;
        if (rc) { rl_set_error("shadow refresh: hip error %d", rc); return RLREP_ERR_HIP; }
    }
    return 0;
}

// regenerate the bf16x3 images of the noise critic's first layers, live and target, from the tensors as they stand (one launch)
static int refresh_x3(rlrep_agent* ag, void* stream) {
    if (!ag->x3_n) return 0;
    const int rc = rl_launch_shadow(ag->x3_refresh, ag->x3_n, ag->x3_tiles, nullptr, 0, (hipStream_t)stream); ++g_rl_launches;
    if (rc) { rl_set_error("x3 shadow refresh: hip error %d", rc); return RLREP_ERR_HIP; }
    return 0;
}

// ================================================================================================
// C ABI
// ================================================================================================
// floats of exchange scratch the batch-coupled feature exchanges of this agent need once attached (rlrep_layout_info.exchange_floats)
static long long exchange_floats(const rlrep_dims& d, int world) {
    if (world <= 1) return 0;
    const long long F = d.feature_dim, B = d.max_batch, W = world;
    if (d.alg == RLREP_ALG_SPEDERSAC) return F <= RL_SLOTS_MAX_F ? 2 * (2 * W * ((F + 63) & ~63ll)) : 0;      // Phibar and v: [2 parities][world][F] each
    if (d.alg == RLREP_ALG_CTRLSAC) return 2 * W * B * F;                                                      // mu(s') of all ranks + its partial gradients
    return 0;
}

extern "C" {

int32_t rlrep_abi_version(void) { return RLREP_ABI_VERSION; }
const char* rlrep_last_error(void) { return g_err; }

int32_t rlrep_layout(const rlrep_dims* dims, rlrep_layout_info* info, rlrep_tensor_desc* descs, int32_t cap) {
    rl_switches_read();
    if (!check_dims(dims) || !info) return RLREP_ERR_ARG;
    rlrep_agent tmp;
    tmp.d = *dims; memset(&tmp.h, 0, sizeof(tmp.h)); memset(&tmp.a, 0, sizeof(tmp.a));
    tmp.h.world_size = dims->world_size > 1 ? dims->world_size : 1;       // (sizes the workspace of the batch-coupled feature steps: ctrlsac's [B, W B] score matrix)
    if (!build_layout(*dims, tmp.L)) return RLREP_ERR_ARG;
    tmp.L.align(RLREP_ARENA_PARAM); tmp.L.align(RLREP_ARENA_TARGET);
    tmp.ws.dry = true;
    static_state(&tmp);
    if (build_programs(&tmp, dims->max_batch) != 0) return RLREP_ERR_ARG;
    if (tmp.h.world_size > 1 && exchange_floats(*dims, tmp.h.world_size) > 0) {
        // ... and the ATTACHED form (rl_agent_attach_dp: exchanges inside the launches, deferred step programs kept): the larger of the two
        const size_t plain = tmp.ws.used;
        tmp.xfold = true; tmp.xscratch_floats = exchange_floats(*dims, tmp.h.world_size); tmp.dp_proto.world = tmp.h.world_size; tmp.dp_proto.rank = dims->rank;
        if (build_programs(&tmp, dims->max_batch) != 0) return RLREP_ERR_ARG;
        if (tmp.ws.used < plain) tmp.ws.used = plain;
    }
    memset(info, 0, sizeof(*info));
    info->param_floats = tmp.L.cur[RLREP_ARENA_PARAM];
    info->target_floats = tmp.L.cur[RLREP_ARENA_TARGET] > 0 ? tmp.L.cur[RLREP_ARENA_TARGET] : 4;
    info->grad_floats = info->param_floats + RLREP_GRAD_TAIL;
    info->workspace_bytes = (int64_t)tmp.ws.used + 4096;
    info->exchange_floats = exchange_floats(*dims, tmp.h.world_size);
    for (int g = 0; g < 4; ++g) { info->group_offset[g] = tmp.L.group_off[g]; info->group_floats[g] = tmp.L.group_n[g]; }
    info->n_tensors = (int32_t)tmp.L.t.size();
    info->n_metrics = M_COUNT;
    if (descs) {
        for (int i = 0; i < (int)tmp.L.t.size() && i < cap; ++i) {
            const LT& e = tmp.L.t[i];
            memset(&descs[i], 0, sizeof(descs[i]));
            snprintf(descs[i].name, sizeof(descs[i].name), "%s", e.name.c_str());
            descs[i].arena = e.arena; descs[i].group = e.group; descs[i].offset = e.off; descs[i].rows = e.rows; descs[i].cols = e.cols;
        }
    }
    return 0;
}

int32_t rlrep_metric_names(int32_t alg, char (*names)[32], int32_t cap) {
    const char* n[M_COUNT];
    for (int i = 0; i < M_COUNT; ++i) n[i] = "";
    n[M_ACTOR_LOSS] = "actor_loss"; n[M_ALPHA_LOSS] = "alpha_loss"; n[M_ALPHA] = "alpha"; n[M_Q1] = "q1"; n[M_Q2] = "q2";
    switch (alg) {
    case RLREP_ALG_SAC: n[M_Q1_LOSS] = "q_loss"; break;
    case RLREP_ALG_VLSAC:
        n[M_FEAT_TOTAL] = "vae_loss"; n[M_FEAT_A] = "ml_loss"; n[M_KL] = "kl_loss"; n[M_S_LOSS] = "s_loss"; n[M_R_LOSS] = "r_loss";
        n[M_Q1_LOSS] = "q1_loss"; n[M_Q2_LOSS] = "q2_loss"; break;
    case RLREP_ALG_CTRLSAC: case RLREP_ALG_SPEDERSAC:
        n[M_FEAT_TOTAL] = "total_loss"; n[M_FEAT_A] = "model_loss"; n[M_R_LOSS] = "r_loss"; n[M_Q1_LOSS] = "q1_loss"; n[M_Q2_LOSS] = "q2_loss"; break;
    case RLREP_ALG_DIFFSRSAC:
        n[M_FEAT_TOTAL] = "score_loss"; n[M_Q1_LOSS] = "q_loss_reg"; n[M_Q2_LOSS] = "q_loss_noreg"; break;
    default: return RLREP_ERR_ARG;
    }
    for (int i = 0; i < M_COUNT && i < cap; ++i) snprintf(names[i], 32, "%s", n[i]);
    return M_COUNT;
}

int32_t rlrep_agent_create(const rlrep_dims* dims, const rlrep_hyper* hyper, const rlrep_arenas* arenas, void* stream, rlrep_agent** out) {
    if (!check_dims(dims) || !hyper || !arenas || !out) return RLREP_ERR_ARG;
    if (!arenas->param_dev || !arenas->grad_dev || !arenas->exp_avg_dev || !arenas->exp_avg_sq_dev || !arenas->workspace_dev ||
        !arenas->alpha_state_dev || !arenas->target_dev) { rl_set_error("null arena pointer"); return RLREP_ERR_ARG; }
    rlrep_layout_info info;
    if (rlrep_layout(dims, &info, nullptr, 0) != 0) return RLREP_ERR_ARG;
    rl_switches_read();            // RLREP_DISABLE / RLREP_ENABLE are read here, once per agent, never per launch
    std::unique_ptr<rlrep_agent> ag(new rlrep_agent());
    ag->d = *dims; ag->h = *hyper; ag->a = *arenas;
    if (ag->h.world_size <= 0) ag->h.world_size = 1;
    if (!build_layout(*dims, ag->L)) return RLREP_ERR_ARG;
    ag->L.align(RLREP_ARENA_PARAM); ag->L.align(RLREP_ARENA_TARGET);
    ag->ws.base = (char*)arenas->workspace_dev; ag->ws.cap = (size_t)info.workspace_bytes; ag->ws.dry = false;
    static_state(ag.get());
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(ag->steps, 0, sizeof(int) * 8 + 0, st);
    {
        GroupCfg gc[4]; memset(gc, 0, sizeof(gc));
        for (int g = 0; g < 4; ++g) {
            gc[g].lr = g == 1 ? hyper->lr_critic : g == 2 ? hyper->lr_actor : hyper->lr_feature;
            gc[g].b1 = hyper->beta1; gc[g].b2 = hyper->beta2; gc[g].eps = hyper->adam_eps;
            gc[g].tau = (g == 0) ? hyper->feature_tau : (g == 1) ? hyper->tau : 0.f;    // Polyak rate of the group's target copy
        }
        if (e == hipSuccess) e = hipMemcpy(ag->adam_step, gc, sizeof(gc), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMemsetAsync(ag->metrics, 0, sizeof(float) * M_COUNT, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { rl_set_error("create: %s", hipGetErrorString(e)); return RLREP_ERR_HIP; }
    if (rl_nc_init() != 0) { rl_set_error("create: cannot reserve LDS for the noise-critic kernels"); return RLREP_ERR_HIP; }
    if (rl_rowprog_init() != 0) { rl_set_error("create: cannot reserve LDS for the row-program kernel"); return RLREP_ERR_HIP; }
    if (rl_replearn_init() != 0) { rl_set_error("create: cannot reserve LDS for the score-matching kernel"); return RLREP_ERR_HIP; }
    int rc = build_programs(ag.get(), dims->max_batch);
    if (rc != 0) return rc;
    *out = ag.release();
    return 0;
}

void rlrep_agent_destroy(rlrep_agent* agent) { delete agent; }

static int ensure_batch(rlrep_agent* ag, int B) {
    if (B <= 0 || B > ag->d.max_batch) { rl_set_error("batch %d outside (0, max_batch=%d]", B, ag->d.max_batch); return RLREP_ERR_ARG; }
    if (B != ag->B) {
        // table re-upload uses blocking copies: quiesce the device first (rare path: batch size changed)
        (void)hipDeviceSynchronize();
        return build_programs(ag, B);
    }
    return 0;
}

int32_t rlrep_set_batch(rlrep_agent* ag, int32_t slot, const rlrep_batch* bt, void* stream) {
    if (!ag || !bt || slot < 0 || slot > 1 || (slot == 1 && ag->d.alg != RLREP_ALG_SPEDERSAC)) { rl_set_error("set_batch: bad argument"); return RLREP_ERR_ARG; }
    if (slot == 0) { ag->pf_done = false; ag->pf_armed = false; } else { ag->pf2_done = false; ag->pf2_armed = false; }
    ag->pi_ready = nullptr; ag->hoist_req = nullptr;           // a new batch invalidates any prefetched policy forward
    ag->early_crit = ag->early_act = ag->early_ready_crit = ag->early_ready_act = nullptr;
    int rc = ensure_batch(ag, bt->batch);
    if (rc) return rc;
    Slot& s = ag->slot[slot];
    SlotFill p; memset(&p, 0, sizeof(p));
    p.s = bt->state_dev; p.a = bt->action_dev; p.r = bt->reward_dev; p.s2 = bt->next_state_dev; p.d = bt->done_dev;
    p.B = ag->B; p.S = ag->d.state_dim; p.A = ag->d.action_dim;
    p.XE = s.XE; p.XF = s.XF; p.XF2 = s.XF2; p.XFpi = s.XFpi; p.R = s.R; p.D = s.D;
    s.filled = true;
    rc = (++g_rl_launches, rl_launch_fill_slot(&p, (hipStream_t)stream));
    if (rc) { rl_set_error("fill_slot: hip error %d", rc); return RLREP_ERR_HIP; }
    return 0;
}

int32_t rlrep_replay_row_floats(const rlrep_dims* d) { return d ? 2 * d->state_dim + d->action_dim + 2 : RLREP_ERR_ARG; }

int32_t rlrep_replay_add(float* ring_dev, int64_t capacity, int32_t row_floats, int64_t ptr, const float* rows_host, int64_t nrows, void* stream) {
    if (!ring_dev || !rows_host || capacity <= 0 || row_floats <= 0 || ptr < 0 || ptr >= capacity || nrows < 0 || nrows > capacity) {
        rl_set_error("replay_add: bad argument"); return RLREP_ERR_ARG;
    }
    const int64_t first = nrows < capacity - ptr ? nrows : capacity - ptr;
    hipError_t e = hipSuccess;
    if (first > 0) e = hipMemcpyAsync(ring_dev + ptr * row_floats, rows_host, sizeof(float) * first * row_floats, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess && first < nrows)        // across the wrap-around
        e = hipMemcpyAsync(ring_dev, rows_host + first * row_floats, sizeof(float) * (nrows - first) * row_floats, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e != hipSuccess) { rl_set_error("replay_add: %s", hipGetErrorString(e)); return RLREP_ERR_HIP; }
    return 0;
}

extern "C" int rl_launch_replay_add(float* ring, long long capacity, int row, long long ptr, const float* rows, long long nrows, int* size_dev, int new_size, hipStream_t st);
// ... the same in ONE launch that also writes the ring's new fill level into `size_dev` (the scalar the device index generator reads): the rows are
// read in place from PINNED (mapped) host memory -- RLREP_ERR_ARG if `rows_host` is not.
int32_t rlrep_replay_add_sized(float* ring_dev, int64_t capacity, int32_t row_floats, int64_t ptr, const float* rows_host, int64_t nrows,
                               int32_t* size_dev, int32_t new_size, void* stream) {
    if (!ring_dev || !rows_host || capacity <= 0 || row_floats <= 0 || ptr < 0 || ptr >= capacity || nrows < 0 || nrows > capacity || new_size < 0 || new_size > capacity) {
        rl_set_error("replay_add_sized: bad argument"); return RLREP_ERR_ARG;
    }
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, const_cast<float*>(rows_host), 0) != hipSuccess || !d) { rl_set_error("replay_add_sized: the staging rows are not mapped (pinned) host memory"); return RLREP_ERR_ARG; }
    ++g_rl_launches;
    const int rc = rl_launch_replay_add(ring_dev, capacity, row_floats, ptr, (const float*)d, nrows, size_dev, new_size, (hipStream_t)stream);
    if (rc) { rl_set_error("replay_add_sized: hip error %d", rc); return RLREP_ERR_HIP; }
    return 0;
}

static void slot_fill_params(rlrep_agent* ag, int slot, const float* ring_dev, const int32_t* idx_dev, SlotFill& p) {
    Slot& s = ag->slot[slot];
    memset(&p, 0, sizeof(p));
    p.ring = ring_dev; p.idx = idx_dev; p.B = ag->B; p.S = ag->d.state_dim; p.A = ag->d.action_dim;
    p.XE = s.XE; p.XF = s.XF; p.XF2 = s.XF2; p.XFpi = s.XFpi; p.R = s.R; p.D = s.D;
}

int32_t rlrep_prefetch_batch(rlrep_agent* ag, const float* ring_dev, const int32_t* idx_dev, int32_t batch) {
    if (!ag || !ring_dev || !idx_dev) { rl_set_error("prefetch_batch: bad argument"); return RLREP_ERR_ARG; }
    ag->pf_armed = false;
    if (batch != ag->B || rl_off("prefetch_batch")) return 0;      // would need a rebuild: let replay_sample do it
    slot_fill_params(ag, 0, ring_dev, idx_dev, ag->pf_fill);
    ag->pf_ring = ring_dev; ag->pf_idx = idx_dev; ag->pf_armed = true;
    return 1;
}

int32_t rlrep_prefetch_batch_slot(rlrep_agent* ag, int32_t slot, const float* ring_dev, const int32_t* idx_dev, int32_t batch) {
    if (slot == 0) return rlrep_prefetch_batch(ag, ring_dev, idx_dev, batch);
    if (!ag || !ring_dev || !idx_dev || slot != 1 || ag->d.alg != RLREP_ALG_SPEDERSAC) { rl_set_error("prefetch_batch_slot: bad argument"); return RLREP_ERR_ARG; }
    ag->pf2_armed = false;
    if (batch != ag->B || rl_off("prefetch_batch")) return 0;
    slot_fill_params(ag, 1, ring_dev, idx_dev, ag->pf2_fill);
    ag->pf2_ring = ring_dev; ag->pf2_idx = idx_dev; ag->pf2_armed = true;
    return 1;
}

int32_t rlrep_train_prologue(rlrep_agent* ag, const float* ring_dev, const int32_t* size_dev, int32_t* idx_pool_dev, int64_t n_idx,
                             float* eps_pool_dev, int64_t n_eps, uint64_t seed, uint64_t idx_offset, uint64_t eps_offset,
                             int32_t batch, void* stream) {
    if (!ag || !ring_dev || !size_dev || !idx_pool_dev || !eps_pool_dev || n_idx < batch || n_eps <= 0 || batch <= 0) {
        rl_set_error("train_prologue: bad argument"); return RLREP_ERR_ARG;
    }
    ag->pi_ready = nullptr; ag->hoist_req = nullptr; ag->pf_armed = false; ag->pf_done = false; ag->pf2_armed = false; ag->pf2_done = false;
    ag->chain_next = ag->chain_bwd_done = ag->l1_done = false;
    ag->early_crit = ag->early_act = ag->early_ready_crit = ag->early_ready_act = nullptr;
    int rc = ensure_batch(ag, batch);
    if (rc) return rc;
    if (ag->mirror_pending) {      // the previous prologue was followed by no optimizer launch that refreshes the counter's mirror: catch up (rare path)
        rc = (++g_rl_launches, rl_launch_counter_sync(ag->steps, 2, (hipStream_t)stream));
        if (rc) { rl_set_error("train_prologue: hip error %d", rc); return RLREP_ERR_HIP; }
    }
    TrainPrologue tp; memset(&tp, 0, sizeof(tp));
    tp.idx.dst_i = idx_pool_dev; tp.idx.n = n_idx; tp.idx.kind = 1; tp.idx.hi = 1; tp.idx.hi_dev = size_dev;
    tp.idx.seed = seed; tp.idx.offset = idx_offset; tp.idx.step_dev = ag->steps + 2; tp.idx.step_add = 1;
    tp.eps.dst_f = eps_pool_dev; tp.eps.n = n_eps; tp.eps.kind = 0; tp.eps.std = 1.0f;
    tp.eps.seed = seed; tp.eps.offset = eps_offset; tp.eps.step_dev = ag->steps + 2; tp.eps.step_add = 1;
    slot_fill_params(ag, 0, ring_dev, nullptr, tp.fill);
    tp.counter = ag->steps; tp.ticket = nullptr;          // (word 2 of the counter block is what the prologue's blocks read: train_prologue_kernel)
    tp.sh = ag->sh_dev[0]; tp.nsh = ag->nsh[0]; tp.nb_tr = ag->sh_tiles[0]; tp.sh_base = ag->a.param_dev + ag->L.group_off[0];
    rc = (++g_rl_launches, rl_launch_train_prologue(&tp, (hipStream_t)stream));
    if (rc) { rl_set_error("train_prologue: hip error %d", rc); return RLREP_ERR_HIP; }
    ag->slot[0].filled = true;
    ag->pf_done = true; ag->pf_ring = ring_dev; ag->pf_idx = idx_pool_dev;     // the first `batch` pool entries are in slot 0
    ag->in_train = true; ag->target_done = false;
    ag->mirror_pending = true;          // cleared by the optimizer launch that refreshes the mirror (engine_internal.h Builder::adam)
    return 0;
}

int32_t rlrep_replay_sample(rlrep_agent* ag, int32_t slot, const float* ring_dev, const int32_t* idx_dev, int32_t batch, void* stream) {
    if (!ag || !ring_dev || !idx_dev || slot < 0 || slot > 1 || (slot == 1 && ag->d.alg != RLREP_ALG_SPEDERSAC)) { rl_set_error("replay_sample: bad argument"); return RLREP_ERR_ARG; }
    if (slot == 0 && ag->pf_done && ring_dev == ag->pf_ring && idx_dev == ag->pf_idx && batch == ag->B && ag->slot[0].filled) {
        ag->pf_done = false;                                   // this very gather already ran (train prologue / optimizer launch)
        return 0;
    }
    if (slot == 1 && ag->pf2_done && ring_dev == ag->pf2_ring && idx_dev == ag->pf2_idx && batch == ag->B && ag->slot[1].filled) {
        ag->pf2_done = false;                                  // gathered by the previous optimizer launch (rlrep_prefetch_batch_slot)
        return 0;
    }
    if (slot == 0) { ag->pf_done = false; ag->l1_done = false; ag->chain_next = false; } else ag->pf2_done = false;
    ag->pi_ready = nullptr; ag->hoist_req = nullptr;           // a new batch invalidates any prefetched policy forward
    ag->early_crit = ag->early_act = ag->early_ready_crit = ag->early_ready_act = nullptr;
    int rc = ensure_batch(ag, batch);
    if (rc) return rc;
    Slot& s = ag->slot[slot];
    SlotFill p; memset(&p, 0, sizeof(p));
    p.ring = ring_dev; p.idx = idx_dev; p.B = ag->B; p.S = ag->d.state_dim; p.A = ag->d.action_dim;
    p.XE = s.XE; p.XF = s.XF; p.XF2 = s.XF2; p.XFpi = s.XFpi; p.R = s.R; p.D = s.D;
    s.filled = true;
    rc = (++g_rl_launches, rl_launch_fill_slot(&p, (hipStream_t)stream));
    if (rc) { rl_set_error("fill_slot: hip error %d", rc); return RLREP_ERR_HIP; }
    return 0;
}

static int philox(float* df, int32_t* di, int64_t n, int kind, float std, int hi, const int* hi_dev, uint64_t seed, uint64_t off,
                  const int* step_dev, void* stream) {
    if (n <= 0) return 0;
    PhiloxFill p; memset(&p, 0, sizeof(p));
    p.dst_f = df; p.dst_i = di; p.n = n; p.kind = kind; p.std = std; p.hi = hi; p.hi_dev = hi_dev; p.seed = seed; p.offset = off;
    p.step_dev = step_dev; p.stream_id = 0;
    int rc = (++g_rl_launches, rl_launch_philox(&p, (hipStream_t)stream));
    if (rc) { rl_set_error("philox: hip error %d", rc); return RLREP_ERR_HIP; }
    return 0;
}
int32_t rlrep_fill_indices(int32_t* dst, int64_t n, int32_t hi, uint64_t seed, uint64_t offset, void* stream) {
    if (!dst || hi <= 0) { rl_set_error("fill_indices: bad argument"); return RLREP_ERR_ARG; }
    return philox(nullptr, dst, n, 1, 0.f, hi, nullptr, seed, offset, nullptr, stream);
}
int32_t rlrep_fill_normal(float* dst, int64_t n, float std, uint64_t seed, uint64_t offset, void* stream) {
    if (!dst) { rl_set_error("fill_normal: bad argument"); return RLREP_ERR_ARG; }
    return philox(dst, nullptr, n, 0, std, 0, nullptr, seed, offset, nullptr, stream);
}
int32_t rlrep_fill_indices_dev(int32_t* dst, int64_t n, const int32_t* hi_dev, uint64_t seed, uint64_t offset, const int32_t* counter_dev, void* stream) {
    if (!dst || !hi_dev) { rl_set_error("fill_indices_dev: bad argument"); return RLREP_ERR_ARG; }
    return philox(nullptr, dst, n, 1, 0.f, 1, hi_dev, seed, offset, counter_dev, stream);
}
int32_t rlrep_fill_normal_dev(float* dst, int64_t n, float std, uint64_t seed, uint64_t offset, const int32_t* counter_dev, void* stream) {
    if (!dst) { rl_set_error("fill_normal_dev: bad argument"); return RLREP_ERR_ARG; }
    return philox(dst, nullptr, n, 0, std, 0, nullptr, seed, offset, counter_dev, stream);
}
int32_t rlrep_philox_raw(const uint32_t* ctr_key_dev, uint32_t* out_dev, int64_t n, void* stream) {
    if (!ctr_key_dev || !out_dev || n <= 0 || n > (1ll << 30)) { rl_set_error("philox_raw: bad argument"); return RLREP_ERR_ARG; }
    const int rc = rl_launch_philox_raw(ctr_key_dev, out_dev, n, (hipStream_t)stream);
    if (rc) { rl_set_error("philox_raw: hip error %d", rc); return RLREP_ERR_HIP; }
    return 0;
}
const int32_t* rlrep_steps_dev(rlrep_agent* ag) { return ag ? ag->steps : nullptr; }
static_assert(sizeof(GroupCfg) == 4 * RLREP_GROUP_CFG_WORDS, "include/rlrep.h documents the GroupCfg layout");
const void* rlrep_group_cfg_dev(rlrep_agent* ag) { return ag ? ag->adam_step : nullptr; }

static int run(rlrep_agent* ag, const Program& p, void* stream) {
    if (!ag->slot[0].filled) { rl_set_error("step before set_batch / replay_sample"); return RLREP_ERR_STATE; }
    ag->last_launches += (int)p.stages.size();
    return p.run((hipStream_t)stream);
}
#define STEP_PROLOGUE(needs_feature) \
    if (!ag) { rl_set_error("null agent"); return RLREP_ERR_ARG; } \
    if ((needs_feature) && ag->d.alg == RLREP_ALG_SAC) { rl_set_error("sac has no feature step"); return RLREP_ERR_ARG; }

int32_t rlrep_feature_backward(rlrep_agent* ag, const float* eps, const int32_t* idx, void* stream) {
    STEP_PROLOGUE(true)
    if (ag->d.alg == RLREP_ALG_VLSAC && !eps) { rl_set_error("vlsac feature step needs eps[B,F]"); return RLREP_ERR_ARG; }
    if (ag->d.alg == RLREP_ALG_DIFFSRSAC && (!eps || !idx)) { rl_set_error("diffsrsac feature step needs noise_idx[B] and eps[B,S]"); return RLREP_ERR_ARG; }
    if (ag->d.alg == RLREP_ALG_SPEDERSAC && !ag->slot[1].filled) { rl_set_error("spedersac feature step needs batch slot 1"); return RLREP_ERR_STATE; }
    ag->cur_eps = eps; ag->cur_idx = idx; ag->last_launches = 0;
    if (!ag->in_train && ag->has_shadows()) { const int rs = refresh_shadows(ag, stream); if (rs) return rs; }     // parameters may have been written by the caller
    if (!ag->in_train && rl_rowprog_enabled() && ag->rp_epoch) (void)(++g_rl_launches, rl_launch_counter_inc(ag->rp_epoch, 0, (hipStream_t)stream));   // a fresh epoch whatever ran before
    ag->pi_ready = nullptr;                                       // f_target is about to change
    ag->early_ready_crit = ag->early_ready_act = nullptr;
    // chained feature steps: the previous step's optimizer launch already ran this step's first stage (encoder.l1 / f.l1)
    const size_t first = ag->l1_done ? 1 : 0;
    const bool chain = ag->chain_next;
    ag->l1_done = false; ag->chain_next = false; ag->chain_bwd_done = false;
    if (ag->early_crit && !ag->feat_bwd_h.stages.empty()) {
        if (first || chain) { rl_set_error("feature step: rlrep_feature_chain_next cannot be combined with rlrep_prefetch_policy_early"); return RLREP_ERR_STATE; }
        ag->cur_eps3 = ag->early_crit; ag->cur_eps2 = ag->early_act;
        ag->early_crit = ag->early_act = nullptr;
        const int rc = run(ag, ag->feat_bwd_h, stream);
        if (rc == 0) { ag->early_ready_crit = ag->cur_eps3; ag->early_ready_act = ag->cur_eps2; }
        return rc;
    }
    ag->early_crit = ag->early_act = nullptr;
    if (!ag->slot[0].filled) { rl_set_error("step before set_batch / replay_sample"); return RLREP_ERR_STATE; }
    const Program& p = chain ? ag->feat_bwd_m : ag->feat_bwd;
    ag->last_launches += (int)(p.stages.size() - first);
    const int rc = p.run((hipStream_t)stream, first);
    if (rc == 0 && chain) ag->chain_bwd_done = true;
    return rc;
}
int32_t rlrep_feature_apply(rlrep_agent* ag, void* stream) {
    STEP_PROLOGUE(true)
    if (ag->chain_bwd_done) {          // the first layers' optimizer already ran (weight-gradient epilogues): the launch that skips them and runs the next step's head
        ag->chain_bwd_done = false;
        const int rc = run(ag, ag->feat_apply_m, stream);
        if (rc == 0) ag->l1_done = true;
        return rc;
    }
    return run(ag, ag->feat_apply, stream);
}
// The NEXT feature step will follow this one directly, on the minibatch armed by rlrep_prefetch_batch: chain them (rlrep_agent::feat_bwd_m).
// Call between rlrep_prefetch_batch and this step's rlrep_feature_backward / rlrep_feature_step.  1: armed; 0: not available for this agent
// (the step then runs as usual).  The caller must indeed run that next step next, with plain rlrep_feature_backward (no early policy).
int32_t rlrep_feature_chain_next(rlrep_agent* ag) {
    if (!ag) { rl_set_error("null agent"); return RLREP_ERR_ARG; }
    ag->chain_next = false;
    if (ag->feat_bwd_m.stages.empty() || ag->feat_apply_m.stages.empty() || !ag->pf_armed || ag->snap_armed) return 0;
    ag->chain_next = true;
    return 1;
}
int32_t rlrep_prefetch_policy_early(rlrep_agent* ag, const float* eps_critic, const float* eps_actor) {
    if (!ag) { rl_set_error("null agent"); return RLREP_ERR_ARG; }
    ag->early_crit = ag->early_act = nullptr;
    if (!eps_critic || !eps_actor || ag->feat_bwd_h.stages.empty() || ag->critic_bwd_h2.stages.empty()) return 0;
    ag->early_crit = eps_critic; ag->early_act = eps_actor;
    return 1;
}
int32_t rlrep_prefetch_policy(rlrep_agent* ag, const float* eps_actor) {
    if (!ag) { rl_set_error("null agent"); return RLREP_ERR_ARG; }
    ag->hoist_req = nullptr;
    if (!eps_actor || ag->critic_bwd_h.stages.empty()) return 0;      // not supported for this agent / shape: nothing armed
    ag->hoist_req = eps_actor;
    return 1;
}
int32_t rlrep_critic_backward(rlrep_agent* ag, const float* eps, void* stream) {
    STEP_PROLOGUE(false)
    if (!eps) { rl_set_error("critic step needs eps[B,A]"); return RLREP_ERR_ARG; }
    ag->cur_eps = eps; ag->last_launches = 0;
    ag->pi_ready = nullptr;
    // the images of critic.l1 / l4 and of their targets, from the tensors as they are now: the target copies have no other writer, and a
    // caller may have written any of them since the last step (the launch rides on the chain that has slack in the pipelined train())
    if (!ag->images_managed) { const int rs = refresh_x3(ag, stream); if (rs) return rs; }
    if (ag->early_ready_crit && ag->early_ready_crit == eps) {      // both policy forwards already ran (last feature step)
        const float* act_eps = ag->early_ready_act;
        ag->early_ready_crit = ag->early_ready_act = nullptr; ag->hoist_req = nullptr;
        const int rc = run(ag, ag->critic_bwd_h2, stream);
        if (rc == 0) ag->pi_ready = act_eps;
        return rc;
    }
    ag->early_ready_crit = ag->early_ready_act = nullptr;
    if (ag->hoist_req) {
        ag->cur_eps2 = ag->hoist_req; ag->hoist_req = nullptr;
        const int rc = run(ag, ag->critic_bwd_h, stream);
        if (rc == 0) ag->pi_ready = ag->cur_eps2;
        return rc;
    }
    return run(ag, ag->critic_bwd, stream);
}
int32_t rlrep_critic_apply(rlrep_agent* ag, void* stream) {
    STEP_PROLOGUE(false)
    if (ag->in_train && !ag->target_done && !ag->critic_apply_f.stages.empty()) {
        ag->target_done = true;
        return run(ag, ag->critic_apply_f, stream);
    }
    return run(ag, ag->critic_apply, stream);
}
int32_t rlrep_actor_backward(rlrep_agent* ag, const float* eps, void* stream) {
    STEP_PROLOGUE(false)
    if (!eps) { rl_set_error("actor step needs eps[B,A]"); return RLREP_ERR_ARG; }
    ag->cur_eps = eps; ag->last_launches = 0;
    // resume after the forward half iff the critic step of this train() ran it with exactly this noise
    const size_t first = (ag->pi_ready && ag->pi_ready == eps) ? (size_t)ag->actor_resume : 0;
    ag->pi_ready = nullptr;
    if (!ag->slot[0].filled) { rl_set_error("step before set_batch / replay_sample"); return RLREP_ERR_STATE; }
    if (!ag->in_train && !ag->images_managed) { const int rs = refresh_x3(ag, stream); if (rs) return rs; }        // (inside a train() the critic's Adam launch kept the live images current)
    ag->last_launches += (int)(ag->actor_bwd.stages.size() - first);
    return ag->actor_bwd.run((hipStream_t)stream, first);
}
int32_t rlrep_actor_apply(rlrep_agent* ag, void* stream) { STEP_PROLOGUE(false) return run(ag, ag->actor_apply, stream); }

int32_t rlrep_feature_step(rlrep_agent* ag, const float* eps, const int32_t* idx, void* stream) {
    int rc = rlrep_feature_backward(ag, eps, idx, stream);
    return rc ? rc : rlrep_feature_apply(ag, stream);
}
int32_t rlrep_critic_step(rlrep_agent* ag, const float* eps, void* stream) {
    int rc = rlrep_critic_backward(ag, eps, stream);
    return rc ? rc : rlrep_critic_apply(ag, stream);
}
int32_t rlrep_actor_alpha_step(rlrep_agent* ag, const float* eps, void* stream) {
    int rc = rlrep_actor_backward(ag, eps, stream);
    return rc ? rc : rlrep_actor_apply(ag, stream);
}
int32_t rlrep_update_target(rlrep_agent* ag, void* stream) {
    if (!ag) return RLREP_ERR_ARG;
    const bool done = ag->target_done;
    ag->in_train = ag->target_done = false;
    if (done) return 0;                                  // already folded into this train()'s critic Adam launch
    ag->last_launches += (int)ag->upd_target.stages.size();
    return ag->upd_target.run((hipStream_t)stream);
}
int32_t rlrep_begin_train(rlrep_agent* ag, void* stream) {
    if (!ag) return RLREP_ERR_ARG;
    int rc = (++g_rl_launches, rl_launch_counter_inc(ag->steps, 2, (hipStream_t)stream));
    if (rc) { rl_set_error("begin_train: hip error %d", rc); return RLREP_ERR_HIP; }
    if (ag->has_shadows() && (rc = refresh_shadows(ag, stream)) != 0) return rc;
    ag->in_train = true; ag->target_done = false;
    return 0;
}
int32_t rlrep_defer_supported(rlrep_agent* ag) {
    if (!ag) return 0;
    int n = 0;
    for (int k = 0; k < rlrep_agent::NSETS; ++k) if (!ag->dset[k].critic_bwd.stages.empty() && !ag->dset[k].actor_bwd.stages.empty()) ++n;
    return n == rlrep_agent::NSETS ? n : 0;
}
// Arm the folded snapshot: the NEXT feature optimizer launch (rlrep_feature_apply) also writes snapshot set `set` -- the minibatch slices
// and the two noise blocks by extra blocks, the f_target (or live f) block by the lanes that produce its new values -- and the
// rlrep_defer_snapshot that follows with the same arguments launches nothing.  To be called before the LAST feature step of a train().
// Returns 1 if armed, 0 if this agent / configuration has no folded form (the caller proceeds as before).
int32_t rlrep_defer_arm(rlrep_agent* ag, int32_t set, const float* eps_critic, const float* eps_actor) {
    if (!ag || set < 0 || set >= rlrep_agent::NSETS) return 0;
    ag->snap_armed = false; ag->snap_done = -1;
    if (!eps_critic || !eps_actor || !rlrep_defer_supported(ag) || ag->dset[set].block_which < 0 || !ag->dset[set].block) return 0;
    if (!ag->sync_prog.stages.empty()) return 0;
    ag->snap_armed = true; ag->snap_set = set; ag->snap_ec = eps_critic; ag->snap_ea = eps_actor;
    return 1;
}
int32_t rlrep_defer_snapshot(rlrep_agent* ag, int32_t set, const float* eps_critic, const float* eps_actor, void* stream) {
    if (!ag || !eps_critic || !eps_actor || set < 0 || set >= rlrep_agent::NSETS) { rl_set_error("defer_snapshot: bad argument"); return RLREP_ERR_ARG; }
    if (!rlrep_defer_supported(ag)) { rl_set_error("deferred critic/actor steps are not built for this agent"); return RLREP_ERR_STATE; }
    if (!ag->slot[0].filled) { rl_set_error("defer_snapshot before set_batch / replay_sample"); return RLREP_ERR_STATE; }
    if (!ag->sync_prog.stages.empty()) {          // ctrlsac: frozen_phi, frozen_phi_target <- phi (ctrlsac_agent.py:344-346) belongs to the end of the feature steps
        const int rs = ag->sync_prog.run((hipStream_t)stream);
        if (rs) return rs;
    }
    if (ag->snap_done == set && ag->snap_ec == eps_critic && ag->snap_ea == eps_actor) {      // the last feature optimizer launch already wrote this set
        ag->snap_done = -1; ag->snap_armed = false;
        ag->dset[set].valid = true;
        return 0;
    }
    ag->snap_done = -1; ag->snap_armed = false;
    CopySegs cs = ag->dset[set].segs;
    cs.src[cs.n - 2] = eps_critic; cs.src[cs.n - 1] = eps_actor;
    const int rc = (++g_rl_launches, rl_launch_copy_segs(&cs, (hipStream_t)stream));
    if (rc) { rl_set_error("defer_snapshot: hip error %d", rc); return RLREP_ERR_HIP; }
    ag->dset[set].valid = true;
    return 0;
}
// part: 0 critic backward, 1 critic apply (+ period-gated critic-target Polyak), 2 actor backward, 3 actor + temperature apply; -1 all.
// Data parallel callers all-reduce the critic / actor gradient slices between 0 and 1 and between 2 and 3.
int32_t rlrep_deferred_part(rlrep_agent* ag, int32_t set, int32_t part, void* stream) {
    if (!ag || set < 0 || set >= rlrep_agent::NSETS || part < -1 || part > 3 || !rlrep_defer_supported(ag)) { rl_set_error("deferred critic/actor steps are not built for this agent"); return RLREP_ERR_STATE; }
    rlrep_agent::DeferSet& D = ag->dset[set];
    if (!D.valid) { rl_set_error("deferred critic/actor steps before rlrep_defer_snapshot of this set"); return RLREP_ERR_STATE; }
    const float* e_crit = D.eps; const float* e_act = D.eps + (size_t)ag->B * ag->d.action_dim;
    const float* keep1 = ag->cur_eps; const float* keep2 = ag->cur_eps2;
    int rc = 0;
    if (part == -1 || part == 0) {
        ag->cur_eps = e_crit; ag->cur_eps2 = e_act; ag->last_launches = 0;
        if (!ag->images_managed) rc = refresh_x3(ag, stream);                 // as in rlrep_critic_backward
        if (!rc) rc = run(ag, D.critic_bwd, stream);
    }
    if (!rc && (part == -1 || part == 1)) rc = run(ag, D.critic_apply, stream);
    if (!rc && (part == -1 || part == 2)) {
        ag->cur_eps = e_act;
        ag->last_launches += (int)(D.actor_bwd.stages.size() - D.actor_resume);
        rc = D.actor_bwd.run((hipStream_t)stream, (size_t)D.actor_resume);
    }
    if (!rc && (part == -1 || part == 3)) rc = run(ag, ag->actor_apply, stream);
    ag->cur_eps = keep1; ag->cur_eps2 = keep2;
    return rc;
}
int32_t rlrep_deferred_critic_actor(rlrep_agent* ag, int32_t set, void* stream) { return rlrep_deferred_part(ag, set, -1, stream); }
// Weight images of the vlsac noise critic (bf16x3).  By default every critic step regenerates them from the tensors with a launch
// of its own (a caller may have written parameters; the target copies have no other writer outside a train()).  A caller that replays
// captured train() graphs can take that launch off the chain: between rlrep_images_managed(agent, 1) and (agent, 0) the step entry points do
// NOT launch it -- inside a train() bracket the critic group's optimizer launch keeps the live AND (with the folded Polyak) the target images
// current -- and the caller runs rlrep_refresh_images itself whenever anything else may have written critic / critic_target (an eager step
// method, rlrep_update_target outside a bracket, a torch write into the arenas).  Returns 1 if the agent keeps such images (else 0: nothing to manage).
int32_t rlrep_images_managed(rlrep_agent* ag, int32_t on) {
    if (!ag) { rl_set_error("null agent"); return RLREP_ERR_ARG; }
    if (!ag->x3_n || ag->critic_apply_f.stages.empty()) { ag->images_managed = false; return 0; }
    ag->images_managed = on != 0;
    return 1;
}
int32_t rlrep_refresh_images(rlrep_agent* ag, void* stream) {
    if (!ag) { rl_set_error("null agent"); return RLREP_ERR_ARG; }
    return refresh_x3(ag, stream);
}
int32_t rlrep_end_train(rlrep_agent* ag) {
    if (!ag) return RLREP_ERR_ARG;
    ag->in_train = ag->target_done = false;
    return 0;
}
int32_t rlrep_feature_exchange_count(rlrep_agent* ag) { return ag ? (int32_t)ag->feat_cuts.size() : RLREP_ERR_ARG; }
int32_t rlrep_feature_exchange(rlrep_agent* ag, int32_t k, int32_t* kind, float** ptr, int64_t* count, int64_t* local_off) {
    if (!ag || k < 0 || k >= (int)ag->feat_cuts.size() || !kind || !ptr || !count || !local_off) { rl_set_error("feature_exchange: bad argument"); return RLREP_ERR_ARG; }
    const Exchange& e = ag->feat_cuts[k];
    *kind = e.kind; *ptr = e.ptr; *count = e.count; *local_off = e.local_off;
    return 0;
}
int32_t rlrep_feature_backward_part(rlrep_agent* ag, int32_t part, const float* eps, const int32_t* idx, void* stream) {
    STEP_PROLOGUE(true)
    const int ncut = (int)ag->feat_cuts.size();
    if (part < 0 || part > ncut) { rl_set_error("feature_backward_part: part %d of %d", part, ncut + 1); return RLREP_ERR_ARG; }
    if (!ag->slot[0].filled) { rl_set_error("step before set_batch / replay_sample"); return RLREP_ERR_STATE; }
    ag->cur_eps = eps; ag->cur_idx = idx; ag->pi_ready = nullptr;
    const int lo = part == 0 ? 0 : ag->feat_cuts[part - 1].after_stage + 1;
    const int hi = part == ncut ? (int)ag->feat_bwd.stages.size() : ag->feat_cuts[part].after_stage + 1;
    if (part == 0) ag->last_launches = 0;
    for (int i = lo; i < hi; ++i) {
        int rc = ag->feat_bwd.stages[i].run((hipStream_t)stream); ++g_rl_launches;
        if (rc) { rl_set_error("stage '%s' failed: hip error %d", ag->feat_bwd.stages[i].what, rc); return RLREP_ERR_HIP; }
    }
    ag->last_launches += hi - lo;
    return 0;
}

int32_t rlrep_sync_frozen(rlrep_agent* ag, void* stream) {
    if (!ag) return RLREP_ERR_ARG;
    ag->last_launches += (int)ag->sync_prog.stages.size();
    return ag->sync_prog.run((hipStream_t)stream);
}

int32_t rlrep_actor_forward(rlrep_agent* ag, const float* obs, int32_t n, const float* eps, float lo, float hi, float* action, void* stream) {
    if (!ag || !obs || !action || n <= 0 || n > ag->d.max_batch) { rl_set_error("actor_forward: bad argument"); return RLREP_ERR_ARG; }
    const int S = ag->d.state_dim, A = ag->d.action_dim, Ha = ag->d.actor_hidden_dim;
    hipStream_t st = (hipStream_t)stream;
    if (ag->infer_n != n) {
        (void)hipDeviceSynchronize();
        ag->infer.stages.clear();
        // inference scratch lives behind the step programs' buffers
        ag->ws.used = ag->prog_end;
        Builder b(ag);
        float* A1 = b.ws.f((size_t)n * Ha); float* A2 = b.ws.f((size_t)n * Ha); float* AO = b.ws.f((size_t)n * 2 * A);
        b.fwd_stage(ag->infer, {Builder::fwd(ag->obs_in, S, n, S, ag->P("actor.trunk.0.weight"), S, ag->P("actor.trunk.0.bias"), Ha, A1, Ha, ACT_ELU)}, "infer l1");
        b.fwd_stage(ag->infer, {Builder::fwd(A1, Ha, n, Ha, ag->P("actor.trunk.2.weight"), Ha, ag->P("actor.trunk.2.bias"), Ha, A2, Ha, ACT_ELU)}, "infer l2");
        b.fwd_stage(ag->infer, {Builder::fwd(A2, Ha, n, Ha, ag->P("actor.trunk.4.weight"), Ha, ag->P("actor.trunk.4.bias"), 2 * A, AO, 2 * A, ACT_NONE)}, "infer head");
        PolicyFwd pf; memset(&pf, 0, sizeof(pf));
        pf.O = AO; pf.B = n; pf.A = A; pf.act = ag->act_out; pf.ld_act = A; pf.logp = nullptr; pf.clamp = 1;
        rlrep_agent* a2 = ag;
        ag->infer.stages.push_back({[=](hipStream_t s2) { PolicyFwd q = pf; q.eps = a2->cur_eps; q.lo = a2->infer_lo; q.hi = a2->infer_hi; return rl_launch_policy_fwd(&q, s2); }, "infer policy"});
        if (!ag->ws.ok()) { ag->ws.used = ag->prog_end; ag->infer.stages.clear(); rl_set_error("workspace too small for inference"); return RLREP_ERR_NOMEM; }
        ag->infer_n = n;
    }
    int rc = rl_launch_copy(obs, ag->obs_in, (long long)n * S, st);
    if (rc) { rl_set_error("actor_forward copy-in: hip error %d", rc); return RLREP_ERR_HIP; }
    ag->cur_eps = eps; ag->infer_lo = lo; ag->infer_hi = hi;
    rc = ag->infer.run(st);
    if (rc) return rc;
    rc = rl_launch_copy(ag->act_out, action, (long long)n * A, st);
    if (rc) { rl_set_error("actor_forward copy-out: hip error %d", rc); return RLREP_ERR_HIP; }
    return 0;
}

extern "C" int rl_launch_select_action(const SelectAct* p, hipStream_t st);
// One observation -> one action in ONE launch.  obs / action: device pointers, or pinned (mapped) host buffers -- resolved here with
// hipHostGetDevicePointer, so that the kernel reads the observation and writes the action in place and no copy launch stands on either side.
int32_t rlrep_select_action(rlrep_agent* ag, const float* obs, int32_t obs_on_host, int32_t explore, uint64_t seed, uint64_t offset,
                            float lo, float hi, float* action, int32_t action_on_host, void* stream) {
    if (!ag || !obs || !action) { rl_set_error("select_action: bad argument"); return RLREP_ERR_ARG; }
    SelectAct p; memset(&p, 0, sizeof(p));
    void* d = nullptr;
    if (obs_on_host) { if (hipHostGetDevicePointer(&d, const_cast<float*>(obs), 0) != hipSuccess || !d) { rl_set_error("select_action: the observation buffer is not mapped (pinned) host memory"); return RLREP_ERR_ARG; } p.obs = (const float*)d; }
    else p.obs = obs;
    if (action_on_host) { if (hipHostGetDevicePointer(&d, action, 0) != hipSuccess || !d) { rl_set_error("select_action: the action buffer is not mapped (pinned) host memory"); return RLREP_ERR_ARG; } p.act = (float*)d; }
    else p.act = action;
    p.W1 = ag->P("actor.trunk.0.weight"); p.b1 = ag->P("actor.trunk.0.bias"); p.W2 = ag->P("actor.trunk.2.weight"); p.b2 = ag->P("actor.trunk.2.bias");
    p.W3 = ag->P("actor.trunk.4.weight"); p.b3 = ag->P("actor.trunk.4.bias");
    p.S = ag->d.state_dim; p.Ha = ag->d.actor_hidden_dim; p.A = ag->d.action_dim; p.explore = explore ? 1 : 0; p.lo = lo; p.hi = hi; p.seed = seed; p.offset = offset;
    ++g_rl_launches;
    const int rc = rl_launch_select_action(&p, (hipStream_t)stream);
    if (rc) { rl_set_error("select_action: launch failed (%d)", rc); return rc == -7 ? RLREP_ERR_ARG : RLREP_ERR_HIP; }
    return 0;
}

static Program* prog_of(rlrep_agent* ag, int id) {
    switch (id) {
    case 0: return &ag->feat_bwd; case 1: return &ag->feat_apply; case 2: return &ag->critic_bwd; case 3: return &ag->critic_apply;
    case 4: return &ag->actor_bwd; case 5: return &ag->actor_apply; case 6: return &ag->upd_target; case 7: return &ag->critic_bwd_h;
    case 8: return &ag->feat_bwd_h; case 9: return &ag->critic_bwd_h2;
    default: return nullptr;
    }
}
int32_t rlrep_stage_count(rlrep_agent* ag, int32_t program) {
    Program* p = ag ? prog_of(ag, program) : nullptr;
    return p ? (int32_t)p->stages.size() : RLREP_ERR_ARG;
}
const char* rlrep_stage_name(rlrep_agent* ag, int32_t program, int32_t stage) {
    Program* p = ag ? prog_of(ag, program) : nullptr;
    if (!p || stage < 0 || stage >= (int)p->stages.size()) return nullptr;
    return p->stages[stage].what;
}
int32_t rlrep_stage_info(rlrep_agent* ag, int32_t program, int32_t stage, int32_t* engine, double* flops, double* bytes) {
    Program* p = ag ? prog_of(ag, program) : nullptr;
    if (!p || stage < 0 || stage >= (int)p->stages.size()) { rl_set_error("stage_info: bad program/stage"); return RLREP_ERR_ARG; }
    const Stage& s = p->stages[stage];
    if (engine) *engine = s.engine;
    if (flops) *flops = s.flops;
    if (bytes) *bytes = s.bytes;
    return 0;
}
int32_t rlrep_run_stage(rlrep_agent* ag, int32_t program, int32_t stage, void* stream) {
    Program* p = ag ? prog_of(ag, program) : nullptr;
    if (!p || stage < 0 || stage >= (int)p->stages.size()) { rl_set_error("run_stage: bad program/stage"); return RLREP_ERR_ARG; }
    if (!ag->slot[0].filled) { rl_set_error("run_stage before a full step"); return RLREP_ERR_STATE; }
    int rc = p->stages[stage].run((hipStream_t)stream); ++g_rl_launches;
    if (rc) { rl_set_error("stage '%s' failed: hip error %d", p->stages[stage].what, rc); return RLREP_ERR_HIP; }
    return 0;
}

int32_t rlrep_gemm(int32_t engine, int32_t la, int32_t lb, const float* A, int32_t lda, const float* B, int32_t ldb,
                   float* Cm, int32_t ldc, int32_t R, int32_t Cn, int32_t K, int32_t epi, int32_t act, int32_t flags,
                   const float* bias, const float* aux, int32_t ldaux, float* out2, int32_t bt, int32_t splits,
                   float* wsp, int64_t ws_floats, void* stream) {
    rl_switches_read();
    if (!A || !B || !Cm || R <= 0 || Cn <= 0 || K <= 0 || (epi != EPI_FWD && epi != EPI_DX && epi != EPI_DW)) { rl_set_error("gemm: bad argument"); return RLREP_ERR_ARG; }
    GemmTask t; memset(&t, 0, sizeof(t));
    t.scale = 1.f; t.A = A; t.lda = lda; t.B = B; t.ldb = ldb; t.C = Cm; t.ldc = ldc; t.R = R; t.Cn = Cn; t.K = K;
    t.epi = epi; t.act = act; t.flags = flags & FLAG_ACCUM;
    if (epi == EPI_FWD) { t.bias = bias; t.out2 = out2; t.ldout2 = ldc; }
    if (epi == EPI_DX) { t.aux = aux; t.ldaux = ldaux; }
    if (epi == EPI_DW && (flags & FLAG_BIASGRAD) && out2) { t.flags |= FLAG_BIASGRAD; t.out2 = out2; }
    GemmBatch gb; memset(&gb, 0, sizeof(gb)); gb.ntasks = 1;
    int rc;
    if (engine == 0) {
        t.tiles_c = (Cn + 15) / 16; t.ntiles = ((R + 15) / 16) * t.tiles_c; t.tile_base = 0;
        gb.t[0] = t;
        rc = rl_launch_gemm16(la, lb, 1, &gb, t.ntiles, (hipStream_t)stream);
    } else {
        t.flags |= rl_gemm_lds_dim_flags(&t, la, lb) | rl_gemm_lds_ptr_flags(&t);
        // (the 64-wide bf16x3 tile has any-alignment loaders; the 128-wide one needs 16-byte-regular operands)
        if (engine == 2 && (t.flags & (FLAG_SCALAR_A | FLAG_SCALAR_B)) && (bt != 64 || R < 4 || Cn < 4 || K < 4)) { rl_set_error("gemm: shape/alignment not eligible for the bf16x3 tile"); return RLREP_ERR_ARG; }
        if (engine == 2 && bt != 64 && bt != 32 && (t.flags & FLAG_SCALAR_C)) { rl_set_error("gemm: shape/alignment not eligible for the bf16x3 tile"); return RLREP_ERR_ARG; }
        if (engine == 2 && bt == 32) {
            // the 32 x 32 tile whose four waves split K (gemm_x3q.h): row-major A, K % 16 == 0, no slabs
            if (la != LD_ROW || (t.flags & (FLAG_SCALAR_A | FLAG_SCALAR_B)) || (K & 15) || K < 32 || (lb == LD_COL && ((Cn & 7) || Cn < 8))) { rl_set_error("gemm: shape/alignment not eligible for the 32 x 32 bf16x3 tile"); return RLREP_ERR_ARG; }
            t.splits = 1; t.kchunk = K;
            t.tiles_c = (Cn + 31) / 32; t.ntiles = ((R + 31) / 32) * t.tiles_c; t.tile_base = 0;
            gb.t[0] = t;
            rc = rl_launch_gemm_lds(33, la, lb, &gb, t.ntiles, 0, (hipStream_t)stream);
            if (rc != 0) { rl_set_error("gemm: launch failed (%d)", rc); return rc < 0 ? RLREP_ERR_ARG : RLREP_ERR_HIP; }
            return 0;
        }
        int pbt = 0, psp = 1, pkc = 0;
        rl_gemm_lds_plan(&t, &pbt, &psp, &pkc);
        if (bt == 64 || bt == 128) pbt = bt;
        if (engine == 2 && bt != 64) pbt = 128;          // bf16x3: the 128-wide tile, or (bt = 64) the 64-wide one
        if (splits > 0) { psp = splits; pkc = ((K + psp - 1) / psp + 31) / 32 * 32; psp = (K + pkc - 1) / pkc; }
        t.splits = psp; t.kchunk = pkc;
        int fin = 0;
        if (psp > 1) {
            const bool bg = (t.flags & FLAG_BIASGRAD) != 0;
            if (!wsp || ws_floats < (int64_t)psp * R * (((Cn + 3) & ~3) + 1)) { rl_set_error("gemm: workspace too small for %d splits", psp); return RLREP_ERR_ARG; }
            t.slab = wsp; t.bslab = wsp + (size_t)psp * R * ((Cn + 3) & ~3); t.fin_base = 0;
            fin = (int)(((long long)R * ((Cn + 3) / 4) + 255) / 256) + (bg ? (R + 255) / 256 : 0);
            if ((flags & 4) && engine == 2 && pbt == 64) {
                // flags & 4: the 64-wide bf16x3 tile finishes its split-K products inside the launch (FLAG_FIN_INLINE: one ticket word per output
                // tile behind the slabs, zero between launches) -- no finishing launch
                const size_t ticks = (size_t)((R + 63) / 64) * ((Cn + 63) / 64);
                if (ws_floats < (int64_t)psp * R * (((Cn + 3) & ~3) + 1) + (int64_t)ticks) { rl_set_error("gemm: workspace too small for %d splits", psp); return RLREP_ERR_ARG; }
                if (hipMemsetAsync(t.bslab, 0, ticks * sizeof(int), (hipStream_t)stream) != hipSuccess) return RLREP_ERR_HIP;
                t.bslab += ticks; t.flags |= FLAG_FIN_INLINE; t.fin_base = 0x7fffffff; fin = 0;
            }
        }
        const bool wide = engine == 2 && bt == 256;          // the 256 x 128 persistent tile (gemm_x3w.h)
        if (wide && (act == ACT_SIN || act == ACT_TANH)) { rl_set_error("gemm: the 256 x 128 tile has no sin / tanh epilogue"); return RLREP_ERR_ARG; }
        const int er = wide ? 256 : pbt, ec = wide ? 128 : pbt;
        t.tiles_c = (Cn + ec - 1) / ec; t.ntiles = ((R + er - 1) / er) * t.tiles_c * psp; t.tile_base = 0;
        gb.t[0] = t;
        rc = rl_launch_gemm_lds(wide ? 257 : engine == 2 ? (pbt == 64 ? 65 : 129) : pbt, la, lb, &gb, t.ntiles, fin, (hipStream_t)stream);
    }
    if (rc != 0) { rl_set_error("gemm: launch failed (%d)", rc); return rc < 0 ? RLREP_ERR_ARG : RLREP_ERR_HIP; }
    return 0;
}

int32_t rlrep_gemm_plan(int32_t la, int32_t lb, int32_t R, int32_t Cn, int32_t K, int32_t lda, int32_t ldb, int32_t ldc,
                        int32_t* engine, int32_t* tile, int32_t* splits, int32_t* kchunk, int32_t* scalar_sides) {
    rl_switches_read();
    if (R <= 0 || Cn <= 0 || K <= 0 || !engine) { rl_set_error("gemm_plan: bad argument"); return RLREP_ERR_ARG; }
    GemmTask t; memset(&t, 0, sizeof(t));
    t.R = R; t.Cn = Cn; t.K = K; t.lda = lda; t.ldb = ldb; t.ldc = ldc; t.epi = la == LD_COL ? EPI_DW : EPI_FWD;
    int sp = 1, kc = 0, fl = 0;
    const int code = rl_gemm_lds_route(&t, la, lb, 0, &sp, &kc, &fl);
    *engine = code == 0 ? 0 : (code == 257 || code == 129 || code == 65 || code == 33) ? 2 : 1;
    if (tile) *tile = code == 0 ? 16 : code == 257 ? 256 : code == 129 ? 128 : code == 65 ? 64 : code == 33 ? 32 : code;
    if (splits) *splits = code ? sp : 1;
    if (kchunk) *kchunk = code ? kc : K;
    if (scalar_sides) *scalar_sides = code ? (((fl & FLAG_SCALAR_A) ? 1 : 0) | ((fl & FLAG_SCALAR_B) ? 2 : 0) | ((fl & FLAG_SCALAR_C) ? 4 : 0)) : 0;
    return 0;
}

int32_t rlrep_nc_fwd_plan(int32_t heads, int32_t B, int32_t F, int32_t H, int32_t* engine, int32_t* rows, int32_t* cols) {
    rl_switches_read();
    if (heads <= 0 || heads > NC_MAX_TASKS || B <= 0 || F <= 0 || H <= 0 || !engine) { rl_set_error("nc_fwd_plan: bad argument"); return RLREP_ERR_ARG; }
    NcFwdTask t[NC_MAX_TASKS]; memset(t, 0, sizeof(t));
    for (int q = 0; q < heads; ++q) { t[q].B = B; t[q].F = F; t[q].H = H; t[q].N = 20; t[q].ld_ml = 2 * F; }
    int e = 0, g2 = 1, c = 128;
    rl_nc_fwd_plan(t, heads, &e, &g2, &c);
    *engine = e;
    if (rows) *rows = 4 * g2;
    if (cols) *cols = c;
    return 0;
}

int32_t rlrep_chain_status(rlrep_agent* ag, uint32_t* status, void* stream) {
    if (!ag || !ag->xc_err) { rl_set_error("chain_status: bad argument"); return RLREP_ERR_ARG; }
    unsigned w = 0;
    hipError_t e = hipMemcpyAsync(&w, ag->xc_err, sizeof(w), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) { rl_set_error("chain_status: %s", hipGetErrorString(e)); return RLREP_ERR_HIP; }
    if (status) *status = w;
    if (w) {
        rl_set_error("a persistent chain launch failed its run-time checks (word %u:%s%s); results are invalid -- leaving xchain out of RLREP_ENABLE runs one launch per stage",
                     w, (w & 5u) ? " wait timed out" : "", (w & 2u) ? " workgroups of a group on different XCDs" : "");
        return RLREP_ERR_STATE;
    }
    return 0;
}

__global__ void debug_stamp_kernel(long long* ring, int cap, int tag) {
    const unsigned long long i = atomicAdd((unsigned long long*)ring, 1ull);
    ring[1 + (long long)(i % (unsigned long long)cap)] = (long long)((wall_clock64() << 8) | (unsigned long long)(tag & 255));
}
int32_t rlrep_debug_stamp(int64_t* ring, int32_t cap, int32_t tag, void* stream) {
    if (!ring || cap <= 0) { rl_set_error("debug_stamp: bad argument"); return RLREP_ERR_ARG; }
    hipLaunchKernelGGL(debug_stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (long long*)ring, (int)cap, (int)tag);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rl_set_error("debug_stamp: %s", hipGetErrorString(e)); return RLREP_ERR_HIP; }
    return 0;
}

int32_t rlrep_history(rlrep_agent* ag, int32_t on) {
    if (!ag) { rl_set_error("history: bad argument"); return RLREP_ERR_ARG; }
    ag->hist_on = on != 0;
    return 0;
}
int32_t rlrep_history_dev(rlrep_agent* ag, const float** ring, const int32_t** seq, int32_t* records, int32_t* record_floats, int32_t* tag_word) {
    if (!ag || !ag->hist) { rl_set_error("history_dev: bad argument"); return RLREP_ERR_ARG; }
    if (ring) *ring = ag->hist;
    if (seq) *seq = ag->hist_seq;
    if (records) *records = RL_HIST_N;
    if (record_floats) *record_floats = RL_HIST_REC;
    if (tag_word) *tag_word = RL_HIST_TAG;
    return 0;
}

int32_t rlrep_build_flags(void) {
#ifdef RL_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

const float* rlrep_metrics_dev(rlrep_agent* ag) { return ag ? ag->metrics : nullptr; }
int64_t rlrep_launch_counter(void) { return g_rl_launches; }
// rlrep_comm_attach (comm.hip): from now on the optimizer launches of the attached groups carry the data-parallel exchange, and -- with
// exchange scratch -- the feature step carries its batch-coupled exchanges (the programs are rebuilt)
extern "C" int rl_adam_dp_occupancy(int* one_shot, int* two_shot);
extern "C" int rl_agent_attach_dp(rlrep_agent* ag, const DpAttach* at, int* attached_mask) {
    if (!ag || !at) return RLREP_ERR_ARG;
    const DpPull* proto = &at->proto;
    if (proto->world != ag->h.world_size) { rl_set_error("comm_attach: the comm spans %d ranks, the agent was created with world_size = %d", proto->world, ag->h.world_size); return RLREP_ERR_ARG; }
    if (proto->world > 1 && proto->rank != ag->d.rank) { rl_set_error("comm_attach: the comm's rank is %d, the agent's dims.rank %d", proto->rank, ag->d.rank); return RLREP_ERR_ARG; }
    if (ag->a.grad_dev != proto->base[proto->rank]) { rl_set_error("comm_attach: the agent's gradient arena is not the comm's arena (create the agent with rlrep_comm_arena() as grad_dev)"); return RLREP_ERR_ARG; }
    rlrep_layout_info info;
    if (rlrep_layout(&ag->d, &info, nullptr, 0) != 0) return RLREP_ERR_ARG;
    if (at->arena_floats < info.grad_floats) { rl_set_error("comm_attach: the comm's arena holds %lld floats, the gradient arena needs %lld", at->arena_floats, (long long)info.grad_floats); return RLREP_ERR_ARG; }
    // Co-residency (dp_pull.h, "Progress with SEVERAL channels in flight"): the blocks of the two largest attached optimizer launches (riders
    // included: two minibatch gathers of at most 2048 blocks are NOT counted -- they wait for nothing and drain) must fit the chip together.
    int occ1 = 0, occ2 = 0, cus = 0, dev = 0;
    if (proto->world > 1) {
        if (rl_adam_dp_occupancy(&occ1, &occ2) != 0 || hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
            rl_set_error("comm_attach: cannot query the occupancy of the optimizer kernels"); return RLREP_ERR_HIP;
        }
    }
    ag->dp_proto = *proto;
    int mask = 0;
    long long blocks[4] = {0, 0, 0, 0};
    for (int g = 0; g < 4; ++g) {
        const bool on = proto->world > 1 && ag->L.group_n[g] > 0 && ag->L.group_n[g] <= at->max_floats;
        ag->dp_on[g] = on;
        ag->dp_two[g] = on && proto->world >= 3 && at->two_shot_floats > 0 && ag->L.group_n[g] >= at->two_shot_floats && proto->red[proto->rank] != nullptr;
        if (on) { mask |= 1 << g; blocks[g] = (ag->L.group_n[g] + 1023) / 1024 + 1 + 256; }       // optimizer blocks + trailing block + a folded snapshot's segments
    }
    // the two largest attached launches must fit the chip together; a group that does not leave room for a second one is NOT attached (its
    // gradients stay with the caller's all-reduce, as for every group above max_floats) -- largest first, until the bound holds
    const long long slots = (long long)cus * std::min(occ1, occ2 > 0 ? occ2 : occ1);
    while (mask) {
        int g0 = -1, g1 = -1;
        for (int g = 0; g < 4; ++g) if (mask & (1 << g)) { if (g0 < 0 || blocks[g] > blocks[g0]) { g1 = g0; g0 = g; } else if (g1 < 0 || blocks[g] > blocks[g1]) g1 = g; }
        if (blocks[g0] + (g1 >= 0 ? blocks[g1] : 0) <= slots) break;
        mask &= ~(1 << g0); ag->dp_on[g0] = ag->dp_two[g0] = false;
    }
    // batch-coupled exchanges: only when the feature group itself is attached (a train() is then one uninterrupted sequence of launches)
    const long long need = exchange_floats(ag->d, proto->world);
    const bool was = ag->xfold;
    ag->xfold = proto->world > 1 && ag->dp_on[0] && need > 0 && at->scratch_floats >= need && at->scratch[proto->rank] != nullptr && !rl_off("dp_fold_exchanges");
    ag->xscratch_floats = at->scratch_floats; ag->xarena_floats = at->arena_floats;
    for (int q = 0; q < RL_DP_MAX_WORLD; ++q) ag->xscratch[q] = q < proto->world ? at->scratch[q] : nullptr;
    if (attached_mask) *attached_mask = mask;
    if ((ag->xfold || was) && ag->B > 0) {
        (void)hipDeviceSynchronize();                      // (table re-upload uses blocking copies: rare path, once per attachment)
        return build_programs(ag, ag->B);
    }
    return 0;
}
int32_t rlrep_front_end_counts(int64_t* out4) {
    if (!out4) return RLREP_ERR_ARG;
    for (int q = 0; q < 4; ++q) out4[q] = g_rl_front[q];
    return 0;
}
int32_t rlrep_last_launch_count(rlrep_agent* ag) { return ag ? ag->last_launches : 0; }

}  // extern "C"

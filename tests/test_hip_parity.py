"""GPU parity: the HIP step programs (through the C ABI) against (a) the golden vectors captured from
the reference and (b) the CPU oracle on the same injected indices/noise.

Tolerance (BASELINE.json north_star): 1e-4 relative, fp32.  Metrics: |x-ref| <= 1e-4*max(|ref|,1e-2);
gradients and parameters: relative L2 <= 1e-4 per tensor (expected ~1e-6).
"""
import numpy as np
import pytest
import torch

from fixture_io import Case, cases, summary, rel_l2

pytestmark = pytest.mark.gpu

BUILT = ('sac', 'vlsac', 'ctrlsac', 'spedersac', 'diffsrsac')
OPT_GROUP = {'feature_optimizer': 0, 'critic_optimizer': 1, 'actor_optimizer': 2, 'phi_optimizer': 0,
             'nablamu_net_optimizer': 3}


class _Space:
    def __init__(self, A, bound):
        self.low, self.high = -bound * np.ones(A, np.float32), bound * np.ones(A, np.float32)


def make_agent(c, **extra):
    from rlrep_amd.agent.sac.sac_agent import SACAgent
    from rlrep_amd.agent.vlsac.vlsac_agent import VLSACAgent
    cls = {'sac': SACAgent, 'vlsac': VLSACAgent}
    try:
        from rlrep_amd.agent.ctrlsac.ctrlsac_agent import CTRLSACAgent
        from rlrep_amd.agent.spedersac.spedersac_agent import SPEDERSACAgent
        from rlrep_amd.agent.diffsrsac.diffsrsac_agent import DIFFSRSACAgent
        cls.update(ctrlsac=CTRLSACAgent, spedersac=SPEDERSACAgent, diffsrsac=DIFFSRSACAgent)
    except ImportError:
        pass
    if c.alg not in cls:
        pytest.skip(f'{c.alg} not built yet')
    kw = dict(c.kw)
    if c.meta.get('patch_vae_hidden'):
        kw['vae_hidden_dim'] = c.meta['patch_vae_hidden']
    agent = cls[c.alg](state_dim=c.S, action_dim=c.A, action_space=_Space(c.A, c.meta['bound']), max_batch=c.B,
                       graph=False, **kw, **extra)
    agent.core.load_state(c.init)
    return agent


def make_buffer(c):
    from rlrep_amd.utils.buffer import ReplayBuffer
    buf = ReplayBuffer(c.S, c.A, max_size=c.meta['replay_n'])
    r = c.replay
    buf.load(r['state'], r['action'], r['next_state'], r['reward'], r['done'])
    return buf


def _cmp(mine, ref, full):
    if full:
        return rel_l2(mine, ref)
    s = summary(mine)
    return max(abs(s[0] - ref[0]) / max(ref[0], 1e-12), rel_l2(s[2:], ref[2:]))


@pytest.mark.parametrize('name', cases())
def test_hip_matches_reference_golden(name):
    c = Case(name)
    agent = make_agent(c)
    buf = make_buffer(c)
    worst = dict(info=0.0, grad=0.0, param=0.0)
    for t, tr in enumerate(c.trains):
        info = agent.train_injected(buf, c.B, tr['idx'], tr['eps'])
        for k, v in tr['info'].items():
            err = abs(info[k] - v) / max(abs(v), 1e-2)
            worst['info'] = max(worst['info'], err)
            assert err < 1e-4, (name, t, k, info[k], v)
        # gradients of the last optimizer step of each group are still in the grad arena
        for optkey, gd in tr['grads'].items():
            opt, j = optkey.split('#')
            if opt not in OPT_GROUP:
                continue
            nsteps = 1 + max(int(k.split('#')[1]) for k in tr['grads'] if k.startswith(opt + '#'))
            if int(j) != nsteps - 1:
                continue
            for pname, g in gd.items():
                if pname not in agent.core.descs:
                    continue
                mine = agent.core.view(pname, 'grad').cpu().numpy()
                gref = g
                err = _cmp(mine, gref, c.full)
                # tiny-norm gradients (cancellation residue in the reference's autograd) get an absolute floor
                nrm = np.linalg.norm(gref) if c.full else gref[0]
                if nrm < 1e-6:
                    continue
                worst['grad'] = max(worst['grad'], err)
                assert err < 2e-4, (name, t, optkey, pname, err)
    st = agent.core.state()
    for k, v in c.final.items():
        if k not in st or k.endswith('noise'):
            continue
        err = _cmp(st[k].numpy(), v, c.full)
        worst['param'] = max(worst['param'], err)
        assert err < (1e-4 if c.T <= 3 else 5e-4), (name, k, err)       # T = 25 free run: Adam amplifies rounding on near-zero gradient elements
        # Polyak'd copies move by tau * (a few Adam steps): compare the MOVEMENT, the absolute check above cannot see
        # a target update that never ran
        if c.full and 'target' in k and k in c.init:
            d_ref = np.asarray(v, np.float64) - np.asarray(c.init[k], np.float64)
            d_mine = st[k].numpy().astype(np.float64) - np.asarray(c.init[k], np.float64)
            if np.linalg.norm(d_ref) > 1e-9:
                assert rel_l2(d_mine, d_ref) < 2e-2, (name, k, 'target movement', rel_l2(d_mine, d_ref))
    # Adam state after the last call (the fixtures' adam/<optimizer>/<name>/{m,v,step} summaries): moments of every tensor the
    # reference's optimizers hold state for, the fp64 temperature moments, and the step counters of the optimizer groups
    worst['adam'] = 0.0
    steps_seen = {}
    for key, ref in c.adam.items():
        opt, pname, which = key.split('/')
        if opt == 'log_alpha_optimizer':
            mine = float(agent.core.alpha_state[{'m': 1, 'v': 2, 'step': 3}[which]].item())
            want = float(ref) if which == 'step' else float(ref[1])          # summary()[1] = the sum = the scalar itself
            assert abs(mine - want) <= 1e-4 * max(abs(want), 1e-12), (name, key, mine, want)
            continue
        if pname not in agent.core.descs or opt not in OPT_GROUP:
            continue
        if which == 'step':
            steps_seen[OPT_GROUP[opt]] = int(ref)
            continue
        mine = agent.core.view(pname, 'exp_avg' if which == 'm' else 'exp_avg_sq').cpu().numpy()
        if ref[0] < 1e-12:
            continue
        err = _cmp(mine, ref, False)
        worst['adam'] = max(worst['adam'], err)
        # (T = 25 free run: the moments are running averages of 100 gradients whose rounding differences Adam has fed back 100 times;
        #  measured up to 9e-4 on the first encoder layer while every metric of every call stays within 1e-4)
        assert err < (1e-4 if which == 'm' else 2e-4) * (1 if c.T <= 3 else 20), (name, key, err)
    cfg_steps = agent.core.group_cfg()[:, 0].contiguous().view(torch.int32).cpu().numpy()
    for g, n in steps_seen.items():
        assert int(cfg_steps[g]) == n, (name, 'optimizer group', g, int(cfg_steps[g]), n)
    print(f'{name}: worst info {worst["info"]:.2e} grad {worst["grad"]:.2e} param {worst["param"]:.2e} adam {worst["adam"]:.2e}')


@pytest.mark.parametrize('alg', ['sac', 'vlsac'])
def test_hip_matches_oracle_fresh_seed(alg):
    """Same comparison against the CPU oracle on inputs no fixture has seen (different seed, B=64)."""
    from oracle import make_oracle
    from oracle.agents import gather_batch
    import synth
    base = Case(alg + '_tiny')
    c = base
    rs = np.random.RandomState(77)
    agent = make_agent(c)
    buf = make_buffer(c)
    agent.core.load_state(c.init)
    o = make_oracle(c.alg, c.S, c.A, c.init, **c.kw)
    torch.set_num_threads(4)
    F = c.kw.get('feature_dim', 0)
    nf = (c.kw.get('extra_feature_steps', 0) + 1) if alg != 'sac' else 0
    for t in range(3):
        nb = max(nf, 1)
        idx = [rs.randint(0, c.meta['replay_n'], size=c.B) for _ in range(nb)]
        eps = [rs.standard_normal((c.B, F)).astype(np.float32) for _ in range(nf)]
        eps += [rs.standard_normal((c.B, c.A)).astype(np.float32) for _ in range(2)]
        info = agent.train_injected(buf, c.B, idx, eps)
        oinfo = o.train([gather_batch(c.replay, i) for i in idx], [torch.as_tensor(e) for e in eps])
        for k, v in oinfo.items():
            assert abs(info[k] - v) <= 1e-4 * max(abs(v), 1e-2), (alg, t, k, info[k], v)
    st = agent.core.state()
    P = o.state()
    for k in st:
        if k in P and not k.endswith('noise'):
            assert rel_l2(st[k].numpy(), P[k].numpy()) < 1e-4, k


@pytest.mark.parametrize('name', ['spedersac_tiny', 'spedersac_ant512', 'ctrlsac_tiny', 'ctrlsac_hc2048'])
def test_theta_head_in_its_own_launches_matches_golden(name, monkeypatch):
    """By default the reward head theta rides in the loss launches: spedersac's gradient of theta.l (sum_i drhat_i phi_i, sum_i drhat_i) in the
    weighted column-sum launch of the spectral loss (colsum_kernel's second set), ctrlsac's rhat = theta . phi + b in the InfoNCE launch.
    RLREP_DISABLE=fold_theta keeps them tasks of the 16-row engine (a launch of their own when the neighbouring GEMMs route to the LDS-tiled engine).
    The golden tests above run the default; this runs the other form against the same reference fixtures."""
    if name not in cases():
        pytest.skip('fixture not present')
    monkeypatch.setenv('RLREP_DISABLE', 'fold_theta')
    test_hip_matches_reference_golden(name)


@pytest.mark.parametrize('name', ['diffsrsac_tiny', 'diffsrsac_hc'])
def test_diffsrsac_score_kernel_for_wide_states_matches_golden_at_small_ones(name, monkeypatch):
    """State dimensions of up to 32 take diffsr_score_small_kernel (a thread per row of U[b], one read of U); RLREP_DISABLE=score_small sends them
    through the lanes-over-s kernel that wider, non-multiple-of-four state dimensions use: same fixtures."""
    if name not in cases():
        pytest.skip('fixture not present')
    monkeypatch.setenv('RLREP_DISABLE', 'score_small')
    test_hip_matches_reference_golden(name)


@pytest.mark.parametrize('alg,B', [('sac', 7), ('vlsac', 100), ('ctrlsac', 33), ('spedersac', 50), ('diffsrsac', 19)])
def test_ragged_batch_sizes_match_oracle(alg, B):
    """Batch sizes that are not multiples of the 16-row MFMA tile (and a batch-size change on a live agent)
    exercise every masked edge of the kernels; compared with the CPU oracle on fresh inputs."""
    from oracle import make_oracle
    from oracle.agents import gather_batch
    c = Case(alg + '_tiny')
    rs = np.random.RandomState(1000 + B)
    agent = make_agent(c, **{})
    # the fixture's agent was created with max_batch = c.B; recreate with room for B
    import importlib
    kw = dict(c.kw)
    if c.meta.get('patch_vae_hidden'):
        kw['vae_hidden_dim'] = c.meta['patch_vae_hidden']
    agent = type(agent)(state_dim=c.S, action_dim=c.A, action_space=_Space(c.A, c.meta['bound']), max_batch=max(B, c.B),
                        graph=False, **kw)
    agent.core.load_state(c.init)
    buf = make_buffer(c)
    o = make_oracle(c.alg, c.S, c.A, c.init, **c.kw)
    torch.set_num_threads(4)
    F = c.kw.get('feature_dim', 0)
    nf = (c.kw.get('extra_feature_steps', 0) + 1) if alg != 'sac' else 0
    for t, bsz in enumerate([B, c.B, B]):          # also switches the batch size on the live agent
        nb = o.n_batches()
        idx = [rs.randint(0, c.meta['replay_n'], size=bsz) for _ in range(nb)]
        eps = []
        if alg == 'vlsac':
            eps = [rs.standard_normal((bsz, F)).astype(np.float32) for _ in range(nf)]
        if alg == 'diffsrsac':
            for _ in range(nf):
                eps += [rs.randint(0, 1000, size=bsz), (0.449 * rs.standard_normal((bsz, c.S))).astype(np.float32)]
        eps += [rs.standard_normal((bsz, c.A)).astype(np.float32) for _ in range(2)]
        info = agent.train_injected(buf, bsz, idx, eps)
        oinfo = o.train([gather_batch(c.replay, i) for i in idx],
                        [torch.as_tensor(e) for e in eps])
        for k, v in oinfo.items():
            assert abs(info[k] - v) <= 1e-4 * max(abs(v), 1e-2), (alg, t, bsz, k, info[k], v)
    st, P = agent.core.state(), o.state()
    for k in st:
        if k in P and not k.endswith('noise') and k != 'noise_alphabars':
            assert rel_l2(st[k].numpy(), P[k].numpy()) < 1e-4, (alg, k)


def test_errors_are_loud():
    """Argument / state errors surface as RuntimeError with the library's message (no silent fallback)."""
    c = Case('sac_tiny')
    agent = make_agent(c)
    with pytest.raises(RuntimeError, match='before set_batch'):
        agent.core.critic_step(torch.zeros(c.B, c.A, device='cuda'))
    buf = make_buffer(c)
    with pytest.raises(RuntimeError, match='max_batch'):
        agent.train_injected(buf, c.B + 1, [np.zeros(c.B + 1, np.int64)], [np.zeros((c.B + 1, c.A), np.float32)] * 2)
    with pytest.raises(RuntimeError, match='no feature step'):
        agent.core.feature_step(None)


def test_select_action_matches_oracle_mean():
    from oracle.agents import actor_mu_std
    c = Case('vlsac_tiny')
    agent = make_agent(c)
    P = {k: torch.as_tensor(v) for k, v in c.init.items()}
    rs = np.random.RandomState(0)
    for _ in range(3):
        s = rs.standard_normal(c.S).astype(np.float32)
        a = agent.select_action(s)
        mu, _ = actor_mu_std(P, torch.as_tensor(s)[None])
        assert np.allclose(a, torch.tanh(mu)[0].numpy(), atol=1e-5), (a, torch.tanh(mu))
        ae = agent.select_action(s, explore=True)
        assert ae.shape == (c.A,) and np.all(np.abs(ae) <= 1.0)


def _tweaked_case(alg, tweak):
    c = Case(alg + '_tiny')
    c.init = {k: np.array(v, copy=True) for k, v in c.init.items()}
    tweak(c.init)
    return c


def _run_vs_oracle(c, trains=2, seed=5):
    from oracle import make_oracle
    from oracle.agents import gather_batch
    agent = make_agent(c)
    buf = make_buffer(c)
    o = make_oracle(c.alg, c.S, c.A, c.init, **c.kw)
    rs = np.random.RandomState(seed)
    F = c.kw.get('feature_dim', 0)
    nf = (c.kw.get('extra_feature_steps', 0) + 1) if c.alg != 'sac' else 0
    for t in range(trains):
        idx = [rs.randint(0, c.meta['replay_n'], size=c.B) for _ in range(o.n_batches())]
        eps = [rs.standard_normal((c.B, F)).astype(np.float32) for _ in range(nf)] if c.alg == 'vlsac' else []
        eps += [rs.standard_normal((c.B, c.A)).astype(np.float32) for _ in range(2)]
        info = agent.train_injected(buf, c.B, idx, eps)
        oinfo = o.train([gather_batch(c.replay, i) for i in idx], [torch.as_tensor(e) for e in eps])
        for k, v in oinfo.items():
            assert np.isfinite(float(info[k])), (k, info[k])
            assert abs(info[k] - v) <= 2e-4 * max(abs(v), 1e-2), (c.alg, t, k, info[k], v)
    st, P = agent.core.state(), o.state()
    for k in st:
        if k in P and not k.endswith('noise'):
            assert rel_l2(st[k].numpy(), P[k].numpy()) < 1e-4, (c.alg, k)
    return agent, o


def test_rare_branches_match_oracle():
    """Inputs that FORCE the data-dependent branches of the fused kernels (SURVEY.md 7.4 'op KAT' edge cases):
    log-std clamp edges (gradient masked above 2 / below -20), exact min(Q1,Q2) ties (sub-gradient split 1/2:1/2),
    tanh saturation with the softplus(-2x) > 20 branch, ELU at its kink."""
    # (1) clamp: half of f's and the encoder's log-std heads far above +2, the other half far below -20
    def clamp(P):
        for m in ('f', 'f_target', 'encoder'):
            b = P[m + '.log_std_linear.bias']
            b[: len(b) // 2] = 7.0
            b[len(b) // 2:] = -30.0
    agent, o = _run_vs_oracle(_tweaked_case('vlsac', clamp))
    g = agent.core.view('f.log_std_linear.weight', 'grad').cpu().numpy()
    assert np.all(g == 0.0), 'clamped log-std heads must receive exactly zero gradient'
    # (2) ties: both Q heads identical -> q1 == q2 bit for bit in the actor step
    def tie(P):
        for k in list(P):
            if k.startswith('critic.Q2.'):
                P[k] = P[k.replace('.Q2.', '.Q1.')].copy()
            if k.startswith('critic_target.Q2.'):
                P[k] = P[k.replace('.Q2.', '.Q1.')].copy()
    _run_vs_oracle(_tweaked_case('sac', tie), trains=1)
    # (3) saturation: huge actor mean -> |x| > 10: tanh(x) == +-1 in fp32, softplus threshold branch, log-prob stays finite
    def saturate(P):
        P['actor.trunk.4.bias'][: len(P['actor.trunk.4.bias']) // 2] = np.array([14.0, -14.0, 11.0])[: len(P['actor.trunk.4.bias']) // 2]
    _run_vs_oracle(_tweaked_case('sac', saturate), trains=1)
    # (4) ELU kink: zero first-layer critic weights and biases -> pre-activations exactly 0
    def kink(P):
        P['critic.Q1.0.weight'][:] = 0.0
        P['critic.Q1.0.bias'][:] = 0.0
    _run_vs_oracle(_tweaked_case('sac', kink), trains=1)


def test_policy_prefetch_is_equivalent():
    """rlrep_prefetch_policy only regroups launches: train() with the actor step's forward half riding in the critic
    step's launches equals train() with the two steps run back to back (and the actor step gets shorter)."""
    c = Case('vlsac_tiny')
    outs, launches = [], []
    for hoist in (True, False):
        agent = make_agent(c)
        buf = make_buffer(c)
        armed = []
        if not hoist:
            agent.core.prefetch_policy = lambda eps: False
        else:
            orig = agent.core.prefetch_policy
            agent.core.prefetch_policy = lambda eps, orig=orig: armed.append(orig(eps)) or armed[-1]
        infos = [agent.train_injected(buf, c.B, tr['idx'], tr['eps']) for tr in c.trains]
        if hoist:
            assert armed and all(armed), 'vlsac_tiny must support the prefetched policy forward'
        launches.append(agent.core.launch_count())
        outs.append((infos, {k: v.numpy().copy() for k, v in agent.core.state().items()}))
    (ia, sa), (ib, sb) = outs
    for x, y in zip(ia, ib):
        for k in y:
            assert abs(x[k] - y[k]) <= 1e-6 * max(abs(y[k]), 1e-2), k
    for k in sb:
        assert rel_l2(sa[k], sb[k]) < 1e-6, k
    assert launches[0] < launches[1], launches


@pytest.mark.parametrize('name', ['vlsac_tiny', 'sac_tiny', 'spedersac_tiny'])
def test_graph_prologue_and_batch_prefetch_are_equivalent(name):
    """Graph-replayed train(): the one-launch prologue (steps += 1, pools, first gather) and the gathers that ride in the
    optimizer launches produce exactly what the separate begin_train / fill / replay_sample launches produce."""
    c = Case(name)
    outs = []
    for fused in (True, False):
        kw = dict(c.kw)
        if c.meta.get('patch_vae_hidden'):
            kw['vae_hidden_dim'] = c.meta['patch_vae_hidden']
        cls = type(make_agent(c))
        agent = cls(state_dim=c.S, action_dim=c.A, action_space=_Space(c.A, c.meta['bound']), max_batch=c.B, graph=True, seed=1234, **kw)
        agent.core.load_state(c.init)
        buf = make_buffer(c)
        core = agent.core
        if not fused:
            def prologue(ring, size_dev, ipool, epool, seed, ioff, eoff, B, core=core):
                core.begin_train()
                core.fill_indices_dev(ipool, size_dev, seed, ioff)
                core.fill_normal_dev(epool, 1.0, seed, eoff)
            core.train_prologue = prologue
            core.prefetch_batch = lambda ring, idx, B, slot=0: False
            core.prefetch_policy_early = lambda e1, e2: False
        infos = [agent.train(buf, c.B) for _ in range(4)]
        torch.cuda.synchronize()
        outs.append((infos, {k: v.numpy().copy() for k, v in core.state().items()}))
    (ia, sa), (ib, sb) = outs
    for x, y in zip(ia, ib):
        for k in y:
            assert abs(float(x[k]) - float(y[k])) <= 1e-6 * max(abs(float(y[k])), 1e-2), (name, k, x[k], y[k])
    for k in sb:
        assert rel_l2(sa[k], sb[k]) < 1e-6, (name, k)


@pytest.mark.parametrize('name', ['vlsac_tiny', 'ctrlsac_tiny', 'spedersac_tiny'])
def test_deferred_pipeline_is_equivalent(name):
    """Pipelined graph mode (critic + actor of train(t) as a graph branch beside the feature steps of train(t+1), against a
    snapshot of f_target / minibatch / noise / step counter) leaves every parameter, Adam moment and target exactly where the
    sequential graph mode leaves it -- also with select_action() (which must see the finished actor) between train() calls."""
    c = Case(name)
    outs = []
    for pipe in (True, False):
        kw = dict(c.kw)
        if c.meta.get('patch_vae_hidden'):
            kw['vae_hidden_dim'] = c.meta['patch_vae_hidden']
        cls = type(make_agent(c))
        agent = cls(state_dim=c.S, action_dim=c.A, action_space=_Space(c.A, c.meta['bound']), max_batch=c.B, graph=True, pipeline=pipe,
                    seed=4321, **kw)
        agent.core.load_state(c.init)
        buf = make_buffer(c)
        acts = []
        for t in range(7):
            info = agent.train(buf, c.B)
            if t in (2, 5):
                acts.append(agent.select_action(np.full(c.S, 0.1 * t, np.float32)))
        last = {k: float(v) for k, v in info.items()}
        if pipe:
            assert agent._pipe is not None, f'{name} must take the pipelined path'
        st = {k: v.numpy().copy() for k, v in agent.core.state().items()}
        m = {k: agent.core.exp_avg.cpu().numpy().copy() for k in ('m',)}
        outs.append((st, m, acts, last))
    (sa, ma, aa, ia), (sb, mb, ab_, ib) = outs
    for k in sb:
        assert rel_l2(sa[k], sb[k]) < 1e-6, k
    assert rel_l2(ma['m'], mb['m']) < 1e-6
    for x, y in zip(aa, ab_):
        assert np.allclose(x, y, atol=1e-6)
    for k in ib:
        assert abs(ia[k] - ib[k]) <= 1e-6 * max(abs(ib[k]), 1e-2), (k, ia[k], ib[k])


@pytest.mark.parametrize('alg,S,A,B,calls,kw', [
    ('vlsac', 17, 6, 256, 400, dict(hidden_dim=256, feature_dim=256, extra_feature_steps=3)),
    ('ctrlsac', 17, 6, 256, 400, dict(hidden_dim=256, feature_dim=256, extra_feature_steps=3)),
    # the configurations whose default-mode oracle check flushes after every call (so the two chains never overlap
    # while being compared) get the overlap covered here, at their BASELINE dimensions (fewer calls: 17 / 4 ms of GPU time per pair)
    ('ctrlsac', 17, 6, 256, 120, dict(hidden_dim=1024, feature_dim=2048, extra_feature_steps=3)),
    ('spedersac', 111, 8, 1024, 120, dict(phi_and_mu_lr=1e-5, phi_hidden_dim=512, phi_hidden_depth=1, mu_hidden_dim=512, mu_hidden_depth=0,
                                          critic_and_actor_lr=3e-4, critic_and_actor_hidden_dim=256, feature_dim=512, hidden_dim=256,
                                          extra_feature_steps=5)),
], ids=['vlsac_hc', 'ctrlsac_hc256', 'ctrlsac_hc2048', 'spedersac_ant512'])
def test_deferred_pipeline_soak_is_bit_identical(alg, S, A, B, calls, kw, monkeypatch):
    """Pipelined train() calls at the BASELINE dimensions (two streams, three snapshot sets, device Philox) end in exactly the
    parameters, moments and targets of as many sequential ones: any missing dependency between the two launch chains would show here
    (tools/exp/pipe_soak.py runs the same check for thousands of calls: 0.0 difference for vlsac, ctrlsac and spedersac)."""
    import importlib
    # both forms on the same POLICY-FORWARD kernels: the sequential form would otherwise run both policy forwards inside the last feature
    # step's launches, the pipelined form inside the critic step's tile launches (different summation order, last-bit differences)
    monkeypatch.setenv('RLREP_DISABLE', 'early_policy')
    import synth
    from rlrep_amd.utils.buffer import ReplayBuffer
    name = {'vlsac': 'VLSACAgent', 'ctrlsac': 'CTRLSACAgent', 'spedersac': 'SPEDERSACAgent'}[alg]
    cls = getattr(importlib.import_module(f'rlrep_amd.agent.{alg}.{alg}_agent'), name)
    data = synth.replay(S, A, 8192, seed=0)
    outs = []
    for pipe in (True, False):
        torch.manual_seed(0)
        agent = cls(state_dim=S, action_dim=A, action_space=_Space(A, 1.0), max_batch=B, pipeline=pipe, seed=99, **kw)
        buf = ReplayBuffer(S, A, max_size=8192)
        buf.load(data['state'], data['action'], data['next_state'], data['reward'], data['done'])
        for i in range(calls):
            agent.train(buf, B)
            if i == (calls * 5) // 8:
                agent.select_action(np.zeros(S, np.float32))
        if pipe:
            assert agent._pipe is not None and agent._pipe.get('mode') == 2, 'the pipelined agent must have taken the two-stream form'
        outs.append({k: v.numpy().copy() for k, v in agent.core.state().items()})
        outs[-1]['exp_avg'] = agent.core.exp_avg.cpu().numpy().copy()
        del agent, buf
        torch.cuda.synchronize()
    for k in outs[1]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_pipelined_and_sequential_default_forms_agree():
    """The two default forms as they ship (the pipelined one runs both policy forwards inside the critic step's tile launches, the
    sequential one inside the last feature step's: different summation order): 40 train() calls at the headline dimensions, parameters
    within 1e-6 relative (measured 5e-9: last-bit rounding)."""
    import synth
    from rlrep_amd.utils.buffer import ReplayBuffer
    from rlrep_amd.agent.vlsac.vlsac_agent import VLSACAgent
    data = synth.replay(17, 6, 8192, seed=0)
    outs = []
    for pipe in (True, False):
        torch.manual_seed(0)
        agent = VLSACAgent(state_dim=17, action_dim=6, action_space=_Space(6, 1.0), max_batch=256, pipeline=pipe, seed=99,
                           hidden_dim=256, feature_dim=256, extra_feature_steps=3)
        buf = ReplayBuffer(17, 6, max_size=8192)
        buf.load(data['state'], data['action'], data['next_state'], data['reward'], data['done'])
        for i in range(40):
            agent.train(buf, 256)
        outs.append({k: v.numpy().copy() for k, v in agent.core.state().items()})
    for k in outs[1]:
        assert rel_l2(outs[0][k], outs[1][k]) < 1e-6, k


def test_deferred_pipeline_batch_change_and_checkpoint():
    """Pipelined graph mode across a batch-size change (graphs and snapshot sets are rebuilt) and a save / load round trip: same final
    state as sequential graph mode doing the same things."""
    c = Case('vlsac_tiny')
    outs = []
    for pipe in (True, False):
        kw = dict(c.kw)
        if c.meta.get('patch_vae_hidden'):
            kw['vae_hidden_dim'] = c.meta['patch_vae_hidden']
        cls = type(make_agent(c))
        agent = cls(state_dim=c.S, action_dim=c.A, action_space=_Space(c.A, c.meta['bound']), max_batch=c.B, graph=True, pipeline=pipe,
                    seed=2024, **kw)
        agent.core.load_state(c.init)
        buf = make_buffer(c)
        for _ in range(3):
            agent.train(buf, c.B)
        for _ in range(3):
            agent.train(buf, c.B // 2)
        snap = agent.state_snapshot()
        for _ in range(2):
            agent.train(buf, c.B)
        mid = {k: v.numpy().copy() for k, v in agent.core.state().items()}
        agent.load(snap)
        for _ in range(2):
            agent.train(buf, c.B)
        end = {k: v.numpy().copy() for k, v in agent.core.state().items()}
        for k in mid:
            assert np.array_equal(mid[k], end[k]), ('resume', pipe, k)
        outs.append(end)
    for k in outs[1]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_deferred_pipeline_sees_rows_added_between_calls():
    """ReplayBuffer.add() stages rows on the caller's stream; a pipelined train() orders its feature chain after them only when the buffer
    has enqueued something since the last call (`device_epoch`).  A ring that starts with B rows and grows by three rows before every
    call: every draw depends on the new size and, soon, on the new rows -- same final state as the sequential graph mode."""
    from rlrep_amd.utils.buffer import ReplayBuffer
    c = Case('vlsac_tiny')
    rs = np.random.RandomState(4)
    extra = [(rs.standard_normal(c.S).astype(np.float32), rs.uniform(-1, 1, c.A).astype(np.float32), rs.standard_normal(c.S).astype(np.float32),
              float(rs.standard_normal()), float(rs.rand() < 0.1)) for _ in range(60)]
    outs = []
    for pipe in (True, False):
        kw = dict(c.kw)
        if c.meta.get('patch_vae_hidden'):
            kw['vae_hidden_dim'] = c.meta['patch_vae_hidden']
        cls = type(make_agent(c))
        agent = cls(state_dim=c.S, action_dim=c.A, action_space=_Space(c.A, c.meta['bound']), max_batch=c.B, graph=True, pipeline=pipe,
                    seed=77, **kw)
        agent.core.load_state(c.init)
        buf = ReplayBuffer(c.S, c.A, max_size=64)
        r = c.replay
        buf.load(r['state'][:c.B], r['action'][:c.B], r['next_state'][:c.B], r['reward'][:c.B], r['done'][:c.B])
        it = iter(extra)
        for t in range(16):
            for _ in range(3):
                buf.add(*next(it))
            agent.train(buf, c.B)
            if t % 5 == 4:
                agent.train(buf, c.B)          # and a call with nothing new in the buffer
        outs.append({k: v.numpy().copy() for k, v in agent.core.state().items()})
    for k in outs[1]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_pipelined_info_early_keys_match_sequential():
    """The feature-step losses of a pipelined train() can be read as soon as its feature chain has ended (LazyInfo early keys), without
    flushing the critic / actor chain; every value read that way, and the critic / actor values read afterwards from the same dict, equal
    the sequential graph mode's."""
    c = Case('vlsac_tiny')
    seen = []
    for pipe in (True, False):
        kw = dict(c.kw)
        if c.meta.get('patch_vae_hidden'):
            kw['vae_hidden_dim'] = c.meta['patch_vae_hidden']
        cls = type(make_agent(c))
        agent = cls(state_dim=c.S, action_dim=c.A, action_space=_Space(c.A, c.meta['bound']), max_batch=c.B, graph=True, pipeline=pipe,
                    seed=5, **kw)
        agent.core.load_state(c.init)
        buf = make_buffer(c)
        vals = []
        for t in range(8):
            info = agent.train(buf, c.B)
            vals.append((info['kl_loss'], info['vae_loss']))            # early keys only: no flush in pipelined mode
            if pipe and t < 7:
                assert agent._pending, 'reading feature losses must not end the overlap'
            if t % 3 == 2:
                vals.append((info['q1_loss'], info['actor_loss'], info['kl_loss']))   # then the rest of the same dict
        seen.append(vals)
    assert seen[0] == seen[1]

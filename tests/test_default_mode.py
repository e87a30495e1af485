"""The BENCHMARKED mode against the oracle, and the device random-number path against its restatement.

`bench.py` times `agent.train(buffer, B)` in the default mode: hipGraph replay, sample indices and Gaussian noise drawn on the
device by Philox4x32-10 from the device-resident train() counter, critic / actor chain of call t running beside the feature chain
of call t+1.  The golden / oracle tests elsewhere drive `train_injected` with `graph=False`.  Here the default mode itself is
checked: after every `train()` the draws it actually used are read back from the pools (`pool_idx`, `pool_eps`; diffsrsac's
per-step `idx_n*` / `eps_pert*`), fed to the CPU oracle in the reference's draw order (reference draw sites
utils/buffer.py:39-48, networks/vae.py:50-57, agent/sac/actor.py:47-60, agent/diffsrsac/diffsrsac_agent.py:276-283) and metrics
and parameters must agree within 1e-4 (BASELINE.json north_star tolerance).  A wrong noise scale, an off-by-one index range or a
gather that used other indices than the pool holds fails these tests.
"""
import numpy as np
import pytest
import torch

from fixture_io import Case, rel_l2

pytestmark = pytest.mark.gpu


def _needs_experiments():
    """Tests of the opt-in engines that were measured and not adopted (row programs, cluster form, fused first layers): they are compiled
    into librlrep_hip_exp.so only (RLREP_BUILD_EXPERIMENTS=1; run them with RLREP_LIB=rlrep_amd/lib/librlrep_hip_exp.so)."""
    from rlrep_amd import _lib
    return pytest.mark.skipif(not _lib.has_experiments(), reason='opt-in engines are not compiled into this library (RLREP_LIB=.../librlrep_hip_exp.so)')


class _Space:
    def __init__(self, A, bound):
        self.low, self.high = -bound * np.ones(A, np.float32), bound * np.ones(A, np.float32)


def _cls(alg):
    import importlib
    name = {'sac': 'SACAgent', 'vlsac': 'VLSACAgent', 'ctrlsac': 'CTRLSACAgent', 'spedersac': 'SPEDERSACAgent',
            'diffsrsac': 'DIFFSRSACAgent'}[alg]
    return getattr(importlib.import_module(f'rlrep_amd.agent.{alg}.{alg}_agent'), name)


def _default_agent(c, **extra):
    kw = dict(c.kw)
    if c.meta.get('patch_vae_hidden'):
        kw['vae_hidden_dim'] = c.meta['patch_vae_hidden']
    extra.setdefault('adaptive', False)       # these tests flush after every call to read the draws back: keep them on the form a train() loop runs
    agent = _cls(c.alg)(state_dim=c.S, action_dim=c.A, action_space=_Space(c.A, c.meta['bound']), max_batch=c.B, seed=20240 + len(c.name),
                        **kw, **extra)                        # no graph= / pipeline= argument: the defaults bench.py runs
    agent.core.load_state(c.init)
    return agent


def _buffer(c):
    from rlrep_amd.utils.buffer import ReplayBuffer
    buf = ReplayBuffer(c.S, c.A, max_size=c.meta['replay_n'])
    r = c.replay
    buf.load(r['state'], r['action'], r['next_state'], r['reward'], r['done'])
    return buf


def _read_back_draws(agent, B):
    """(idx list, eps list) of the train() call that has just been flushed, in the oracle's consumption order."""
    idx_keys, eps_specs = agent._plan(B)
    ipool = agent._bufs['pool_idx'].cpu().numpy()
    epool = agent._bufs['pool_eps'].cpu().numpy()
    assert ipool.size == len(idx_keys) * B
    idx = [ipool[q * B:(q + 1) * B].astype(np.int64) for q in range(len(idx_keys))]
    eps, o = {}, 0
    for k, sh in eps_specs:
        n = int(np.prod(sh))
        eps[k] = epool[o:o + n].reshape(sh).copy()
        o += n
    assert o == epool.size
    nf = agent._feature_iters()
    out = []
    if agent.ALG == 'vlsac':
        out += [eps[f'feat{i}'] for i in range(nf)]
    if agent.ALG == 'diffsrsac':
        for i in range(nf):
            out += [agent._bufs[f'idx_n{i}'].cpu().numpy().astype(np.int64), agent._bufs[f'eps_pert{i}'].cpu().numpy().copy()]
    out += [eps['crit'], eps['act']]
    return idx, out


def _check_against_oracle(c, calls, expect_pipeline):
    from oracle import make_oracle
    from oracle.agents import gather_batch
    torch.set_num_threads(4)
    agent, buf = _default_agent(c), _buffer(c)
    assert agent.use_graph, 'the default mode is graph replay'
    o = make_oracle(c.alg, c.S, c.A, c.init, **c.kw)
    n = c.meta['replay_n']
    seen = []
    for t in range(calls):
        info = agent.train(buf, c.B)
        agent.flush()
        torch.cuda.synchronize()
        idx, eps = _read_back_draws(agent, c.B)
        for i in idx:
            assert i.min() >= 0 and i.max() < n, (c.name, t, i.min(), i.max())
        seen.append(np.concatenate([e.ravel() for e in eps if e.dtype == np.float32]))
        oinfo = o.train([gather_batch(c.replay, i) for i in idx], [torch.as_tensor(e) for e in eps])
        for k, v in oinfo.items():
            assert abs(info[k] - v) <= 1e-4 * max(abs(v), 1e-2), (c.name, t, k, info[k], v)
    if expect_pipeline:
        assert agent._pipe is not None and agent._pipe.get('mode') != 1, f'{c.name} must take the two-stream pipelined path'
    # fresh draws every call (the device counter advanced): no two calls saw the same noise
    for a in range(len(seen)):
        for b in range(a + 1, len(seen)):
            assert not np.array_equal(seen[a], seen[b]), (c.name, a, b)
    st, P = agent.core.state(), o.state()
    worst = 0.0
    for k in st:
        if k in P and not k.endswith('noise') and k != 'noise_alphabars':
            e = rel_l2(st[k].numpy(), P[k].numpy())
            worst = max(worst, e)
            assert e < 1e-4, (c.name, k, e)
    return worst


@pytest.mark.parametrize('alg,pipe', [('sac', False), ('vlsac', True), ('ctrlsac', True), ('spedersac', True), ('diffsrsac', False)])
def test_default_mode_matches_oracle_tiny(alg, pipe):
    worst = _check_against_oracle(Case(alg + '_tiny'), calls=4, expect_pipeline=pipe)
    print(f'{alg}_tiny default mode vs oracle: worst param rel-L2 {worst:.2e}')


def test_default_mode_matches_oracle_headline_dims():
    """BASELINE config[1] (vlsac, HalfCheetah dims, F = H = 256, B = 256, 4 feature steps): exactly what bench.py times."""
    worst = _check_against_oracle(Case('vlsac_hc'), calls=3, expect_pipeline=True)
    print(f'vlsac_hc default mode vs oracle: worst param rel-L2 {worst:.2e}')


def test_default_mode_sequential_graph_matches_oracle(monkeypatch):
    """RLREP_PIPELINE=0: the one-graph sequential form of the same train()."""
    monkeypatch.setenv('RLREP_PIPELINE', '0')
    _check_against_oracle(Case('vlsac_tiny'), calls=3, expect_pipeline=False)


# ---- the generator itself ------------------------------------------------------------------------------------------------------
def test_philox_known_answers_on_device():
    """Random123's published Philox4x32-10 vectors through the device function the fills are built on."""
    import ctypes as C
    from rlrep_amd._lib import lib, check
    from oracle.philox import KAT, philox4x32_10
    rs = np.random.RandomState(3)
    extra = rs.randint(0, 2 ** 32, size=(64, 6), dtype=np.uint64)
    ck = np.array([list(c) + list(k) for c, k, _ in KAT] + extra.tolist(), dtype=np.uint32)
    want = np.array([list(e) for _, _, e in KAT], dtype=np.uint32)
    d_in = torch.from_numpy(ck.view(np.int32)).cuda()
    d_out = torch.zeros(len(ck), 4, dtype=torch.int32, device='cuda')
    check(lib.rlrep_philox_raw(C.c_void_p(d_in.data_ptr()), C.c_void_p(d_out.data_ptr()), len(ck),
                               C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'philox_raw')
    got = d_out.cpu().numpy().view(np.uint32)
    assert np.array_equal(got[:3], want), (got[:3], want)
    assert np.array_equal(got[3:], philox4x32_10(ck[3:, :4], ck[3:, 4:]))


def _core():
    c = Case('sac_tiny')
    from rlrep_amd.agent.sac.sac_agent import SACAgent
    return SACAgent(state_dim=c.S, action_dim=c.A, action_space=_Space(c.A, 1.0), max_batch=c.B, **c.kw).core


@pytest.mark.parametrize('n,hi,seed,offset', [(1, 7, 0, 0), (1023, 1000, 12345, (1 << 40) + 17), (4096, 65536, 2 ** 31 - 1, 5 << 20),
                                              (100003, 1000000, 987654321, (1 << 40) + 123456)])
def test_index_stream_is_bit_exact_and_in_range(n, hi, seed, offset):
    from oracle import philox
    core = _core()
    t = torch.full((n,), -1, dtype=torch.int32, device='cuda')
    core.fill_indices(t, hi, seed, offset)
    got = t.cpu().numpy()
    assert got.min() >= 0 and got.max() < hi
    assert np.array_equal(got, philox.indices(n, hi, seed, offset))


def test_index_stream_is_uniform():
    """chi-square of 2^20 draws over 1000 cells (999 degrees of freedom: mean 999, sd 44.7; bound = mean + 5 sd) and the
    extreme cells 0 and hi - 1 are both reached (an off-by-one range would lose or overshoot one of them)."""
    core = _core()
    n, hi = 1 << 20, 1000
    t = torch.empty(n, dtype=torch.int32, device='cuda')
    core.fill_indices(t, hi, 424242, 9 << 20)
    cnt = np.bincount(t.cpu().numpy(), minlength=hi).astype(np.float64)
    assert cnt.size == hi and cnt[0] > 0 and cnt[-1] > 0
    chi2 = ((cnt - n / hi) ** 2 / (n / hi)).sum()
    assert chi2 < 999 + 5 * 44.7, chi2


def test_device_counter_variant_reads_size_and_step_on_the_device():
    """rlrep_fill_indices_dev: range from a device scalar, stream offset = offset + the agent's train() counter."""
    from oracle import philox
    c = Case('sac_tiny')
    from rlrep_amd.agent.sac.sac_agent import SACAgent
    agent = SACAgent(state_dim=c.S, action_dim=c.A, action_space=_Space(c.A, 1.0), max_batch=c.B, **c.kw)
    core = agent.core
    hi_dev = torch.full((1,), 777, dtype=torch.int32, device='cuda')
    t = torch.empty(500, dtype=torch.int32, device='cuda')
    for step in range(3):
        core.fill_indices_dev(t, hi_dev, 99, 1 << 40)
        assert np.array_equal(t.cpu().numpy(), philox.indices(500, 777, 99, (1 << 40) + step)), step
        core.begin_train()                          # steps += 1 on the device


@pytest.mark.parametrize('n,std,seed,offset', [(5, 1.0, 1, 0), (4099, 0.449, 77, 2 << 40), (1 << 16, 1.0, 31337, (2 << 40) + 9)])
def test_normal_stream_matches_restatement(n, std, seed, offset):
    """Box-Muller on the Philox words, element by element (float32 libm differences between device and host: <= 2e-6 absolute at
    |x| <= 6)."""
    from oracle import philox
    core = _core()
    t = torch.zeros(n, dtype=torch.float32, device='cuda')
    core.fill_normal(t, std, seed, offset)
    got, want = t.cpu().numpy(), philox.normals(n, std, seed, offset)
    assert np.all(np.isfinite(got))
    assert np.max(np.abs(got - want)) <= 4e-6 * max(std, 1.0), np.max(np.abs(got - want))


def test_normal_stream_moments():
    """N(0, std^2) moments over 2^22 draws: mean 0 +- 5 sd/sqrt(n), variance, skewness 0, kurtosis 3, tail mass beyond 3 sigma."""
    core = _core()
    n = 1 << 22
    for std in (1.0, 0.449):
        t = torch.empty(n, dtype=torch.float32, device='cuda')
        core.fill_normal(t, std, 2718281828, 3 << 40)
        x = t.cpu().numpy().astype(np.float64) / std
        assert abs(x.mean()) < 5 / np.sqrt(n)
        assert abs(x.var() - 1.0) < 5 * np.sqrt(2.0 / n)
        assert abs((x ** 3).mean()) < 5 * np.sqrt(15.0 / n)
        assert abs((x ** 4).mean() - 3.0) < 5 * np.sqrt(96.0 / n)
        tail = (np.abs(x) > 3).mean()
        assert abs(tail - 0.0026998) < 5 * np.sqrt(0.0027 / n), tail
        # consecutive elements (the cos / sin pair of one Box-Muller draw, and neighbours across blocks) are uncorrelated
        assert abs(np.mean(x[:-1] * x[1:])) < 5 / np.sqrt(n)


def test_train_pools_are_the_documented_streams():
    """The pools a graph-replayed train() draws are exactly stream (seed, (1 << 40) + steps) for indices and (seed, (2 << 40) + steps)
    for noise, steps = the value of the device train() counter AFTER this call's increment (train_prologue_kernel: step_add = 1)."""
    from oracle import philox
    c = Case('vlsac_tiny')
    agent, buf = _default_agent(c), _buffer(c)
    for call in range(1, 4):
        agent.train(buf, c.B)
        agent.flush()
        torch.cuda.synchronize()
        ip, ep = agent._bufs['pool_idx'].cpu().numpy(), agent._bufs['pool_eps'].cpu().numpy()
        assert np.array_equal(ip, philox.indices(ip.size, c.meta['replay_n'], agent._seed, (1 << 40) + call))
        assert np.max(np.abs(ep - philox.normals(ep.size, 1.0, agent._seed, (2 << 40) + call))) <= 4e-6


def test_train_prologue_without_a_feature_step_still_advances_the_streams():
    """A C-ABI caller that runs rlrep_train_prologue and then only the critic step (no launch that refreshes the train() counter's mirror:
    advisor r05) must not draw the same indices and noise again: the next prologue catches the mirror up itself (one extra launch)."""
    from oracle import philox
    c = Case('vlsac_tiny')
    agent, buf = _default_agent(c), _buffer(c)
    core = agent.core
    ip = torch.zeros(c.B, dtype=torch.int32, device='cuda')
    ep = torch.zeros(2 * c.B * c.A, device='cuda')
    seen = []
    for call in range(1, 4):
        core.train_prologue(buf.ring, buf.size_dev(), ip, ep, 99, 1 << 40, 2 << 40, c.B)
        core.critic_step(ep[:c.B * c.A].view(c.B, c.A))
        core.end_train()
        torch.cuda.synchronize()
        got = ip.cpu().numpy()
        assert np.array_equal(got, philox.indices(got.size, c.meta['replay_n'], 99, (1 << 40) + call)), call
        seen.append(got.copy())
    assert not np.array_equal(seen[0], seen[1]) and not np.array_equal(seen[1], seen[2])


# ---- the opt-in row-program form of the vlsac feature step (rowprog.hip) ---------------------------------------------------------
@_needs_experiments()
@pytest.mark.parametrize('name,single', [('vlsac_tiny', False), ('vlsac_hc', False), ('vlsac_tiny', True)])
def test_row_program_feature_step_matches_oracle(name, single, monkeypatch):
    """RLREP_ENABLE=rowprog: forward + dX chains of a feature step as one launch of row-block programs (two workgroups per 16-row block that
    hand the Gaussian heads / KL gradients to each other through flags; RLREP_DISABLE=rowprog_pair: one workgroup per block, no hand-off),
    reading the forward layers' weights from transposed shadows that the Adam launch keeps current -- in the default (graph, pipelined)
    mode against the oracle on the read-back draws, like every other default-mode test."""
    monkeypatch.setenv('RLREP_ENABLE', 'rowprog')
    if single:
        monkeypatch.setenv('RLREP_DISABLE', 'rowprog_pair')
    c = Case(name)
    worst = _check_against_oracle(c, calls=3, expect_pipeline=True)
    print(f'{name} row programs vs oracle: worst param rel-L2 {worst:.2e}')


@_needs_experiments()
def test_row_program_shadows_follow_external_parameter_writes(monkeypatch):
    """The transposed shadows are regenerated at the head of every train() and before an eager feature step: parameters overwritten by
    the caller between calls (load_state / checkpoint load) are what the next step uses."""
    from oracle import make_oracle
    from oracle.agents import gather_batch
    monkeypatch.setenv('RLREP_ENABLE', 'rowprog')
    c = Case('vlsac_tiny')
    agent, buf = _default_agent(c), _buffer(c)
    agent.train(buf, c.B)
    agent.flush()
    agent.core.load_state(c.init)                                 # back to the fixture's parameters, behind the library's back
    agent.core.exp_avg.zero_(); agent.core.exp_avg_sq.zero_()
    agent.core.group_cfg()[:, 0] = 0                              # Adam step counters (int32 zero == float zero bit pattern)
    agent.core.alpha_state[1:] = 0
    o = make_oracle(c.alg, c.S, c.A, c.init, **c.kw)
    rs = np.random.RandomState(3)
    idx = rs.randint(0, c.meta['replay_n'], size=c.B)
    eps = rs.standard_normal((c.B, c.kw['feature_dim'])).astype(np.float32)
    batch = buf.gather(torch.as_tensor(idx, device='cuda'))
    info = agent.feature_step(batch, eps=torch.as_tensor(eps, device='cuda'))
    oinfo = o.feature_step(gather_batch(c.replay, idx), torch.as_tensor(eps))
    for k, v in oinfo.items():
        assert abs(info[k] - v) <= 1e-4 * max(abs(v), 1e-2), (k, info[k], v)


def test_snapshot_rides_in_the_last_feature_optimizer_launch(monkeypatch):
    """rlrep_defer_arm: the deferred chain's snapshot is written by the last feature step's optimizer launch (one dependent launch less on
    the chain that bounds train()); RLREP_DISABLE=fold_snapshot keeps the separate copy launch.  Both forms against the oracle, and the
    launch counts of the captured feature graph differ by exactly one."""
    c = Case('vlsac_hc')
    counts = {}
    for fold in (True, False):
        if not fold:
            monkeypatch.setenv('RLREP_DISABLE', 'fold_snapshot')
        agent, buf = _default_agent(c), _buffer(c)
        agent.train(buf, c.B)
        agent.flush()
        counts[fold] = agent._pipe['launches'][0]
        del agent
        _check_against_oracle(c, calls=3, expect_pipeline=True)
    assert counts[False] == counts[True] + 1, counts


def test_decoder_heads_and_mse_ride_in_the_decoder_dx_launch(monkeypatch):
    """FLAG_PRE_MSE (vlsac_agent.py:137-140's s_loss / r_loss): every 16-row tile of the decoder.l1 dX launch computes the heads' forward,
    the mse gradient and its squared-error partials itself, so the 'dec.heads + mse' launch leaves each feature step.
    RLREP_DISABLE=fold_mse keeps the separate launch.  Both forms against the oracle (s_loss / r_loss are among the compared metrics), and the
    captured feature graph is one launch per feature step shorter."""
    c = Case('vlsac_hc')
    counts = {}
    for fold in (True, False):
        if not fold:
            monkeypatch.setenv('RLREP_DISABLE', 'fold_mse')
        agent, buf = _default_agent(c), _buffer(c)
        agent.train(buf, c.B)
        agent.flush()
        counts[fold] = agent._pipe['launches'][0]
        steps = agent.extra_feature_steps + 1
        del agent
        _check_against_oracle(c, calls=3, expect_pipeline=True)
    assert counts[False] == counts[True] + steps, counts


def test_chained_feature_steps_change_nothing(monkeypatch):
    """rlrep_feature_chain_next (vlsac_agent.py:250-256's loop of feature steps, chained): the first layers' optimizer runs in the weight-gradient
    epilogues and the step's optimizer launch carries the next step's encoder.l1 / f.l1 on rows read straight from the ring.  Against
    RLREP_DISABLE=chain_next (every step its own ten launches): three launches fewer in the captured feature graph, and parameters, moments and
    targets BIT-identical after 30 pipelined train() calls at the headline dims (adam_elem's operations are pinned for exactly this); both
    forms against the oracle."""
    import synth
    from rlrep_amd.utils.buffer import ReplayBuffer
    from rlrep_amd.agent.vlsac.vlsac_agent import VLSACAgent
    c = Case('vlsac_hc')
    data = synth.replay(17, 6, 8192, seed=0)
    outs, counts = [], {}
    for chained in (True, False):
        if not chained:
            monkeypatch.setenv('RLREP_DISABLE', 'chain_next')
        torch.manual_seed(0)
        agent = VLSACAgent(state_dim=17, action_dim=6, action_space=_Space(6, 1.0), max_batch=256, pipeline=True, seed=77,
                           hidden_dim=256, feature_dim=256, extra_feature_steps=3)
        buf = ReplayBuffer(17, 6, max_size=8192)
        buf.load(data['state'], data['action'], data['next_state'], data['reward'], data['done'])
        for _ in range(30):
            agent.train(buf, 256)
        agent.flush()
        counts[chained] = agent._pipe['launches'][0]
        st = {k: v.numpy().copy() for k, v in agent.core.state().items()}
        st['exp_avg'] = agent.core.exp_avg.cpu().numpy().copy(); st['exp_avg_sq'] = agent.core.exp_avg_sq.cpu().numpy().copy()
        outs.append(st)
        del agent, buf
        _check_against_oracle(c, calls=3, expect_pipeline=True)
    assert counts[False] == counts[True] + 3, counts
    for k in outs[1]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_fast_front_end_changes_nothing(monkeypatch):
    """gemm16_fast_kernel / gemm16_fast4_kernel / gemm16_fastpre_kernel (operand loads issued from preloaded scalars, the record's loads under them)
    run the same tile body in the same order: against RLREP_DISABLE=gemm16_fast (every launch fetches its record first) parameters, moments and
    targets are BIT-identical after 20 pipelined train() calls at the headline dims."""
    import synth
    from rlrep_amd.utils.buffer import ReplayBuffer
    from rlrep_amd.agent.vlsac.vlsac_agent import VLSACAgent
    data = synth.replay(17, 6, 8192, seed=0)
    outs, routes = [], []
    for fast in (True, False):
        if not fast:
            monkeypatch.setenv('RLREP_DISABLE', 'gemm16_fast')
        torch.manual_seed(0)
        agent = VLSACAgent(state_dim=17, action_dim=6, action_space=_Space(6, 1.0), max_batch=256, pipeline=True, seed=78,
                           hidden_dim=256, feature_dim=256, extra_feature_steps=3)
        buf = ReplayBuffer(17, 6, max_size=8192)
        buf.load(data['state'], data['action'], data['next_state'], data['reward'], data['done'])
        for _ in range(20):
            agent.train(buf, 256)
        agent.flush()
        # the routing itself (rlrep_front_end_counts, taken around the capture of the running graphs): the comparison below is vacuous if the
        # default run silently fell back to the record front end (a launch qualifies only if all its operands lie within 16 GiB of one base:
        # HipCore carves every arena out of ONE allocation for that)
        fe = agent._pipe['front_ends']
        if fast:
            assert fe['fast'] + fe['fast4'] + fe['fastpre'] >= 20 and fe['fast'] >= 8 and fe['fast4'] >= 1 and fe['fastpre'] >= 4, fe
        else:
            assert fe['fast'] == fe['fast4'] == fe['fastpre'] == 0 and fe['record'] >= 30, fe
        routes.append(fe)
        st = {k: v.numpy().copy() for k, v in agent.core.state().items()}
        st['exp_avg'] = agent.core.exp_avg.cpu().numpy().copy(); st['exp_avg_sq'] = agent.core.exp_avg_sq.cpu().numpy().copy()
        outs.append(st)
        del agent, buf
    assert sum(routes[0].values()) == sum(routes[1].values()), routes        # same launches, different front ends
    for k in outs[1]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_noise_critic_weight_images_follow_external_parameter_writes():
    """vlsac at the headline dims runs its noise critic from bf16x3 images of critic.l1 / l4 and of their target copies (ShadowEnt kind 1).
    The live images are kept by the critic group's Adam launch and ALL of them are regenerated at the head of every critic step: parameters
    and targets overwritten by the caller between calls are what the next critic and actor steps use."""
    from oracle import make_oracle
    from oracle.agents import gather_batch
    c = Case('vlsac_hc')
    agent, buf = _default_agent(c), _buffer(c)
    agent.train(buf, c.B)
    agent.flush()
    agent.core.load_state(c.init)                                 # parameters AND targets back to the fixture's, behind the library's back
    agent.core.exp_avg.zero_(); agent.core.exp_avg_sq.zero_()
    agent.core.group_cfg()[:, 0] = 0
    agent.core.alpha_state[1:] = 0
    o = make_oracle(c.alg, c.S, c.A, c.init, **c.kw)
    rs = np.random.RandomState(5)
    idx = rs.randint(0, c.meta['replay_n'], size=c.B)
    batch = buf.gather(torch.as_tensor(idx, device='cuda'))
    obatch = gather_batch(c.replay, idx)
    for step in ('critic_step', 'update_actor_and_alpha'):
        eps = rs.standard_normal((c.B, c.A)).astype(np.float32)
        info = getattr(agent, step)(batch, eps=torch.as_tensor(eps, device='cuda'))
        oinfo = getattr(o, step)(obatch, torch.as_tensor(eps))
        for k, v in oinfo.items():
            assert abs(float(info[k]) - v) <= 1e-4 * max(abs(v), 1e-2), (step, k, float(info[k]), v)


def test_managed_weight_images_follow_external_writes_between_graph_replays(monkeypatch):
    """Captured train() graphs leave the noise critic's bf16x3 weight images to the optimizer launches (no refresh launch at the head of the critic
    step: rlrep_images_managed).  Anything else that writes the critic -- here the caller overwrites parameters, targets and optimizer state between
    two replays -- is noticed (torch's version counter of the arena block, the agent's dirty flag) and the images are regenerated before the next
    replay: the run ends bit-identical to the same run with the refresh launch kept (RLREP_DISABLE=managed_images), at the headline dimensions."""
    import synth
    from rlrep_amd.utils.buffer import ReplayBuffer
    from rlrep_amd.agent.vlsac.vlsac_agent import VLSACAgent
    c = Case('vlsac_hc')
    data = synth.replay(17, 6, 8192, seed=0)
    outs = []
    for managed in (True, False):
        if not managed:
            monkeypatch.setenv('RLREP_DISABLE', 'managed_images')
        torch.manual_seed(0)
        agent = VLSACAgent(state_dim=17, action_dim=6, action_space=_Space(6, 1.0), max_batch=256, pipeline=True, seed=79,
                           hidden_dim=256, feature_dim=256, extra_feature_steps=3)
        assert agent._img_on is False
        buf = ReplayBuffer(17, 6, max_size=8192)
        buf.load(data['state'], data['action'], data['next_state'], data['reward'], data['done'])
        for _ in range(5):
            agent.train(buf, 256)
        agent.flush()
        assert agent._img_on == managed
        agent.core.load_state(c.init)                                 # parameters AND targets replaced behind the library's back
        agent.core.exp_avg.zero_(); agent.core.exp_avg_sq.zero_()
        for _ in range(5):
            agent.train(buf, 256)
        agent.select_action(np.zeros(17, np.float32))                 # (the one-graph form's capture goes through the same bracket)
        for _ in range(4):
            agent.train(buf, 256); agent.select_action(np.zeros(17, np.float32))
        agent.flush()
        st = {k: v.numpy().copy() for k, v in agent.core.state().items()}
        outs.append(st)
        del agent, buf
    for k in outs[1]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize('name', ['vlsac_tiny_noft', 'ctrlsac_tiny_noft', 'spedersac_tiny_noft'])
def test_default_mode_without_feature_target(name):
    """use_feature_target=False in the default (graph, pipelined) mode: vlsac's deferred critic / actor chain then runs against a snapshot
    of the LIVE f (vlsac_agent.py:176-179, 214-219), ctrlsac / spedersac drop the Polyak copies."""
    c = Case(name)
    assert c.kw.get('use_feature_target') is False
    worst = _check_against_oracle(c, calls=4, expect_pipeline=True)
    print(f'{name} default mode vs oracle: worst param rel-L2 {worst:.2e}')


@_needs_experiments()
def test_fused_first_layers_match_oracle(monkeypatch):
    """RLREP_ENABLE=fuse_l1 (opt-in, measured slower): encoder.l1 / f.l1 recomputed inside the encoder.l2 / f.l2 launch from transposed weight
    shadows (gemm16.hip FLAG_PRE_FWD); K1 = 13 and 8 at the tiny dimensions, 40 and 23 at the headline ones."""
    monkeypatch.setenv('RLREP_ENABLE', 'fuse_l1')
    for name in ('vlsac_tiny', 'vlsac_hc'):
        c = Case(name)
        agent = _default_agent(c)
        assert any('enc.l1+l2' in n for n in agent.core.stages(0)), agent.core.stages(0)
        del agent
        _check_against_oracle(c, calls=3, expect_pipeline=True)


@_needs_experiments()
@pytest.mark.parametrize('name', ['vlsac_tiny', 'vlsac_hc'])
def test_cluster_row_programs_match_oracle(name, monkeypatch):
    """RLREP_ENABLE=rowprog=2: the cluster form -- C workgroups per row block and chain (4 at the tiny dimensions, 8 at the headline ones), each
    owning a column slice of every layer, completing each other's activation vectors through tagged 8-byte granules (rowprog.hip
    RP_XCHG / RP_PUBLISH / RP_GATHER) -- in the default mode against the oracle on the read-back draws."""
    monkeypatch.setenv('RLREP_ENABLE', 'rowprog=2')
    c = Case(name)
    worst = _check_against_oracle(c, calls=3, expect_pipeline=True)
    print(f'{name} cluster row programs vs oracle: worst param rel-L2 {worst:.2e}')


@_needs_experiments()
@pytest.mark.parametrize('name,mpg', [('vlsac_tiny', 32), ('vlsac_hc', 32), ('vlsac_hc', 64)])
def test_xcd_chain_feature_step_matches_oracle(name, mpg, monkeypatch):
    """RLREP_ENABLE=xchain (opt-in, csrc/xchain.hip): the forward + dX stages of a feature step as ONE persistent launch whose workgroups hand
    their tiles over inside one XCD's L2 (plain stores, a flag per workgroup, sc1 loads; 32 or 64 workgroups per XCD) -- in the default
    (graph, pipelined) mode against the oracle on the read-back draws; the launch's own device-side checks (every flag carries its
    writer's XCC id; bounded waits) must stay clean."""
    import ctypes as C
    from rlrep_amd._lib import lib
    monkeypatch.setenv('RLREP_ENABLE', f'xchain,xchain_mpg={mpg}')
    c = Case(name)
    worst = _check_against_oracle(c, calls=3, expect_pipeline=True)
    print(f'{name} XCD chain ({mpg} workgroups per XCD) vs oracle: worst param rel-L2 {worst:.2e}')

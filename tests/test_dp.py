"""Data-parallel path: replay sharded over ranks, gradients all-reduced per optimizer step.

* CPU (`-m "not gpu"`): 2 gloo ranks check, on the oracle, that the sum over ranks of the gradients of the
  locally-scaled losses equals the full-batch gradient (the identity the HIP path's 1/(B*world) scaling relies on).
* GPU (`-m gpu`): 2 ranks share cuda:0 with the gloo backend (RCCL refuses two ranks on one device) and run the
  real HIP `train()` with world_size=2 (backward -> all_reduce -> apply); both replicas must stay bit-identical
  and match the CPU oracle run on the concatenated global batch.
"""
import os
import sys
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE, os.path.join(HERE, 'golden')]


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(c, rs, nranks):
    """Per-rank sample indices / noise for ONE train() of a tiny fixture's agent, in the reference's draw order:
    sac: one minibatch; vlsac: one latent noise per feature step; spedersac: two minibatches per feature step; diffsrsac: per feature step
    the noise-level indices (diffsrsac_agent.py:276) and the sigma-scaled perturbation (:283); then the two policy noises."""
    F = c.kw.get('feature_dim', 0)
    nf = 0 if c.alg == 'sac' else c.kw['extra_feature_steps'] + 1
    nb = 2 * nf if c.alg == 'spedersac' else max(nf, 1)
    out = []
    for _ in range(nranks):
        idx = [rs.randint(0, c.meta['replay_n'], size=c.B) for _ in range(nb)]
        eps = [rs.standard_normal((c.B, F)).astype(np.float32) for _ in range(nf)] if c.alg == 'vlsac' else []
        if c.alg == 'diffsrsac':
            for _i in range(nf):
                eps.append(rs.randint(0, c.kw.get('num_noises', 1000), size=c.B).astype(np.int64))
                eps.append((rs.standard_normal((c.B, c.S)) * c.kw.get('sigma_scale_factor', 0.449)).astype(np.float32))
        eps += [rs.standard_normal((c.B, c.A)).astype(np.float32) for _ in range(2)]
        out.append((idx, eps))
    return out


def _oracle_global(c, per_rank, trains=1):
    from oracle import make_oracle
    from oracle.agents import gather_batch
    o = make_oracle(c.alg, c.S, c.A, c.init, **c.kw)
    for t in range(trains):
        pr = per_rank[t]
        nf = len(pr[0][0])
        batches = [gather_batch(c.replay, np.concatenate([r[0][i] for r in pr])) for i in range(nf)]
        eps = [torch.as_tensor(np.concatenate([r[1][i] for r in pr], axis=0)) for i in range(len(pr[0][1]))]
        o.train(batches, eps)
    return o


# ------------------------------------------------------------------------------------------------
# CPU: gradient-sum identity over gloo
# ------------------------------------------------------------------------------------------------
def _cpu_worker(rank, world, port, q):
    from fixture_io import Case
    from oracle import make_oracle
    from oracle.agents import gather_batch, Batch
    import torch.nn.functional as Fn
    torch.set_num_threads(1)
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    c = Case('sac_tiny')
    rs = np.random.RandomState(3)
    idx = rs.randint(0, c.meta['replay_n'], size=c.B * world)
    eps = rs.standard_normal((c.B * world, c.A)).astype(np.float32)
    full = gather_batch(c.replay, idx)
    sl = slice(rank * c.B, (rank + 1) * c.B)
    local = Batch(*(t[sl] for t in full))

    def critic_grads(batch, e, scale):
        o = make_oracle(c.alg, c.S, c.A, c.init, **c.kw)
        from oracle.agents import actor_mu_std, squashed_rsample_logp, double_q
        P = o.P
        with torch.no_grad():
            mu, std = actor_mu_std(P, batch.next_state)
            a2, logp = squashed_rsample_logp(mu, std, torch.as_tensor(e))
            t1, t2 = double_q(P, 'critic_target', batch.next_state, a2)
            y = batch.reward + (1 - batch.done) * o.discount * (torch.min(t1, t2) - o.alpha.detach() * logp)
        q1, q2 = double_q(P, 'critic', batch.state, batch.action)
        loss = (Fn.mse_loss(q1, y) + Fn.mse_loss(q2, y)) * scale
        names = o.names('critic')
        return names, torch.autograd.grad(loss, [P[n] for n in names])

    names, g_local = critic_grads(local, eps[sl], 1.0 / world)      # mean over local B, then /world
    flat = torch.cat([g.reshape(-1) for g in g_local])
    dist.all_reduce(flat)                                           # SUM over ranks == global-mean gradient
    _, g_full = critic_grads(full, eps, 1.0)
    ref = torch.cat([g.reshape(-1) for g in g_full])
    err = float((flat - ref).norm() / ref.norm())
    q.put((rank, err))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_gradient_sum_identity_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_cpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err in res:
        assert err < 1e-5, (rank, err)


def _early_worker(rank, world, port, q):
    """Host logic of the early (exchange kind 3) gradient all-reduce: SACAgent._feature_backward_dp / _allreduce_rest on a stub core."""
    from rlrep_amd.agent.sac.sac_agent import SACAgent
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)

    class Lay:
        group_offset, group_floats, grad_floats = [0, 100, 0, 40], [40, 0, 0, 60], 100

    class Core:
        layout = Lay()

        def __init__(self):
            self.grads = torch.full((100,), float(rank + 1))
            self.parts = []

        def feature_exchange_count(self):
            return 1

        def feature_exchange(self, k):                    # group 3 = [40, 100): its tail [70, 100) is final after part 0
            return 3, self.grads[70:100], 30, 70

        def feature_backward_part(self, k, eps, idx):
            self.parts.append(k)
            if k == 1:
                self.grads[40:70] += 10.0                 # the rest of the backward still writes the remainder of the group ...
                assert self.grads[70].item() in (float(rank + 1), 3.0)     # ... while the early slice is untouched or already summed

    a = SACAgent.__new__(SACAgent)
    a.core, a.world_size, a.rank = Core(), world, rank
    a._seg, a._seg_capture_colls, a._n_captured_colls = None, False, 0
    a._early_works, a._early_slices = [], []
    a._feature_backward_dp(None, None)
    assert a.core.parts == [0, 1] and a._early_slices == [(70, 100)] and len(a._early_works) == 1
    a._allreduce_rest(3)
    a._allreduce(0)
    assert not a._early_works and not a._early_slices
    g = a.core.grads
    ok = bool((g[0:40] == 3.0).all() and (g[40:70] == 23.0).all() and (g[70:100] == 3.0).all())      # every element summed exactly once
    q.put((rank, ok, g[[0, 40, 69, 70, 99]].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_early_slice_allreduce_covers_every_gradient_once_gloo():
    """diffsrsac's head-first gradient exchange (csrc/agents2.hip, exchange kind 3): the early slice is reduced asynchronously from inside the
    backward, the remainder of the group after it -- two gloo ranks on the CPU, every gradient element summed exactly once."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_early_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, sample in res:
        assert ok, (rank, sample)


# ------------------------------------------------------------------------------------------------
# GPU: the HIP train() with world_size 2
# ------------------------------------------------------------------------------------------------
def _gpu_worker(rank, world, port, q, case='vlsac_tiny'):
    from fixture_io import Case
    from test_hip_parity import make_agent, make_buffer
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    c = Case(case)
    agent = make_agent(c)
    assert agent.world_size == world
    buf = make_buffer(c)
    rs = np.random.RandomState(11)
    for t in range(2):
        per_rank = _inputs(c, rs, world)
        agent.train_injected(buf, c.B, *per_rank[rank])
    torch.cuda.synchronize()
    st = {k: v.numpy() for k, v in agent.core.state().items()}
    q.put((rank, st))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['sac_tiny', 'vlsac_tiny', 'ctrlsac_tiny', 'spedersac_tiny', 'diffsrsac_tiny',
                                  # the two agents BASELINE shards (configs 4 and 5), at config dimensions: spedersac Ant F = 512 with B = 1024 per
                                  # rank (the LDS-tiled engine, split-K, the Phibar / v exchange at full width) and diffsrsac at HalfCheetah dims
                                  # (the bf16x3 head, the early all-reduce of the head's gradient slice)
                                  'spedersac_ant512', 'diffsrsac_hc'])
def test_hip_dp_two_ranks_match_global_batch_oracle(case):
    """ctrlsac (in-batch negatives over BOTH ranks' minibatches) and spedersac (global Phibar / v) are exact too:
    the oracle sees one batch of 2B rows.  diffsrsac (BASELINE config 5's agent: two optimizers per feature step, all-reduce of the
    nabla-mu group and of the phi group, diffsrsac_agent.py:271-318) and plain sac run the same backward -> all-reduce -> apply form."""
    from fixture_io import Case, rel_l2
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q, case)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    c = Case(case)
    rs = np.random.RandomState(11)
    per_rank = [_inputs(c, rs, world) for _ in range(2)]
    o = _oracle_global(c, per_rank, trains=2)
    P = o.state()
    for k, v in res[0].items():
        assert np.array_equal(v, res[1][k]), f'replicas diverged at {k}'
        if k in P and not k.endswith('noise'):
            assert rel_l2(v, P[k].numpy()) < 1e-4, k


def _gpu_graph_worker(rank, world, port, q):
    from fixture_io import Case
    from test_hip_parity import make_agent, make_buffer
    os.environ['RLREP_DP_FUSED'] = '0'       # this test is about the torch.distributed form: graph segments around eager all-reduces (the in-launch exchange: tests/test_comm.py)
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    c = Case('vlsac_tiny')
    agent = make_agent(c)
    agent.use_graph = True                   # segmented hipGraph capture around the eager all-reduces
    agent.use_pipeline_dp = False            # the sequential form (the pipelined one has its own test below)
    buf = make_buffer(c)
    infos = [agent.train(buf, c.B) for _ in range(5)]
    torch.cuda.synchronize()
    nseg = sum(1 for k, _ in agent._graph if k == 'graph')
    st = {k: v.numpy() for k, v in agent.core.state().items()}
    q.put((rank, st, nseg, float(infos[-1]['vae_loss'])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_hip_dp_segmented_graph_keeps_replicas_identical():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_graph_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r for r in (q.get(timeout=600) for _ in range(world))}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][2] == 7, 'vlsac train() = 6 all-reduces (4 feature, critic, actor+alpha) -> 7 graph segments'
    for k, v in res[0][1].items():
        assert np.all(np.isfinite(v)), k
        assert np.array_equal(v, res[1][1][k]), f'replicas diverged at {k}'
    assert np.isfinite(res[0][3]) and np.isfinite(res[1][3])


def _gpu_pipe_worker(rank, world, port, q, pipelined):
    from fixture_io import Case
    from test_hip_parity import make_agent, make_buffer
    os.environ['RLREP_PIPELINE_DP'] = '1' if pipelined else '0'
    os.environ['RLREP_DP_FUSED'] = '0'       # the torch.distributed forms (the in-launch exchange takes the single-GPU graph forms: tests/test_comm.py)
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    c = Case('vlsac_tiny')
    agent = make_agent(c, seed=31)
    agent.use_graph = True
    buf = make_buffer(c)
    for t in range(7):
        info = agent.train(buf, c.B)
        if t == 3:
            agent.select_action(np.zeros(c.S, np.float32))
        if rank == 0 and t in (1, 4):
            float(info['q1_loss'])         # ONE rank reads a critic metric (a flush): flush() issues no collective, the ranks stay in step
    last = float(info['q1_loss'])
    torch.cuda.synchronize()
    took = bool(agent._pipe is not None and agent._pipe.get('mode') == 3)
    st = {k: v.numpy() for k, v in agent.core.state().items()}
    q.put((rank, st, took, last))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_hip_dp_pipelined_equals_sequential_dp():
    """Data parallel + deferred critic/actor chain (two streams, graph segments between the six all-reduces, one process group, fixed
    interleaved issue order) ends in exactly the state of the sequential data-parallel train(), on both ranks."""
    out = {}
    for pipelined in (True, False):
        world, port = 2, _free_port()
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        procs = [ctx.Process(target=_gpu_pipe_worker, args=(r, world, port, q, pipelined)) for r in range(world)]
        for p in procs:
            p.start()
        res = {r[0]: r for r in (q.get(timeout=600) for _ in range(world))}
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        assert res[0][2] == pipelined and res[1][2] == pipelined
        for k, v in res[0][1].items():
            assert np.array_equal(v, res[1][1][k]), f'replicas diverged at {k} (pipelined={pipelined})'
        out[pipelined] = res[0]
    for k, v in out[True][1].items():
        assert np.array_equal(v, out[False][1][k]), f'pipelined != sequential at {k}'
    assert out[True][3] == out[False][3]


def _gpu_rccl_worker(port, q, mode, case='vlsac_tiny'):
    """mode: 'single' (no process group), 'dp' (sequential DP over a one-rank RCCL group, all-reduces captured into the graph), 'dp_seg'
    (the same with graph segments around eager all-reduces), 'dp_pipe' (pipelined DP, same group)."""
    from fixture_io import Case
    from test_hip_parity import make_agent, make_buffer
    try:
        if mode != 'single':
            os.environ['RLREP_FORCE_DP'] = '1'
            os.environ['RLREP_PIPELINE_DP'] = '1' if mode == 'dp_pipe' else '0'
            os.environ['RLREP_DP_CAPTURE'] = '0' if mode == 'dp_seg' else '1'
            torch.cuda.set_device(0)
            dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
        c = Case(case)
        agent = make_agent(c, seed=17)
        agent.use_graph = True
        buf = make_buffer(c)
        for t in range(9):
            info = agent.train(buf, c.B)
            if t == 4:
                agent.select_action(np.zeros(c.S, np.float32))
        last = float(info['q1_loss' if 'q1_loss' in info else 'actor_loss'])
        torch.cuda.synchronize()
        form = 'pipe' if agent._pipe is not None else 'graph'
        if mode == 'dp_pipe':
            assert agent._pipe['mode'] == 3
        if mode == 'dp' and case == 'vlsac_tiny':          # the six all-reduces of a vlsac train() are inside ONE captured graph
            assert sum(1 for k, _ in agent._graph if k == 'coll') == 0 and len(agent._graph) == 1 and agent._n_captured_colls == 6
        if mode == 'dp_seg' and case == 'vlsac_tiny':      # RLREP_DP_CAPTURE=0: graph segments around eager collectives
            assert sum(1 for k, _ in agent._graph if k == 'coll') == 6
        if mode != 'single' and case.startswith('diffsrsac'):
            # the nabla-mu head's gradient slice is reduced from inside the backward (exchange kind 3), asynchronously
            assert agent.core.feature_exchange_count() == 1 and agent.core.feature_exchange(0)[0] == 3
            assert not agent._early_works and not agent._early_slices
            if mode == 'dp':
                assert len(agent._graph) == 1 and agent._n_captured_colls > 0
        st = {k: v.numpy() for k, v in agent.core.state().items()}
        q.put((mode, st, last, form))
        if mode != 'single':
            dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((mode, traceback.format_exc(), None, None))


@pytest.mark.gpu
def test_hip_dp_diffsrsac_early_head_allreduce_over_rccl_one_rank():
    """diffsrsac under data parallelism takes the nabla-mu head's weight gradient FIRST and all-reduces that slice (99 % of the group's bytes)
    asynchronously while the rest of the backward runs (csrc/agents2.hip build_diffsrsac, exchange kind 3).  Over a one-rank RCCL
    group (the all-reduce is the identity) both the captured and the segmented form must end in exactly the single-GPU state."""
    out = {}
    ctx = mp.get_context('spawn')
    for mode in ('single', 'dp', 'dp_seg'):
        q = ctx.Queue()
        p = ctx.Process(target=_gpu_rccl_worker, args=(_free_port(), q, mode, 'diffsrsac_tiny'))
        p.start()
        res = q.get(timeout=280)
        p.join(timeout=120)
        assert isinstance(res[1], dict), res[1]
        assert p.exitcode == 0
        out[mode] = res
    for mode in ('dp', 'dp_seg'):
        for k, v in out['single'][1].items():
            assert np.array_equal(v, out[mode][1][k]), f'{mode} != single GPU at {k}'
        assert out[mode][2] == out['single'][2]


@pytest.mark.gpu
def test_hip_dp_over_rccl_one_rank_equals_single_gpu():
    """The data-parallel train() forms (graph segments around eager all-reduces; and the two-stream pipelined one) run over the REAL
    RCCL backend -- a one-rank group, the only RCCL configuration a one-GPU box allows -- and, the all-reduce being the identity there,
    must end in exactly the single-GPU state.  Rehearses ProcessGroupNCCL's stream hand-offs, its watchdog thread beside hipGraph
    capture, and the two issuing streams sharing one communicator."""
    out = {}
    ctx = mp.get_context('spawn')
    for mode in ('single', 'dp', 'dp_seg', 'dp_pipe'):
        q = ctx.Queue()
        p = ctx.Process(target=_gpu_rccl_worker, args=(_free_port(), q, mode))
        p.start()
        res = q.get(timeout=280)
        p.join(timeout=120)
        assert isinstance(res[1], dict), res[1]
        assert p.exitcode == 0
        out[mode] = res
    for mode in ('dp', 'dp_seg', 'dp_pipe'):
        for k, v in out['single'][1].items():
            assert np.array_equal(v, out[mode][1][k]), f'{mode} != single GPU at {k}'
        assert out[mode][2] == out['single'][2]


def _gpu_pipe_big_worker(rank, world, port, q, pipelined):
    import synth
    from rlrep_amd.utils.buffer import ReplayBuffer
    from rlrep_amd.agent.vlsac.vlsac_agent import VLSACAgent
    os.environ['RLREP_PIPELINE_DP'] = '1' if pipelined else '0'
    os.environ['RLREP_DP_FUSED'] = '0'       # the torch.distributed forms (the in-launch exchange takes the single-GPU graph forms: tests/test_comm.py)
    try:
        dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
        torch.cuda.set_device(0)
        torch.manual_seed(0)

        class Sp:
            low, high = -np.ones(6, np.float32), np.ones(6, np.float32)
        agent = VLSACAgent(state_dim=17, action_dim=6, action_space=Sp(), max_batch=256, seed=41, hidden_dim=256, feature_dim=256,
                           extra_feature_steps=3)
        data = synth.replay(17, 6, 4096, seed=10 + rank)                 # every rank its own shard
        buf = ReplayBuffer(17, 6, max_size=4096)
        buf.load(data['state'], data['action'], data['next_state'], data['reward'], data['done'])
        for t in range(30):
            info = agent.train(buf, 256)
            if t == 11:
                float(info['kl_loss'])                                   # an early key: must not disturb the schedule
        agent.flush()
        torch.cuda.synchronize()
        st = {k: v.numpy() for k, v in agent.core.state().items()}
        q.put((rank, st, bool(agent._pipe is not None and agent._pipe.get('mode') == 3)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc(), None))


@pytest.mark.gpu
def test_hip_dp_pipelined_headline_dims():
    """Two ranks (gloo, one GPU), BASELINE config-2 dimensions (the bf16x3 noise-critic kernels, split-K weight gradient, two-stream
    schedule with graph segments between the six all-reduces), 30 train() calls on per-rank replay shards: replicas bit-identical, and
    the pipelined schedule ends exactly where the sequential data-parallel one does."""
    out = {}
    for pipelined in (True, False):
        world, port = 2, _free_port()
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        procs = [ctx.Process(target=_gpu_pipe_big_worker, args=(r, world, port, q, pipelined)) for r in range(world)]
        for p in procs:
            p.start()
        res = {}
        for _ in range(world):
            r = q.get(timeout=400)
            assert isinstance(r[1], dict), r[1]
            res[r[0]] = r
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        assert res[0][2] == pipelined
        for k, v in res[0][1].items():
            assert np.all(np.isfinite(v)), k
            assert np.array_equal(v, res[1][1][k]), f'replicas diverged at {k} (pipelined={pipelined})'
        out[pipelined] = res[0][1]
    for k, v in out[True].items():
        assert np.array_equal(v, out[False][k]), f'pipelined != sequential at {k}'

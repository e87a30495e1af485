"""The data-parallel gradient exchange INSIDE the optimizer launches (rlrep_amd/csrc/comm.hip, dp_pull.h, rlrep_amd/comm.py).
The reference is a single process: there is nothing to compare with but arithmetic -- the sum in RANK ORDER, which every rank must produce bit
for bit -- and the same train() over torch.distributed (gloo), which must end in exactly the same state.  Several processes share the one GPU
of the test box and map each other's gradient arena (a multi-GPU node only changes the wires: there the block is fine-grained memory)."""
import os
import socket
import sys
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE, os.path.join(HERE, 'golden')]

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _data(rank, n, call):
    rs = np.random.RandomState(1000 * call + rank)
    return (rs.standard_normal(n) * (10.0 ** rs.randint(-3, 4, size=n))).astype(np.float32)


SIZES = [1, 3, 64, 1000, 4096 + 3, 262144, 481810]        # ... 481 810: the vlsac feature group at the headline dims (1.93 MB)
OFFSETS = [0, 5, 64, 0, 128, 0, 16]                        # (odd offsets take the element-wise path)


def _spawn(fn, world, *args, timeout=400):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_entry, args=(r, world, port, q, fn) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=timeout)
        assert not isinstance(r[1], str), r[1]
        res[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def _entry(rank, world, port, q, body, *args):
    """Process entry point (top level: the spawn context pickles it by name); `body` names one of the *_body functions below."""
    try:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
        torch.cuda.set_device(0)
        globals()[body](rank, world, q, *args)
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))


def _pull_body(rank, world, q):
    from rlrep_amd.comm import GradientExchange
    ex = GradientExchange(max(SIZES) + 256)
    assert ex.usable and ex.same_device
    assert ex.probe()
    out = []
    for call, (n, off) in enumerate(list(zip(SIZES, OFFSETS)) * 2):           # twice over the same addresses, back to back without a host sync
        ex.arena[off:off + n].copy_(torch.from_numpy(_data(rank, n, call)))
        out.append(ex.all_reduce(off, n))
    torch.cuda.synchronize()
    assert ex.status() == 0
    res = [t.cpu().numpy() for t in out]
    # a rank that runs ahead: rank 0 starts the next exchange immediately, the others late -- the bounded wait must simply wait
    if rank != 0:
        time.sleep(0.3)
    ex.arena[:1024].fill_(float(rank + 1))
    res.append(ex.all_reduce(0, 1024).cpu().numpy())
    assert ex.status() == 0
    q.put((rank, res, ex.fine_grained))
    dist.barrier()
    ex.close()



@pytest.mark.parametrize('world', [2, 3])
def test_in_launch_exchange_is_the_rank_ordered_sum_on_every_rank(world):
    res = _spawn('_pull_body', world)
    for call, (n, off) in enumerate(list(zip(SIZES, OFFSETS)) * 2):
        want = _data(0, n, call).copy()
        for r in range(1, world):
            want = want + _data(r, n, call)                     # float32, rank order: ((r0 + r1) + r2)
        for r in range(world):
            assert np.array_equal(res[r][0][call], want), (world, call, n, r)
    for r in range(world):
        assert np.array_equal(res[r][0][-1], np.full(1024, world * (world + 1) / 2, np.float32))
        assert res[r][1], 'the shared block should be fine-grained device memory on this runtime (hipExtMallocWithFlags + hipIpcGetMemHandle)'


def _late_body(rank, world, q):
    """Rank 1 never calls: rank 0's bounded wait runs out, sets rank 1's bit, the launch drains and status() raises -- no hang, no re-exec."""
    from rlrep_amd.comm import GradientExchange
    ex = GradientExchange(4096)
    mask, raised = 0, False
    if rank == 0:
        ex.arena[:1024].fill_(1.0)
        t0 = time.time()
        ex.all_reduce(0, 1024, timeout_spins=20000)
        torch.cuda.synchronize()
        took = time.time() - t0
        mask = ex.status(raise_on_error=False)
        try:
            ex.status(clear=True)
        except RuntimeError as e:
            raised = 'did not arrive' in str(e)
        assert ex.status() == 0                                  # cleared
        q.put((rank, mask, raised, took))
    else:
        q.put((rank, 0, True, 0.0))
    dist.barrier()
    ex.close()



def test_a_rank_that_never_arrives_sets_the_mask_and_raises_within_the_timeout():
    res = _spawn('_late_body', 2)
    mask, raised, took = res[0]
    assert mask == 0b10 and raised, (mask, raised)
    assert took < 30.0, took


def _agent_body(rank, world, q, case, fused, graph, calls, pipe_dp=False):
    os.environ['RLREP_DP_FUSED'] = '1' if fused else '0'
    os.environ['RLREP_PIPELINE_DP'] = '1' if pipe_dp else '0'
    from fixture_io import Case
    from test_hip_parity import make_agent, make_buffer
    from test_dp import _inputs
    c = Case(case)
    agent = make_agent(c, seed=17) if graph else make_agent(c)
    assert bool(agent.core.fused_groups) == fused, agent.core.fused_groups
    buf = make_buffer(c)
    took = None
    if graph:
        agent.use_graph = True
        for t in range(calls):
            info = agent.train(buf, c.B)
            if t == calls // 2:
                agent.select_action(np.zeros(c.S, np.float32))
            if rank == 0 and t == 1:
                float(next(iter(info.values())))             # ONE rank reads its metrics (a flush): no collective behind it, the ranks stay in step
        agent.flush()
        took = ('pipe' if agent._pipe.get('mode') != 3 else 'pipe_dp') if agent._pipe is not None else ('graph' if not isinstance(agent._graph, list) else 'segments:%d' % sum(1 for k, _ in agent._graph if k == 'coll'))
    else:
        rs = np.random.RandomState(11)
        for t in range(calls):
            per_rank = _inputs(c, rs, world)
            agent.train_injected(buf, c.B, *per_rank[rank])
    torch.cuda.synchronize()
    if fused:
        assert agent.core.exchange.status() == 0
    st = {k: v.numpy() for k, v in agent.core.state().items()}
    q.put((rank, st, took, sorted(agent.core.fused_groups)))
    dist.barrier()
    del agent



@pytest.mark.parametrize('case,world', [('vlsac_tiny', 2), ('spedersac_tiny', 2), ('sac_tiny', 3)])
def test_train_with_gradients_summed_in_the_optimizer_launches_equals_gloo(case, world):
    """Eager step programs, injected draws: every gradient slice summed inside its optimizer launch (spedersac: its Phibar / v exchange still
    over gloo) ends in exactly the state of the same run with torch.distributed all-reduces between backward and apply; replicas bit-identical.
    (Two ranks: a + b in either order.  Three ranks, sac: gloo's ring order is not the rank order -- replicas identical, states equal to 1e-6.)"""
    out = {}
    for fused in (True, False):
        res = _spawn('_agent_body', world, case, fused, False, 2)
        for r in range(1, world):
            for k, v in res[0][0].items():
                assert np.array_equal(v, res[r][0][k]), f'replicas diverged at {k} (fused={fused}, rank {r})'
        out[fused] = res[0]
    assert out[True][2], 'no group was attached'
    for k, v in out[True][0].items():
        if world == 2:
            assert np.array_equal(v, out[False][0][k]), f'in-launch exchange != gloo at {k}'
        else:
            assert np.allclose(v, out[False][0][k], rtol=1e-5, atol=1e-6), k


@pytest.mark.parametrize('case', ['vlsac_tiny', 'vlsac_hc'])
def test_default_graph_forms_with_two_ranks_equal_the_gloo_forms(case):
    """The default train() with two ranks: with the exchange inside the optimizer launches it is the single-GPU form -- two chains on two streams,
    whole hipGraphs, no collective anywhere -- and ends in exactly the state of BOTH torch.distributed forms: graph segments around six gloo
    all-reduces (sequential), and the two-chain form with one process group per chain (RLREP_PIPELINE_DP=1); select_action and a one-sided
    metrics read in between.  vlsac_hc = the headline dimensions."""
    out = {}
    for form, (fused, pipe_dp) in (('fused', (True, False)), ('segments', (False, False)), ('pipe_dp', (False, True))):
        res = _spawn('_agent_body', 2, case, fused, True, 6, pipe_dp, timeout=600)
        for k, v in res[0][0].items():
            assert np.array_equal(v, res[1][0][k]), f'replicas diverged at {k} ({form})'
        out[form] = res[0]
    assert out['fused'][1] == 'pipe', out['fused'][1]
    assert out['segments'][1].startswith('segments:6'), out['segments'][1]
    assert out['pipe_dp'][1] == 'pipe_dp', out['pipe_dp'][1]
    for form in ('segments', 'pipe_dp'):
        for k, v in out['fused'][0].items():
            assert np.array_equal(v, out[form][0][k]), f'in-launch exchange != gloo {form} at {k}'


def test_four_ranks_match_the_global_batch_oracle():
    """Four processes on the one GPU, every gradient slice summed inside its optimizer launch: replicas bit-identical, and equal (1e-4) to the
    CPU oracle run on the concatenated global batch of 4 B rows -- the rank-ordered sum of four arenas against autograd on one batch."""
    from fixture_io import Case, rel_l2
    from test_dp import _inputs, _oracle_global
    world, case = 4, 'vlsac_tiny'
    res = _spawn('_agent_body', world, case, True, False, 2, timeout=600)
    for r in range(1, world):
        for k, v in res[0][0].items():
            assert np.array_equal(v, res[r][0][k]), f'replicas diverged at {k} (rank {r})'
    c = Case(case)
    rs = np.random.RandomState(11)
    per_rank = [_inputs(c, rs, world) for _ in range(2)]
    P = _oracle_global(c, per_rank, trains=2).state()
    for k, v in res[0][0].items():
        if k in P and not k.endswith('noise'):
            assert rel_l2(v, P[k].numpy()) < 1e-4, k
